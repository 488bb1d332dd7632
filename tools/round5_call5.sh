#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c5"; mkdir -p $OUT; export TMPDIR=/tmp
python tools/diag_chain.py 0 8 2>&1 | grep -v amdgpu.ids | tee $OUT/diag.txt
bash tools/ab_env.sh chain "LS3D_TILE_CHAIN=1" costorder "LS3D_CHAIN_ABLATE=8" nodma "LS3D_CHAIN_ABLATE=32" nohalo "LS3D_CHAIN_ABLATE=16" nodma_nohalo "LS3D_CHAIN_ABLATE=48" chain0 "LS3D_TILE_CHAIN=0" | tee $OUT/ab.txt
