#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c10"; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/ab_env.sh chain "LS3D_TILE_CHAIN=1" nodmawait "LS3D_CHAIN_ABLATE=1" nodma "LS3D_CHAIN_ABLATE=32" nodmawait_nohalo "LS3D_CHAIN_ABLATE=17" nodma_nohalo "LS3D_CHAIN_ABLATE=48" chain_b "LS3D_TILE_CHAIN=1" | tee $OUT/ab.txt
