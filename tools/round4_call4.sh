#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/call4"; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/ab_env.sh e0 "LS3D_EARLY_ORDER=0" e2 "LS3D_EARLY_ORDER=2" e0b "LS3D_EARLY_ORDER=0" e2b "LS3D_EARLY_ORDER=2" e1 "LS3D_EARLY_ORDER=1" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_e0 "LS3D_EARLY_ORDER=0" m_e2 "LS3D_EARLY_ORDER=2" m_e0b "LS3D_EARLY_ORDER=0" m_e2b "LS3D_EARLY_ORDER=2" | tee -a $OUT/ab.txt
