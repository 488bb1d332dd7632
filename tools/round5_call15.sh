#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c15"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision bf16x6 --profile-ops 60 > $OUT/ops.txt 2>&1
head -c 14000 $OUT/ops.txt
echo finished
