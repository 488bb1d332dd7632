#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c22"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "wgrad or backward or training or linear_layer" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python tools/probe_wgrad_sparse.py > $OUT/wgrad_sparse.txt 2>&1; tail -12 $OUT/wgrad_sparse.txt
timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > $OUT/train.json 2> $OUT/train.err; cat $OUT/train.json | head -c 900; echo
echo finished
