#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c19"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "token_attention or column_sums or training or linear_layer" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep "token attention fwd" $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > $OUT/train.json 2> $OUT/train.err; cat $OUT/train.json | head -c 1500; echo
echo finished
