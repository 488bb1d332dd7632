#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c26"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "token_attention or training" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep "token attention fwd" $OUT/pytest.log; tail -3 $OUT/pytest.log
for i in 1 2; do
timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > $OUT/train$i.json 2> $OUT/train.err; python -c "
import json; j=json.load(open('$OUT/train$i.json')); print('step %.2f ms fwd %.2f bwd %.2f loss %.6f' % (j['step_ms'], j['forward_ms'], j['backward_ms'], j['loss_last']))"
done
echo finished
