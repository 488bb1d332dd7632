#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c11"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "training or batch_norm or syncbn or linear_layer" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > $OUT/train.json 2> $OUT/train.err; cat $OUT/train.json | head -c 1500; echo
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 --ddp --syncbn 2>> $OUT/train.err | grep "^{" | tail -1 > $OUT/train_ddp.json; head -c 1500 $OUT/train_ddp.json; echo
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o tr -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > /dev/null 2>&1; cp $(find /tmp/ptr -name tr_kernel_stats.csv | head -1) $OUT/train_kernel_stats.csv)
head -45 $OUT/train_kernel_stats.csv | cut -c1-150
echo finished
