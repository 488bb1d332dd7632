#!/bin/bash
# Training-step records of HEAD on the MI355X box (one short gpurun call): the training -m gpu tests, then the Waymo step in bf16x6 / f32 /
# DDP + SyncBN, the SDSeg3D nuScenes step and the per-kernel statistics of the Waymo step -> gpurun_out/train_r$ROUND/ (copy into profiles/).
ROUND=${ROUND:-4}; R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/train_r$ROUND"; mkdir -p $OUT; export TMPDIR=/tmp
W="--model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2"
timeout 150 python -m pytest tests/test_gpu_parity.py -q -x -k "training_step or linear_weight or layer_norm" -p no:cacheprovider > $OUT/pytest_training.log 2>&1; tail -2 $OUT/pytest_training.log
timeout 60 python tools/bench_train_step.py $W --precision bf16x6 > $OUT/round${ROUND}_train_step_mseg3d_waymo_2frames_bf16x6.json 2> $OUT/err.log
timeout 60 python tools/bench_train_step.py $W --precision f32 > $OUT/round${ROUND}_train_step_mseg3d_waymo_2frames_f32.json 2>> $OUT/err.log
timeout 60 python tools/bench_train_step.py --steps 5 --warmup 2 --precision bf16x6 > $OUT/round${ROUND}_train_step_sdseg3d_nusc_bf16x6.json 2>> $OUT/err.log
timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/bench_train_step.py $W --precision bf16x6 --ddp --syncbn 2>> $OUT/err.log | grep "^{" | tail -1 > $OUT/round${ROUND}_train_step_mseg3d_waymo_2frames_ddp_syncbn_bf16x6.json
(cd /tmp; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o tr -- python $R/tools/bench_train_step.py $W --precision bf16x6 > /dev/null 2>&1; cp $(find /tmp/ptr -name tr_kernel_stats.csv | head -1) $OUT/round${ROUND}_train_step_mseg3d_waymo_kernel_stats.csv)
for f in $OUT/*.json; do echo $(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(d['step_ms'], d['forward_ms'], d['backward_ms'])" 2>&1 | tail -1); done
