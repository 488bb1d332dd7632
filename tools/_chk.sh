#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/wg; export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "wgrad_on_bf16 or sparse_conv_backward_gpu or layer_norm or linear" -p no:cacheprovider > gpurun_out/wg/t.log 2>&1; tail -2 gpurun_out/wg/t.log
timeout 200 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > gpurun_out/wg/on.json 2>gpurun_out/wg/err.log
cat gpurun_out/wg/on.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o tr -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > /dev/null 2>&1; cp $(find /tmp/ptr -name tr_kernel_stats.csv | head -1) $R/gpurun_out/wg/stats.csv)
head -12 gpurun_out/wg/stats.csv | cut -c1-150
