#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "backward or training" 2>&1 | tail -2
timeout 300 python tools/bench_backbone_train.py 2>&1 | tail -1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -o t -- python $R/tools/bench_backbone_train.py --steps 2 --warmup 1 > $OUT/kt_train.log 2>&1
head -8 $OUT/kt_train/t_kernel_stats.csv | cut -c1-60,150-260
