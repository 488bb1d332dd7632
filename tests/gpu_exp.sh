#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mseg -o bench -- python $R/bench.py --model mseg3d --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode > $OUT/kt_mseg.log 2>&1
grep -h "k_cross_attn\|k_grid_gather\|k_nchw" $OUT/kt_mseg/bench_kernel_stats.csv | cut -c1-200
