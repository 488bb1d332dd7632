#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tv -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode > $OUT/kt_tv.log 2>&1
grep -h "k_transvfe" $OUT/kt_tv/*kernel_stats* | cut -c1-160
