#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c20"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_m -o bench -- python $R/bench.py --model mseg3d --steps 5 --warmup 3 --no-extra-modes --no-cpu-baseline --no-train-leg --no-graph > /dev/null 2>&1
cd $R
python tools/timeline.py $(find /tmp/tl_m -name bench_kernel_trace.csv | head -1) > $OUT/timeline_mseg3d_eager.txt 2>&1
grep -n "k_tile_conv<1, 6, false, 0, false>" $OUT/timeline_mseg3d_eager.txt | tail -2
awk '/k_gather_gemm_x6<1>/{f=1} f' $OUT/timeline_mseg3d_eager.txt | head -60 | cut -c1-130
echo finished
