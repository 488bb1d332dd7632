#!/bin/bash
# the last GPU call of round 4: kernel stats of HEAD (default and lateral stream off) and the full bench record
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/final"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extra-modes --precision bf16x6 --steps 10 --warmup 3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o bench -- $BENCH > $OUT/prof_a.log 2>&1
cp $(find /tmp/prof_a -name 'bench_kernel_stats.csv' | head -1) $OUT/round4_bench_bf16x6_kernel_stats.csv
LS3D_LATERAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- $BENCH > $OUT/prof_b.log 2>&1
cp $(find /tmp/prof_b -name 'bench_kernel_stats.csv' | head -1) $OUT/round4_bench_bf16x6_kernel_stats_lateral_stream_off.csv
tail -1 $OUT/prof_b.log | grep -o '"sparse_conv_ms_per_frame": {[^}]*}' > $OUT/lateral_off_bracket.txt
cd "$R"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/round4_bench.json 2> $OUT/bench.err; echo "bench rc=$?"
head -c 300 $OUT/round4_bench.json; cat $OUT/lateral_off_bracket.txt
