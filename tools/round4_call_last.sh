#!/bin/bash
# last GPU call of round 4: the serial (lateral stream off) kernel-stats cross-check of HEAD, and the bench record on a second box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/last"; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/ab_env.sh default "LS3D_NOP=1" lateral0 "LS3D_LATERAL_STREAM=0" | tee $OUT/ab.txt
cd /tmp
LS3D_LATERAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --precision bf16x6 --steps 10 --warmup 3 > $OUT/prof_serial.log 2>&1
cp $(find /tmp/prof_serial -name 'bench_kernel_stats.csv' | head -1) $OUT/round4_bench_bf16x6_kernel_stats_lateral_stream_off.csv
cd "$R"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/round4_bench_second_box.json 2> $OUT/bench.err; echo "bench rc=$?"
head -c 400 $OUT/round4_bench_second_box.json
