#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c9"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --no-train-leg --precision bf16x6 --steps 6 --warmup 3 > $OUT/tl.log 2>&1
python $R/tools/timeline.py $(find /tmp/tl -name 'bench_kernel_trace.csv' | head -1) 0 > $OUT/timeline_graph.txt 2>&1
tail -30 $OUT/timeline_graph.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlm -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --no-train-leg --model mseg3d --precision bf16x6 --steps 6 --warmup 3 > $OUT/tlm.log 2>&1
python $R/tools/timeline.py $(find /tmp/tlm -name 'bench_kernel_trace.csv' | head -1) 0 > $OUT/timeline_graph_mseg3d.txt 2>&1
cd $R; timeout 200 python tools/probe_stack_brackets.py 2>/dev/null | tee $OUT/brackets.txt
timeout 300 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision bf16x6 --steps 5 --warmup 2 2>/dev/null | tail -1 | tee $OUT/train_bf16x6.json
