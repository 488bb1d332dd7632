#!/bin/bash
# second GPU call of round 4: the batched FrameGraph test, A/Bs of the gather-layer knobs, the serial (lateral stream off) kernel-stats record
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/call2"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider -k "frame_graph_of_a_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
bash tools/ab_env.sh base "LS3D_NOP=1" order0 "LS3D_ORDER_MIN_CC=0" tb256 "LS3D_TARGET_BLOCKS=256" order0_tb256 "LS3D_ORDER_MIN_CC=0 LS3D_TARGET_BLOCKS=256" lateral0 "LS3D_LATERAL_STREAM=0" base2 "LS3D_NOP=2" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_base "LS3D_NOP=1" m_order0 "LS3D_ORDER_MIN_CC=0" | tee -a $OUT/ab.txt
for E in "LS3D_NOP=1" "LS3D_ORDER_MIN_CC=0" "LS3D_ORDER_MIN_CC=0 LS3D_TARGET_BLOCKS=256"; do
  echo "== $E" >> $OUT/layers.txt
  env $E timeout 200 python tools/bench_layers.py --out $OUT/layers.json 2>/dev/null | grep -E "gather|sum of" >> $OUT/layers.txt
done
cat $OUT/layers.txt
cd /tmp
LS3D_LATERAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --precision bf16x6 --steps 10 --warmup 3 > $OUT/prof_serial.log 2>&1
cp $(find /tmp/prof_serial -name 'bench_kernel_stats.csv' | head -1) $OUT/round4_bench_bf16x6_kernel_stats_lateral_stream_off.csv
tail -1 $OUT/prof_serial.log | head -c 600
