#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; OUT="$R/gpurun_out"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
lscpu | egrep 'Model name|^CPU\(s\)' >> $OUT/gpu.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --steps ${STEPS:-20} --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
timeout 600 python bench.py --precision bf16x3 --no-cpu-baseline --steps ${STEPS:-20} --warmup 5 > $OUT/bench_bf16x3.log 2> $OUT/bench_bf16x3.err; echo "bench bf16x3 rc=$?" >> $OUT/summary.txt
timeout 600 python bench.py --model mseg3d --steps ${STEPS:-20} --warmup 5 > $OUT/bench_mseg3d.log 2> $OUT/bench_mseg3d.err; echo "bench mseg3d rc=$?" >> $OUT/summary.txt
if [ "${PROFILE:-1}" = "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof.log 2>&1
  echo "rocprof rc=$?" >> $OUT/summary.txt
  if [ "${PMC:-0}" = "1" ]; then
    # HBM traffic counters: separate passes (FETCH_SIZE takes 3 of the 4 TCC slots), kernel-trace only
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
    echo "pmc SQ rc=$?" >> $OUT/summary.txt
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
      echo "pmc $c rc=$?" >> $OUT/summary.txt
    done
  fi
  cd "$R"
fi
tail -5 $OUT/pytest_gpu.log; cat $OUT/summary.txt; tail -3 $OUT/smoke.log; cat $OUT/bench.log; cat $OUT/bench_bf16x3.log; tail -3 $OUT/bench_bf16x3.err; cat $OUT/bench_mseg3d.log; tail -3 $OUT/bench_mseg3d.err
