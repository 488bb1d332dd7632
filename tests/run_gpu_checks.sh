#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel stats (+ PMC passes with PMC=1).  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; OUT="$R/gpurun_out"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
lscpu | egrep 'Model name|^CPU\(s\)' >> $OUT/gpu.txt
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
  timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt
fi
timeout 900 python bench.py --steps ${STEPS:-20} --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
timeout 600 python bench.py --model mseg3d --no-cpu-baseline --no-extra-modes --steps ${STEPS:-20} --warmup 5 > $OUT/bench_mseg3d.log 2> $OUT/bench_mseg3d.err; echo "bench mseg3d rc=$?" >> $OUT/summary.txt
if [ "${TRAIN:-1}" = "1" ]; then  # BASELINE configs[3]: Waymo-geometry MSeg3D training step, 2 x 180k points per GPU (+ the nuScenes SDSeg3D step)
  for P in f32 bf16x6; do
    timeout 300 python tools/bench_train_step.py --steps 5 --warmup 2 --precision $P > $OUT/train_nusc_$P.json 2> $OUT/train.err
    timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision $P > $OUT/train_waymo_$P.json 2>> $OUT/train.err
  done
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 --ddp --syncbn > $OUT/train_waymo_ddp_syncbn_bf16x6.json 2>> $OUT/train.err
  echo "train rc=$?" >> $OUT/summary.txt
fi
if [ "${PROFILE:-1}" = "1" ]; then
  cd /tmp
  for P in bf16x6 f32; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -o bench -- python $R/bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes > $OUT/prof_$P.log 2>&1
    echo "rocprof $P rc=$?" >> $OUT/summary.txt
  done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mseg3d -o bench -- python $R/bench.py --model mseg3d --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes > $OUT/prof_mseg3d.log 2>&1
  if [ "${PMC:-0}" = "1" ]; then
    # counters in their own passes, kernel-trace only (FETCH_SIZE takes 3 of the 4 TCC slots)
    for P in bf16x6 f32; do
      timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ_$P -o bench -- python $R/bench.py --precision $P --steps 3 --warmup 2 --no-cpu-baseline --no-extra-modes > $OUT/pmc_SQ_$P.log 2>&1
      echo "pmc SQ $P rc=$?" >> $OUT/summary.txt
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$P -o bench -- python $R/bench.py --precision $P --steps 3 --warmup 2 --no-cpu-baseline --no-extra-modes > $OUT/pmc_${c}_$P.log 2>&1
        echo "pmc $c $P rc=$?" >> $OUT/summary.txt
      done
    done
  fi
  cd "$R"
fi
tail -5 $OUT/pytest_gpu.log; cat $OUT/summary.txt; tail -3 $OUT/smoke.log; cat $OUT/bench.log | cut -c1-1500; tail -3 $OUT/bench.err; cat $OUT/bench_mseg3d.log | cut -c1-600
