#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c7"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider --durations=12 -k "every_schedule_switch or linear_layer_backward or interpolate_rows_backward or semantickitti or waymo_config or mseg3d_absolute or bf16_mode_tolerance or sdseg3d_120k_frame or batch_norm_train or training_step or chained_tile" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest.log
for P in bf16x6 f32; do
  timeout 300 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision $P --steps 5 --warmup 2 2>/dev/null | tail -1 | tee $OUT/train_$P.json
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision bf16x6 --steps 5 --warmup 2 --ddp --syncbn 2>/dev/null | tail -1 | tee $OUT/train_ddp.json
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-modes 2>$OUT/bench.err | tail -1 > $OUT/bench.json; python - <<'PY'
import json
j=json.load(open("gpurun_out/r5c7/bench.json"))
print("value", j["value"], "train_step", json.dumps(j.get("train_step"))[:1500])
print("reference_outputs_mode", json.dumps(j.get("reference_outputs_mode"))[:600])
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision bf16x6 --steps 5 --warmup 2 > $OUT/prof_train.log 2>&1
cp $(find /tmp/prof_train -name 'train_kernel_stats.csv' | head -1) $OUT/train_kernel_stats.csv
head -25 $OUT/train_kernel_stats.csv | cut -c1-80,150-400
