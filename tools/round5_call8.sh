#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c8"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "every_schedule_switch or linear_layer_backward or lazy_encoded or training_step" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 300 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --precision bf16x6 --steps 5 --warmup 2 2>/dev/null | tail -1 | tee $OUT/train_bf16x6.json
timeout 300 python tools/probe_wgrad_sparse.py 2>/dev/null | tee $OUT/wgrad_sparse.txt
# clocks / power while the bench loops (the power-limit claim of profiles/round5_experiments.md)
(for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo; sleep 0.5; done) > $OUT/smi.txt 2>&1 &
SMI=$!
timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>$OUT/bench.err | tail -1 > $OUT/bench.json
kill $SMI 2>/dev/null
python - <<'PY'
import json
j=json.load(open("gpurun_out/r5c8/bench.json"))
print("value", j["value"], "ms", j["ms_per_step"], "roofline frac", j["roofline"]["frac"])
print("reference_outputs_mode", json.dumps(j.get("reference_outputs_mode"))[:700])
print("train_step", json.dumps(j.get("train_step"))[:1800])
print("mseg3d", j["mseg3d"]["value"], json.dumps(j["mseg3d"].get("reference_outputs_mode"))[:300])
print("exact f32", j["exact_f32_mode"]["value"], "batched", [(l["frames_per_step"], round(l["frames_per_s"],1), round(l.get("graph",{}).get("frames_per_s",0),1)) for l in j["batched"]["legs"]])
PY
sort -u $OUT/smi.txt | head -20
