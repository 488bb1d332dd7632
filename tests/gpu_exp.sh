#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT
run() {  # label, model, env...
  local label=$1; local model=$2; shift; shift
  env "$@" timeout 300 python bench.py --model $model --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/exp_err.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l)
        print('$label $model f32', round(j['value'], 2), round(j['ms_per_step'], 2), 'x6', round(j['f32_grade_mode']['value'],2), 'x3', round(j['fast_mode']['value'],2))
"
}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "end_to_end or 120k or drop_in or devox" 2>&1 | tail -2
run now sdseg3d
run now mseg3d
run now sdseg3d
