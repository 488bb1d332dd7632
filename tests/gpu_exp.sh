#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/exp_err.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l)
        print('$label f32', round(j['value'], 2), round(j['ms_per_step'], 2), 'x6', round(j['f32_grade_mode']['value'],2), 'x3', round(j['fast_mode']['value'],2))
"
}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "sdseg3d_end_to_end or 120k_properties" 2>&1 | tail -2
run overlap1 LS3D_OVERLAP=1
run overlap0 LS3D_OVERLAP=0
run overlap1 LS3D_OVERLAP=1
