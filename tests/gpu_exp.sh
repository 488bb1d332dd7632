#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>$OUT/exp_err.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); f = j.get('fast_mode') or {}
        print('$label', round(j['value'], 2), round(j['ms_per_step'], 2), 'conv_ms', round(j['roofline']['sparse_conv_ms_per_frame'],2), 'fast', round(f.get('value', 0), 2), 'conv_ms', round(f.get('sparse_conv_ms_per_frame',0),2))
" >> $OUT/exp.txt
}
: > $OUT/exp.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3
run default
run default
cat $OUT/exp.txt
