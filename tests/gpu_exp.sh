#!/bin/bash
# ad-hoc A/B: workgroup->(tile,slab) mapping, region-blocked mask order, workgroup targets
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>$OUT/exp_err.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); f = j.get('fast_mode') or {}
        print('$label', round(j['value'], 2), round(j['ms_per_step'], 2), round(j['roofline']['frac'], 3), 'fast', round(f.get('value', 0), 2), round((f.get('roofline') or {}).get('frac', 0), 3))
" >> $OUT/exp.txt
}
: > $OUT/exp.txt
run legacy_slabmajor LS3D_XCD_MAP=2
run new_default LS3D_XCD_MAP=0
run new_contig LS3D_XCD_MAP=1
run legacy_slabmajor LS3D_XCD_MAP=2
run new_default LS3D_XCD_MAP=0
run contig_region64 LS3D_XCD_MAP=1 LS3D_REGION=64
run contig_region32 LS3D_XCD_MAP=1 LS3D_REGION=32
run interl_region64 LS3D_XCD_MAP=0 LS3D_REGION=64
run new_tb512 LS3D_XCD_MAP=0 LS3D_TARGET_BLOCKS=512
run new_tb1024 LS3D_XCD_MAP=0 LS3D_TARGET_BLOCKS=1024
run new_tb4000 LS3D_XCD_MAP=0 LS3D_TARGET_BLOCKS=4000
cat $OUT/exp.txt; tail -3 $OUT/exp_err.log
