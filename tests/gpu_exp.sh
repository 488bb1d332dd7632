#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT="$R/gpurun_out"; mkdir -p $OUT
timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>$OUT/exp_err.log | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l)
        print('f32', round(j['value'], 2), round(j['ms_per_step'], 2), 'conv_ms', round(j['roofline']['sparse_conv_ms_per_frame'],2))
        for k in ('f32_grade_mode','fast_mode'):
            f = j[k]; print(k, round(f['value'],2), round(f['ms_per_step'],2), 'conv_ms', round(f['sparse_conv_ms_per_frame'],2), 'roof', round(f['roofline_frac'],3), 'relerr', f['max_rel_logit_diff_vs_f32'], 'agree', f['argmax_agreement_vs_f32'])
"
tail -2 $OUT/exp_err.log
for tb in 192 1024 2000; do LS3D_TARGET_BLOCKS=$tb timeout 300 python bench.py --precision bf16x6 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('bf16x6 tb$tb', round(j['value'], 2), round(j['ms_per_step'], 2), 'conv_ms', round(j['roofline']['sparse_conv_ms_per_frame'],2))
"; done
