R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; OUT=$R/gpurun_out; export TMPDIR=/tmp
rm -f $OUT/exp_*.log
for x in 0 1; do for ro in mask none; do for pr in f32 bf16x3; do
  LS3D_XCD_MAP=$x python bench.py --precision $pr --row-order $ro --no-cpu-baseline --no-fast-mode --steps 15 --warmup 4 > $OUT/exp_${pr}_${ro}_xcd$x.log 2>&1
done; done; done
cd $R; for f in $OUT/exp_*.log; do echo -n "$f "; python -c "
import json,sys
l=[x for x in open('$f') if x.startswith('{')][-1]; d=json.loads(l); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['sparse_conv_ms_per_frame'],2))"; done
