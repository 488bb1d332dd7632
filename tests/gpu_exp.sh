R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; OUT=$R/gpurun_out; export TMPDIR=/tmp
rm -f $OUT/exp_*.log
for a in 1 0 1 0; do
  LS3D_ORDER_ASC=$a python bench.py --no-cpu-baseline --steps 15 --warmup 4 > $OUT/exp_asc${a}_$RANDOM.log 2>&1
done
cd $R; for f in $OUT/exp_*.log; do echo -n "$f "; python -c "
import json,sys
l=[x for x in open('$f') if x.startswith('{')][-1]; d=json.loads(l); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), 'fast', round(d['fast_mode']['value'],2), round(d['fast_mode']['roofline_frac'],3))"; done
