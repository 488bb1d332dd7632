#!/bin/bash
# timing ablations of k_spconv_wgrad_lds (compile-time WGL_ABLATE builds in gpurun_in_ab/): 1 no MFMAs, 2 no split, 4 no row loads, 8 no LDS stores
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c25"; mkdir -p $OUT; export TMPDIR=/tmp
echo "default"; timeout 200 python tools/probe_wgrad_sparse.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print('  level %d ch %3d: planes %.3f ms (f32 %.3f)' % (j['level'], j['ch'], j['planes_forced_ms'], j['f32_ms']))"
for A in 1 2 4 8 3 6; do
echo "WGL_ABLATE=$A"; timeout 200 python tools/ab_library.py gpurun_in_ab/libls3d_wgl$A.so tools/probe_wgrad_sparse.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): j=json.loads(l); print('  level %d ch %3d: planes %.3f ms' % (j['level'], j['ch'], j['planes_forced_ms']))"
done
echo finished
