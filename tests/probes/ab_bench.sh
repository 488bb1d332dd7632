#!/bin/bash
# A/B helper for gpurun: prints value / ms per step / sparse-conv ms of the f32 leg and the two split-bf16 legs of bench.py
# usage: bash tests/probes/ab_bench.sh label [ENV=VALUE ...] [-- extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
label=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${envs[@]}" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l)
        s = '$label f32 %.2f fps %.2f ms conv %.2f' % (j['value'], j['ms_per_step'], j['roofline']['sparse_conv_ms_per_frame'])
        for k, t in (('f32_grade_mode', 'x6'), ('fast_mode', 'x3')):
            if k in j:
                s += ' | %s %.2f fps conv %.2f' % (t, j[k]['value'], j[k]['sparse_conv_ms_per_frame'])
        if 'throughput_mode' in j:
            t = j['throughput_mode']; s += ' | 2-stream f32 %.2f x6 %.2f x3 %.2f' % (t['f32_frames_per_s'], t['bf16x6_frames_per_s'], t['bf16x3_frames_per_s'])
        print(s)
"
