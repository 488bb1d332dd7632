#!/bin/bash
# A/B: ls3d_split_pair3_rne with plain v_sub_f32 (gpurun_in_ab/libls3d_scalar_sub.so, built with -DLS3D_SPLIT_SCALAR_SUB) against the default build, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c23"; mkdir -p $OUT; export TMPDIR=/tmp
V=gpurun_in_ab/libls3d_scalar_sub.so
run() { # name, then command
  name=$1; shift
  "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    if "step_ms" in j: print("%-28s step %.2f ms (fwd %.2f bwd %.2f)" % (sys.argv[2], j["step_ms"], j["forward_ms"], j["backward_ms"]))
    else: print("%-28s value %.1f f/s  %.3f ms  stack %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], j.get("roofline", {}).get("sparse_conv_ms_per_frame", {}).get("mean", 0)))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
B="--steps 30 --warmup 5 --no-extra-modes --no-cpu-baseline --no-train-leg"
run sd_default python bench.py $B
run sd_scalar python tools/ab_library.py $V bench.py $B
run sd_default_b python bench.py $B
run sd_scalar_b python tools/ab_library.py $V bench.py $B
run ms_default python bench.py --model mseg3d $B
run ms_scalar python tools/ab_library.py $V bench.py --model mseg3d $B
T="--model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6"
run train_default python tools/bench_train_step.py $T
run train_scalar python tools/ab_library.py $V tools/bench_train_step.py $T
timeout 200 python tools/bench_decoder.py > $OUT/dec_default.txt 2>&1; tail -2 $OUT/dec_default.txt
timeout 200 python tools/ab_library.py $V tools/bench_decoder.py > $OUT/dec_scalar.txt 2>&1; tail -2 $OUT/dec_scalar.txt
echo finished
