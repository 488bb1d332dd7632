#!/bin/bash
# A/B of module constants on the inference bench (LS3D_EXPERIMENT="ops._TILE_CHAIN=0,..." - lidarseg3d_amd/experiments.py - or the four user switches): one short bench run per setting, one line per run.
#   bash tools/ab_env.sh name1 "ENV1=a ENV2=b" name2 "ENV3=c" ...      (EXTRA="--model mseg3d" for other bench arguments)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/ab"; mkdir -p $OUT
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  env $envs timeout 200 python bench.py --steps 30 --warmup 5 --no-extra-modes --no-cpu-baseline --no-train-leg $EXTRA > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j.get("roofline",{}).get("sparse_conv_ms_per_frame",{})
    sr=j.get("stage_rooflines",{})
    print("%-22s value %.1f f/s  graph %.3f ms  eager %.3f ms  stack %.3f ms  reader %.3f  decoder %.3f  bit-identical %s" % (sys.argv[2], j["value"], j["ms_per_step"], j.get("eager_mode",{}).get("ms_per_step",0), r.get("mean",0), sr.get("reader",{}).get("ms",0), sr.get("sffm_decoder",{}).get("ms",0), j.get("graph_mode",{}).get("logits_bit_identical_to_eager")))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
