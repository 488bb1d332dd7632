#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c13"; mkdir -p $OUT; export TMPDIR=/tmp
LS3D_TILE_COLOR=0 timeout 300 python tools/bench_layers.py --zero-input --out $OUT/layers_color0.json > $OUT/layers_color0.txt 2>&1
LS3D_TILE_COLOR=1 timeout 300 python tools/bench_layers.py --zero-input --out $OUT/layers_color1.json > $OUT/layers_color1.txt 2>&1
grep "128->128\|64-> 64" $OUT/layers_color0.txt | head -12; echo; grep "128->128\|64-> 64" $OUT/layers_color1.txt | head -12
cd /tmp
for C in 0 1; do
LS3D_TILE_COLOR=$C timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm$C -o b -- python $R/bench.py --steps 3 --warmup 2 --no-extra-modes --no-cpu-baseline --no-train-leg --no-graph > $OUT/pmc$C.log 2>&1
python - $(find /tmp/pm$C -name b_counter_collection.csv | head -1) $C <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_tile_conv" not in k: continue
    k = k.split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"]);
    if (k, key) not in seen: seen.add((k, key)); n[k] += 1
for k, v in acc.items():
    print("color", sys.argv[2], k, "launches", n[k], " ".join("%s %.3g" % kv for kv in sorted(v.items())), "conflict/active %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done
echo finished
