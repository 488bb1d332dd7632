#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c14"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pt -o tr -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 2 --warmup 1 --precision bf16x6 > $OUT/pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pt2 -o tr -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 2 --warmup 1 --precision bf16x6 > $OUT/pmc2.log 2>&1
for d in /tmp/pt /tmp/pt2; do
python - $(find $d -name tr_counter_collection.csv | head -1) $(find $d -name tr_kernel_trace.csv | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(s in k for s in ("wgrad", "k_tile_conv<4", "k_gather_gemm<32, 3")): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, v in sorted(acc.items()):
    print(k[:60], "launches", n[k], "us %.0f" % dur[k], " ".join("%s %.4g" % kv for kv in sorted(v.items())))
PY
done
echo finished
