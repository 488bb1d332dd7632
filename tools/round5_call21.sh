#!/bin/bash
# clock of the tile kernel on random vs all-zero inputs: per-dispatch GRBM_GUI_ACTIVE (cycles summed over the 8 XCDs) / duration, bench_layers --zero-input
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c21"; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pz -o z -- python $R/tools/bench_layers.py --zero-input --reps 10 --out $OUT/layers.json > $OUT/layers.txt 2>&1
python - $(find /tmp/pz -name z_counter_collection.csv | head -1) $(find /tmp/pz -name z_kernel_trace.csv | head -1) > $OUT/clock_zero_vs_random.txt <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = sorted((v[1], k) for k, v in dur.items() if "k_tile_conv<4, 6, false, 1, false>" in v[0])
# bench_layers: per layer 2 + reps random launches, then 2 + reps zero-input launches: runs of 12 dispatches alternate random / zero
per_frame = len(rows) // 28  # 4 model frames (3 warm + the recorded one) + per recorded launch 12 random + 12 zero-input replays
rows = rows[4 * per_frame:]
runs, cur = [], []
for _, k in rows:
    cur.append(k)
    if len(cur) == 12:
        runs.append(cur); cur = []
print("k_tile_conv<4, 6, false, 1, false>: %d dispatches, %d runs of 12" % (len(rows), len(runs)))
print("run  input    us/launch   cycles/launch (GRBM_GUI_ACTIVE / 8)   clock GHz   MFMA busy")
for i, run in enumerate(runs[:40]):
    us = sum(dur[k][2] for k in run[2:]) / 10
    cyc = sum(cnt[k].get("GRBM_GUI_ACTIVE", 0) for k in run[2:]) / 10 / 8
    mf = sum(cnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for k in run[2:]) / 10
    print("%3d  %-7s %9.1f   %12.0f   %6.3f   %5.1f %%" % (i, "zero" if i % 2 else "random", us, cyc, cyc / us / 1e3, 100.0 * mf / max(cyc * 1024, 1)))
PY
head -45 $OUT/clock_zero_vs_random.txt
echo finished
