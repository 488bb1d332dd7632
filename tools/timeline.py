#!/usr/bin/env python
"""Print the kernel timeline of the LAST frame of a rocprofv3 --kernel-trace CSV (measurement helper): per stream/queue, start offset,
duration, name; plus per-kernel totals.  usage: python tools/timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows), key=lambda t: t[0])
marks = [i for i, e in enumerate(ev) if "k_vox_insert" in e[2]]
a = marks[-2] if len(marks) >= 2 else 0
b = marks[-1] if len(marks) >= 2 else len(ev)
t0 = ev[a][0]
tot = defaultdict(float)
cnt = defaultdict(int)
last_end = {}
for s, e, name, q in ev[a:b]:
    short = name.split("(")[0].replace("void ", "")[:58]
    tot[short] += (e - s) / 1e3
    cnt[short] += 1
    gap = (s - last_end.get(q, s)) / 1e3
    last_end[q] = e
    if (e - s) / 1e3 >= min_us:
        print("q%-3s +%9.1f us  dur %8.1f  gap %7.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short))
print("frame span: %.1f us, %d launches" % ((max(e for _, e, _, _ in ev[a:b]) - t0) / 1e3, b - a))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print("%9.1f us  x%-3d %s" % (v, cnt[k], k))
