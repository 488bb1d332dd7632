#!/usr/bin/env python
"""Summarise an `rocprofv3 --kernel-trace --pmc SQ_...` output directory per kernel (measurement helper; run on the GPU box, the raw CSVs of a
training step exceed gpurun's pull limit).  usage: python tools/pmc_summary.py <dir> <prefix> <kernel-regex>"""
import sys

import pandas as pd

d, pre, pat = sys.argv[1], sys.argv[2], sys.argv[3]
df = pd.read_csv("%s/%s_counter_collection.csv" % (d, pre))
kt = pd.read_csv("%s/%s_kernel_trace.csv" % (d, pre))
kt["dur"] = kt.End_Timestamp - kt.Start_Timestamp
dur = kt.set_index("Dispatch_Id").dur
df["k"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 40)
piv = df.pivot_table(index=["Dispatch_Id", "k"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
piv["dur"] = piv.Dispatch_Id.map(dur)
for k, g in piv[piv.k.str.contains(pat)].groupby("k"):
    t = g.dur.sum() / 1e9
    clk = g.GRBM_GUI_ACTIVE.sum() / 8 / t / 1e9
    simd = t * clk * 1e9 * 1024
    line = "%-36s n=%3d total %.1f ms clk %.2f GHz | MFMA busy %.1f %% | waves/SIMD %.2f | WAIT_ANY %.0f %% WAIT_INST %.0f %% ACTIVE %.0f %%" % (
        k, len(g), t * 1e3, clk, 100 * g.SQ_VALU_MFMA_BUSY_CYCLES.sum() / simd, 4 * g.SQ_WAVE_CYCLES.sum() / simd,
        100 * g.SQ_WAIT_ANY.sum() / g.SQ_WAVE_CYCLES.sum(), 100 * g.SQ_WAIT_INST_ANY.sum() / g.SQ_WAVE_CYCLES.sum(),
        100 * g.SQ_ACTIVE_INST_ANY.sum() / g.SQ_WAVE_CYCLES.sum())
    for extra in ("SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if extra in g:
            line += " | %s %.3g" % (extra, g[extra].sum() / max(len(g), 1))
    print(line)
