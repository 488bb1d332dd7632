#!/usr/bin/env python
"""Turn the rocprofv3 counter passes of tests/run_gpu_checks.sh (PMC=1) into profiles/round<N>_pmc.{md,json} and round<N>_pmc_sq.md.

  python profiles/summarize_pmc.py gpurun_out [N=4] [commit]

HBM-side bytes per kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB), the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md
(FETCH_SIZE reports half of the bytes of wide coalesced reads); the counters sit on the L2 -> fabric side, Infinity-Cache hits are
included: an upper bound of DRAM traffic.  Sparse-conv kernels = k_tile_conv<*>, k_gather_gemm_x6<*>, k_gather_gemm<*, true> and k_gather_gemm_bf16x3<*, true, *> (the table-driven
launches); bytes per launch = their summed bytes / their launch count, per precision mode of `bench.py --precision P`."""
import json
import os
import sys

import pandas as pd

SPARSE = r"k_tile_conv<|k_gather_gemm_x6<|k_gather_gemm(_bf16x3)?<.*true"


def load(d, sub, counter):
    df = pd.read_csv(os.path.join(d, sub, "bench_counter_collection.csv"))
    df = df[df.Counter_Name == counter].copy()
    df["k"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 60)
    return df


def main(d, rnd="4", commit=""):
    here = os.environ.get("LS3D_PROFILE_OUT") or os.path.dirname(os.path.abspath(__file__))  # on the GPU box: a directory under gpurun_out/
    os.makedirs(here, exist_ok=True)
    out, lines = {}, ["# HBM-side traffic of the sparse-conv launches (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)", "",
                      "`python bench.py --precision P --steps 3 --warmup 2 --no-cpu-baseline --no-extra-modes` (120k-pt SDSeg3D frame).  bytes = (2 x FETCH_SIZE +",
                      "WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section).  Algorithmic pair-model bytes: 24.4 GB/frame = 659 MB per layer (37 layers; since round 5 the 12 SubM layers of levels 2 and 3 run inside two chained kernel launches: 27 kernel launches per frame).", ""]
    for prec in ("bf16x6", "bf16x8", "f32"):
        try:
            f, w = load(d, "pmc_FETCH_SIZE_" + prec, "FETCH_SIZE"), load(d, "pmc_WRITE_SIZE_" + prec, "WRITE_SIZE")
        except Exception as e:
            continue  # no passes for this arithmetic in this run
        fk, wk = f.groupby("k").Counter_Value.agg(["sum", "count"]), w.groupby("k").Counter_Value.agg(["sum", "count"])
        t = fk.join(wk, lsuffix="_f", rsuffix="_w", how="outer").fillna(0.0)
        t["bytes"] = (2 * t.sum_f + t.sum_w) * 1024
        sp = t[t.index.str.contains(SPARSE)]
        launches = float(sp.count_f.sum())
        out[prec] = dict(traffic_bytes_per_launch=float(sp.bytes.sum() / max(launches, 1)), sparse_launches_counted=launches,
                         correction="(2*FETCH_SIZE+WRITE_SIZE)*1024 (MI355X_MICROARCH.md HBM section)", kernels=sorted(sp.index.tolist()),
                         commit=commit)
        lines += ["## precision %s: %.1f MB per sparse-conv launch (%d launches counted)" % (prec, out[prec]["traffic_bytes_per_launch"] / 1e6, launches), "",
                  "| kernel | launches | FETCH_SIZE KiB / launch | WRITE_SIZE KiB / launch | MB / launch |", "|---|---|---|---|---|"]
        for k, r in t.sort_values("bytes", ascending=False).head(14).iterrows():
            n = max(r.count_f, 1)
            lines.append("| `%s` | %d | %.0f | %.0f | %.1f |" % (k, r.count_f, r.sum_f / n, r.sum_w / max(r.count_w, 1), r.bytes / n / 1e6))
        lines.append("")
    open(os.path.join(here, "round%s_pmc.md" % rnd), "w").write("\n".join(lines) + "\n")
    # ---- SQ counters: MFMA busy etc. per kernel
    sq = ["# SQ counters of the sparse-conv kernels (rocprofv3 --kernel-trace --pmc ..., bench.py --steps 3 --warmup 2, MI355X)", "",
          "SIMD cycles available = kernel time x clock (GRBM_GUI_ACTIVE / 8 XCDs / time) x 1024 SIMDs.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count",
          "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md).", "",
          "| precision | kernel | launches | total ms | clock GHz | MFMA busy / SIMD cycles | waves/SIMD resident | WAIT_ANY | WAIT_INST_ANY | ACTIVE |", "|---|---|---|---|---|---|---|---|---|---|"]
    for prec in ("bf16x6", "bf16x8", "f32", "mseg3d"):  # "mseg3d": the SQ pass of `bench.py --model mseg3d` (k_sffm_decoder_rt, k_sffm_memory)
        try:
            sub = "pmc_SQ_" + prec
            df = pd.read_csv(os.path.join(d, sub, "bench_counter_collection.csv"))
            kt = pd.read_csv(os.path.join(d, sub, "bench_kernel_trace.csv"))
        except Exception as e:
            continue
        kt["dur"] = kt.End_Timestamp - kt.Start_Timestamp
        dur = kt.set_index("Dispatch_Id").dur
        df["k"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 60)
        piv = df.pivot_table(index=["Dispatch_Id", "k"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
        piv["dur"] = piv.Dispatch_Id.map(dur)
        sel = piv[piv.k.str.contains(SPARSE)]
        if len(sel) and prec in out:  # time-weighted MFMA busy of the sparse-conv stack (bench.py: roofline.mfma.mfma_busy)
            t_all = sel.dur.sum()
            clk_all = sel.GRBM_GUI_ACTIVE.sum() / 8.0 / t_all
            out[prec]["mfma_busy"] = float(sel.SQ_VALU_MFMA_BUSY_CYCLES.sum() / (t_all * clk_all * 1024))
            out[prec]["clock_ghz"] = float(clk_all)
        for k, g in piv[piv.k.str.contains(SPARSE + "|k_transvfe|k_sffm")].groupby("k"):
            t_ns = g.dur.sum()
            clk = g.GRBM_GUI_ACTIVE.sum() / 8.0 / t_ns  # GHz
            simd_cycles = t_ns * clk * 1024
            wc = 4.0 * g.SQ_WAVE_CYCLES.sum()
            sq.append("| %s | `%s` | %d | %.2f | %.2f | %.1f %% | %.2f | %.0f %% | %.0f %% | %.0f %% |" % (
                prec, k, len(g), t_ns / 1e6, clk, 100 * g.SQ_VALU_MFMA_BUSY_CYCLES.sum() / simd_cycles, wc / simd_cycles,
                100 * g.SQ_WAIT_ANY.sum() / g.SQ_WAVE_CYCLES.sum(), 100 * g.SQ_WAIT_INST_ANY.sum() / g.SQ_WAVE_CYCLES.sum(),
                100 * g.SQ_ACTIVE_INST_ANY.sum() / g.SQ_WAVE_CYCLES.sum()))
    open(os.path.join(here, "round%s_pmc_sq.md" % rnd), "w").write("\n".join(sq) + "\n")
    json.dump(out, open(os.path.join(here, "round%s_pmc.json" % rnd), "w"), indent=1)
    print("\n".join(lines[-20:]))
    print("\n".join(sq))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "4", sys.argv[3] if len(sys.argv) > 3 else "")
