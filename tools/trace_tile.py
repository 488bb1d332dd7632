#!/usr/bin/env python
"""In-kernel trace of ls3d_tile_conv on the 120k-point SDSeg3D frame (measurement helper; needs the MI355X).

Every 6-product tile_conv launch of one eager frame runs the tracing build of the kernel (flags bit 5, include/ls3d.h): each wave of
each work unit records when (100 MHz wall clock) and where (XCC / SE / SH / CU) it ran and how its shader cycles split into
prologue / halo staging / waits at the step barriers / epilogue.  From the records, per launch: the makespan, how many units ran
at once over time and how many CUs held two / one / no unit, unit durations alone and beside a partner, and the phase split.

usage: trace_tile.py [--flags F1,F2,...] [--out gpurun_out/trace_tile]   (flags: values of ls3d_tile_conv's `flags`, one traced frame each)
"""
import os, sys, json, argparse
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from lidarseg3d_amd import ops, synth

U32 = 0xFFFFFFFF


def analyse(t, bin_us=10.0):
    r = t["records"].astype(np.int64) & U32                      # [units][4 waves][16]
    live = r[:, 0, 10] > 0
    r = r[live]
    if r.shape[0] == 0:
        return None
    w0 = (r[..., 0] | (r[..., 1] << 32)).min(1) / 100.0          # us
    w1 = (r[..., 2] | (r[..., 3] << 32)).max(1) / 100.0
    t0 = w0.min()
    w0 -= t0; w1 -= t0
    dur = w1 - w0
    hw, xcc = r[:, 0, 4], r[:, 0, 5] & 15
    cu = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15))
    shared = np.zeros_like(dur)
    for c in np.unique(cu):
        idx = np.nonzero(cu == c)[0]
        for i in idx:
            for j in idx:
                if i != j:
                    shared[i] += max(0.0, min(w1[i], w1[j]) - max(w0[i], w0[j]))
    share_frac = shared / np.maximum(dur, 1e-9)
    span = w1.max()
    nb = int(np.ceil(span / bin_us))
    run, cu2, cu1 = [], [], []
    for b in range(nb):
        a, e = b * bin_us, (b + 1) * bin_us
        ov = np.clip(np.minimum(w1, e) - np.maximum(w0, a), 0, None) / bin_us   # fraction of the bin each unit runs
        run.append(float(ov.sum()))
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1], weights=ov)
        cu2.append(int((per_cu > 1.5).sum())); cu1.append(int(((per_cu > 0.5) & (per_cu <= 1.5)).sum()))
    cyc = r[..., 10].astype(np.float64)
    mhz = float((cyc[:, 0] / np.maximum(dur, 1e-9)).mean())       # cycles per us
    ph = dict(total=cyc, prologue=r[..., 11], staging=r[..., 12], step_barriers=r[..., 13], epilogue=r[..., 14])
    ph["mfma_loop"] = ph["total"] - ph["prologue"] - ph["staging"] - ph["step_barriers"] - ph["epilogue"]
    full = ((r[:, 0, 7] >> 8) & 255) == 1
    first = w0 < 5.0
    groups = {"first_round": first & full, "later_full_units": ~first & full, "half_units": ~full}
    out = dict(rows=t["rows"], cin=t["cin"], cout=t["cout"], units=int(r.shape[0]), full_units=int(full.sum()), cus_used=int(np.unique(cu).size),
               makespan_us=float(span), clock_mhz=mhz, bin_us=bin_us, units_running=[round(x, 1) for x in run], cus_with_2=cu2, cus_with_1=cu1,
               unit_us=dict(mean=float(dur.mean()), p10=float(np.percentile(dur, 10)), p50=float(np.percentile(dur, 50)), p90=float(np.percentile(dur, 90)),
                            max=float(dur.max())), steps_mean=float(r[:, 0, 15].mean()), halo_rows_mean=float(r[:, 0, 8].mean()), groups={})
    for name, m in groups.items():
        if m.sum() == 0:
            continue
        g = dict(n=int(m.sum()), start_us_mean=float(w0[m].mean()), dur_us_mean=float(dur[m].mean()), shared_frac_mean=float(share_frac[m].mean()))
        # units that ran (almost) alone on their CU / (almost) always beside a partner
        for tag, mm in (("alone", m & (share_frac < 0.2)), ("paired", m & (share_frac > 0.8))):
            if mm.sum():
                g[tag] = dict(n=int(mm.sum()), dur_us=float(dur[mm].mean()),
                              phases_us={k: float(v[mm].mean() / mhz) for k, v in ph.items()},
                              wave_spread_us=float((cyc[mm].max(1) - cyc[mm].min(1)).mean() / mhz))
        out["groups"][name] = g
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="0")
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace_tile"))
    a = ap.parse_args()
    dev = torch.device("cuda:0"); torch.cuda.set_device(0)
    ops.set_precision("bf16x6")
    ops.set_tile_chain(False)  # every layer as its own launch (the chained launches of levels 2 / 3 would hide six layers in one call)
    model, _ = bench.build_model(dev)
    f = synth.lidar_frame(a.points, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1)).to(dev)
    ex = dict(points=pts, batch_size=1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    summary = {}
    with torch.no_grad():
        for _ in range(4):
            model(dict(ex), return_loss=False)
        for fl in [int(x, 0) for x in a.flags.split(",")]:
            ops.set_tile_flags(conv=fl)
            model(dict(ex), return_loss=False)
            torch.cuda.synchronize()
            tr = ops.trace_tile_convs(True)
            model(dict(ex), return_loss=False)
            torch.cuda.synchronize()
            ops.trace_tile_convs(False)
            layers = []
            for i, t in enumerate(tr):
                t = dict(t, records=t["records"].cpu().numpy())
                s = analyse(t)
                if s is not None:
                    s["launch"] = i
                    layers.append(s)
            summary["flags_%d" % fl] = layers
            np.savez_compressed("%s_flags%d.npz" % (a.out, fl), **{"launch%02d_%dx%d_%drows" % (i, t["cin"], t["cout"], t["rows"]): t["records"].cpu().numpy()
                                                                  for i, t in enumerate(tr)})
            print("flags %d: %d traced launches" % (fl, len(layers)))
            for s in layers:
                g = s["groups"]
                def grp(n, k):
                    return g.get(n, {}).get(k, {})
                fa, fp, la = grp("first_round", "alone"), grp("first_round", "paired"), grp("later_full_units", "alone")
                print("  #%02d %3d->%3d rows %6d units %4d  makespan %6.1f us  clock %4.0f MHz  unit p50 %5.1f max %5.1f | first round paired: %s | later alone: %s"
                      % (s["launch"], s["cin"], s["cout"], s["rows"], s["units"], s["makespan_us"], s["clock_mhz"], s["unit_us"]["p50"], s["unit_us"]["max"],
                         ("%d x %.0f us" % (fp["n"], fp["dur_us"])) if fp else "-", ("%d x %.0f us" % (la["n"], la["dur_us"])) if la else "-"))
    with open(a.out + ".json", "w") as fh:
        json.dump(summary, fh, indent=1)
    # the most expensive launch in full
    for key, layers in summary.items():
        big = max(layers, key=lambda s: s["makespan_us"] if s["cin"] == s["cout"] else 0)
        print(key, json.dumps(big, indent=1))


if __name__ == "__main__":
    main()
