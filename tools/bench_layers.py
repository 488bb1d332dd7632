#!/usr/bin/env python
"""Per-layer timing of the sparse-convolution stack on the 120k-point SDSeg3D frame (measurement helper; needs the MI355X).

One eager frame runs with ops.tile_conv / ops.gather_gemm wrapped: every launch is recorded with its arguments, then replayed REPS times
back to back between two events (same inputs, same output buffer).  Per launch: kernel path, rows, channels, kernel offsets, active
pairs, microseconds, pair-model GB/s (pairs x (cin + cout) x 4 bytes) and useful TFLOP/s.

usage: bench_layers.py [--model sdseg3d|mseg3d] [--reps 20] [--out gpurun_out/layers.json]; LS3D_EXPERIMENT="ops._TILE_FLAGS=1,..." (lidarseg3d_amd/experiments.py) applies
"""
import os, sys, json, argparse
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from lidarseg3d_amd import ops, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--precision", default="bf16x6")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "layers.json"))
    ap.add_argument("--zero-input", action="store_true", help="time every launch on an all-zero input matrix as well (same instructions, less switching: the power side of the clock)")
    a = ap.parse_args()
    dev = torch.device("cuda:0"); torch.cuda.set_device(0)
    ops.set_precision(a.precision)
    ops.set_tile_chain(False)  # every layer as its own launch (the chained launches of levels 2 / 3 would hide six layers in one call)
    model, _ = bench.build_model(dev)
    f = synth.lidar_frame(a.points, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1)).to(dev)
    ex = dict(points=pts, batch_size=1)
    calls = []
    g, t = ops.gather_gemm, ops.tile_conv

    def wg(x, w, tbl=None, **kw):
        out = g(x, w, tbl=tbl, **kw)
        if tbl is not None:
            calls.append(("gather", g, (x, w), dict(kw, tbl=tbl, out=out), tbl, kw.get("n_dev"), w.shape[1], kw.get("cout") or w.cout))
        return out

    def wt(x, w, plan, **kw):
        out = t(x, w, plan, **kw)
        calls.append(("tile", t, (x, w, plan), dict(kw, out=out), plan.tbl, plan.n_dev, w.shape[1], kw.get("cout") or w.cout))
        return out
    with torch.no_grad():
        for _ in range(3):
            model(dict(ex), return_loss=False)
        torch.cuda.synchronize()
        ops.gather_gemm, ops.tile_conv = wg, wt
        try:
            model(dict(ex), return_loss=False)
            torch.cuda.synchronize()
        finally:
            ops.gather_gemm, ops.tile_conv = g, t
        rows_out, total = [], 0.0
        for i, (kind, fn, args, kw, tbl, n_dev, cin, cout) in enumerate(calls):
            n = int(n_dev.item()) if n_dev is not None else tbl.shape[0]
            pairs = int((tbl[:n] >= 0).sum().item())
            for _ in range(2):
                fn(*args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn(*args, **kw)
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / a.reps
            total += us
            us_zero = None
            if a.zero_input:
                keep = args[0].clone()
                args[0].zero_()
                for _ in range(2):
                    fn(*args, **kw)
                e0.record()
                for _ in range(a.reps):
                    fn(*args, **kw)
                e1.record(); torch.cuda.synchronize()
                us_zero = 1e3 * e0.elapsed_time(e1) / a.reps
                args[0].copy_(keep)
            r = dict(launch=i, path=kind, rows=n, kvol=int(tbl.shape[1]), cin=cin, cout=cout, pairs=pairs, pairs_per_row=pairs / max(n, 1), us=us,
                     pair_model_GBps=pairs * (cin + cout) * 4.0 / us / 1e3, useful_TFLOPs=2.0 * pairs * cin * cout / us / 1e6,
                     fused=[k for k in ("scale", "res_pre", "pair", "relu") if kw.get(k) is not None and kw.get(k) is not False])
            if us_zero is not None:
                r["us_zero_input"] = us_zero
            rows_out.append(r)
            print("#%02d %-6s rows %6d kvol %2d %3d->%3d pairs/row %5.2f  %7.1f us  %6.0f GB/s (pair model)  %5.1f TF useful  %s"
                  % (i, kind, n, r["kvol"], cin, cout, r["pairs_per_row"], us, r["pair_model_GBps"], r["useful_TFLOPs"], ",".join(r["fused"]))
                  + ("  zero input %7.1f us" % us_zero if us_zero is not None else ""), flush=True)
        print("sum of the %d launches, each alone on the GPU: %.1f us" % (len(rows_out), total))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(precision=a.precision, points=a.points, reps=a.reps, total_us=total, launches=rows_out), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
