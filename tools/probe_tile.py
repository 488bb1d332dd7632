#!/usr/bin/env python
"""GPU probe (measurement helper, not part of the product path): per-layer A/B of the tile-halo convolution against the gather-GEMM
kernels on the rulebooks of the bench frame, GEMM error of every arithmetic against float64, plan build cost.
    python tools/probe_tile.py [--points 120000] [--reps 10]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--focus", action="store_true", help="only the 128->128 SubM layers, tile path, with the timing ablations (for --pmc runs)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from bench import build_model
    from lidarseg3d_amd import ops, spconv, synth
    from lidarseg3d_amd.packing import PackedWeight
    out = {}
    # ---- 1. GEMM error vs float64, wide dynamic range operands
    rng = np.random.default_rng(0)
    m, k, n = 4096, 128, 128
    a = (rng.normal(size=(m, k)) * np.exp(rng.normal(size=(m, k)) * 2)).astype(np.float32)
    b = (rng.normal(size=(27, k, n)) * np.exp(rng.normal(size=(27, k, n)) * 2)).astype(np.float32)
    A, Bm = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    tbl = torch.from_numpy(rng.integers(0, m, size=(m, 27)).astype(np.int32)).to(dev)
    want = torch.zeros((m, n), dtype=torch.float64, device=dev)
    mag = torch.zeros((m, n), dtype=torch.float64, device=dev)
    for kk in range(27):
        want += A.double()[tbl[:, kk].long()] @ Bm[kk].double()
        mag += A.double().abs()[tbl[:, kk].long()] @ Bm[kk].double().abs()
    pw = PackedWeight(Bm, 27, k, k, n)
    errs = {}
    for prec in ("f32", "bf16x8", "bf16x6", "bf16x3"):
        ops.set_precision(prec)
        o = ops.gather_gemm(A, pw, tbl=tbl, cout=n)
        e = ((o.double() - want).abs() / mag)
        errs["gather_" + prec] = [float(e.max()), float(e.pow(2).mean().sqrt())]
    ops.set_precision("f32")
    coords = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    plan = ops.tile_plan(tbl, coords, (1, 8, 8), 1)
    for p in (8, 6):
        o = ops.tile_conv(A, pw, plan, cout=n, products=p)
        e = ((o.double() - want).abs() / mag)
        errs["tile_%d" % p] = [float(e.max()), float(e.pow(2).mean().sqrt())]
    out["gemm_err_vs_f64_max_rms"] = errs
    print(json.dumps(errs), flush=True)

    # ---- 2. per-layer A/B on the bench frame's rulebooks
    model, sd = build_model(dev)
    frame = synth.lidar_frame(args.points, seed=100, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((frame.shape[0], 1), np.float32), frame], 1)).to(dev)
    seen = []
    orig = spconv.SparseConvolution.conv

    def rec(self, x, rb, **kw):
        feats = x.features if isinstance(x, spconv.SparseConvTensor) else x
        seen.append((self, rb, feats.shape[1]))
        return orig(self, x, rb, **kw)
    spconv.SparseConvolution.conv = rec
    with torch.no_grad():
        model(dict(points=pts, batch_size=1), return_loss=False)
    spconv.SparseConvolution.conv = orig
    rows = []
    done = set()
    for layer, rb, cin in seen:
        kind = "inverse" if layer.inverse else rb.kind
        tblx = rb.tbl_inv if layer.inverse else rb.tbl
        key = (id(rb), kind, cin, layer.out_channels)
        if key in done or cin % 32:
            continue
        done.add(key)
        nrow, kvol = tblx.shape
        pairs = int((tblx >= 0).sum())
        if args.focus and not (kind == "subm" and cin == 128 and layer.out_channels == 128):
            continue
        x = torch.randn((rb.in_indices.shape[0] if not layer.inverse else rb.out_indices.shape[0], cin), device=dev)
        w = torch.randn((kvol, cin, layer.out_channels), device=dev) * 0.05
        pw = PackedWeight(w, kvol, cin, cin, layer.out_channels)
        order = rb.order(bool(layer.inverse))
        rb.batch_size = 1
        t_plan = timed(lambda: ops.tile_plan(tblx, (rb.in_indices if layer.inverse else rb.out_indices)[:nrow],
                                             rb.in_shape if layer.inverse else rb.out_shape, 1), 3)
        plan = rb.tile_plan(bool(layer.inverse))
        H = plan.buf  # noqa
        r = dict(kind=kind, rows=nrow, kvol=kvol, cin=cin, cout=layer.out_channels, pairs=pairs, plan_us=t_plan)
        if args.focus:
            for name, fl in (("full", 0), ("no_mfma", 4), ("no_dma", 8), ("no_halo", 16), ("no_mfma_no_dma", 12), ("no_dma_no_halo", 24), ("only_sync", 28)):
                ops._L().ls3d_set_tile_map(fl)
                r["tile8_" + name] = timed(lambda: ops.tile_conv(x, pw, plan, cout=layer.out_channels, products=8), args.reps)
            ops._L().ls3d_set_tile_map(0)
            rows.append(r)
            print(json.dumps(r), flush=True)
            continue
        ref = None
        for prec in ("f32", "bf16x6", "bf16x8"):
            ops.set_precision(prec)
            r["gather_%s_us" % prec] = timed(lambda: ops.gather_gemm(x, pw, tbl=tblx, order=order, cout=layer.out_channels), args.reps)
            if prec == "f32":
                ref = ops.gather_gemm(x, pw, tbl=tblx, order=order, cout=layer.out_channels)
        ops.set_precision("f32")
        for p in (8, 6):
            for flat in (0, 1):
                ops._L().ls3d_set_tile_map(flat)
                r["tile_%d_map%d_us" % (p, flat)] = timed(lambda: ops.tile_conv(x, pw, plan, cout=layer.out_channels, products=p), args.reps)
            ops._L().ls3d_set_tile_map(0)
            got = ops.tile_conv(x, pw, plan, cout=layer.out_channels, products=p)
            r["tile_%d_maxdiff_vs_f32" % p] = float((got - ref).abs().max() / ref.abs().max())
        gb = pairs * (cin + layer.out_channels) * 4 / 1e9
        r["pair_GB"] = gb
        r["tile_8_frac_of_8TBs"] = gb / (r["tile_8_map0_us"] * 1e-6) / 8000
        rows.append(r)
        print(json.dumps(r), flush=True)
    out["layers"] = rows
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_focus.json" if args.focus else "probe_tile.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
