#!/usr/bin/env python
"""Device check + A/B timing of ls3d_sffm_memory (the class-embedding side of the SF-Phase decoder in one launch; needs the MI355X).

The kernel was built after round 3's GPU budget was spent: tests/hipsim pins it to the layer-by-layer form, this script is the first thing
to run on the device before LS3D_FUSED_SFFM_MEMORY becomes the default:

    python tools/check_sffm_memory.py [--cls 17] [--layers 6] [--batch 1] [--points 120000] [--reps 50]

prints one JSON line: max |kv difference|, max |decoder output difference|, microseconds of the memory side layer by layer / in one
launch, and of the whole SemanticFeatureFusionModule forward both ways.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarseg3d_amd import ops, point_heads  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cls", type=int, default=17)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = point_heads.SemanticFeatureFusionModule(64, 48, 32, d_model=96, nhead=4, num_decoder_layers=a.layers, dim_feedforward=192).to(dev).eval()
    B, L, E = a.batch, 2 * a.cls, 96
    pk = m.packed()
    mem = torch.randn(B * L, E, device=dev)

    def by_layer():
        kvs, mf = [], mem
        for lp in pk["layers"]:
            att = ops.mha_core(point_heads._lin(mf, lp["sa_qkv"]), B, L, E, 4)
            mf = point_heads._lin(att, lp["sa_out"], res=mf, ln=lp["n1"])
            kvs.append(point_heads._lin(mf, lp["k"]).view(B, L, E).permute(0, 2, 1))
            kvs.append(point_heads._lin(mf, lp["v"]).view(B, L, E).permute(0, 2, 1))
        return torch.stack(kvs).contiguous()

    want, got = by_layer(), ops.sffm_memory(mem, B, L, pk["memory"])
    n = a.points * B
    x = torch.randn(n, 64, device=dev)
    e1, e2 = torch.randn(B, 48, a.cls, 1, device=dev), torch.randn(B, 32, a.cls, 1, device=dev)
    bidx = torch.arange(n, device=dev) // a.points
    pts = torch.cat([bidx[:, None].float(), torch.randn(n, 3, device=dev)], 1).contiguous()
    with torch.no_grad():
        ref = m(x, e1, e2, bidx, B, points=pts)
        t_ref = timed(lambda: m(x, e1, e2, bidx, B, points=pts), a.reps)
        point_heads.set_fused_sffm_memory(True)
        out = m(x, e1, e2, bidx, B, points=pts)
        t_one = timed(lambda: m(x, e1, e2, bidx, B, points=pts), a.reps)
        point_heads.set_fused_sffm_memory(False)
    rec = dict(what="ls3d_sffm_memory vs layer by layer", cls=a.cls, layers=a.layers, batch=B, points=a.points,
               kv_max_abs_diff=float((got - want).abs().max()), out_max_abs_diff=float((out - ref).abs().max()),
               memory_side_us_layer_by_layer=timed(by_layer, a.reps), memory_side_us_one_launch=timed(lambda: ops.sffm_memory(mem, B, L, pk["memory"]), a.reps),
               sffm_forward_us_layer_by_layer=t_ref, sffm_forward_us_one_launch=t_one)
    print(json.dumps(rec))
    return 0 if rec["kv_max_abs_diff"] < 1e-4 and rec["out_max_abs_diff"] < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
