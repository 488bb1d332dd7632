#!/usr/bin/env python
"""GPU probe (measurement helper): where does the end-to-end error of the 3-plane split modes come from?
Backbone features and logits of a 30k-point frame against the float64 evaluation, per arithmetic and kernel family."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = torch.device("cuda", 0)
    from bench import build_model
    from lidarseg3d_amd import ops, synth
    from lidarseg3d_amd.packing import PackedWeight
    from oracle import ref as orc
    from tests.test_gpu_parity import _f64_sdseg3d
    orc.build_c()
    cfg = synth.NUSC
    model, sd = build_model(dev)
    frame = synth.lidar_frame(30000, seed=12, **cfg)
    want64, feat64 = _f64_sdseg3d(sd, frame, cfg)
    pts = torch.from_numpy(np.concatenate([np.zeros((frame.shape[0], 1), np.float32), frame], 1)).to(dev)
    out = {}
    for name, prec, tile, kinds in (("f32", "f32", True, "subm"), ("bf16x8 tile(subm)", "bf16x8", True, "subm"), ("bf16x8 gather only", "bf16x8", False, "subm"),
                                    ("bf16x8 tile(all kinds)", "bf16x8", True, "subm,conv,inverse"), ("bf16x6 tile(subm)", "bf16x6", True, "subm"),
                                    ("bf16x6 gather only", "bf16x6", False, "subm")):
        ops.set_precision(prec)
        ops.set_tile(tile, kinds=kinds)
        with torch.no_grad():
            model(dict(points=pts, batch_size=1), return_loss=False)
        feat = model.point_head.forward_ret_dict
        logits = feat["out_logits"].double().cpu()
        d = (logits - want64)
        out[name] = dict(logit_rms=float(d.pow(2).mean().sqrt()), logit_max=float(d.abs().max()), logit_mean_signed=float(d.mean()),
                         logit_scale=float(want64.abs().max()))
        print(name, json.dumps(out[name]), flush=True)
    ops.set_precision("f32")
    ops.set_tile(True, kinds="subm")
    # signed bias of one sparse layer on post-ReLU-like (non-negative) activations
    rng = np.random.default_rng(1)
    m, k, n = 8192, 128, 128
    a = np.maximum(rng.normal(size=(m, k)), 0).astype(np.float32)
    b = (rng.normal(size=(27, k, n)) * 0.05).astype(np.float32)
    A, B = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    tbl = torch.from_numpy(rng.integers(0, m, size=(m, 27)).astype(np.int32)).to(dev)
    want = torch.zeros((m, n), dtype=torch.float64, device=dev)
    for kk in range(27):
        want += A.double()[tbl[:, kk].long()] @ B[kk].double()
    pw = PackedWeight(B, 27, k, k, n)
    plan = ops.tile_plan(tbl, torch.zeros((m, 4), dtype=torch.int32, device=dev), (1, 8, 8), 1)
    res = {}
    for name, fn in (("gather f32", lambda: ops.gather_gemm(A, pw, tbl=tbl, cout=n)), ("tile x8", lambda: ops.tile_conv(A, pw, plan, cout=n, products=8)),
                     ("tile x6", lambda: ops.tile_conv(A, pw, plan, cout=n, products=6))):
        if name.startswith("gather"):
            ops.set_precision("f32")
        o = fn().double()
        rel = (o - want) / want.abs().clamp_min(1e-3)
        res[name] = dict(rms=float(rel.pow(2).mean().sqrt()), mean_signed_times_sign=float(((o - want) * torch.sign(want)).mean()), max=float(rel.abs().max()))
        print(name, json.dumps(res[name]), flush=True)
    out["layer_bias"] = res
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_accuracy.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
