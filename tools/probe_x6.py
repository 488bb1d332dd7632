#!/usr/bin/env python
"""GPU probe (measurement helper): end-to-end logit error against float64 of the exact-f32 path and of the 3-plane split modes with 8 / 6 plane
products, with round-to-nearest or truncated weight planes (ls3d_set_tile_map bit 5), on several frames."""
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
    from oracle import ref as orc
    from tests.test_gpu_parity import _f64_sdseg3d
    orc.build_c()
    cfg = synth.NUSC
    model, sd = build_model(dev)
    out = {}
    for seed, n in ((12, 30000), (5, 30000), (7, 60000), (3, 30000), (21, 45000)):
        frame = synth.lidar_frame(n, seed=seed, **cfg)
        want64, _ = _f64_sdseg3d(sd, frame, cfg)
        scale = float(want64.abs().max())
        pts = torch.from_numpy(np.concatenate([np.zeros((frame.shape[0], 1), np.float32), frame], 1)).to(dev)
        for name, prec, flags in (("f32", "f32", 0), ("bf16x8 rne", "bf16x8", 0), ("bf16x6 rne", "bf16x6", 0), ("bf16x6 trunc", "bf16x6", 32)):
            ops.set_precision(prec)
            ops.set_tile(True, kinds="subm")
            ops.set_tile_map(flags)
            for m in model.modules():
                if hasattr(m, "_packed"):
                    m._packed = None
            with torch.no_grad():
                model(dict(points=pts, batch_size=1), return_loss=False)
            d = model.point_head.forward_ret_dict["out_logits"].double().cpu() - want64
            r = dict(rms=float(d.pow(2).mean().sqrt()) / scale, max=float(d.abs().max()) / scale, mean_signed=float(d.mean()) / scale)
            out["%s seed %d n %d" % (name, seed, n)] = r
            print("%-14s seed %2d n %6d  rms %.3e  max %.3e  signed mean %+.2e  (relative to |logit|max %.1f)" % (name, seed, n, r["rms"], r["max"], r["mean_signed"], scale),
                  flush=True)
    ops.set_tile_map(0)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_x6.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
