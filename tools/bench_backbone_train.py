#!/usr/bin/env python
"""Forward + backward of the sparse UNet backbone in training mode on one synthetic 120k-point frame (SURVEY.md 8f rank 1:
the first piece of the training step).  Prints one JSON line; not the headline benchmark (bench.py is)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from lidarseg3d_amd import ops, scn_unet, synth
    cfg = synth.NUSC
    dev = "cuda:0"
    frame = synth.lidar_frame(args.points, seed=0, **cfg)
    pts = torch.from_numpy(np.concatenate([np.zeros((frame.shape[0], 1), np.float32), frame], 1)).to(dev)
    v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 200000, batched=True)
    V = int(nv)
    coords = c[:V].contiguous()
    feats0 = torch.randn((V, 16), device=dev)
    net = scn_unet.UNetSCN3D(num_input_features=16, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                             model_cfg=dict(SCALING_RATIO=2), ds_factor=8, us_factor=8).to(dev).train()
    shape = np.asarray(ops.make_grid(cfg["voxel_size"], cfg["pc_range"])[1])
    tf = tb = 0.0
    for it in range(args.warmup + args.steps):
        for p in net.parameters():
            p.grad = None
        f = feats0.clone().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = net(dict(voxel_features=f, voxel_coords=coords, batch_size=1, input_shape=shape))["conv_point_features"]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.square().mean().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= args.warmup:
            tf += t1 - t0; tb += t2 - t1
    print(json.dumps({"what": "UNetSCN3D train-mode forward + backward (f32), 1 frame", "points": args.points, "active_voxels": V,
                      "forward_ms": 1e3 * tf / args.steps, "backward_ms": 1e3 * tb / args.steps, "steps": args.steps}))


if __name__ == "__main__":
    main()
