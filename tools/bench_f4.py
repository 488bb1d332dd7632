#!/usr/bin/env python
"""Per-module forward times of the other model families on one 120k-point sweep, eager, bf16x6 (f32-grade) and exact f32:
the two dynamic cylindrical readers, Cylinder3D_Asymm_3d_spconv_v2p / UNetCylinder3D on the Cylinder3D reader's voxels, SpMiddleResNetFHD on the
hard-voxelized nuScenes grid.   python tools/bench_f4.py [--iters 10]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import lidarseg3d_amd as L
from lidarseg3d_amd import ops, synth


def seeded(m, seed, dev):
    m.load_state_dict({k: torch.from_numpy(a) for k, a in synth.random_state_dict({k: tuple(t.shape) for k, t in m.state_dict().items()}, seed).items()})
    return m.to(dev).eval()


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, grid, rng_ = 120000, [480, 360, 32], [0.0, -np.pi, -4.0, 50.0, np.pi, 2.0]
    f = synth.lidar_frame(n, seed=52, **synth.NUSC)
    pts = torch.from_numpy(np.concatenate([np.zeros((n, 1), np.float32), f], 1)).to(dev)
    rkw = dict(grid_size=grid, point_cloud_range=rng_, num_input_features=5, num_output_features=64, fea_compre=16)
    rd_c = seeded(L.build_from_cfg(dict(type="Cylinder3DDynamicVoxelFeatureExtractor", average_points=False, **rkw), L.READERS), 3, dev)
    rd_p = seeded(L.build_from_cfg(dict(type="PolarNetDynamicVoxelFeatureExtractor", average_points=True, **rkw), L.READERS), 5, dev)
    v2p = seeded(L.build_from_cfg(dict(type="Cylinder3D_Asymm_3d_spconv_v2p", num_input_features=16, grid_size=grid, point_cloud_range=rng_,
                                       model_cfg=dict(init_size=16)), L.BACKBONES), 4, dev)
    vs = [(rng_[3 + i] - rng_[i]) / grid[i] for i in range(3)]
    unet = seeded(L.build_from_cfg(dict(type="UNetCylinder3D", num_input_features=16, voxel_size=vs, point_cloud_range=rng_, model_cfg=dict(init_size=16)),
                                   L.BACKBONES), 6, dev)
    cfg = synth.NUSC
    v, c, npv, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 10, 120000, batched=True)
    V = int(nv)
    coords = c[:V].contiguous()
    g3 = [int(x) for x in ops.make_grid(cfg["voxel_size"], cfg["pc_range"])[1]]
    feats = torch.randn((V, 16), generator=torch.Generator().manual_seed(2)).to(dev)
    fhd = seeded(L.build_from_cfg(dict(type="SpMiddleResNetFHD", num_input_features=16, ds_factor=8), L.BACKBONES), 7, dev)
    with torch.no_grad():
        rdo = rd_c(dict(points=pts, batch_size=1))
    vox = int(rdo["voxel_coords"].shape[0])
    print("120000 points -> %d cylindrical voxels (grid %s); %d hard voxels on the nuScenes grid" % (vox, grid, V))
    for prec in ("bf16x6", "f32"):
        ops.set_precision(prec)
        with torch.no_grad():
            rows = [("Cylinder3DDynamicVoxelFeatureExtractor", lambda: rd_c(dict(points=pts, batch_size=1))),
                    ("PolarNetDynamicVoxelFeatureExtractor", lambda: rd_p(dict(points=pts, batch_size=1))),
                    ("Cylinder3D_Asymm_3d_spconv_v2p (37 convolutions, init_size 16)", lambda: v2p(dict(rdo))),
                    ("UNetCylinder3D (init_size 16)", lambda: unet(dict(rdo))),
                    ("SpMiddleResNetFHD (19 convolutions)", lambda: fhd(feats, coords, 1, g3))]
            for name, fn in rows:
                print("%-8s %-68s %8.3f ms" % (prec, name, timed(fn, a.iters)), flush=True)
    ops.set_precision("f32")


if __name__ == "__main__":
    main()
