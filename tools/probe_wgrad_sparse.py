#!/usr/bin/env python
"""sparse weight gradient of the level-2 / level-3 SubM layers of a 2 x 180k Waymo batch: exact-f32 kernel vs the bf16-plane kernel (forced)"""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lidarseg3d_amd import ops, synth
dev = "cuda:0"
cfg = synth.WAYMO
frames = [synth.lidar_frame(180000, seed=b, **cfg) for b in range(2)]
pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)])).to(dev)
v, c, n, nv = ops.voxelize_hard(pts, cfg["voxel_size"], cfg["pc_range"], 5, 600000, batched=True)
V = int(nv)
coords = c[:V].contiguous()
shape = [int(s) for s in np.asarray(ops.make_grid(cfg["voxel_size"], cfg["pc_range"])[1])[::-1]]
shape[0] += 1
for lvl, ch in ((2, 64), (3, 128), (1, 32)):
    cc, sh = coords, shape
    for _ in range(lvl - 1):
        oc, cnt, nbr_out, nbr_inv, osh = ops.rulebook_conv(cc, 2, sh, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        cc, sh = oc[:int(cnt[0])].contiguous(), osh
    tbl = ops.rulebook_subm(cc, sh, (3, 3, 3))
    order = ops.rulebook_order(tbl, cc)
    pairs = ops.spconv_pairs(tbl, order)
    rows = tbl.shape[0]
    x = torch.randn(rows, ch, device=dev).relu_()
    g = torch.randn(rows, ch, device=dev) * 0.1
    rec = dict(level=lvl, rows=rows, ch=ch, pairs=int((tbl >= 0).sum()))
    for name, prod in (("f32", 0), ("planes_rule", 6), ("planes_forced", 6 | 64)):
        for _ in range(2):
            r = ops.spconv_wgrad(x, g, tbl, order, ch, ch, products=prod, pairs=pairs)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            r = ops.spconv_wgrad(x, g, tbl, order, ch, ch, products=prod, pairs=pairs)
        b.record(); torch.cuda.synchronize()
        rec[name + "_ms"] = a.elapsed_time(b) / 5
        if name == "f32":
            ref = r
        else:
            rec[name + "_rel_vs_f32"] = float((r - ref).norm() / ref.norm())
    print(json.dumps(rec), flush=True)
