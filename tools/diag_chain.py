#!/usr/bin/env python
"""diagnostic (GPU): the chained launch in cost order (LS3D_CHAIN_ABLATE=8) against layer-by-layer launches - which rows of which level differ"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarseg3d_amd as L
from lidarseg3d_amd import models_cfg, ops, synth, detectors

dev = torch.device("cuda:0")
cfg = synth.NUSC
model = L.build_detector(models_cfg.sdseg3d(), train_cfg=None, test_cfg={}).eval()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.random_state_dict(shapes, 5).items()})
model.to(dev)
f = synth.lidar_frame(120000, seed=100, **cfg)
pts = torch.from_numpy(np.concatenate([np.zeros((f.shape[0], 1), np.float32), f], 1)).to(dev)
ops.set_precision("bf16x6")
detectors.CAPACITY_MODE = False


def run():
    with torch.no_grad():
        data = model.forward_features(dict(points=pts, batch_size=1))
    ms = data["multi_scale_3d_features"]
    return dict(x_conv2=ms["x_conv2"].features.clone(), x_conv3=ms["x_conv3"].features.clone(), out=data["conv_point_features"].clone())


ops.set_tile_chain(False)
ref = run()
for ab in [int(a) for a in (sys.argv[1:] or ["0", "8"])]:
    ops.set_tile_chain(True)
    ops._CHAIN_ABLATE = ab
    st = ops.collect_chain_states(True)
    for rep in range(3):
        got = run()
        torch.cuda.synchronize()
        msg = []
        for k in ref:
            bad = (got[k] != ref[k]).any(1)
            d = (got[k] - ref[k]).abs().max().item()
            msg.append("%s: %d/%d rows differ, max |d| %.3g (scale %.3g)" % (k, int(bad.sum()), bad.numel(), d, ref[k].abs().max().item()))
        print("ablate %d rep %d | %s | watchdog %s" % (ab, rep, " | ".join(msg), [int(s[1]) for s in st][-2:]))
    ops.collect_chain_states(False)
