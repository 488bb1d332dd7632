import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lidarseg3d_amd as L
from lidarseg3d_amd import ops, synth
DEV = torch.device("cuda:0")
n, grid, rng_ = 120000, [480, 360, 32], [0.0, -np.pi, -4.0, 50.0, np.pi, 2.0]
f = synth.lidar_frame(n, seed=52, **synth.NUSC)
pts = torch.from_numpy(np.concatenate([np.zeros((n, 1), np.float32), f], 1)).to(DEV)
rd = L.build_from_cfg(dict(type="Cylinder3DDynamicVoxelFeatureExtractor", grid_size=grid, point_cloud_range=rng_, average_points=False, num_input_features=5, num_output_features=64, fea_compre=16), L.READERS)
bb = L.build_from_cfg(dict(type="Cylinder3D_Asymm_3d_spconv_v2p", num_input_features=16, grid_size=grid, point_cloud_range=rng_, model_cfg=dict(init_size=16)), L.BACKBONES)
for m, seed in ((rd, 3), (bb, 4)):
    m.load_state_dict({k: torch.from_numpy(a) for k, a in synth.random_state_dict({k: tuple(t.shape) for k, t in m.state_dict().items()}, seed).items()})
    m.to(DEV).eval()
res = {}
for prec in ("f32", "bf16x6", "bf16x8", "bf16x3"):
    ops.set_precision(prec)
    with torch.no_grad():
        rdo = rd(dict(points=pts, batch_size=1))
        res[prec + "_reader"] = rdo["voxel_features"].clone()
        o = bb(rdo)
    res[prec] = o["conv_point_features"].clone()
sc = float(res["f32"].abs().max())
print("scale", sc, "reader scale", float(res["f32_reader"].abs().max()))
for a, b in (("bf16x6", "f32"), ("bf16x8", "f32"), ("bf16x6", "bf16x8"), ("bf16x3", "f32")):
    print(a, "vs", b, float((res[a] - res[b]).abs().max()) / sc, " reader:", float((res[a + "_reader"] - res[b + "_reader"]).abs().max()) / float(res["f32_reader"].abs().max()))
# rows: per-row relative
d = (res["bf16x6"] - res["f32"]).abs().max(1)[0] / res["f32"].abs().max(1)[0].clamp_min(1e-30)
print("per-row rel: median %.2e p99 %.2e max %.2e" % (float(d.median()), float(d.quantile(0.99)), float(d.max())))
