"""debug probe: UNetSCN3D train-mode forward+backward, HIP vs torch-f32 vs torch-f64 restatements of the sparse convolutions.
usage: train_dbg.py n_voxels scaling_ratio bn_mode(train|eval) [sim]"""
import sys, os, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lidarseg3d_amd import _lib, ops, scn_unet, synth, spconv
from oracle import ref as orc
from tests.util import golden
n, ratio, bnmode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sim = len(sys.argv) > 4
if sim:
    _lib.use_library_for_testing(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipsim", "libls3d_sim.so")); ops.set_sim(True)
DEV = "cpu" if sim else "cuda:0"
cfg = synth.NUSC; g = golden("unet_nusc_c13.npz")
coords = torch.from_numpy(g["coords"][:n]).to(DEV); feats0 = torch.from_numpy(g["voxel_features"][:n]).to(DEV)
torch.manual_seed(1)
net = scn_unet.UNetSCN3D(num_input_features=13, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"], model_cfg=dict(SCALING_RATIO=ratio), ds_factor=8, us_factor=8).to(DEV).train()
if bnmode == "eval":
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.eval()
shape = np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))

def ref(feats, w, tbl):
    kvol = tbl.shape[1]; w = w.reshape(kvol, w.shape[-2], w.shape[-1]); out = torch.zeros((tbl.shape[0], w.shape[-1]), dtype=feats.dtype, device=feats.device)
    for k in range(kvol):
        o = torch.nonzero(tbl[:, k] >= 0)[:, 0]
        if o.numel():
            out = out.index_add(0, o, feats[tbl[o, k].long()] @ w[k])
    return out

def run(model, dtype):
    for p in model.parameters():
        p.grad = None
    f = feats0.detach().to(dtype).clone().requires_grad_(True)
    out = model(dict(voxel_features=f, voxel_coords=coords, batch_size=1, input_shape=shape))["conv_point_features"]
    w = torch.linspace(-1, 1, out.numel(), dtype=dtype, device=DEV).reshape(out.shape)
    (out * w).sum().backward()
    return out.detach().double(), f.grad.double(), {k: p.grad.double() for k, p in model.named_parameters() if p.grad is not None}

net64 = copy.deepcopy(net).double()
a = run(net, torch.float32)
class RefFn:
    @staticmethod
    def apply(feats, weight, bias, rb, inverse, subm):
        y = ref(feats, weight, (rb.tbl_inv if inverse else rb.tbl)); return y if bias is None else y + bias
spconv._SparseConvFn = RefFn
b = run(net, torch.float32); c = run(net64, torch.float64)
d = lambda x, y: float((x - y).abs().max())
print("n", n, "ratio", ratio, "bn", bnmode, "sim" if sim else "gpu")
print(" out  hip %.2e torch32 %.2e  max %.3g" % (d(a[0], c[0]), d(b[0], c[0]), float(c[0].abs().max())))
print(" gin  hip %.2e torch32 %.2e  max %.3g" % (d(a[1], c[1]), d(b[1], c[1]), float(c[1].abs().max())))
worst = sorted(((d(a[2][k], c[2][k]) / (float(c[2][k].abs().max()) + 1e-30), d(b[2][k], c[2][k]) / (float(c[2][k].abs().max()) + 1e-30), k) for k in c[2]), reverse=True)[:6]
for w_ in worst:
    print("  rel hip %.2e torch32 %.2e  %s" % w_)
