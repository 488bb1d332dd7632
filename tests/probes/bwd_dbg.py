"""debug probe: per-layer backward error of the HIP path vs a float64 torch restatement (and torch f32 as the yardstick)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lidarseg3d_amd import ops, spconv
DEV = "cuda:0"

def ref(feats, w, tbl):
    kvol = tbl.shape[1]; w = w.reshape(kvol, w.shape[-2], w.shape[-1])
    out = torch.zeros((tbl.shape[0], w.shape[-1]), dtype=feats.dtype, device=feats.device)
    for k in range(kvol):
        o = torch.nonzero(tbl[:, k] >= 0)[:, 0]
        if o.numel():
            out = out.index_add(0, o, feats[tbl[o, k].long()] @ w[k])
    return out

for nsite, cin, cout in ((4000, 64, 128), (30000, 64, 128), (30000, 128, 128), (4000, 128, 64)):
    rng = np.random.default_rng(1)
    shape = [21, 200, 200]
    cells = rng.choice(shape[0] * shape[1] * shape[2], size=nsite, replace=False)
    coords = np.stack([np.zeros_like(cells), cells // 40000, (cells // 200) % 200, cells % 200], 1).astype(np.int32)
    coords = coords[np.lexsort((coords[:, 3], coords[:, 2], coords[:, 1]))]
    f0 = torch.from_numpy(rng.normal(size=(nsite, cin)).astype(np.float32)).to(DEV)
    torch.manual_seed(0)
    c1 = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="s1").to(DEV).train()
    c2 = spconv.SparseConv3d(cout, cout, 3, stride=2, padding=1, bias=False, indice_key="d1").to(DEV).train()
    c3 = spconv.SparseInverseConv3d(cout, cin, 3, indice_key="d1", bias=False).to(DEV).train()
    feats = f0.clone().requires_grad_(True)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(coords).to(DEV), shape, 1)
    y1 = c1(x); y2 = c2(y1); y3 = c3(y2)
    r = torch.from_numpy(rng.normal(size=tuple(y3.features.shape)).astype(np.float32)).to(DEV)
    (y3.features * r).sum().backward()
    got = dict(y1=y1.features.detach(), y2=y2.features.detach(), y3=y3.features.detach(), gin=feats.grad, gw1=c1.weight.grad, gw2=c2.weight.grad, gw3=c3.weight.grad)
    rb1, rb2 = x.find_indice_pair("s1"), x.find_indice_pair("d1")
    res = {}
    for dt in (torch.float32, torch.float64):
        f2 = f0.to(dt).clone().requires_grad_(True)
        w1, w2, w3 = (t.detach().to(dt).clone().requires_grad_(True) for t in (c1.weight, c2.weight, c3.weight))
        z1 = ref(f2, w1, rb1.tbl); z2 = ref(z1, w2, rb2.tbl); z3 = ref(z2, w3, rb2.tbl_inv)
        (z3 * r.to(dt)).sum().backward()
        res[dt] = dict(y1=z1.detach(), y2=z2.detach(), y3=z3.detach(), gin=f2.grad, gw1=w1.grad, gw2=w2.grad, gw3=w3.grad)
    print("sites", nsite, "cin", cin, "cout", cout)
    for k in got:
        c = res[torch.float64][k]
        m = float(c.abs().max())
        print("  %-4s rel err hip %.2e  torch32 %.2e" % (k, float((got[k].double() - c).abs().max()) / m, float((res[torch.float32][k].double() - c).abs().max()) / m))
