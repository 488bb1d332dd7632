"""Replay of the sparse-convolution calls the reference's Cylinder3D_Asymm_3d_spconv / SpMiddleResNetFHD issue (fixtures written by
tests/golden/make_golden_f4.py from the reference's own files over the oracle shim) on lidarseg3d_amd.spconv, in the recorded order
with one shared indice_dict - hipsim on CPU and the MI355X through the same code."""
import numpy as np
import torch

from tests.util import golden


def conv_weight(seed, shape):
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return (rng.normal(size=shape) * (2.0 / (shape[3] * 9)) ** 0.5).astype(np.float32)


def replay(name, device, max_calls=None):
    from lidarseg3d_amd import spconv
    g = golden(name)
    n, batch = int(g["n_calls"]), int(g["batch"])
    indice_dict = {}
    seen = set()
    worst = 0.0
    for i in range(n if max_calls is None else min(n, max_calls)):
        p = "c%02d_" % i
        meta = g[p + "meta"].tolist()
        kind, ks, st, pd = meta[0], tuple(meta[1:4]), tuple(meta[4:7]), tuple(meta[7:10])
        key = bytes(g[p + "key"]).decode()
        key = None if key == "None" else key  # layers without an indice_key do not share rulebooks
        ws = g[p + "wseed"].tolist()
        w = conv_weight(ws[0], tuple(ws[1:]))
        cin, cout = w.shape[3], w.shape[4]
        bias = g[p + "bias"] if (p + "bias") in g.files else None
        if kind == 0:
            m = spconv.SubMConv3d(cin, cout, ks, bias=bias is not None, indice_key=key)
        elif kind == 1:
            m = spconv.SparseConv3d(cin, cout, ks, stride=st, padding=pd, bias=bias is not None, indice_key=key)
        else:
            m = spconv.SparseInverseConv3d(cin, cout, ks, indice_key=key, bias=bias is not None)
        m = m.to(device).eval()
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w))
            if bias is not None:
                m.bias.copy_(torch.from_numpy(bias))
            x = spconv.SparseConvTensor(torch.from_numpy(g[p + "in_feats"]).to(device), torch.from_numpy(g[p + "in_idx"]).to(device),
                                        [int(v) for v in g[p + "in_shape"]], batch)
            x.indice_dict = indice_dict  # spconv: every tensor derived from the network input shares it
            y = m(x)
        want = g[p + "out_feats"]
        assert np.array_equal(y.indices.cpu().numpy(), g[p + "out_idx"]), (name, i, "output sites")
        assert [int(v) for v in y.spatial_shape] == [int(v) for v in g[p + "out_shape"]]
        err = float(np.abs(y.features.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-6))
        worst = max(worst, err)
        assert err <= 3e-6, (name, i, kind, ks, st, key, err)
        seen.add((kind, ks, st))
    return seen, worst
