"""The oracle's spconv restatement (oracle/ref.py: subm_rulebook, conv_rulebook, spconv_fwd) and the product's spconv modules
(lidarseg3d_amd.spconv on tests/hipsim) against DENSE convolutions of the densified grids (tests/dense_cases.py): kernel-offset
order, (kD,kH,kW,Cin,Cout) filter indexing, strided output sites and their order, inverse-conv pairing, asymmetric kernels and
strides.  This is the independent pin of the conv primitives (the third-party spconv source is absent from the reference tree)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref as orc
from tests import dense_cases as dc

HERE = os.path.dirname(os.path.abspath(__file__))
CIN, COUT, BATCH = 5, 7, 2


@pytest.mark.parametrize("name,ks,st,pd,grid", dc.CASES)
def test_oracle_spconv_equals_dense_convolution(name, ks, st, pd, grid):
    coords, feats = dc.random_sparse(grid, BATCH, cin=CIN, seed=len(name))
    w = dc.weight(ks, CIN, COUT, seed=3)
    if name.endswith("subm"):
        nbr = orc.subm_rulebook(coords, grid, ks)
        got = orc.spconv_fwd(torch.from_numpy(feats).double(), torch.from_numpy(w).double(), nbr).numpy()
        np.testing.assert_allclose(got, dc.dense_subm(coords, feats, w, grid, BATCH), rtol=0, atol=1e-10)
        return
    oc, oshape, nbr = orc.conv_rulebook(coords, grid, ks, st, pd)
    want_oc, want_f, want_shape = dc.dense_conv(coords, feats, w, st, pd, grid, BATCH)
    assert tuple(oshape) == want_shape
    np.testing.assert_array_equal(oc, want_oc)  # the same sites in the same (ascending linear index) order
    got = orc.spconv_fwd(torch.from_numpy(feats).double(), torch.from_numpy(w).double(), nbr).numpy()
    np.testing.assert_allclose(got, want_f, rtol=0, atol=1e-10)
    # SparseInverseConv3d on the forward conv's pairs: back onto the input sites
    w_inv = dc.weight(ks, COUT, CIN + 1, seed=4)
    back = orc.spconv_fwd(torch.from_numpy(got), torch.from_numpy(w_inv).double(), nbr, inverse=True, n_out=coords.shape[0]).numpy()
    want_back = dc.dense_inverse(oc, got.astype(np.float32).astype(np.float64).astype(np.float32), w_inv, st, pd, oshape, coords, grid, BATCH)
    np.testing.assert_allclose(back, want_back, rtol=0, atol=1e-5)


@pytest.fixture(scope="module")
def sim():
    from lidarseg3d_amd import _lib, ops
    sys.path.insert(0, os.path.join(HERE, "hipsim"))
    import build_sim
    _lib.use_library_for_testing(build_sim.build())
    ops.set_sim(True)
    yield
    ops.set_sim(False)
    _lib.use_library_for_testing(None)


def run_modules(name, ks, st, pd, grid, device):
    """the product's SubMConv3d / SparseConv3d / SparseInverseConv3d on one case; -> max deviation from dense / max |dense|"""
    from lidarseg3d_amd import spconv
    cin, cout = 16, 32  # the kernels' channel granularity
    coords, feats = dc.random_sparse(grid, BATCH, cin=cin, seed=len(name) + 1)
    w = dc.weight(ks, cin, cout, seed=5)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(device), torch.from_numpy(coords).to(device), list(grid), BATCH)
    out = {}
    with torch.no_grad():
        if name.endswith("subm"):
            m = spconv.SubMConv3d(cin, cout, ks, bias=False, indice_key="a").to(device)
            m.weight.copy_(torch.from_numpy(w))
            y = m(x)
            want = dc.dense_subm(coords, feats, w, grid, BATCH)
            out["subm"] = float(np.abs(y.features.cpu().numpy() - want).max() / np.abs(want).max())
            return out
        m = spconv.SparseConv3d(cin, cout, ks, stride=st, padding=pd, bias=False, indice_key="d").to(device)
        m.weight.copy_(torch.from_numpy(w))
        y = m(x)
        want_oc, want_f, want_shape = dc.dense_conv(coords, feats, w, st, pd, grid, BATCH)
        assert tuple(y.spatial_shape) == want_shape
        assert np.array_equal(y.indices.cpu().numpy(), want_oc)
        yf = y.features.cpu().numpy()
        out["conv"] = float(np.abs(yf - want_f).max() / np.abs(want_f).max())
        inv = spconv.SparseInverseConv3d(cout, 16, ks, indice_key="d", bias=False).to(device)
        w_inv = dc.weight(ks, cout, 16, seed=6)
        inv.weight.copy_(torch.from_numpy(w_inv))
        z = inv(y)
        assert np.array_equal(z.indices.cpu().numpy(), coords)
        want_back = dc.dense_inverse(want_oc, yf, w_inv, st, pd, want_shape, coords, grid, BATCH)
        out["inverse"] = float(np.abs(z.features.cpu().numpy() - want_back).max() / np.abs(want_back).max())
    return out


@pytest.mark.parametrize("name,ks,st,pd,grid", dc.CASES)
def test_spconv_modules_equal_dense_convolution_hipsim(sim, name, ks, st, pd, grid):
    for what, err in run_modules(name, ks, st, pd, grid, "cpu").items():
        assert err <= 2e-6, (name, what, err)  # f32 summation order only


def test_spconv_handcomputed_case():
    """a case small enough to check by hand: two active sites, one kernel tap each way.
    sites a = (0,0,0,0), b = (0,0,0,1) (x neighbours), Cin = Cout = 1, 3x3x3 weight with W[kz,ky,kx] = 100 kz + 10 ky + kx.
    SubM: out[a] = W[1,1,1] f(a) + W[1,1,2] f(b) (b sits at offset +1 in x: kx = 2); out[b] = W[1,1,1] f(b) + W[1,1,0] f(a)."""
    coords = np.array([[0, 0, 0, 0], [0, 0, 0, 1]], np.int32)
    f = np.array([[2.0], [3.0]], np.float32)
    w = np.zeros((3, 3, 3, 1, 1), np.float32)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                w[kz, ky, kx, 0, 0] = 100 * kz + 10 * ky + kx
    nbr = orc.subm_rulebook(coords, (1, 1, 2), 3)
    got = orc.spconv_fwd(torch.from_numpy(f), torch.from_numpy(w), nbr).numpy()
    np.testing.assert_array_equal(got[:, 0], [111 * 2 + 112 * 3, 111 * 3 + 110 * 2])
    # strided: k = 2, s = 2 over a 1x1x2 grid -> one output at (0,0,0) = W[0,0,0] f(a) + W[0,0,1] f(b); inverse hands it back
    # to a through W'[0,0,0] and to b through W'[0,0,1]
    oc, oshape, nb = orc.conv_rulebook(coords, (1, 1, 2), (1, 1, 2), (1, 1, 2), 0)
    w2 = np.array([5.0, 7.0], np.float32).reshape(1, 1, 2, 1, 1)
    y = orc.spconv_fwd(torch.from_numpy(f), torch.from_numpy(w2), nb).numpy()
    assert oc.tolist() == [[0, 0, 0, 0]] and y[0, 0] == 5 * 2 + 7 * 3
    back = orc.spconv_fwd(torch.from_numpy(y), torch.from_numpy(w2), nb, inverse=True, n_out=2).numpy()
    np.testing.assert_array_equal(back[:, 0], [5 * 31, 7 * 31])
