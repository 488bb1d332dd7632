"""Shared body of the pointnet2_utils drop-in tests (hipsim on CPU, MI355X through the same C ABI): the autograd Functions that
replace det3d/ops/pointnet2_batch/pointnet2_utils.py:76-153 against the C oracle (oracle/c/ls3d_oracle.c, a restatement of
interpolate_gpu.cu:16-149)."""
import numpy as np
import torch

from oracle import ref as orc


def run(device, b=2, n=300, m=70, c=11, seed=0):
    from lidarseg3d_amd import ops, pointnet2_utils as pu
    rng = np.random.default_rng(seed)
    unknown = rng.uniform(-5, 5, size=(b, n, 3)).astype(np.float32)
    known = rng.uniform(-5, 5, size=(b, m, 3)).astype(np.float32)
    known[0, 5] = known[0, 3]  # a duplicated centre: the lower index must win the tie
    unknown[b - 1, 7] = known[b - 1, 2]  # a query on top of a centre: distance 0
    feats = rng.normal(size=(b, c, m)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(device)
    dist, idx = pu.three_nn(T(unknown), T(known))
    for i in range(b):
        d2, ix = orc.three_nn(unknown[i], known[i])
        assert np.array_equal(idx[i].cpu().numpy(), ix)                       # bit-exact indices
        # ... and distances: ThreeNN returns torch.sqrt of the kernel's squared distances, as pointnet2_utils.py:95 does
        assert torch.equal(dist[i].cpu(), torch.sqrt(torch.from_numpy(d2).to(device)).cpu())
    recip = 1.0 / (dist + 1e-8)
    weight = (recip / recip.sum(dim=2, keepdim=True)).contiguous()            # point_utils.py:29-31
    f = T(feats).requires_grad_(True)
    out = pu.three_interpolate(f, idx, weight)
    w_np, i_np = weight.cpu().numpy(), idx.cpu().numpy()
    for i in range(b):
        want = orc.three_interpolate_cm(feats[i], i_np[i], w_np[i])
        assert np.array_equal(out[i].detach().cpu().numpy(), want)            # fma(w2,f2, fma(w1,f1, w0*f0)): bit-exact
    gout = rng.normal(size=(b, c, n)).astype(np.float32)
    out.backward(T(gout))
    for i in range(b):
        want = orc.three_interpolate_grad_cm(gout[i], i_np[i], w_np[i], m)
        got = f.grad[i].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)               # atomic / point-order f32 sums
    # the raw C-ABI entry points too (what INTEGRATION.md binds), incl. m < 3 and empty batches
    d2s, ixs = ops.three_nn(T(np.ascontiguousarray(unknown[:, :4])), T(np.ascontiguousarray(known[:, :2])))
    d2o, ixo = orc.three_nn(np.ascontiguousarray(unknown[0, :4]), np.ascontiguousarray(known[0, :2]))
    assert np.array_equal(ixs[0].cpu().numpy(), ixo) and np.array_equal(d2s[0].cpu().numpy(), d2o)
    g = ops.three_interpolate_grad(T(gout), idx, weight, m)
    assert torch.allclose(g, f.grad, rtol=0, atol=2e-5)
    # torch autograd of the same interpolation (independent of the oracle)
    f2 = T(feats).requires_grad_(True)
    gathered = torch.stack([f2[i][:, idx[i].long()] for i in range(b)])       # (B, C, n, 3)
    (gathered * weight[:, None]).sum(-1).backward(T(gout))
    assert torch.allclose(f2.grad, f.grad, rtol=0, atol=2e-5)
