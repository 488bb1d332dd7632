"""Shared body of the det3d/ops/voxel module tests (hipsim on CPU, MI355X through the same C ABI): Voxelization (voxelize.py:65-123), HardSimpleVFE /
DynamicSimpleVFE (voxel_encoder.py:13-68) and DynamicScatterWithDistance (scatter_points.py:132-213) against fixtures written by the reference's own
C++ (tests/golden/voxelize_*.npz: cpp_hard_*, cpp_dyn_coors, cpp_scatter_* come from det3d/ops/voxel/src compiled in the build container) and, for
the pooling arithmetic, against the reference's torch expressions on that padded tensor."""
import numpy as np
import torch

from oracle import ref as orc
from tests.util import golden


def run(device, tags=("nusc", "nusc_cap", "kitti")):
    from lidarseg3d_amd import voxel_ops
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    for tag in tags:
        g = golden("voxelize_%s.npz" % tag)
        vs, rng, mv = [float(v) for v in g["voxel_size"]], [float(v) for v in g["pc_range"]], int(g["max_voxels"])
        pts = T(g["points"])
        # ---- Voxelization, hard: (voxels, coors, num_points) of hard_voxelize, bit for bit (eval mode takes max_voxels[1])
        vox = voxel_ops.Voxelization(vs, rng, 5, max_voxels=(mv + 7, mv)).eval()
        v, c, n = vox(pts)
        assert np.array_equal(c.cpu().numpy(), g["cpp_hard_coors"]) and np.array_equal(n.cpu().numpy(), g["cpp_hard_num"])
        assert np.array_equal(v.cpu().numpy(), g["cpp_hard_voxels"])
        assert list(vox.grid_size) == [int(x) for x in orc.grid_size(g["voxel_size"], g["pc_range"])]
        # ---- Voxelization, dynamic (max_num_points = -1): per-point coordinates, -1 outside the range
        dyn = voxel_ops.Voxelization(vs, rng, -1)(pts)
        assert dyn.dtype == torch.int32 and np.array_equal(dyn.cpu().numpy(), g["cpp_dyn_coors"])
        # ---- HardSimpleVFE: mean of the first four features over the points of a voxel
        want = torch.from_numpy(g["cpp_hard_voxels"])[:, :, :4].sum(1) / torch.from_numpy(g["cpp_hard_num"]).float().view(-1, 1)
        got = voxel_ops.HardSimpleVFE()(v, n, c)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-6)
        # ---- DynamicSimpleVFE: mean of ALL points of a voxel, voxels in first-appearance order
        f, vc = voxel_ops.DynamicSimpleVFE(vs, rng)(pts, dyn)
        assert np.array_equal(vc.cpu().numpy(), g["cpp_scatter_coors"])
        want = torch.from_numpy(g["cpp_scatter_voxels"]).sum(1) / torch.from_numpy(g["cpp_scatter_num"]).float()[:, None]
        np.testing.assert_allclose(f.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-6)
        # ---- DynamicScatterWithDistance: column 0 = weight; the reference's expressions on its padded [V, M, 1 + C] tensor
        w = np.random.default_rng(3).uniform(0.05, 2.0, size=(g["points"].shape[0], 1)).astype(np.float32)
        wp = np.concatenate([w, g["points"]], 1)
        padded, num = orc.dynamic_scatter_padded(torch.from_numpy(wp), g["cpp_dyn_coors"], g["voxel_size"], g["pc_range"])
        dist, feat = padded[..., 0], padded[..., 1:]
        ref = dict(max=feat.max(1)[0], avg=feat.sum(1) / num.view(-1, 1),
                   weighted_avg=(feat * (dist / (dist.sum(1, keepdim=True) + 1e-8)).unsqueeze(-1)).sum(1))
        for method, want in ref.items():
            f, vc = voxel_ops.DynamicScatterWithDistance(vs, rng, method)(T(wp), dyn)
            assert np.array_equal(vc.cpu().numpy(), g["cpp_scatter_coors"]), method
            if method == "max":
                np.testing.assert_array_equal(f.cpu().numpy(), want.numpy())
            else:
                np.testing.assert_allclose(f.cpu().numpy(), want.numpy(), rtol=3e-6, atol=2e-6, err_msg=method)
    # batched coordinates (batch, z, y, x): two frames in one call == frame by frame, concatenated (scatter_points.py:195-209)
    g = golden("voxelize_nusc.npz")
    vs, rng = [float(v) for v in g["voxel_size"]], [float(v) for v in g["pc_range"]]
    dyn = g["cpp_dyn_coors"]
    half = dyn.shape[0] // 2
    w = np.random.default_rng(4).uniform(0.05, 2.0, size=(dyn.shape[0], 1)).astype(np.float32)
    wp = np.concatenate([w, g["points"]], 1)
    bc = np.concatenate([np.concatenate([np.zeros((half, 1), np.int32), dyn[:half]], 1), np.concatenate([np.ones((dyn.shape[0] - half, 1), np.int32), dyn[half:]], 1)])
    bc[dyn[:, 0] < 0] = -1
    m = voxel_ops.DynamicScatterWithDistance(vs, rng, "weighted_avg")
    fb, cb = m(T(wp), T(bc))
    f0, c0 = m(T(wp[:half]), T(dyn[:half]))
    f1, c1 = m(T(wp[half:]), T(dyn[half:]))
    assert torch.equal(fb.cpu(), torch.cat([f0, f1]).cpu())
    assert np.array_equal(cb.cpu().numpy()[:, 1:], torch.cat([c0, c1]).cpu().numpy()) and np.array_equal(cb.cpu().numpy()[:, 0], [0] * len(c0) + [1] * len(c1))
