"""Module-level cases of the 'other backbones + dynamic readers' row, shared by the hipsim and the GPU suites:
each module is built through lidarseg3d_amd's registry from the cfg dict the reference's config would carry, loads the seeded
state_dict STRICT (same keys / shapes as the reference module: tests/golden/manifests.json), runs one forward on `device` and is compared
with the fixture tests/golden/make_golden_f4.py --modules wrote from the reference's own file: output sites / cell rows / counts / labels
bit-exact, features within `TOL` of the fixture's largest magnitude."""
import numpy as np
import torch

import lidarseg3d_amd as L
from tests.util import golden, seeded_sd

TOL = 1e-3   # the bar of north_star, relative to max|reference| (the seeded networks are O(1) - O(10^3) at their outputs)
MEASURED = {}  # name -> worst relative error seen (the GPU test prints it)


def _close(name, got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)) if want.size else 0.0
    MEASURED[name] = max(MEASURED.get(name, 0.0), err)
    assert err <= TOL, (name, err)
    return err


def _build(registry, cfg, key, seed, device):
    m = L.build_from_cfg(cfg, registry)
    m.load_state_dict(seeded_sd(key, int(seed)), strict=True)
    return m.to(device).eval()


def _dense_from(g, prefix="dense"):
    out = np.zeros(tuple(int(v) for v in g[prefix + "_shape"]), np.float32)
    nz = g[prefix + "_nonzero"]
    out[tuple(nz[:, i] for i in range(nz.shape[1]))] = g[prefix + "_values"]
    return out


def spmiddleresnetfhd(device):
    g = golden("f4_mod_spmiddleresnetfhd.npz")
    net = _build(L.BACKBONES, dict(type="SpMiddleResNetFHD", num_input_features=16, ds_factor=8), "backbone.SpMiddleResNetFHD", g["seed"], device)
    with torch.no_grad():
        ret, scales = net(torch.from_numpy(g["feats"]).to(device), torch.from_numpy(g["coords"]).to(device), 2, g["input_shape"])
    for k in ("conv1", "conv2", "conv3", "conv4"):
        t = scales[k]
        assert np.array_equal(t.indices.cpu().numpy(), g[k + "_idx"]), (k, "output sites")
        assert [int(v) for v in t.spatial_shape] == [int(v) for v in g[k + "_shape"]]
        _close("SpMiddleResNetFHD." + k, t.features.cpu().numpy(), g[k + "_feats"])
    want = _dense_from(g)
    got = ret.cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got != 0, want != 0) or np.abs(got[(got != 0) != (want != 0)]).max() < 1e-6  # the ReLU's zeros: same sites
    _close("SpMiddleResNetFHD.dense", got, want)


def _cyl_batch(g, device):
    return dict(voxel_features=torch.from_numpy(g["feats"]).to(device), voxel_coords=torch.from_numpy(g["coords"]).to(device), batch_size=2,
                input_shape=g["grid"])


def unetcylinder3d(device):
    g = golden("f4_mod_unetcylinder3d.npz")
    net = _build(L.BACKBONES, dict(type="UNetCylinder3D", num_input_features=16, voxel_size=[float(v) for v in g["voxel_size"]],
                                   point_cloud_range=[float(v) for v in g["pc_range"]], model_cfg=dict(init_size=8)),
                 "backbone.UNetCylinder3D", g["seed"], device)
    with torch.no_grad():
        o = net(_cyl_batch(g, device))
    _close("UNetCylinder3D.conv_point_features", o["conv_point_features"].cpu().numpy(), g["conv_point_features"])
    assert np.array_equal(o["conv_point_coords"].cpu().numpy(), g["conv_point_coords"])  # batch index + cell centres: f32 bit-exact


def cylinder3d_v2p(device):
    g = golden("f4_mod_cylinder3d_v2p.npz")
    net = _build(L.BACKBONES, dict(type="Cylinder3D_Asymm_3d_spconv_v2p", num_input_features=16, grid_size=[int(v) for v in g["grid"]],
                                   point_cloud_range=[float(v) for v in g["pc_range"]], model_cfg=dict(init_size=8)),
                 "backbone.Cylinder3D_Asymm_3d_spconv_v2p", g["seed"], device)
    with torch.no_grad():
        o = net(_cyl_batch(g, device))
    _close("Cylinder3D_v2p.conv_point_features", o["conv_point_features"].cpu().numpy(), g["conv_point_features"])
    # rho * cos(phi), rho * sin(phi): the device's cos / sin differ from glibc's by an ulp
    np.testing.assert_allclose(o["conv_point_coords"].cpu().numpy(), g["conv_point_coords"], rtol=0, atol=2e-5)


def cylinder3d_asymm(device):
    g = golden("f4_mod_cylinder3d_asymm.npz")
    grid = [int(v) for v in g["grid"]]
    net = _build(L.BACKBONES, dict(type="Cylinder3D_Asymm_3d_spconv", output_shape=[grid[0], grid[1], grid[2] + 1], num_input_features=16, nclasses=7,
                                   init_size=8), "backbone.Cylinder3D_Asymm_3d_spconv", g["seed"], device)
    with torch.no_grad():
        o = net(_cyl_batch(g, device))
    _close("Cylinder3D_Asymm_3d_spconv.logits", o["voxel_features"].cpu().numpy(), _dense_from(g))


def _reader(kind, tag, avg, device):
    g = golden("f4_mod_reader_%s.npz" % tag)
    rd = _build(L.READERS, dict(type=kind, grid_size=[int(v) for v in g["grid"]], point_cloud_range=[float(v) for v in g["pc_range"]], average_points=avg,
                                num_input_features=5, num_output_features=64, fea_compre=16, voxel_label_enc="major"), "reader." + kind, g["seed"], device)
    with torch.no_grad():
        o = rd(dict(points=torch.from_numpy(g["points"]).to(device), batch_size=2, point_sem_labels=torch.from_numpy(g["labels"]).to(device)))
    assert np.array_equal(o["point_vcoors"].cpu().numpy(), g["point_vcoors"]), "cell of every point"
    assert np.array_equal(o["num_points_in_voxel"].cpu().numpy(), g["num_points_in_voxel"])
    assert np.array_equal(o["voxel_sem_labels"].cpu().numpy(), g["voxel_sem_labels"]), "majority labels"
    assert [int(v) for v in o["input_shape"]] == [int(v) for v in g["grid"]]
    return g, o


def reader_cylinder3d(device):
    g, o = _reader("Cylinder3DDynamicVoxelFeatureExtractor", "cylinder3d", False, device)
    assert np.array_equal(o["voxel_coords"].cpu().numpy(), g["voxel_coords"]), "torch.unique's rows"
    _close("Cylinder3DDynamicVFE.voxel_features", o["voxel_features"].cpu().numpy(), g["voxel_features"])


def reader_polarnet(device):
    g, o = _reader("PolarNetDynamicVoxelFeatureExtractor", "polarnet", True, device)
    _close("PolarNetDynamicVFE.bev", o["voxel_features"].cpu().numpy(), _dense_from(g, "bev"))


CASES = [spmiddleresnetfhd, unetcylinder3d, cylinder3d_v2p, cylinder3d_asymm, reader_cylinder3d, reader_polarnet]


def tta_merge(device):
    """point_head.predict with tta_flag on `device` against the labels the REFERENCE head's predict() gave for the same logits
    (tests/golden/tta_merge.npz; 2 samples x 4 variants, one row of exact ties)"""
    from lidarseg3d_amd import point_heads
    g = golden("tta_merge.npz")
    head = point_heads.PointSegBatchlossHead(False, 17, dict(CONV_IN_DIM=32, CONV_CLS_FC=[64], CONV_ALIGN_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0))
    head.forward_ret_dict["out_logits"] = torch.from_numpy(g["logits"]).to(device)
    ex = dict(points=torch.from_numpy(g["points"]).to(device), num_voxels=torch.zeros(8), metadata=[dict(token="f%d" % b) for b in range(8)])
    out = head.predict(example=ex, test_cfg=dict(tta_flag=True, merge_type="ArithmeticMean", num_tta_tranforms=4))
    assert [o["metadata"]["token"] for o in out] == ["f0", "f4"]
    for o, key in zip(out, ("labels0", "labels1")):
        got, want = o["pred_point_sem_labels"].cpu().numpy(), g[key]
        assert got.dtype == want.dtype and got.shape == want.shape
        # the merged probabilities of two classes can differ by less than the ulp two exp() implementations disagree on: allow 1 in 10^4
        assert (got != want).mean() <= 1e-4, (key, int((got != want).sum()))
    assert int(out[0]["pred_point_sem_labels"][5]) == 0  # the row of exact ties: first class


def dynamic_point_to_voxel(device, tag="nusc"):
    """ls3d_dynamic_point_to_voxel_index / _forward / _backward against the reference's own C++ (cpp_scatter_* of voxelize_*.npz come from
    det3d/ops/voxel/src/scatter_points_cpu.cpp compiled in the build container): the padded [V, M, C] tensor, coordinates and counts
    bit-exact; backward = the gather map_voxel_to_point_kernel does (scatter_points_cuda.cu:50-69)"""
    from lidarseg3d_amd import ops
    from oracle import ref as orc
    g = golden("voxelize_%s.npz" % tag)
    gs = orc.grid_size(g["voxel_size"], g["pc_range"])
    pts, coors = torch.from_numpy(g["points"]).to(device), torch.from_numpy(g["cpp_dyn_coors"]).to(device)
    p2v, c2v, num, vc, counts = ops.dynamic_point_to_voxel_index(coors, [int(gs[2]), int(gs[1]), int(gs[0])])
    V, M = (int(v) for v in counts.tolist())
    want = g["cpp_scatter_voxels"]
    assert (V, M) == want.shape[:2]
    assert np.array_equal(vc[:V].cpu().numpy(), g["cpp_scatter_coors"]) and np.array_equal(num[:V].cpu().numpy(), g["cpp_scatter_num"])
    vox = ops.dynamic_point_to_voxel_forward(pts, p2v, c2v, V, M)
    assert np.array_equal(vox.cpu().numpy(), want)
    inside = (g["cpp_dyn_coors"][:, 0] >= 0)
    assert np.array_equal(c2v.cpu().numpy() >= 0, inside) and np.array_equal(p2v.cpu().numpy() >= 0, inside)
    gv = torch.randn(vox.shape, generator=torch.Generator().manual_seed(3))
    gp = ops.dynamic_point_to_voxel_backward(torch.zeros(pts.shape, device=device), gv.to(device), p2v, c2v).cpu().numpy()
    ref = np.zeros(gp.shape, np.float32)
    ci, pi = c2v.cpu().numpy()[inside], p2v.cpu().numpy()[inside]
    ref[inside] = gv.numpy()[ci, pi]
    assert np.array_equal(gp, ref)
    # the points of a voxel fill its slots in point order
    assert np.array_equal(vox.cpu().numpy()[ci, pi], g["points"][inside])
