"""The oracle (oracle/) pinned against golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from lidarseg3d_amd import synth
from oracle import ref as orc
from tests.util import golden, seeded_sd


@pytest.mark.parametrize("tag", ["nusc", "nusc_cap", "kitti"])
def test_hard_voxelize_matches_numba_and_cpp(tag):
    g = golden("voxelize_%s.npz" % tag)
    mv = int(g["max_voxels"])
    v, c, n = orc.hard_voxelize(g["points"], g["voxel_size"], g["pc_range"], 5, mv, overflow="numba")
    assert np.array_equal(c, g["numba_coors"]) and np.array_equal(n, g["numba_num"])
    assert np.array_equal(v, g["numba_voxels"])  # bit-exact copies of the input points
    v, c, n = orc.hard_voxelize(g["points"], g["voxel_size"], g["pc_range"], 5, mv, overflow="break")
    assert np.array_equal(c, g["cpp_hard_coors"]) and np.array_equal(n, g["cpp_hard_num"])
    assert np.array_equal(v, g["cpp_hard_voxels"])


@pytest.mark.parametrize("tag", ["nusc", "kitti"])
def test_dynamic_voxelize_and_scatter_match_cpp(tag):
    g = golden("voxelize_%s.npz" % tag)
    coors = orc.dynamic_voxelize(g["points"], g["voxel_size"], g["pc_range"])
    assert np.array_equal(coors, g["cpp_dyn_coors"])
    feats, vc = orc.dynamic_scatter(g["points"], coors, g["voxel_size"], g["pc_range"], average_points=True)
    assert np.array_equal(vc, g["cpp_scatter_coors"])
    sv, sn = g["cpp_scatter_voxels"], g["cpp_scatter_num"]
    want = torch.from_numpy(sv).sum(dim=1).div(torch.from_numpy(sn).float().view(-1, 1)).numpy()
    np.testing.assert_allclose(feats, want, rtol=2e-6, atol=1e-6)  # f32 summation order (torch sums dim 1 pairwise)
    fmax, _ = orc.dynamic_scatter(g["points"], coors, g["voxel_size"], g["pc_range"], average_points=False)
    np.testing.assert_array_equal(fmax, sv.max(axis=1))


def test_vfe_readers_match_reference():
    g = golden("vfe_nusc.npz")
    vx, num = torch.from_numpy(g["voxels"]), torch.from_numpy(g["num"])
    np.testing.assert_allclose(orc.mean_vfe(vx, num).numpy(), g["mean"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(orc.improved_mean_vfe(vx, num).numpy(), g["improved"], rtol=0, atol=1e-6)
    sd = seeded_sd("reader.TransformerVoxelFeatureExtractor", g["trans_seed"])
    got = orc.trans_vfe(sd, vx, num).numpy()
    np.testing.assert_allclose(got, g["trans"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("tag", ["c13", "c16"])
def test_unet_matches_reference_wiring(tag):
    g = golden("unet_nusc_%s.npz" % tag)
    sd = seeded_sd("backbone.UNetSCN3D.%s" % tag, g["seed"])
    feat, ctr, aux = orc.unet_scn3d(sd, torch.from_numpy(g["voxel_features"]), g["coords"],
                                    orc.spatial_shape(synth.NUSC["voxel_size"], synth.NUSC["pc_range"]),
                                    synth.NUSC["voxel_size"], synth.NUSC["pc_range"], return_all=True)
    np.testing.assert_allclose(feat.numpy(), g["conv_point_features"], rtol=0, atol=1e-4)
    if tag == "c13":
        np.testing.assert_array_equal(ctr.numpy(), g["conv_point_coords"])
        np.testing.assert_array_equal(aux["rb"].c4, g["x_conv4_indices"])
        np.testing.assert_array_equal(aux["rb"].c3, g["x_up4_indices"])
        np.testing.assert_array_equal(aux["rb"].c2, g["x_up3_indices"])
        np.testing.assert_array_equal(aux["rb"].c5, g["enc_indices"])
        np.testing.assert_allclose(aux["x4"].numpy(), g["x_conv4_features"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(aux["enc"].numpy(), g["enc_features"], rtol=0, atol=1e-4)


def test_batchloss_head_matches_reference():
    g = golden("head_batchloss_nusc.npz")
    sd = seeded_sd("point_head.PointSegBatchlossHead", g["seed"])
    conv_logits, out = orc.batchloss_head(sd, torch.from_numpy(g["conv_point_features"]),
                                          torch.from_numpy(g["conv_point_coords"]), torch.from_numpy(g["points"]), 1)
    np.testing.assert_allclose(out.numpy(), g["out_logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(conv_logits.numpy(), g["conv_logits"], rtol=0, atol=1e-4)


def test_mseg3d_head_matches_reference():
    g = golden("head_mseg3d_nusc.npz")
    sd = seeded_sd("point_head.PointSegMSeg3DHead", g["seed"])
    pts = torch.from_numpy(g["points"])
    h, w = (int(v) for v in g["cam_hw"])
    img, emb, cuv = synth.camera_inputs(pts.shape[0], seed=int(g["cam_seed"]), ncam=6, c_img=48, h=h, w=w, batch=2)
    vl, out = orc.mseg3d_head(sd, torch.from_numpy(g["conv_point_features"]), torch.from_numpy(g["conv_point_coords"]),
                              pts, torch.from_numpy(cuv), torch.from_numpy(img), torch.from_numpy(emb), 2)
    np.testing.assert_allclose(vl.numpy(), g["voxel_logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.numpy(), g["out_logits"], rtol=0, atol=2e-4)


def test_camera_sfam_vs_reference():
    """oracle restatement of CameraSemanticFeatureAggregationModule vs the reference class's output"""
    g = golden("camera_sfam.npz")
    got = orc.camera_sfam(torch.from_numpy(g["feats"]), torch.from_numpy(g["probs"]), int(g["batch_size"]))
    np.testing.assert_allclose(got.numpy(), g["emb"], rtol=0, atol=1e-6)


def test_view_points_vs_reference():
    g = golden("view_points.npz")
    np.testing.assert_array_equal(orc.view_points(g["points"], g["view"], normalize=True), g["out"])
