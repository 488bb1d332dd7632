"""Drop-in check in the build container: the REFERENCE's own backbone files (det3d/models/backbones/scn_unet.py, scn.py),
imported unmodified from /root/reference, run on `lidarseg3d_amd.spconv` (kernels on tests/hipsim) and on the oracle's spconv
restatement, and must agree.  This is what "import lidarseg3d_amd.spconv as spconv" in INTEGRATION.md promises; it also covers
SpMiddleResNetFHD and the Cylinder3D backbones with their asymmetric (1,3,3)/(3,1,3)/(3,1,1)/... kernels and shared indice_keys
(other sparse backbones on the same kernels, no new kernel classes).
Skipped where /root/reference does not exist (the GPU box)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "det3d")), reason="reference tree only exists in the build container")

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    from lidarseg3d_amd import _lib, ops
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "spconv" or k.startswith("det3d")}
    _lib.use_library_for_testing(os.path.join(HERE, "hipsim", "libls3d_sim.so"))
    ops.set_sim(True)
    mg.setup_reference_imports()
    shim = mg.make_spconv_shim()

    def dense(self, channels_first=True):  # the only spconv tensor method the backbones use that the shim lacks
        shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        out = torch.zeros(shape, dtype=self.features.dtype)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out
    shim.SparseConvTensor.dense = dense
    # det3d.models.utils.build_norm_layer (scn.py): its own file drags in the distributed helpers; for type "BN1d" it returns
    # (name, nn.BatchNorm1d(n, eps, momentum)) - stubbed as exactly that
    mg._pkg("det3d.models.utils")
    sys.modules["det3d.models.utils"].build_norm_layer = \
        lambda cfg, n, postfix="": ("bn" + str(postfix), torch.nn.BatchNorm1d(n, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1)))

    def load(fname, sp, tag):
        sys.modules["spconv"] = sp
        sys.modules["det3d.models.registry"].BACKBONES._module_dict.clear()  # the file is imported once per spconv implementation
        name = "det3d.models.backbones.%s__%s" % (fname, tag)
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "det3d/models/backbones/%s.py" % fname))
        m = importlib.util.module_from_spec(spec)
        m.__package__ = "det3d.models.backbones"
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    yield shim, load
    ops.set_sim(False)
    _lib.use_library_for_testing(None)
    for k in [k for k in sys.modules if k == "spconv" or k.startswith("det3d")]:
        del sys.modules[k]
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v


def _inputs(n):
    from lidarseg3d_amd import synth
    from oracle import ref as orc
    from tests.util import golden
    cfg = synth.NUSC
    g = golden("unet_nusc_c13.npz")
    return cfg, torch.from_numpy(g["coords"][:n]), torch.from_numpy(g["voxel_features"][:n]), \
        np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"]))


def _randomise_bn(sd, seed):
    gen = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.1
    return sd


def test_reference_unetscn3d_file_runs_on_our_spconv(harness):
    import lidarseg3d_amd.spconv as ours
    shim, load = harness
    cfg, coords, feats, shape = _inputs(150)
    outs, sd = {}, None
    for tag, sp in (("oracle", shim), ("ours", ours)):
        m = load("scn_unet", sp, tag)
        torch.manual_seed(0)
        net = m.UNetSCN3D(num_input_features=13, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                          model_cfg=dict(SCALING_RATIO=1), ds_factor=8, us_factor=8).eval()
        if sd is None:
            sd = _randomise_bn({k: v.clone() for k, v in net.state_dict().items()}, 1)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            bd = net(dict(voxel_features=feats.clone(), voxel_coords=coords, batch_size=1, input_shape=shape))
        outs[tag] = bd
    a, b = outs["oracle"], outs["ours"]
    assert torch.equal(a["conv_point_coords"], b["conv_point_coords"])
    scale = float(a["conv_point_features"].abs().max())
    assert float((a["conv_point_features"] - b["conv_point_features"]).abs().max()) <= 1e-5 * scale + 1e-4
    ea, eb = a["encoded_spconv_tensor"], b["encoded_spconv_tensor"]
    assert torch.equal(ea.indices.int(), eb.indices.int())
    assert float((ea.features - eb.features).abs().max()) <= 1e-5 * float(ea.features.abs().max()) + 1e-4


def test_reference_spmiddleresnetfhd_file_runs_on_our_spconv(harness):
    """CenterPoint-style encoder (scn.py:83-176): SubM + three stride-2 SparseConv3d WITHOUT indice_key + a (3,1,1)/(2,1,1)
    conv + .dense(): none of it is touched by our own model code, only by the spconv replacement"""
    import lidarseg3d_amd.spconv as ours
    shim, load = harness
    cfg, coords, feats, shape = _inputs(200)
    outs, sd = {}, None
    for tag, sp in (("oracle", shim), ("ours", ours)):
        m = load("scn", sp, tag)
        torch.manual_seed(0)
        net = m.SpMiddleResNetFHD(num_input_features=13).eval()
        if sd is None:
            sd = _randomise_bn({k: v.clone() for k, v in net.state_dict().items()}, 2)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            outs[tag] = net(feats.clone(), coords, 1, shape)
    (da, ma), (db, mb) = outs["oracle"], outs["ours"]
    assert tuple(da.shape) == tuple(db.shape)
    assert float((da - db).abs().max()) <= 1e-5 * float(da.abs().max()) + 1e-4
    for k in ma:
        assert torch.equal(ma[k].indices.int(), mb[k].indices.int()), k
        assert float((ma[k].features - mb[k].features).abs().max()) <= 1e-5 * float(ma[k].features.abs().max()) + 1e-4, k


def _cyl_inputs(n, cin, seed=0):
    """voxels of a small two-frame cloud in a cylinder-style grid (z,y,x) = (rho, phi, height)-like extents"""
    gen = np.random.default_rng(seed)
    shape = np.array([16, 48, 40])  # input_shape is (x,y,z)-ordered in batch_dict; the backbones flip it
    box = np.array([8, 12, 10])  # occupied sub-box: ~27 % full, so every kernel offset of every shape has pairs
    lin = gen.choice(box.prod(), size=n, replace=False)
    x, y, z = lin % box[0] + 3, (lin // box[0]) % box[1] + 17, lin // (box[0] * box[1]) + 11
    b = (np.arange(n) >= n * 2 // 3).astype(np.int64)
    coords = torch.from_numpy(np.stack([b, z, y, x], 1).astype(np.int32))
    feats = torch.from_numpy(gen.normal(size=(n, cin)).astype(np.float32))
    return coords, feats, shape


def test_reference_unetcylinder3d_file_runs_on_our_spconv(harness):
    """UNetCylinder3D (scn_unet_cylinder3d.py:257-335): asymmetric SubM kernels, several kernel shapes under ONE indice_key
    (the stored pairs of the first layer are reused, spconv v1 semantics), stride-(2,2,1) pooling convs and their inverses,
    features assigned in place between layers"""
    import lidarseg3d_amd.spconv as ours
    shim, load = harness
    coords, feats, shape = _cyl_inputs(260, 16)
    outs, sd = {}, None
    for tag, sp in (("oracle", shim), ("ours", ours)):
        m = load("scn_unet_cylinder3d", sp, tag)
        torch.manual_seed(0)
        net = m.UNetCylinder3D(num_input_features=16, voxel_size=[0.2, 0.2, 0.2], point_cloud_range=[0, 0, 0, 3.2, 9.6, 8.0],
                               model_cfg=dict(init_size=8)).eval()
        if sd is None:
            sd = _randomise_bn({k: v.clone() for k, v in net.state_dict().items()}, 3)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            outs[tag] = net(dict(voxel_features=feats.clone(), voxel_coords=coords, batch_size=2, input_shape=shape))
    a, b = outs["oracle"], outs["ours"]
    assert torch.equal(a["conv_point_coords"], b["conv_point_coords"])
    fa, fb = a["conv_point_features"], b["conv_point_features"]
    assert tuple(fa.shape) == (260, 32) and float(fa.abs().max()) > 1e-3
    assert float((fa - fb).abs().max()) <= 1e-5 * float(fa.abs().max()) + 1e-4


def test_reference_cylinder3d_asymm_file_runs_on_our_spconv(harness):
    """Cylinder3D_Asymm_3d_spconv (cylinder3d_backbone.py:254-338): the same asymmetric blocks + a biased 3x3x3 SubM logits
    layer + SparseConvTensor.dense()"""
    import lidarseg3d_amd.spconv as ours
    shim, load = harness
    coords, feats, shape = _cyl_inputs(240, 16, seed=1)
    outs, sd = {}, None
    for tag, sp in (("oracle", shim), ("ours", ours)):
        m = load("cylinder3d_backbone", sp, tag)
        torch.manual_seed(0)
        net = m.Cylinder3D_Asymm_3d_spconv(output_shape=[int(v) for v in shape], num_input_features=16, nclasses=7, init_size=8).eval()
        if sd is None:
            sd = _randomise_bn({k: v.clone() for k, v in net.state_dict().items()}, 4)
            sd["logits.bias"] = torch.linspace(-1, 1, 7)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            outs[tag] = net(dict(voxel_features=feats.clone(), voxel_coords=coords, batch_size=2))["voxel_features"]
    a, b = outs["oracle"], outs["ours"]
    assert tuple(a.shape) == tuple(b.shape) == (2, 7, 16, 48, 40) and float(a.abs().max()) > 1e-3
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-4


@pytest.mark.parametrize("cls", ["Cylinder3DDynamicVoxelFeatureExtractor", "PolarNetDynamicVoxelFeatureExtractor"])
@pytest.mark.parametrize("average", [True, False])
def test_reference_dynamic_readers_run_on_our_scatter(harness, average, cls):
    """Cylinder3DDynamicVoxelFeatureExtractor (voxel_encoder.py:504-720), the reference's file unmodified, with `torch_scatter`
    = lidarseg3d_amd.scatter (kernels on the simulator) vs `torch_scatter` = the oracle's torch restatement: cylindrical
    voxelization, per-voxel mean for the point descriptor, the point MLP, mean / max pooling per voxel, majority-vote labels"""
    import types
    from lidarseg3d_amd import scatter as ours, synth
    from oracle import ref as orc
    oracle_ts = types.ModuleType("torch_scatter")
    oracle_ts.scatter_mean, oracle_ts.scatter_max = orc.scatter_mean, orc.scatter_max
    cfg = synth.NUSC
    frames = [synth.lidar_frame(900, seed=3, **cfg), synth.lidar_frame(500, seed=4, **cfg)]
    pts = torch.from_numpy(np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)]))
    labels = torch.randint(0, 17, (pts.shape[0],), generator=torch.Generator().manual_seed(1))
    outs, sd = {}, None
    for tag, ts in (("oracle", oracle_ts), ("ours", ours)):
        sys.modules["torch_scatter"] = ts
        sys.modules["det3d.models.registry"].READERS._module_dict.clear()
        name = "det3d.models.readers.voxel_encoder__" + tag
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "det3d/models/readers/voxel_encoder.py"))
        m = importlib.util.module_from_spec(spec)
        m.__package__ = "det3d.models.readers"
        sys.modules[name] = m
        spec.loader.exec_module(m)
        torch.manual_seed(0)
        net = getattr(m, cls)(grid_size=[48, 36, 8], point_cloud_range=[0, -3.1415926, -5, 50, 3.1415926, 3],
                                                       average_points=average, num_input_features=5, num_output_features=32,
                                                       fea_compre=16, voxel_label_enc="major").eval()
        if sd is None:
            sd = _randomise_bn({k: v.clone() for k, v in net.state_dict().items()}, 5)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            outs[tag] = net(dict(points=pts.clone(), batch_size=2, point_sem_labels=labels.clone()))
    a, b = outs["oracle"], outs["ours"]
    assert torch.equal(a["point_vcoors"], b["point_vcoors"])
    if cls.startswith("PolarNet"):  # dense BEV map [B, C, rho, phi] out of the per-pillar features
        assert tuple(a["voxel_features"].shape) == (2, 16, 48, 36) and float(a["voxel_features"].abs().max()) > 1e-3
    else:
        assert torch.equal(a["voxel_coords"], b["voxel_coords"])
        assert a["voxel_features"].shape[1] == 16 and a["voxel_features"].shape[0] == a["voxel_coords"].shape[0] > 100
    assert float((a["voxel_features"] - b["voxel_features"]).abs().max()) <= 1e-5 * float(a["voxel_features"].abs().max()) + 1e-5
    assert torch.equal(a["voxel_sem_labels"], b["voxel_sem_labels"])


def test_collate_kitti_equals_reference_on_segmentation_keys(harness):
    """lidarseg3d_amd.collate.collate_kitti vs the reference's function (torchie/parallel/collate.py:91-170), imported
    unmodified, on per-frame examples shaped like the segmentation pipeline's output (incl. a TTA list-of-variants frame)"""
    import types
    from lidarseg3d_amd import collate as ours
    dc = types.ModuleType("det3d.torchie.parallel.data_container")
    dc.DataContainer = type("DataContainer", (), {})
    for name in ("det3d.torchie.parallel",):
        pk = types.ModuleType(name)
        pk.__path__ = []
        sys.modules[name] = pk
    sys.modules["det3d.torchie.parallel.data_container"] = dc
    spec = importlib.util.spec_from_file_location("det3d.torchie.parallel.collate", os.path.join(REF, "det3d/torchie/parallel/collate.py"))
    ref = importlib.util.module_from_spec(spec)
    ref.__package__ = "det3d.torchie.parallel"
    sys.modules["det3d.torchie.parallel.collate"] = ref
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(0)

    def frame(n, v, tag):
        return dict(metadata=dict(token=tag), points=rng.normal(size=(n, 5)).astype(np.float32),
                    voxels=rng.normal(size=(v, 5, 5)).astype(np.float32), num_points=rng.integers(1, 6, size=v).astype(np.int32),
                    coordinates=rng.integers(0, 40, size=(v, 3)).astype(np.int32), num_voxels=np.array([v], dtype=np.int64),
                    shape=np.array([1024, 1024, 40]), voxel_sem_labels=rng.integers(0, 17, size=v).astype(np.uint8),
                    point_sem_labels=rng.integers(0, 17, size=n).astype(np.uint8), points_cuv=rng.normal(size=(n, 4)).astype(np.float32),
                    points_cp=rng.normal(size=(n, 3)).astype(np.float32), images=rng.normal(size=(6, 3, 8, 12)).astype(np.float32),
                    images_sem_labels=rng.integers(0, 17, size=(6, 8, 12)).astype(np.uint8))
    batch = [frame(50, 30, "a"), [frame(20, 11, "b0"), frame(20, 12, "b1")], frame(1, 1, "c")]
    want, got = ref.collate_kitti(batch), ours.collate_kitti(batch)
    assert set(want) == set(got)
    for k in want:
        if isinstance(want[k], torch.Tensor):
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
        elif isinstance(want[k], np.ndarray):
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
        else:
            assert got[k] == want[k], k
    assert got["points"].shape == (91, 6) and got["points"][:, 0].unique().tolist() == [0.0, 1.0, 2.0, 3.0]
    dev = ours.collate_points([torch.from_numpy(f["points"]) for f in (batch[0], batch[1][0], batch[1][1], batch[2])])
    assert torch.equal(dev, got["points"])
    with pytest.raises(NotImplementedError):
        ours.collate_kitti([dict(metadata={}, gt_boxes=[np.zeros((1, 7))])])


@pytest.mark.parametrize("tta", [False, True])
def test_predict_equals_reference(harness, tta):
    """point_head.predict (per-frame argmax; TTA: mean of the variants' softmax) vs the reference head's own method
    (point_seg_batchloss_head.py:171-271, the same code in point_seg_mseg3d_head.py:379-479) on the same logits"""
    import make_golden as mg
    from lidarseg3d_amd import point_heads
    mg._load("det3d.models.point_heads.point_utils", "det3d/models/point_heads/point_utils.py")
    sys.modules["det3d.models.registry"].POINT_HEADS._module_dict.clear()
    bh = mg._load("det3d.models.point_heads.point_seg_batchloss_head__predict", "det3d/models/point_heads/point_seg_batchloss_head.py")
    mcfg = dict(CONV_IN_DIM=32, CONV_CLS_FC=[64], CONV_ALIGN_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0)
    ref, ours = bh.PointSegBatchlossHead(class_agnostic=False, num_class=17, model_cfg=mcfg), point_heads.PointSegBatchlossHead(False, 17, mcfg)
    gen = torch.Generator().manual_seed(2)
    k, groups = 4, 2
    sizes = [37, 37, 37, 37, 52, 52, 52, 52] if tta else [40, 1, 63]   # TTA variants of one frame have the same point count
    B = len(sizes)
    pts = torch.cat([torch.cat([torch.full((n, 1), float(b)), torch.randn(n, 4, generator=gen)], 1) for b, n in enumerate(sizes)])
    logits = torch.randn(pts.shape[0], 17, generator=gen)
    labels = torch.randint(0, 17, (pts.shape[0],), generator=gen)
    example = dict(points=pts, num_voxels=torch.zeros(B), metadata=[dict(token="f%d" % b) for b in range(B)], point_sem_labels=labels)
    cfg = dict(tta_flag=True, merge_type="ArithmeticMean", num_tta_tranforms=k) if tta else dict()
    ref.forward_ret_dict["out_logits"] = logits
    ours.forward_ret_dict["out_logits"] = logits
    want, got = ref.predict(example=example, test_cfg=cfg), ours.predict(example=example, test_cfg=cfg)
    assert len(want) == len(got) == (groups if tta else B)
    for w, g in zip(want, got):
        assert w["metadata"] == g["metadata"]
        assert torch.equal(w["pred_point_sem_labels"], g["pred_point_sem_labels"])
        assert torch.equal(w["point_sem_labels"], g["point_sem_labels"])
