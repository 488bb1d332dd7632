#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Runs only where /root/reference exists (never on the GPU box).  Nothing from the reference is
copied: its files are imported where they lie (importlib + a package skeleton of empty stub modules,
SURVEY.md Appendix B) and only inputs / outputs (plain arrays) are written to .npz.

What is pinned by which reference code:
  voxelize_*.npz   det3d/ops/point_cloud/point_cloud_ops.py (numba kernel under an identity-jit stub)
                   + the reference's compiled C++ det3d/ops/voxel/src/*.cpp (oracle/_ref)
  vfe_*.npz        det3d/models/readers/voxel_encoder.py (Mean / ImprovedMean / Transformer VFE)
  unet_*.npz       det3d/models/backbones/scn_unet.py wiring, run over a `spconv` shim whose three conv
                   primitives are the oracle restatement (spconv itself is absent => parity unpinned)
  head_*.npz       det3d/models/point_heads/{point_seg_batchloss_head,point_seg_mseg3d_head,
                   context_module,point_utils}.py with three_nn/three_interpolate = oracle C restatement
  manifest_*.json  state_dict key -> shape of the reference modules (Appendix A)

Weights are never stored: fixtures carry a seed for lidarseg3d_amd.synth.random_state_dict.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from lidarseg3d_amd import synth  # noqa: E402
from oracle import ref as orc  # noqa: E402


# ---------------------------------------------------------------- import harness (Appendix B)
def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def setup_reference_imports():
    for p in ["det3d", "det3d.models", "det3d.models.readers", "det3d.models.backbones",
              "det3d.models.point_heads", "det3d.core", "det3d.core.utils", "det3d.utils", "det3d.torchie",
              "det3d.ops", "det3d.ops.pointnet2_batch", "det3d.ops.point_cloud", "det3d.core.input"]:
        _pkg(p)
    sys.modules["det3d.torchie"].is_str = lambda x: isinstance(x, str)
    sys.modules["det3d"].torchie = sys.modules["det3d.torchie"]
    reg = _load("det3d.utils.registry", "det3d/utils/registry.py")
    sys.modules["det3d.utils"].Registry = reg.Registry
    sys.modules["det3d.utils"].build_from_cfg = reg.build_from_cfg
    _load("det3d.models.registry", "det3d/models/registry.py")
    _pkg("det3d.core.utils.box_utils")
    cu = _load("det3d.core.utils.common_utils", "det3d/core/utils/common_utils.py")
    sys.modules["det3d.core.utils"].common_utils = cu
    _load("det3d.core.utils.loss_utils", "det3d/core/utils/loss_utils.py")
    # numba: identity jit
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = nb
    sys.modules["torch_scatter"] = types.ModuleType("torch_scatter")
    # pointnet2 three_nn / three_interpolate: CUDA-only in the reference -> oracle restatement
    pn = types.ModuleType("det3d.ops.pointnet2_batch.pointnet2_utils")

    def three_nn(unknown, known):
        d2, idx = orc.three_nn(unknown[0].numpy(), known[0].numpy())
        return torch.sqrt(torch.from_numpy(d2))[None], torch.from_numpy(idx)[None]

    def three_interpolate(features, idx, weight):
        o = orc.three_interpolate_cm(features[0].numpy(), idx[0].numpy(), weight[0].numpy())
        return torch.from_numpy(o)[None]

    pn.three_nn, pn.three_interpolate = three_nn, three_interpolate
    sys.modules["det3d.ops.pointnet2_batch.pointnet2_utils"] = pn
    # spconv shim: containers + three conv primitives backed by the oracle restatement
    sys.modules["spconv"] = make_spconv_shim()


def make_spconv_shim():
    sp = types.ModuleType("spconv")
    nn = torch.nn

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
            self.features, self.indices = features, indices
            self.spatial_shape = tuple(int(v) for v in spatial_shape)
            self.batch_size = batch_size
            self.indice_dict = {}

        def like(self, features, indices=None, spatial_shape=None):
            t = SparseConvTensor(features, self.indices if indices is None else indices,
                                 self.spatial_shape if spatial_shape is None else spatial_shape, self.batch_size)
            t.indice_dict = self.indice_dict
            return t

    class SparseModule(nn.Module):
        pass

    class _Conv(SparseModule):
        kind = None

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
            super().__init__()
            ks = orc._triple(kernel_size)
            self.ks, self.stride, self.padding, self.key = ks, stride, padding, indice_key
            self.weight = nn.Parameter(torch.zeros(*ks, in_channels, out_channels))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

        def forward(self, x):
            c = x.indices.numpy()
            if self.kind == "subm":
                if self.key not in x.indice_dict:
                    x.indice_dict[self.key] = ("subm", orc.subm_rulebook(c, x.spatial_shape, self.ks))
                nbr = x.indice_dict[self.key][1]
                return x.like(self._b(orc.spconv_fwd(x.features, self.weight, nbr)))
            if self.kind == "conv":
                oc, oshape, nbr = orc.conv_rulebook(c, x.spatial_shape, self.ks, self.stride, self.padding)
                x.indice_dict[self.key] = ("conv", nbr, x.indices, x.spatial_shape)
                return x.like(self._b(orc.spconv_fwd(x.features, self.weight, nbr)), torch.from_numpy(oc), oshape)
            _, nbr, in_idx, in_shape = x.indice_dict[self.key]
            f = orc.spconv_fwd(x.features, self.weight, nbr, inverse=True, n_out=in_idx.shape[0])
            return x.like(self._b(f), in_idx, in_shape)

        def _b(self, f):
            return f if self.bias is None else f + self.bias

    class SubMConv3d(_Conv):
        kind = "subm"

    class SparseConv3d(_Conv):
        kind = "conv"

    class SparseInverseConv3d(_Conv):
        kind = "inv"

        def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=False):
            super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    class SparseSequential(SparseModule):
        def __init__(self, *mods):
            super().__init__()
            for i, m in enumerate(mods):
                self.add_module(str(i), m)

        def forward(self, x):
            for m in self._modules.values():
                if isinstance(m, SparseModule):
                    x = m(x)
                elif x.indices.shape[0] != 0:
                    x.features = m(x.features)
            return x

    for k, v in dict(SparseConvTensor=SparseConvTensor, SparseModule=SparseModule, SubMConv3d=SubMConv3d,
                     SparseConv3d=SparseConv3d, SparseInverseConv3d=SparseInverseConv3d,
                     SparseSequential=SparseSequential).items():
        setattr(sp, k, v)
    return sp


def load_sd(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.random_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    module.eval()
    return shapes


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def build_ref_ext():
    """compile the reference's own C++ voxel ops where they lie (oracle/_ref/, gitignored)."""
    from torch.utils.cpp_extension import load
    out = os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(REF, "det3d/ops/voxel/src")
    return load(name="ref_voxel_layer", build_directory=out, with_cuda=False, verbose=False,
                sources=[os.path.join(src, f) for f in ("voxelization.cpp", "voxelization_cpu.cpp", "scatter_points_cpu.cpp")])


def main():
    setup_reference_imports()
    torch.manual_seed(0)
    manifests = {}

    # ---------------------------------------------------------------- voxelization
    pco = _load("det3d.ops.point_cloud.point_cloud_ops", "det3d/ops/point_cloud/point_cloud_ops.py")
    sys.modules["det3d.ops.point_cloud"].point_cloud_ops = pco
    vg = _load("det3d.core.input.voxel_generator", "det3d/core/input/voxel_generator.py")
    ext = build_ref_ext()
    cases = [("nusc", synth.NUSC, 6000, 300000, 0), ("nusc_cap", synth.NUSC, 3000, 500, 1),
             ("kitti", synth.KITTI, 5000, 300000, 2)]
    for tag, cfg, n, max_vox, seed in cases:
        pts = synth.lidar_frame(n, seed=seed, **cfg)
        if tag == "nusc":  # edge cases: out-of-range points, exact boundaries, duplicates, a dense cluster
            pts[:40, 0] = 60.0
            pts[40:50, 2] = 3.0
            pts[50:60, 2] = -5.0
            pts[60:70] = pts[70:80]
            pts[100:200, :3] = np.float32([1.234, -2.345, -1.0]) + np.random.RandomState(1).uniform(0, 0.05, (100, 3)).astype(np.float32)
        gen = vg.VoxelGenerator(cfg["voxel_size"], cfg["pc_range"], 5, max_voxels=max_vox)
        v, c, k = gen.generate(pts, max_vox)
        t = torch.from_numpy(pts)
        dyn = torch.zeros((n, 3), dtype=torch.int32)
        ext.dynamic_voxelize(t, dyn, cfg["voxel_size"], cfg["pc_range"], 3)
        hv = torch.zeros((max_vox, 5, pts.shape[1]))
        hc = torch.zeros((max_vox, 3), dtype=torch.int32)
        hn = torch.zeros((max_vox,), dtype=torch.int32)
        hnum = ext.hard_voxelize(t, hv, hc, hn, cfg["voxel_size"], cfg["pc_range"], 5, max_vox, 3)
        # the reference's CPU scatter indexes out of bounds for coors == -1 (scatter_points_cpu.cpp:48-50),
        # so it is only fed the in-range points
        inr = dyn[:, 0] >= 0
        sc = ext.dynamic_point_to_voxel_forward(t[inr].contiguous(), dyn[inr].contiguous(), cfg["voxel_size"], cfg["pc_range"])
        save("voxelize_%s.npz" % tag, points=pts, voxel_size=np.float32(cfg["voxel_size"]),
             pc_range=np.float32(cfg["pc_range"]), max_voxels=max_vox,
             numba_voxels=v, numba_coors=c, numba_num=k,
             cpp_dyn_coors=dyn.numpy(), cpp_hard_voxels=hv[:hnum].numpy(), cpp_hard_coors=hc[:hnum].numpy(),
             cpp_hard_num=hn[:hnum].numpy(), cpp_scatter_voxels=sc[0].numpy(), cpp_scatter_coors=sc[1].numpy(),
             cpp_scatter_num=sc[2].numpy())
        if tag == "nusc":
            base = dict(points=pts, voxels=v, coors=c, num=k, cfg=cfg)

    # ---------------------------------------------------------------- VFE readers
    ve = _load("det3d.models.readers.voxel_encoder", "det3d/models/readers/voxel_encoder.py")
    vx, num = torch.from_numpy(base["voxels"]), torch.from_numpy(base["num"])
    with torch.no_grad():
        mean = ve.MeanVoxelFeatureExtractor(num_input_features=5)(vx, num)
        imp = ve.ImprovedMeanVoxelFeatureExtractor(num_input_features=5)(vx, num)
        tv = ve.TransformerVoxelFeatureExtractor(num_input_features=5, num_compressed_features=16, num_embed=64,
                                                 num_head=4, num_layers=3)
        manifests["reader.TransformerVoxelFeatureExtractor"] = load_sd(tv, 11)
        # TransVFE.forward with the encoder layers iterated by hand (torch>=2.0 breaks nn.TransformerEncoder
        # with this custom layer); everything else is the reference's own module code.
        P = vx.shape[1]
        pm = vx.sum(dim=1) / num.type_as(vx).view(-1, 1)
        desc = imp  # the descriptor is literally ImprovedMeanVFE's output (same code, voxel_encoder.py:210-246)
        assert torch.equal(desc[:, :3], pm[:, :3])
        pf = torch.cat([vx, desc[:, None, :].expand(-1, P, -1)], dim=-1).permute(0, 2, 1)
        pf = tv.feature_conv(pf).permute(2, 0, 1)
        for layer in tv.chunck.layers:
            pf = layer(pf)
        tvo = tv.compress_layer(torch.max(pf.permute(1, 2, 0), dim=2)[0])
    save("vfe_nusc.npz", voxels=base["voxels"], num=base["num"], mean=mean.numpy(), improved=imp.numpy(),
         trans=tvo.numpy(), trans_seed=11)

    # ---------------------------------------------------------------- UNetSCN3D (reference wiring over the shim)
    scn = _load("det3d.models.backbones.scn_unet", "det3d/models/backbones/scn_unet.py")
    cfg = base["cfg"]
    for cin, feats, tag in ((13, imp, "c13"), (16, tvo, "c16")):
        net = scn.UNetSCN3D(num_input_features=cin, voxel_size=cfg["voxel_size"], point_cloud_range=cfg["pc_range"],
                            model_cfg=dict(SCALING_RATIO=2), ds_factor=8, us_factor=8)
        manifests["backbone.UNetSCN3D.%s" % tag] = load_sd(net, 23)
        coords = np.concatenate([np.zeros((base["coors"].shape[0], 1), np.int32), base["coors"]], axis=1)
        bd = dict(voxel_features=feats, voxel_coords=torch.from_numpy(coords), batch_size=1,
                  input_shape=np.asarray(orc.grid_size(cfg["voxel_size"], cfg["pc_range"])))
        with torch.no_grad():
            out = net(bd)
        ms = out["multi_scale_3d_features"]
        extra = {}
        if cin == 13:  # intermediates only once (fixture size)
            extra = dict(conv_point_coords=out["conv_point_coords"].numpy(),
                         x_conv4_features=ms["x_conv4"].features.numpy(), x_conv4_indices=ms["x_conv4"].indices.numpy(),
                         x_up4_indices=ms["x_conv3"].indices.numpy(), x_up3_indices=ms["x_conv2"].indices.numpy(),
                         enc_features=out["encoded_spconv_tensor"].features.numpy(),
                         enc_indices=out["encoded_spconv_tensor"].indices.numpy())
        save("unet_nusc_%s.npz" % tag, voxel_features=feats.numpy(), coords=coords, seed=23,
             conv_point_features=out["conv_point_features"].numpy(), **extra)
        if cin == 16:
            cpf16, cpc = out["conv_point_features"], out["conv_point_coords"]
        else:
            cpf13 = out["conv_point_features"]

    # ---------------------------------------------------------------- point heads
    _load("det3d.models.point_heads.point_utils", "det3d/models/point_heads/point_utils.py")
    _load("det3d.models.point_heads.context_module", "det3d/models/point_heads/context_module.py")
    bh = _load("det3d.models.point_heads.point_seg_batchloss_head", "det3d/models/point_heads/point_seg_batchloss_head.py")
    mh = _load("det3d.models.point_heads.point_seg_mseg3d_head", "det3d/models/point_heads/point_seg_mseg3d_head.py")
    pts_b = torch.from_numpy(np.concatenate([np.zeros((base["points"].shape[0], 1), np.float32), base["points"]], 1))
    head = bh.PointSegBatchlossHead(class_agnostic=False, num_class=17, model_cfg=dict(
        CONV_IN_DIM=32, CONV_CLS_FC=[64], CONV_ALIGN_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0))
    manifests["point_head.PointSegBatchlossHead"] = load_sd(head, 31)
    bd = dict(batch_size=1, conv_point_features=cpf16, conv_point_coords=cpc, points=pts_b[:, 0:4])
    with torch.no_grad():
        head(bd, return_loss=False)
    save("head_batchloss_nusc.npz", conv_point_features=cpf16.numpy(), conv_point_coords=cpc.numpy(),
         points=pts_b.numpy(), seed=31, out_logits=bd["out_logits"].numpy(),
         conv_logits=head.forward_ret_dict["conv_logits"].numpy())

    # MSeg3D head, batch of 2 frames (second frame = a shifted copy of a subset) to exercise per-frame loops
    n0 = base["points"].shape[0]
    pts1 = base["points"][: n0 // 2].copy()
    pts1[:, 0] += 0.37
    v1, c1, k1 = orc.hard_voxelize(pts1, cfg["voxel_size"], cfg["pc_range"], 5, 300000)
    coords2 = np.concatenate([np.concatenate([np.zeros((base["coors"].shape[0], 1), np.int32), base["coors"]], 1),
                              np.concatenate([np.ones((c1.shape[0], 1), np.int32), c1], 1)], 0)
    rng = np.random.Generator(np.random.PCG64(5))
    vf2 = torch.from_numpy(rng.normal(0, 1, (coords2.shape[0], 32)).astype(np.float32))
    cpc2 = orc.voxel_centers(coords2, cfg["voxel_size"], cfg["pc_range"])
    pts2 = torch.from_numpy(np.concatenate([
        np.concatenate([np.zeros((n0, 1), np.float32), base["points"]], 1),
        np.concatenate([np.ones((pts1.shape[0], 1), np.float32), pts1], 1)], 0))
    img, emb, cuv = synth.camera_inputs(pts2.shape[0], seed=3, ncam=6, c_img=48, h=40, w=60, batch=2)
    mcfg = dict(VOXEL_IN_DIM=32, VOXEL_CLS_FC=[64], VOXEL_ALIGN_DIM=64, IMAGE_IN_DIM=48, IMAGE_ALIGN_DIM=64,
                GEO_FUSED_DIM=64, OUT_CLS_FC=[64, 64], IGNORED_LABEL=0, DP_RATIO=0.25, MIMIC_FC=[64, 64],
                SFPhase_CFG=dict(embeddings_proj_kernel_size=1, d_model=96, n_head=4, n_layer=6, n_ffn=192,
                                 drop_ratio=0, activation="relu", pre_norm=False))
    head = mh.PointSegMSeg3DHead(class_agnostic=False, num_class=17, model_cfg=mcfg)
    manifests["point_head.PointSegMSeg3DHead"] = load_sd(head, 37)
    bd = dict(batch_size=2, conv_point_features=vf2, conv_point_coords=cpc2, points=pts2[:, 0:4],
              image_features=torch.from_numpy(img), points_cuv=torch.from_numpy(cuv),
              camera_semantic_embeddings=torch.from_numpy(emb))
    with torch.no_grad():
        head(bd, return_loss=False)
    save("head_mseg3d_nusc.npz", conv_point_features=vf2.numpy(), conv_point_coords=cpc2.numpy(), coords=coords2,
         points=pts2.numpy(), cam_seed=3, cam_hw=np.int64([40, 60]), seed=37,  # camera inputs: synth.camera_inputs(seed)
         out_logits=bd["out_logits"].numpy(), voxel_logits=head.forward_ret_dict["voxel_logits"].numpy())

    # ---- camera SFAM (img_heads/fcn_mseg3d_head.py:17-51): the class is plain torch; its file's other imports are stubbed
    for name in ("mmcv", "mmcv.cnn", "det3d.models.builder", "det3d.models.img_heads", "det3d.models.img_heads.decode_head",
                 "det3d.ops.mmseg_ops", "det3d.models.img_heads.sc_conv"):
        if name not in sys.modules:
            _pkg(name)
    sys.modules["mmcv.cnn"].ConvModule = object
    sys.modules["det3d.models.builder"].IMG_HEADS = types.SimpleNamespace(register_module=lambda c: c)
    sys.modules["det3d.models.img_heads.decode_head"].BaseDecodeHead = torch.nn.Module
    sys.modules["det3d.ops.mmseg_ops"].resize = None
    sys.modules["det3d.models.img_heads.sc_conv"].SCBottleneck = object
    fh = _load("det3d.models.img_heads.fcn_mseg3d_head", "det3d/models/img_heads/fcn_mseg3d_head.py")
    rng = np.random.default_rng(41)
    cf = torch.from_numpy(rng.normal(size=(2 * 6, 48, 10, 15)).astype(np.float32))
    cp = torch.from_numpy((rng.normal(size=(2 * 6, 17, 10, 15)) * 3).astype(np.float32))
    with torch.no_grad():
        cemb = fh.CameraSemanticFeatureAggregationModule()(cf, cp, 2)
    save("camera_sfam.npz", feats=cf.numpy(), probs=cp.numpy(), batch_size=2, emb=cemb.numpy())

    # ---- view_points (datasets/pipelines/loading.py:67-103), the projection helper under points_cp
    for name in ("turtle", "pycocotools", "pycocotools.mask", "cv2", "det3d.core.box_np_ops", "det3d.datasets", "det3d.datasets.pipelines",
                 "det3d.datasets.registry"):
        if name not in sys.modules:
            _pkg(name)
    sys.modules["turtle"].shape = None
    sys.modules["det3d.core"].box_np_ops = sys.modules["det3d.core.box_np_ops"]
    sys.modules["det3d.datasets.registry"].PIPELINES = types.SimpleNamespace(register_module=lambda c: c)
    ld = _load("det3d.datasets.pipelines.loading", "det3d/datasets/pipelines/loading.py")
    rng = np.random.default_rng(43)
    pc = rng.normal(size=(3, 500)) * np.array([[20.0], [20.0], [30.0]])
    K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    save("view_points.npz", points=pc, view=K, out=ld.view_points(pc, K, normalize=True))

    # ---- Lovasz-Softmax + CE as the batch-loss head applies them (loss_utils.py:217-291, point_seg_batchloss_head.py:77-121)
    lu = sys.modules["det3d.core.utils.loss_utils"]
    rng = np.random.default_rng(47)
    lg = torch.from_numpy((rng.normal(size=(700, 17)) * 2).astype(np.float32)).requires_grad_(True)
    lb = torch.from_numpy(rng.integers(0, 17, size=700).astype(np.int64))
    lb[rng.uniform(size=700) < 0.2] = 0  # ignored label
    lb[lb == 5] = 6                      # a class that is absent
    lv = lu.lovasz_softmax(torch.softmax(lg, dim=-1), lb, ignore=0)
    ce = torch.nn.CrossEntropyLoss(ignore_index=0)(lg, lb)
    (lv + ce).backward()
    save("seg_loss.npz", logits=lg.detach().numpy(), labels=lb.numpy(), ignore=0, lovasz=float(lv), ce=float(ce), grad=lg.grad.numpy())

    with open(os.path.join(HERE, "manifests.json"), "w") as f:
        json.dump({k: {n: list(s) for n, s in v.items()} for k, v in manifests.items()}, f, indent=0, sort_keys=True)
    print("wrote manifests.json")


if __name__ == "__main__":
    main()
