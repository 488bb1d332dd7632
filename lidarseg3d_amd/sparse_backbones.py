"""The reference's other sparse-convolution backbones on the same kernels, registered under the reference's names
with its constructor signatures, attribute names (=> state_dict keys) and forward contracts:

  SpMiddleResNetFHD               det3d/models/backbones/scn.py:84-176              (CenterPoint's 3-D encoder)
  UNetCylinder3D                  det3d/models/backbones/scn_unet_cylinder3d.py:257-335
  Cylinder3D_Asymm_3d_spconv      det3d/models/backbones/cylinder3d_backbone.py:254-338   (dense logits volume)
  Cylinder3D_Asymm_3d_spconv_v2p  det3d/models/backbones/cylinder3d_backbone.py:341-442   (voxel -> point features)

Own code on lidarseg3d_amd.spconv: every convolution is ls3d_gather_gemm / ls3d_tile_conv over the output-major rulebooks; the residual
blocks of SpMiddleResNetFHD fuse BatchNorm(eval) / bias / ReLU / identity into the convolution's epilogue; the Cylinder3D blocks apply
their activation BEFORE the BatchNorm (conv -> LeakyReLU -> BN, a ReLU-after-BN epilogue cannot express it) and run that tail, the
residual sums and ReconBlock's sigmoid gates as one ls3d_act_affine launch per convolution.  Layers of different kernel shapes that
share an indice_key reuse the first layer's pairs exactly as spconv v1 does (spconv.SparseConvolution._weight_for).

Training mode / gradient flow takes the differentiable composition (convolutions through spconv._SparseConvFn, torch for the
point-wise tails)."""
import numpy as np
import torch
from torch import nn

from . import ops
from . import spconv
from .registry import BACKBONES
from .scn_unet import SparseBasicBlock

_NORMS = {"BN": nn.BatchNorm2d, "BN1d": nn.BatchNorm1d, "GN": nn.GroupNorm}


def build_norm_layer(cfg, num_features, postfix=""):
    """-> (name, layer): the cfg-driven norm factory the detection backbones use (det3d/models/utils/norm.py:67-108)"""
    assert isinstance(cfg, dict) and "type" in cfg
    args = dict(cfg)
    kind = args.pop("type")
    if kind not in _NORMS:
        raise KeyError("Unrecognized norm type {}".format(kind))
    trainable = args.pop("requires_grad", True)
    args.setdefault("eps", 1e-5)
    layer = _NORMS[kind](num_channels=num_features, **args) if kind == "GN" else _NORMS[kind](num_features, **args)
    for p in layer.parameters():
        p.requires_grad = trainable
    return {"BN": "bn", "BN1d": "bn1d", "GN": "gn"}[kind] + str(postfix), layer


def _stage(cin, cout, norm_cfg, key, down=None):
    """[strided conv + BN + ReLU,] two residual blocks on one rulebook: an encoder level of scn.py:104-142"""
    mods = []
    if down is not None:
        mods += [spconv.SparseConv3d(cin, cout, 3, 2, padding=down, bias=False), build_norm_layer(norm_cfg, cout)[1], nn.ReLU(inplace=True)]
    norm_fn = lambda c: build_norm_layer(norm_cfg, c)[1]  # noqa: E731
    mods += [SparseBasicBlock(cout, cout, indice_key=key, norm_fn=norm_fn, bias=True) for _ in range(2)]
    return spconv.SparseSequential(*mods)


@BACKBONES.register_module
class SpMiddleResNetFHD(nn.Module):
    """scn.py:84-176.  forward(voxel_features, coors, batch_size, input_shape) -> (dense [N, C * D, H, W], {conv1..conv4: SparseConvTensor})"""

    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super().__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.conv_input = spconv.SparseSequential(spconv.SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"),
                                                  build_norm_layer(norm_cfg, 16)[1], nn.ReLU(inplace=True))
        self.conv1 = _stage(16, 16, norm_cfg, "res0")
        self.conv2 = _stage(16, 32, norm_cfg, "res1", down=1)
        self.conv3 = _stage(32, 64, norm_cfg, "res2", down=1)
        self.conv4 = _stage(64, 128, norm_cfg, "res3", down=[0, 1, 1])
        self.extra_conv = spconv.SparseSequential(spconv.SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False),
                                                  build_norm_layer(norm_cfg, 128)[1], nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        x = spconv.SparseConvTensor(voxel_features, coors.int(), sparse_shape, batch_size)
        x = self.conv_input(x)
        scales = {}
        for name in ("conv1", "conv2", "conv3", "conv4"):
            x = scales[name] = getattr(self, name)(x)
        ret = self.extra_conv(x).dense()
        n, c, d, h, w = ret.shape
        return ret.view(n, c * d, h, w), scales


# ---------------------------------------------------------------------------------------------------------------- Cylinder3D
# asymmetric SubM kernels of the blocks: name -> (kernel, padding); spconv ignores a SubMConv3d's padding, kept for the signature
_K = {"133": ((1, 3, 3), (0, 1, 1)), "313": ((3, 1, 3), (1, 0, 1)), "333": (3, 1), "113": ((1, 1, 3), (0, 0, 1)), "131": ((1, 3, 1), (0, 1, 0)),
      "311": ((3, 1, 1), (1, 0, 0))}


def _subm(kind, cin, cout, key):
    k, p = _K[kind]
    return spconv.SubMConv3d(cin, cout, kernel_size=k, stride=1, padding=p, bias=False, indice_key=key)


def _fused(conv, bn, x):
    """the HIP tail applies: inference, no gradient wanted through this layer"""
    return not (bn.training or spconv.needs_grad(conv, x.features))


def _conv_act_bn(conv, act, bn, x, add=None):
    """bn(act(conv(x))) [+ add] -> SparseConvTensor; act = the block's LeakyReLU"""
    if _fused(conv, bn, x):
        y = conv(x)
        scale, shift = spconv.cached_bn_scale_shift(conv, bn)
        y.features = ops.act_affine(y.features, pre="leaky", slope=act.negative_slope, scale=scale, shift=shift, add=add, n_dev=y.n_dev)
        return y
    y = conv(x)
    f = bn(act(y.features))
    y.features = f if add is None else f + add
    return y


def _init_bn(module):
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _TwoBranch(nn.Module):
    """two chains of two asymmetric convolutions over one rulebook, summed: ResContextBlock (first kernels 133 | 313) and the body of
    ResBlock (313 | 133) - scn_unet_cylinder3d.py:52-158.  Attribute names are the reference's: (conv1, bn0, conv1_2, bn0_2) is the
    shortcut chain, (conv2, bn1, conv3, bn2) the residual one."""

    def __init__(self, cin, cout, first, second, key):
        super().__init__()
        self.conv1, self.bn0, self.act1 = _subm(first, cin, cout, key), nn.BatchNorm1d(cout), nn.LeakyReLU()
        self.conv1_2, self.bn0_2, self.act1_2 = _subm(second, cout, cout, key), nn.BatchNorm1d(cout), nn.LeakyReLU()
        self.conv2, self.act2, self.bn1 = _subm(second, cin, cout, key), nn.LeakyReLU(), nn.BatchNorm1d(cout)
        self.conv3, self.act3, self.bn2 = _subm(first, cout, cout, key), nn.LeakyReLU(), nn.BatchNorm1d(cout)

    def branches(self, x):
        s = _conv_act_bn(self.conv1, self.act1, self.bn0, x)
        s = _conv_act_bn(self.conv1_2, self.act1_2, self.bn0_2, s)
        r = _conv_act_bn(self.conv2, self.act2, self.bn1, x)
        return _conv_act_bn(self.conv3, self.act3, self.bn2, r, add=s.features)


class ResContextBlock(_TwoBranch):
    def __init__(self, in_filters, out_filters, kernel_size=(3, 3, 3), stride=1, indice_key=None):
        super().__init__(in_filters, out_filters, "133", "313", indice_key + "bef")
        _init_bn(self)

    def forward(self, x):
        return self.branches(x)


class ResBlock(_TwoBranch):
    def __init__(self, in_filters, out_filters, dropout_rate, kernel_size=(3, 3, 3), stride=1, pooling=True, drop_out=True, height_pooling=False,
                 indice_key=None):
        super().__init__(in_filters, out_filters, "313", "133", indice_key + "bef")
        self.pooling, self.drop_out = pooling, drop_out
        if pooling:
            self.pool = spconv.SparseConv3d(out_filters, out_filters, kernel_size=3, stride=2 if height_pooling else (2, 2, 1), padding=1,
                                            indice_key=indice_key, bias=False)
        _init_bn(self)

    def forward(self, x):
        res = self.branches(x)
        return (self.pool(res), res) if self.pooling else res


class UpBlock(nn.Module):
    """scn_unet_cylinder3d.py:161-218: 3x3x3 conv, inverse convolution back onto the encoder's sites + skip, three convolutions"""

    def __init__(self, in_filters, out_filters, kernel_size=(3, 3, 3), indice_key=None, up_key=None):
        super().__init__()
        self.trans_dilao, self.trans_act, self.trans_bn = _subm("333", in_filters, out_filters, indice_key + "new_up"), nn.LeakyReLU(), nn.BatchNorm1d(out_filters)
        self.conv1, self.act1, self.bn1 = _subm("133", out_filters, out_filters, indice_key), nn.LeakyReLU(), nn.BatchNorm1d(out_filters)
        self.conv2, self.act2, self.bn2 = _subm("313", out_filters, out_filters, indice_key), nn.LeakyReLU(), nn.BatchNorm1d(out_filters)
        self.conv3, self.act3, self.bn3 = _subm("333", out_filters, out_filters, indice_key), nn.LeakyReLU(), nn.BatchNorm1d(out_filters)
        self.up_subm = spconv.SparseInverseConv3d(out_filters, out_filters, kernel_size=3, indice_key=up_key, bias=False)
        _init_bn(self)

    def forward(self, x, skip):
        up = _conv_act_bn(self.trans_dilao, self.trans_act, self.trans_bn, x)
        if spconv.needs_grad(self.up_subm, up.features):
            up = self.up_subm(up)
            up.features = up.features + skip.features
        else:  # the skip connection is the inverse convolution's residual operand
            rb = self.up_subm.rulebook(up)
            f = self.up_subm.conv(up, rb, res_pre=skip.features)
            up = up._like(f, rb.in_indices, rb.in_shape, n_dev=rb.rows_dev(True))
        up = _conv_act_bn(self.conv1, self.act1, self.bn1, up)
        up = _conv_act_bn(self.conv2, self.act2, self.bn2, up)
        return _conv_act_bn(self.conv3, self.act3, self.bn3, up)


class ReconBlock(nn.Module):
    """scn_unet_cylinder3d.py:221-252: x * (sigmoid(bn(conv311 x)) + sigmoid(bn(conv131 x)) + sigmoid(bn(conv113 x)))"""

    def __init__(self, in_filters, out_filters, kernel_size=(3, 3, 3), stride=1, indice_key=None):
        super().__init__()
        key = indice_key + "bef"
        self.conv1, self.bn0, self.act1 = _subm("311", in_filters, out_filters, key), nn.BatchNorm1d(out_filters), nn.Sigmoid()
        self.conv1_2, self.bn0_2, self.act1_2 = _subm("131", in_filters, out_filters, key), nn.BatchNorm1d(out_filters), nn.Sigmoid()
        self.conv1_3, self.bn0_3, self.act1_3 = _subm("113", in_filters, out_filters, key), nn.BatchNorm1d(out_filters), nn.Sigmoid()

    def forward(self, x, out=None):
        """out (optional): a [rows, C] column view that receives the gated features (the left half of the caller's concat buffer)"""
        gates = ((self.conv1, self.bn0), (self.conv1_2, self.bn0_2), (self.conv1_3, self.bn0_3))
        if all(_fused(c, b, x) for c, b in gates):
            acc = None
            for i, (conv, bn) in enumerate(gates):
                scale, shift = spconv.cached_bn_scale_shift(conv, bn)
                y = conv(x)
                last = i == len(gates) - 1
                acc = ops.act_affine(y.features, post="sigmoid", scale=scale, shift=shift, add=acc, mul=x.features if last else None,
                                     out=out if last else None, n_dev=y.n_dev)
            y.features = acc
            return y
        total = None
        for conv, bn in gates:
            y = conv(x)
            g = torch.sigmoid(bn(y.features))
            total = g if total is None else total + g
        y.features = total * x.features
        if out is not None:
            out.copy_(y.features)
            y.features = out
        return y


class _AsymmTrunk(nn.Module):
    """encoder (context block + four pooled residual blocks), decoder (four UpBlocks), ReconBlock: the part the three Cylinder3D
    variants share (scn_unet_cylinder3d.py:270-285, cylinder3d_backbone.py:277-292,363-378).  trunk(x) -> SparseConvTensor with
    features = cat(recon(up1e), up1e) [rows, 4 * init_size]"""

    def _build_trunk(self, cin, s):
        self.downCntx = ResContextBlock(cin, s, indice_key="pre")
        self.resBlock2 = ResBlock(s, 2 * s, 0.2, height_pooling=True, indice_key="down2")
        self.resBlock3 = ResBlock(2 * s, 4 * s, 0.2, height_pooling=True, indice_key="down3")
        self.resBlock4 = ResBlock(4 * s, 8 * s, 0.2, pooling=True, height_pooling=False, indice_key="down4")
        self.resBlock5 = ResBlock(8 * s, 16 * s, 0.2, pooling=True, height_pooling=False, indice_key="down5")
        self.upBlock0 = UpBlock(16 * s, 16 * s, indice_key="up0", up_key="down5")
        self.upBlock1 = UpBlock(16 * s, 8 * s, indice_key="up1", up_key="down4")
        self.upBlock2 = UpBlock(8 * s, 4 * s, indice_key="up2", up_key="down3")
        self.upBlock3 = UpBlock(4 * s, 2 * s, indice_key="up3", up_key="down2")
        self.ReconNet = ReconBlock(2 * s, 2 * s, indice_key="recon")

    def trunk(self, x):
        x = self.downCntx(x)
        skips = []
        for blk in (self.resBlock2, self.resBlock3, self.resBlock4, self.resBlock5):
            x, skip = blk(x)
            skips.append(skip)
        for blk in (self.upBlock0, self.upBlock1, self.upBlock2, self.upBlock3):
            x = blk(x, skips.pop())
        # cat(recon, up1e): ReconNet writes its half of the concat buffer, the other half is one strided copy
        c = x.features.shape[1]
        cat = torch.empty((x.features.shape[0], 2 * c), dtype=x.features.dtype, device=x.features.device)
        cat[:, c:] = x.features
        y = self.ReconNet(x, out=cat[:, :c])
        y.features = cat
        return y


def _sparse_input(batch_dict, shape):
    return spconv.SparseConvTensor(features=batch_dict["voxel_features"], indices=batch_dict["voxel_coords"].int(), spatial_shape=shape,
                                   batch_size=batch_dict["batch_size"])


@BACKBONES.register_module
class UNetCylinder3D(_AsymmTrunk):
    """scn_unet_cylinder3d.py:257-335: Cylinder3D's asymmetric UNet behind UNetSCN3D's batch_dict contract"""

    def __init__(self, num_input_features=128, name="UNetCylinder3D", voxel_size=[], point_cloud_range=[], model_cfg={}, **kwargs):
        super().__init__()
        self.model_cfg, self.voxel_size, self.point_cloud_range = model_cfg, voxel_size, point_cloud_range
        self._build_trunk(num_input_features, model_cfg["init_size"])

    def forward(self, batch_dict):
        out = self.trunk(_sparse_input(batch_dict, np.array(batch_dict["input_shape"][::-1]) + [1, 0, 0]))
        batch_dict["conv_point_features"] = out.features
        batch_dict["conv_point_coords"] = ops.voxel_centers(out.indices, self.voxel_size, self.point_cloud_range, n_dev=out.n_dev)
        return batch_dict


@BACKBONES.register_module
class Cylinder3D_Asymm_3d_spconv(_AsymmTrunk):
    """cylinder3d_backbone.py:254-338: the original network - a biased 3x3x3 logits convolution, densified to [B, classes, X, Y, Z]"""

    def __init__(self, output_shape, use_norm=True, num_input_features=128, nclasses=20, n_height=32, strict=False, init_size=16):
        super().__init__()
        self.nclasses, self.nheight, self.strict = nclasses, n_height, False
        self.sparse_shape = np.array(np.array(output_shape)[::-1])
        self._build_trunk(num_input_features, init_size)
        self.logits = spconv.SubMConv3d(4 * init_size, nclasses, indice_key="logit", kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, batch_dict):
        y = self.logits(self.trunk(_sparse_input(batch_dict, self.sparse_shape))).dense()
        batch_dict["voxel_features"] = y.permute(0, 1, 4, 3, 2)  # the PolarNet heads' [B, C, X, Y, Z] layout
        return batch_dict


@BACKBONES.register_module
class Cylinder3D_Asymm_3d_spconv_v2p(_AsymmTrunk):
    """cylinder3d_backbone.py:341-442: voxel features + the voxel centres mapped back from (rho, phi, z) to Cartesian coordinates"""

    def __init__(self, num_input_features=128, name="Cylinder3D_Asymm_3d_spconv_v2p", grid_size=[], point_cloud_range=[], model_cfg={}, **kwargs):
        super().__init__()
        self.model_cfg, self.grid_size, self.point_cloud_range = model_cfg, grid_size, point_cloud_range
        self.voxel_size = [(point_cloud_range[3 + i] - point_cloud_range[i]) / grid_size[i] for i in range(3)]
        self._build_trunk(num_input_features, model_cfg["init_size"])

    def forward(self, batch_dict):
        out = self.trunk(_sparse_input(batch_dict, np.array(batch_dict["input_shape"][::-1]) + [1, 0, 0]))
        batch_dict["conv_point_features"] = out.features
        cyl = ops.voxel_centers(out.indices, self.voxel_size, self.point_cloud_range, n_dev=out.n_dev)  # (b, rho, phi, z) of the cell centres
        batch_dict["conv_point_coords"] = torch.stack([cyl[:, 0], cyl[:, 1] * torch.cos(cyl[:, 2]), cyl[:, 1] * torch.sin(cyl[:, 2]), cyl[:, 3]], dim=1)
        return batch_dict
