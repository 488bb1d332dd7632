"""Voxel feature extractors registered under the reference's names and constructor signatures
(det3d/models/readers/voxel_encoder.py:39-270).  call: reader(features[V,P,C], num_voxels[V], coors=None) -> [V,C']

Forward runs on libls3d kernels only: the per-voxel descriptor / attention / pooling kernels of csrc/vfe.hip and
the MFMA gather-GEMM (csrc/spconv.hip) for every projection."""
import torch
from torch import nn

from . import ops
from .packing import PackedModule, pack_linear
from .registry import READERS

import os as _os
_FUSED = True  # False: compose the TransVFE from the individual ops (A/B, tests)


@READERS.register_module
class MeanVoxelFeatureExtractor(nn.Module):
    """voxel_encoder.py:39-58"""

    def __init__(self, num_input_features=4, name="MeanVoxelFeatureExtractor"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None, n_dev=None):
        """n_dev (all readers; not in the reference): device count of the valid voxels when `features` has spare rows (capacity mode)"""
        assert self.num_input_features == features.shape[-1]
        return ops.vfe_mean(features.contiguous(), num_voxels.to(torch.int32).contiguous(), n_dev=n_dev)


@READERS.register_module
class ImprovedMeanVoxelFeatureExtractor(nn.Module):
    """voxel_encoder.py:62-124 — 13-/12-channel descriptor (mean, max, min, density, std)."""

    def __init__(self, num_input_features=4, norm_cfg=None, name="ImprovedMeanVoxelFeatureExtractor"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None, out_ld=None, n_dev=None):
        assert self.num_input_features == features.shape[-1]
        return ops.vfe_improved_mean(features.contiguous(), num_voxels.to(torch.int32).contiguous(), out_ld=out_ld, n_dev=n_dev)


class TransformerEncoderLayerPreNorm(nn.Module):
    """parameter container with the reference's attribute names (voxel_encoder.py:128-147);
    the arithmetic lives in TransformerVoxelFeatureExtractor.forward."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _LayerStack(nn.Module):
    """stands in for nn.TransformerEncoder: same `.layers.{i}` state_dict keys, norm=None."""

    def __init__(self, make_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])


@READERS.register_module
class TransformerVoxelFeatureExtractor(PackedModule):
    """TransVFE (voxel_encoder.py:166-270): per voxel, its <=5 points are tokens [point feats | descriptor];
    Conv1d(k=1) embed -> num_layers x pre-norm transformer layers (residual taken from the NORMED tensor,
    :154-161, no padding mask) -> max over the tokens -> Linear+ReLU compression."""

    def __init__(self, num_input_features=4, num_compressed_features=16, num_embed=64, num_head=4, num_layers=2,
                 norm_cfg=None, name="TransformerVoxelFeatureExtractor"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features
        self.num_embed, self.num_head = num_embed, num_head
        n_desc = num_input_features + 3 + 3 + 1 + 1
        self.feature_conv = nn.Sequential(nn.Conv1d(num_input_features + n_desc, num_embed, 1, bias=True))
        self.chunck = _LayerStack(lambda: TransformerEncoderLayerPreNorm(num_embed, num_head, num_embed * 2, dropout=0),
                                  num_layers)
        if num_compressed_features > 0:
            self.compress_layer = nn.Sequential(nn.Linear(num_embed, num_compressed_features), nn.ReLU())
            self.num_out_features = num_compressed_features
        else:
            self.compress_layer = None
            self.num_out_features = num_embed

    def _pack(self):
        conv = self.feature_conv[0]
        p = dict(embed=pack_linear(conv.weight, conv.bias), layers=[])
        for l in self.chunck.layers:
            p["layers"].append(dict(
                qkv=pack_linear(l.self_attn.in_proj_weight, l.self_attn.in_proj_bias),
                out=pack_linear(l.self_attn.out_proj.weight, l.self_attn.out_proj.bias),
                ff1=pack_linear(l.linear1.weight, l.linear1.bias), ff2=pack_linear(l.linear2.weight, l.linear2.bias),
                n1=(l.norm1.weight.detach().contiguous(), l.norm1.bias.detach().contiguous(), l.norm1.eps),
                n2=(l.norm2.weight.detach().contiguous(), l.norm2.bias.detach().contiguous(), l.norm2.eps)))
        if self.compress_layer is not None:
            p["compress"] = pack_linear(self.compress_layer[0].weight, self.compress_layer[0].bias)
        p["fused"] = self._fused_model(p)
        return p

    def _fused_model(self, p):
        """description for the one-kernel path (ops.transvfe): 64-column-slab f32 packing of every matrix + plain biases"""
        def mat(pk):
            return pk[0].for_nt(2, ops.F32)

        def vec(pk):  # pack_linear folds the bias into `shift` (no BatchNorm here: scale is None)
            assert pk[1] is None and pk[2] is not None
            return pk[2].contiguous()
        conv = self.feature_conv[0]
        if conv.bias is None or any(l.self_attn.in_proj_bias is None or l.linear1.bias is None for l in self.chunck.layers):
            return None
        layers = [dict(wqkv=mat(l["qkv"]), bqkv=vec(l["qkv"]), wo=mat(l["out"]), bo=vec(l["out"]), w1=mat(l["ff1"]), b1=vec(l["ff1"]),
                       w2=mat(l["ff2"]), b2=vec(l["ff2"]), n1=l["n1"], n2=l["n2"]) for l in p["layers"]]
        comp = None
        if self.compress_layer is not None:
            lin = self.compress_layer[0]
            comp = (lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous())
        ffn = self.chunck.layers[0].linear1.out_features if len(self.chunck.layers) else 2 * self.num_embed
        return ops.TransVFEModel((mat(p["embed"]), vec(p["embed"])), layers, comp, self.num_embed, self.num_head, ffn,
                                 p["embed"][0].shape[1])

    @staticmethod
    def _lin(x, pk, relu=False, res=None, ln=None):
        W, scale, shift, cout = pk
        return ops.gather_gemm(x, W, cout=cout, scale=scale, shift=shift, relu=relu, res_pre=res, ln=ln)

    def _forward_train(self, features, num_voxels):
        """training mode: the same computation on torch modules / autograd (dense, small GEMMs on the
        library path); the tokens (no gradient: they are inputs) still come from ls3d_vfe_tokens"""
        V, P, C = features.shape
        kt = 2 * C + 8
        tok = ops.vfe_tokens(features.contiguous(), num_voxels.to(torch.int32).contiguous(), (kt + 15) // 16 * 16)[:, :kt]
        conv = self.feature_conv[0]
        x = torch.nn.functional.linear(tok, conv.weight.squeeze(-1), conv.bias).view(V, P, self.num_embed).permute(1, 0, 2)
        for l in self.chunck.layers:  # voxel_encoder.py:149-163: the residuals start from the NORMED tensor
            x = l.norm1(x)
            x = x + l.self_attn(x, x, x, need_weights=False)[0]
            x = l.norm2(x)
            x = x + l.linear2(torch.relu(l.linear1(x)))
        x = x.max(dim=0)[0]
        return self.compress_layer(x) if self.compress_layer is not None else x

    def forward(self, features, num_voxels, coors=None, n_dev=None):
        assert self.num_input_features == features.shape[-1]
        if self.training:
            return self._forward_train(features, num_voxels)
        self._require_eval()
        pk = self.packed()
        V, P, C = features.shape
        E, H = self.num_embed, self.num_head
        if pk["fused"] is not None and _FUSED:
            y = ops.transvfe(features.contiguous(), num_voxels.to(torch.int32).contiguous(), pk["fused"], n_dev=n_dev)
            if y is not None:
                return y
        if n_dev is not None:
            raise ops.CapacityModeUnsupported("the layer-by-layer TransVFE path needs the voxel count on the host")
        # configurations the fused kernel is not specialised for: the same computation layer by layer
        tok = ops.vfe_tokens(features.contiguous(), num_voxels.to(torch.int32).contiguous(), pk["embed"][0].shape[1])
        # every LayerNorm runs in the epilogue of the GEMM that produces its input (norm1 of layer l+1 in layer l's
        # ff2 GEMM, norm1 of layer 0 in the embedding GEMM): no separate pass over the [V*5, 64] token matrix
        layers = pk["layers"]
        x = self._lin(tok, pk["embed"], ln=layers[0]["n1"] if layers else None)
        for li, lp in enumerate(layers):
            att = ops.mha_core(self._lin(x, lp["qkv"]), V, P, E, H)
            x = self._lin(att, lp["out"], res=x, ln=lp["n2"])
            nxt = layers[li + 1]["n1"] if li + 1 < len(layers) else None
            x = self._lin(self._lin(x, lp["ff1"], relu=True), lp["ff2"], res=x, ln=nxt)
        x = ops.group_max(x, V, P)
        if self.compress_layer is not None:
            x = self._lin(x, pk["compress"], relu=True)
        return x


# ---------------------------------------------------------------------------------------------------------------- dynamic (point-wise) readers
class _CylindricalDynamicReader(PackedModule):
    """What PolarNetDynamicVoxelFeatureExtractor and Cylinder3DDynamicVoxelFeatureExtractor share (voxel_encoder.py:275-497,503-720):
    points -> (rho, phi, z) -> cells of a cylindrical grid (clamped, so every point is kept) -> torch.unique of the cell rows ->
    per point [cyl, x, y, extras, offsets from the voxel mean and the cell centre] -> BatchNorm + 3 x (Linear, BN, ReLU) + Linear ->
    scatter mean / max per voxel -> optional compression.  On the device: ls3d_cyl_voxelize, the in-library radix sort +
    ls3d_unique_sorted, ls3d_segment_reduce, ls3d_dyn_point_features and the MFMA gather-GEMM for every Linear (csrc/dynreader.hip,
    voxelize.hip, spconv.hip); the leading BatchNorm is folded into the feature kernel, the others into the GEMM epilogues."""
    _reverse = False   # cell columns stored (c2, c1, c0): the (z, y, x) order spconv wants
    _collapse = False  # group whole (rho, phi) columns (PolarNet's bird's-eye-view pillars)

    def __init__(self, grid_size, point_cloud_range, average_points, num_input_features, num_output_features, fea_compre=None, voxel_label_enc=None,
                 **kwargs):
        super().__init__()
        fea_dim = num_input_features + 2 + 8
        self.PPmodel = nn.Sequential(nn.BatchNorm1d(fea_dim),
                                     nn.Linear(fea_dim, 64), nn.BatchNorm1d(64), nn.ReLU(inplace=True),
                                     nn.Linear(64, 128), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                     nn.Linear(128, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                     nn.Linear(256, num_output_features))
        self.pool_dim = num_output_features
        self.fea_compre = fea_compre
        if fea_compre is not None:
            self.fea_compression = nn.Sequential(nn.Linear(self.pool_dim, fea_compre), nn.ReLU())
            self.pt_fea_dim = fea_compre
        else:
            self.pt_fea_dim = self.pool_dim
        self.grid_size, self.point_cloud_range, self.average_points = grid_size, point_cloud_range, average_points
        self.voxel_size = [(point_cloud_range[3 + i] - point_cloud_range[i]) / grid_size[i] for i in range(3)]
        self.voxel_label_enc = voxel_label_enc
        self._fea_dim = fea_dim

    def _pack(self):
        from .packing import fold_bn
        pp = self.PPmodel
        s0, t0 = fold_bn(None, pp[0], self._fea_dim, pp[0].weight.device)
        p = dict(in_scale=s0, in_shift=t0, mlp=[pack_linear(pp[1].weight, pp[1].bias, pp[2]), pack_linear(pp[4].weight, pp[4].bias, pp[5]),
                                                 pack_linear(pp[7].weight, pp[7].bias, pp[8]), pack_linear(pp[10].weight, pp[10].bias)])
        if self.fea_compre is not None:
            p["compress"] = pack_linear(self.fea_compression[0].weight, self.fea_compression[0].bias)
        return p

    def voxelize_labels(self, point_labels, point_vcoors):
        """majority label per voxel (voxel_encoder.py:388-406,618-636), in torch.unique's voxel order; ties go to the smallest label"""
        from . import scatter
        lbxyz = torch.cat([point_labels.reshape(-1, 1).to(point_vcoors.dtype), point_vcoors], dim=-1)
        unq, count = torch.unique(lbxyz, return_counts=True, dim=0)
        inv = torch.unique(unq[:, 1:], return_inverse=True, dim=0)[1]
        if self.voxel_label_enc != "major":
            raise AssertionError("voxel_label_enc %r" % (self.voxel_label_enc,))
        return unq[:, 0][scatter.scatter_max(count, inv)[1]]

    def _point_rows(self, points, batch_size):
        g = [int(v) for v in self.grid_size]
        cyl5, vcoors, keys = ops.cyl_voxelize(points, g, self.point_cloud_range, self._reverse, self._collapse, batch_size)
        rows, inverse, counts = ops.unique_rows(keys, g[::-1] if self._reverse else g, batch_size)
        mean5 = ops.segment_reduce(cyl5, inverse, rows.shape[0], "mean")
        return cyl5, vcoors, rows, inverse, counts, mean5

    def _voxel_features(self, points, batch_size):
        """-> voxel features [V, C], unique cell rows [V, 4], per-point rows [n, 4], inverse [n], counts [V]"""
        cyl5, vcoors, rows, inverse, counts, mean5 = self._point_rows(points, batch_size)
        g = [int(v) for v in self.grid_size]
        if self.training:
            feats = ops.dyn_point_features(points, cyl5, vcoors, inverse, mean5, g, self.point_cloud_range)[:, :self._fea_dim]
            x = self.PPmodel(feats)
            idx = inverse[:, None].expand(-1, x.shape[1])
            pooled = torch.zeros((rows.shape[0], x.shape[1]), dtype=x.dtype, device=x.device).scatter_reduce(
                0, idx, x, "mean" if self.average_points else "amax", include_self=False)
            if self.fea_compre:
                pooled = self.fea_compression(pooled)
            return pooled, rows, vcoors, inverse, counts
        pk = self.packed()
        ld = pk["mlp"][0][0].shape[1]
        x = ops.dyn_point_features(points, cyl5, vcoors, inverse, mean5, g, self.point_cloud_range, pk["in_scale"], pk["in_shift"], ld)
        for i, (W, scale, shift, cout) in enumerate(pk["mlp"]):
            x = ops.gather_gemm(x, W, cout=cout, scale=scale, shift=shift, relu=i < 3)
        pooled = ops.segment_reduce(x, inverse, rows.shape[0], "mean" if self.average_points else "max")
        if self.fea_compre:
            W, scale, shift, cout = pk["compress"]
            pooled = ops.gather_gemm(pooled, W, cout=cout, scale=scale, shift=shift, relu=True)
        return pooled, rows, vcoors, inverse, counts

    def _labels(self, batch_dict, vcoors):
        labels = batch_dict.get("point_sem_labels")
        if self.voxel_label_enc is not None and labels is not None:
            batch_dict["voxel_sem_labels"] = self.voxelize_labels(labels, vcoors)


@READERS.register_module
class PolarNetDynamicVoxelFeatureExtractor(_CylindricalDynamicReader):
    """voxel_encoder.py:275-497: pillars of a polar bird's-eye-view grid -> dense [B, C, grid0, grid1] map for PolarNet's 2-D UNet"""
    _collapse = True

    def forward(self, batch_dict):
        points, batch_size = batch_dict["points"].float().contiguous(), batch_dict["batch_size"]
        feats, rows, vcoors, inverse, counts = self._voxel_features(points, batch_size)
        g = [int(v) for v in self.grid_size]
        bev = torch.zeros((batch_size, g[0], g[1], feats.shape[-1]), dtype=feats.dtype, device=feats.device)
        bev[rows[:, 0], rows[:, 1], rows[:, 2], :] = feats
        batch_dict["voxel_features"] = bev.permute(0, 3, 1, 2)
        batch_dict["point_vcoors"] = vcoors
        batch_dict["input_shape"] = self.grid_size
        batch_dict["num_points_in_voxel"] = counts
        self._labels(batch_dict, vcoors)
        return batch_dict


@READERS.register_module
class Cylinder3DDynamicVoxelFeatureExtractor(_CylindricalDynamicReader):
    """voxel_encoder.py:503-720: sparse voxels of the cylindrical grid, rows (batch, z, y, x) in torch.unique's sorted order"""
    _reverse = True

    def forward(self, batch_dict):
        points, batch_size = batch_dict["points"].float().contiguous(), batch_dict["batch_size"]
        feats, rows, vcoors, inverse, counts = self._voxel_features(points, batch_size)
        batch_dict["voxel_features"] = feats
        batch_dict["point_vcoors"] = vcoors[:, [0, 3, 2, 1]]  # (b, vz, vy, vx) -> (b, vx, vy, vz) for the dense-to-sparse mapping of the head
        batch_dict["input_shape"] = self.grid_size
        batch_dict["voxel_coords"] = rows
        batch_dict["num_points_in_voxel"] = counts
        self._labels(batch_dict, vcoors)
        return batch_dict
