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
