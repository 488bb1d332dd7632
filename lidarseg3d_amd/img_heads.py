"""Camera-side glue of MSeg3D that sits next to the point head.  The HRNet backbone stays outside (dense MIOpen
work); what turns its multi-level maps into the inputs of the SF-Phase is here: the FCN head's 1x1 convolutions (per-pixel GEMMs over
the 6 x 160 x 240 pixels of a frame), the pixel classifier and the camera semantic-feature aggregation."""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .packing import PackedModule, pack_linear
from .registry import IMG_HEADS


class CameraSemanticFeatureAggregationModule(nn.Module):
    """det3d/models/img_heads/fcn_mseg3d_head.py:17-51: class-wise semantic embeddings of the multi-camera feature maps,
    softmax over all pixels of all cameras of a frame.  forward(_feats [B*ncam, C, h, w], _probs [B*ncam, cls, h, w],
    batch_size) -> [B, C, cls, 1]"""

    def forward(self, _feats, _probs, batch_size):
        return ops.camera_sfam(_feats.contiguous(), _probs.contiguous(), int(batch_size))


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule as fcn_mseg3d_head.py:84-137 instantiates it (conv without bias -> BatchNorm2d -> ReLU; attribute names
    `conv` / `bn` so that the reference checkpoints load)"""

    def __init__(self, cin, cout, kernel_size, padding=0, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):  # torch path: kernel sizes other than 1 and training
        return F.relu(self.bn(self.conv(x)))


@IMG_HEADS.register_module
class FCNMSeg3DHead(PackedModule):
    """det3d/models/img_heads/fcn_mseg3d_head.py:54-200 (+ decode_head.py:57-141,213-218) with the shipped configuration's shape
    (configs/semanticnusc/MSeg3D/fcn_cfg.py + semnusc_avgvfe_unetscn3d_hrnetw18_lr1en2_e12.py:41-52: in_channels [18, 36, 72, 144],
    input_transform 'resize_concat', kernel_size 1, num_convs 2, channels 48, concat_input False, dropout off).

    forward(batch_dict) reads `inputs` (the list of HRNet level maps [B*ncam, Ci, hi, wi]) and `batch_size`, writes `image_features`
    [B*ncam, channels, h, w], `image_logits` [B*ncam, num_classes, h, w] and `camera_semantic_embeddings` [B, channels, num_classes, 1].
    In eval mode with kernel_size 1 the whole head after the bilinear resize is per-pixel work on the HIP kernels: the concatenated
    maps go channels-last once, ConvModule = ls3d_gather_gemm (dense) with the BatchNorm + ReLU epilogue, conv_seg the same with its
    bias, and the aggregation is the SFAM kernels on the same rows (no NCHW round trip in between).  Other kernel sizes, use_sc_conv
    and training run the torch composition of the same modules (dense convolutions: MIOpen's job)."""

    def __init__(self, num_convs=2, kernel_size=3, concat_input=True, dilation=1, ignore_index=0, loss_weight=1.0, lovasz_loss_weight=-1.0,
                 use_sc_conv=False, in_channels=None, channels=None, num_classes=None, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), in_index=-1, input_transform=None, align_corners=False, **kwargs):
        super().__init__()
        assert num_convs >= 0 and dilation > 0 and isinstance(dilation, int)
        if use_sc_conv:
            raise NotImplementedError("SCBottleneck (use_sc_conv=True) is camera-CNN work outside this package")
        assert input_transform in (None, "resize_concat"), "multiple_select feeds no FCN head"
        self.num_convs, self.kernel_size, self.concat_input = num_convs, kernel_size, concat_input
        self.in_index, self.input_transform, self.align_corners = in_index, input_transform, align_corners
        self.in_channels = sum(in_channels) if input_transform == "resize_concat" else in_channels
        self.channels, self.num_classes, self.ignore_index = channels, num_classes, ignore_index
        self.loss_weight, self.lovasz_loss_weight = loss_weight, lovasz_loss_weight
        if num_convs == 0:
            assert self.in_channels == channels
        pad = (kernel_size // 2) * dilation
        convs = [_ConvModule(self.in_channels if i == 0 else channels, channels, kernel_size, pad, dilation) for i in range(num_convs)]
        self.convs = nn.Sequential(*convs) if num_convs else nn.Identity()
        if concat_input:
            self.conv_cat = _ConvModule(self.in_channels + channels, channels, kernel_size, kernel_size // 2)
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.camera_sfam = CameraSemanticFeatureAggregationModule()
        self.forward_ret_dict = {}

    # ---- decode_head.py:141-165
    def _transform_inputs(self, inputs):
        if self.input_transform == "resize_concat":
            inputs = [inputs[i] for i in self.in_index]
            size = inputs[0].shape[2:]
            return torch.cat([x if x.shape[2:] == size else F.interpolate(x, size=size, mode="bilinear", align_corners=self.align_corners)
                              for x in inputs], dim=1)
        return inputs[self.in_index]

    def _pack(self):
        mods = list(self.convs) if self.num_convs else []
        return dict(convs=[pack_linear(m.conv.weight.reshape(m.conv.out_channels, -1), None, m.bn) for m in mods],
                    cat=pack_linear(self.conv_cat.conv.weight.reshape(self.channels, -1), None, self.conv_cat.bn) if self.concat_input else None,
                    seg=pack_linear(self.conv_seg.weight.reshape(self.num_classes, -1), self.conv_seg.bias))

    @staticmethod
    def _gemm(rows, packed, relu):
        w, scale, shift, cout = packed
        if rows.shape[1] != w.shape[1]:
            rows = F.pad(rows, (0, w.shape[1] - rows.shape[1]))
        return ops.gather_gemm(rows, w, cout=cout, scale=scale, shift=shift, relu=relu)

    def forward(self, batch_dict, return_loss=True, **kwargs):
        x = self._transform_inputs(batch_dict["inputs"])
        bn, _, h, w = x.shape
        bs = int(batch_dict["batch_size"])
        # the fused HIP path is not differentiable: a frozen head in eval mode whose input carries a gradient (camera backbone being
        # trained, input-gradient analysis) takes the torch composition below
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()) and self.training)
        if self.kernel_size == 1 and not self.training and not wants_grad and (x.is_cuda or ops.sim_mode()):
            pk = self.packed()
            rows0 = ops.nchw_to_nhwc(x.contiguous())  # [bn * h * w, Cin]: pixels are rows from here on
            rows = rows0
            for p in pk["convs"]:
                rows = self._gemm(rows, p, True)
            if self.concat_input:
                rows = self._gemm(torch.cat([rows0, rows], 1), pk["cat"], True)
            logit_rows = self._gemm(rows, pk["seg"], False)
            per = (bn // bs) * h * w
            off = torch.arange(0, (bs + 1) * per, per, dtype=torch.int32, device=x.device)
            emb = ops.sfam(rows, logit_rows, off, bs, per).permute(0, 2, 1).contiguous().unsqueeze(3)
            # contiguous NCHW like the reference's outputs (fcn_mseg3d_head.py:175-200): downstream .view() calls work
            feature = rows.view(bn, h, w, self.channels).permute(0, 3, 1, 2).contiguous()
            output = logit_rows.view(bn, h, w, self.num_classes).permute(0, 3, 1, 2).contiguous()
        else:
            feature = self.convs(x)
            if self.concat_input:
                feature = self.conv_cat(torch.cat([x, feature], dim=1))
            output = self.conv_seg(self.dropout(feature) if self.dropout is not None else feature)
            emb = self.camera_sfam(feature, output, bs)
        self.forward_ret_dict.update({"image_logits": output})
        if return_loss:
            self.forward_ret_dict.update({"image_sem_labels": batch_dict["images_sem_labels"]})
        batch_dict["image_logits"], batch_dict["image_features"], batch_dict["camera_semantic_embeddings"] = output, feature, emb
        return batch_dict

    def get_loss(self, image_loss_dict=None):
        """fcn_mseg3d_head.py:203-241: point-to-pixel cross-entropy on the camera logits resized to the label maps (+ Lovasz when
        weighted; the logged entries carry their weights, as in the reference)"""
        from .losses import lovasz_softmax
        parts = {} if image_loss_dict is None else image_loss_dict
        logits, labels = self.forward_ret_dict["image_logits"], self.forward_ret_dict["image_sem_labels"]
        logits = F.interpolate(logits, size=labels.shape[2:], mode="bilinear", align_corners=self.align_corners)
        target = labels.squeeze(1).long()
        ce = self.loss_weight * F.cross_entropy(logits, target, ignore_index=self.ignore_index)
        loss = ce
        parts["image_ce_loss"] = ce.detach()
        if self.lovasz_loss_weight > 0:
            flat = torch.softmax(logits, dim=1).permute(0, 2, 3, 1).reshape(-1, logits.shape[1])
            lv = self.lovasz_loss_weight * lovasz_softmax(flat, target.reshape(-1))
            loss = loss + lv
            parts["image_lvsz_loss"] = lv.detach()
        return loss, parts
