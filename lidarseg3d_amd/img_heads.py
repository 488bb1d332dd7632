"""Camera-side glue of MSeg3D that sits next to the point head.  The HRNet backbone and the FCN head's
convolutions stay outside (dense MIOpen work); what feeds the SF-Phase is here."""
from torch import nn

from . import ops


class CameraSemanticFeatureAggregationModule(nn.Module):
    """det3d/models/img_heads/fcn_mseg3d_head.py:17-51: class-wise semantic embeddings of the multi-camera feature maps,
    softmax over all pixels of all cameras of a frame.  forward(_feats [B*ncam, C, h, w], _probs [B*ncam, cls, h, w],
    batch_size) -> [B, C, cls, 1]"""

    def forward(self, _feats, _probs, batch_size):
        return ops.camera_sfam(_feats.contiguous(), _probs.contiguous(), int(batch_size))
