"""build_* helpers mirroring det3d/models/builder.py:19-63 (list cfg -> nn.Sequential; build_detector
injects train_cfg / test_cfg)."""
from torch import nn

from .registry import (BACKBONES, DETECTORS, HEADS, IMG_BACKBONES, IMG_HEADS, LOSSES, NECKS, POINT_HEADS, READERS,
                       ROI_HEAD, SECOND_STAGE, build_from_cfg)


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def _maker(registry):
    return lambda cfg: build(cfg, registry)


build_second_stage_module = _maker(SECOND_STAGE)
build_roi_head = _maker(ROI_HEAD)
build_reader = _maker(READERS)
build_backbone = _maker(BACKBONES)
build_img_backbone = _maker(IMG_BACKBONES)
build_img_head = _maker(IMG_HEADS)
build_neck = _maker(NECKS)
build_head = _maker(HEADS)
build_loss = _maker(LOSSES)
build_point_head = _maker(POINT_HEADS)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
