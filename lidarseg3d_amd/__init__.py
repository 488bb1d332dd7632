"""lidarseg3d_amd — the MSeg3D / SDSeg3D segmentation forward path of jialeli1/lidarseg3d as hand-written HIP
kernels for MI355X (gfx950), behind the reference's registry / module API.

    from lidarseg3d_amd import build_detector        # == det3d.models.build_detector
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg).cuda().eval()

Layout: csrc/ (HIP kernels + C ABI, built into libls3d.so by lidarseg3d_amd.build), _lib/ops (ctypes binding),
registry/builder/readers/spconv/scn_unet/point_heads/detectors (host-side mirror of the reference interface).
"""
from . import registry  # noqa: F401
from .builder import (build_backbone, build_detector, build_point_head, build_reader)  # noqa: F401
from .registry import (BACKBONES, DETECTORS, POINT_HEADS, READERS, Registry, build_from_cfg)  # noqa: F401


def register_all():
    """import the modules that register the hot-path components (needs torch; loads libls3d lazily on first op)"""
    from . import detectors, img_heads, point_heads, readers, scn_unet  # noqa: F401


register_all()
