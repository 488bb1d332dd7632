"""Make reference-side code find this package under the reference's import names.

    import lidarseg3d_amd.compat as compat; compat.install()
    from det3d.models import build_detector          # -> lidarseg3d_amd.builder.build_detector
    from det3d.torchie import Config
    import spconv                                     # -> lidarseg3d_amd.spconv (v1-shaped namespace)
    from det3d.ops.pointnet2_batch import pointnet2_utils
    from det3d.ops.voxel import Voxelization, DynamicScatter
    import torch_scatter                              # -> lidarseg3d_amd.scatter, only when torch_scatter is not installed

Only the hot-path surface is provided; asking for anything else raises ImportError as usual.  Refuses to install over
a real `det3d` / `spconv` already imported."""
import sys
import types


def install(force=False):
    from . import builder, checkpoint, config, pointnet2_utils, registry, spconv, voxel_ops
    if not force:
        for name in ("det3d", "spconv"):
            if name in sys.modules and not getattr(sys.modules[name], "__ls3d_alias__", False):
                raise RuntimeError("%s is already imported; refusing to shadow it" % name)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__ls3d_alias__ = True
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    reg_names = ("READERS", "BACKBONES", "IMG_BACKBONES", "IMG_HEADS", "NECKS", "HEADS", "LOSSES", "DETECTORS", "SECOND_STAGE",
                 "ROI_HEAD", "POINT_HEADS")
    regs = {n: getattr(registry, n) for n in reg_names}
    builders = {n: getattr(builder, n) for n in dir(builder) if n.startswith("build")}
    det3d = mod("det3d")
    det3d.utils = mod("det3d.utils", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    mod("det3d.utils.registry", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    det3d.models = mod("det3d.models", **regs, **builders)
    mod("det3d.models.registry", **regs)
    mod("det3d.models.builder", **builders)
    det3d.torchie = mod("det3d.torchie", Config=config.Config, is_str=lambda x: isinstance(x, str))
    mod("det3d.torchie.trainer", load_checkpoint=checkpoint.load_checkpoint)
    from . import collate
    mod("det3d.torchie.parallel", collate_kitti=collate.collate_kitti)
    det3d.ops = mod("det3d.ops")
    mod("det3d.ops.pointnet2_batch", pointnet2_utils=pointnet2_utils)
    sys.modules["det3d.ops.pointnet2_batch.pointnet2_utils"] = pointnet2_utils
    mod("det3d.ops.voxel", Voxelization=voxel_ops.Voxelization, DynamicScatter=voxel_ops.DynamicScatter)
    from . import img_heads
    mod("det3d.models.img_heads", CameraSemanticFeatureAggregationModule=img_heads.CameraSemanticFeatureAggregationModule,
        FCNMSeg3DHead=img_heads.FCNMSeg3DHead)
    mod("det3d.models.img_heads.fcn_mseg3d_head", CameraSemanticFeatureAggregationModule=img_heads.CameraSemanticFeatureAggregationModule,
        FCNMSeg3DHead=img_heads.FCNMSeg3DHead)
    sys.modules["spconv"] = spconv
    spconv.__ls3d_alias__ = True
    try:  # the dynamic readers' scatter_mean / scatter_max
        import torch_scatter  # noqa: F401
    except ImportError:
        from . import scatter
        sys.modules["torch_scatter"] = scatter
    try:  # the reference's config files do `from addict.addict import Dict` only to have it in scope
        import addict  # noqa: F401
    except ImportError:
        a = mod("addict", Dict=config.ConfigDict)
        a.addict = mod("addict.addict", Dict=config.ConfigDict)
    return det3d
