"""SegNet (SDSeg3D) and SegMSeg3DNet (MSeg3D) detectors with the reference's registry names and constructor
signatures (det3d/models/detectors/seg_net.py:11-107, seg_mseg3d_net.py:6-147, single_stage.py:10-33).

forward(example, return_loss=False) consumes the collated `example` dict of the reference dataloader
(det3d/torchie/parallel/collate.py:91-170) and returns point_head.predict(...)'s per-frame list.
Two input modes:
  * drop-in: example carries the dataloader's CPU-voxelised `voxels / coordinates / num_points` -> used as is;
  * MI355X-native: example carries only `points` ([N,1+C], batch index in column 0) -> hard voxelization runs
    on the GPU (csrc/voxelize.hip), bit-exact with the dataloader's numba kernel.
The camera CNN (HRNet + FCN head) is outside the hot-path scope: SegMSeg3DNet takes `image_features`
[B,ncam,C,h,w] and `camera_semantic_embeddings` [B,C,num_cls,1] from `example` unless an img_backbone /
img_head registered by the user is configured."""
import numpy as np
import torch
from torch import nn

from . import builder, ops
from .registry import DETECTORS


import os as _os
# Capacity mode (inference from `points`): the frame runs on device-side row counts - voxel count, strided-rulebook counts - with
# tensors sized by capacities, so the host never waits for the GPU inside a frame and can submit the next frame while this one
# runs.  LS3D_CAPACITY_MODE=0 restores host-side counts (three synchronisations per frame).  Results are bit-identical.
CAPACITY_MODE = _os.environ.get("LS3D_CAPACITY_MODE", "1") != "0"


def _voxel_inputs(example, voxel_cfg, capacity=False):
    """-> voxels, coordinates[V,4], num_points, batch_size, input_shape(x,y,z), n_dev (device voxel count when the tensors carry
    spare rows - capacity mode - else None)"""
    if "voxels" in example:
        shape = example["shape"][0] if "shape" in example else ops.make_grid(voxel_cfg["voxel_size"], voxel_cfg["range"])[1]
        return (example["voxels"], example["coordinates"], example["num_points"], len(example["num_voxels"]),
                np.asarray(shape), None)
    if voxel_cfg is None:
        raise KeyError("example has no 'voxels' and the detector was built without a voxel_generator cfg")
    points = example["points"].contiguous()
    batch_size = int(example["batch_size"]) if "batch_size" in example else int(points[:, 0].max().item()) + 1
    mv = voxel_cfg.get("max_voxel_num", 300000)
    mv = mv[1] if isinstance(mv, (list, tuple)) else mv
    mp = voxel_cfg.get("max_points_in_voxel", 5)
    _, grid = ops.make_grid(voxel_cfg["voxel_size"], voxel_cfg["range"])
    off = None
    if points.shape[0] > int(mv) and batch_size > 1 and not capacity:
        off = ops.frame_offsets(points, batch_size).tolist()  # one host sync: can any single frame reach the per-frame cap?
    if off is None or max(b - a for a, b in zip(off[:-1], off[1:])) <= int(mv):
        # no frame can reach the dataloader's per-frame cap (a frame has at most as many voxels as points): one batched launch
        # is bit-identical to voxelising frame by frame
        v, c, n, nv = ops.voxelize_hard(points, voxel_cfg["voxel_size"], voxel_cfg["range"], mp, int(mv) * batch_size, batched=True)
        if capacity:
            # capacity mode: all min(N, cap) rows stay, the count stays on the device (batches of more points than one frame's voxel cap paid
            # one host read of the frame offsets above - before anything of the frame was submitted)
            # per-frame voxel counts (predict() takes its batch size from their number): one frame's is the device count itself
            example["num_voxels"] = nv.reshape(1) if (batch_size == 1 and _LEAN_START) else ops.frame_offsets(c, batch_size, n_dev=nv).diff()
            if points.shape[0] > int(mv) and batch_size > 1:
                # the batch has more points than ONE frame's voxel cap: whether a single frame exceeds it is decided on the device (no host read, so
                # the batch can be captured: graph.FrameGraph) - the flag joins the rulebooks' overflow flags, and a frame that trips it is run
                # again on host-side counts, which caps each frame as the reference's dataloader does
                example["_voxel_overflow_dev"] = (example["num_voxels"] > int(mv)).any().to(torch.int32).reshape(1)
            return v, c, n, batch_size, np.asarray(grid), nv
        V = int(nv.item())  # one host sync per batch: downstream tensor shapes depend on it
        v, c, n = v[:V], c[:V], n[:V]
    else:
        # a frame may overflow max_voxel_num: the reference caps EACH frame (segpreprocess.py:148-177 runs per sample), so the
        # frames are voxelised one by one with that cap and concatenated as collate_kitti does (collate.py:141-150)
        parts = []
        for b in range(batch_size):
            fv, fc, fn, fnv = ops.voxelize_hard(points[off[b]:off[b + 1]].contiguous(), voxel_cfg["voxel_size"], voxel_cfg["range"], mp, int(mv),
                                                batched=True)
            k = int(fnv.item())
            parts.append((fv[:k], fc[:k], fn[:k]))
        v, c, n = (torch.cat([p[i] for p in parts]) for i in range(3))
        V = c.shape[0]
    example["num_voxels"] = ops.frame_offsets(c, batch_size).diff()
    return v, c, n, batch_size, np.asarray(grid), None


class SingleStageDetector(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        if bbox_head is not None:
            self.bbox_head = builder.build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def init_weights(self, pretrained=None):
        if pretrained is None:
            return
        from .checkpoint import load_checkpoint
        load_checkpoint(self, pretrained, strict=False)


def _with_training_kernels(model, example, return_loss):
    """training forward (autograd records): the tall-skinny weight gradients of the nn.Linear layers - reader tokens, head and
    decoder MLPs: 10^5..10^6 rows reduced into <= 256 x 256 matrices - go to ls3d_spconv_wgrad instead of hipBLASLt's 32 x 32 macro
    tiles (ops.fast_linear_backward patches torch.nn.functional.linear for the duration of the forward)"""
    if (return_loss or model.training) and torch.is_grad_enabled():
        with ops.fast_linear_backward(model):
            return model._forward(example, return_loss)
    return model._forward(example, return_loss)


def _capacity_ok(model, example, return_loss):
    """capacity mode applies to inference from raw points on the device (or under the host emulation of the tests)"""
    return (CAPACITY_MODE and not return_loss and not model.training and "voxels" not in example
            and "points" in example and hasattr(model.backbone, "geometry_check") and getattr(model, "voxel_generator", None) is not None)


def _capacity_forward(model, example, features):
    """one inference frame on device-side row counts.  -> predict()'s list, or None when the frame has to be run again with host-side
    counts (a stage without capacity support, or a rulebook that overflowed the capacity learned from earlier frames - the next
    frames start from the worst case again)"""
    ex = dict(example)
    try:
        data = features(ex)
        model.point_head(batch_dict=data, return_loss=False)
    except ops.CapacityModeUnsupported:
        return None
    if not model.backbone.geometry_check(data):
        return None
    example["num_voxels"] = ex["num_voxels"]
    return model.point_head.predict(example=example, test_cfg=model.test_cfg)


_LEAN_START = True  # A/B (module constant; experiments.py): without the three small launches between the voxelization and the reader


def _points_bxyz(points, training):
    """the (batch, x, y, z) columns of the sweep for the point head (seg_net.py:60: example["points"][:, 0:4]).  Every kernel behind it takes a row
    stride, so at inference a contiguous [N, 4 + k] sweep is handed on as it is: no copy kernel between the voxelization and the reader"""
    if _LEAN_START and not training and not torch.is_grad_enabled() and points.dim() == 2 and points.shape[1] >= 4 and points.is_contiguous():
        return points
    return points[:, 0:4].contiguous()


def _coords_ready(coords):
    """event on the current stream marking "voxel coordinates are final": the backbone builds its rulebooks on a side stream
    from this point on, concurrently with the reader (extra batch_dict key, not in the reference)"""
    if not coords.is_cuda:
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(coords.device))
    return ev


@DETECTORS.register_module
class SegNet(SingleStageDetector):
    def __init__(self, reader, backbone, point_head, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, voxel_generator=None, **kwargs):
        super().__init__(reader, backbone, neck, bbox_head, train_cfg, test_cfg, pretrained=None)
        self.point_head = builder.build_point_head(point_head)
        self.voxel_generator = voxel_generator
        self.init_weights(pretrained=pretrained)

    def forward_features(self, example, capacity=False):
        voxels, coords, num, batch_size, shape, n_dev = _voxel_inputs(example, self.voxel_generator, capacity)
        data = dict(features=voxels, num_voxels=num, voxel_coords=coords, batch_size=batch_size, input_shape=shape,
                    points=_points_bxyz(example["points"], self.training))
        data["voxel_coords_ready"] = _coords_ready(coords)
        if n_dev is not None:
            data["num_active_voxels_dev"] = n_dev
            data["voxel_overflow_dev"] = example.pop("_voxel_overflow_dev", None)
            data["voxel_features"] = self.reader(data["features"], data["num_voxels"], data["voxel_coords"], n_dev=n_dev)
        else:
            data["voxel_features"] = self.reader(data["features"], data["num_voxels"], data["voxel_coords"])
        return self.backbone(data)

    def forward(self, example, return_loss=True, **kwargs):
        return _with_training_kernels(self, example, return_loss)

    def _forward(self, example, return_loss):
        if _capacity_ok(self, example, return_loss):
            ret = _capacity_forward(self, example, lambda ex: self.forward_features(ex, capacity=True))
            if ret is not None:
                return ret
        data = self.forward_features(example)
        if return_loss:  # seg_net.py:86-103: labels in, per-task loss list + detached parts for the logger out
            data["voxel_sem_labels"], data["point_sem_labels"] = example["voxel_sem_labels"], example["point_sem_labels"]
            self.point_head(batch_dict=data, return_loss=True)
            loss, parts = self.point_head.get_loss()
            ret = dict(loss=[loss])
            ret.update({k: [v] for k, v in parts.items()})
            return ret
        self.point_head(batch_dict=data, return_loss=False)
        return self.point_head.predict(example=example, test_cfg=self.test_cfg)


@DETECTORS.register_module
class SegMSeg3DNet(SingleStageDetector):
    def __init__(self, reader, backbone, point_head, img_backbone=None, img_head=None, neck=None, bbox_head=None,
                 train_cfg=None, test_cfg=None, pretrained=None, voxel_generator=None, **kwargs):
        super().__init__(reader, backbone, neck, bbox_head, train_cfg, test_cfg, pretrained=None)
        self.img_backbone = self._build_camera(builder.build_img_backbone, img_backbone)
        self.img_head = self._build_camera(builder.build_img_head, img_head)
        self.point_head = builder.build_point_head(point_head)
        self.voxel_generator = voxel_generator
        self.init_weights(pretrained=pretrained)

    def _capacity_features(self, example):
        voxels, coords, num, batch_size, shape, n_dev = _voxel_inputs(example, self.voxel_generator, True)
        data = dict(features=voxels, num_voxels=num, voxel_coords=coords, batch_size=batch_size, input_shape=shape,
                    points=_points_bxyz(example["points"], self.training))
        data["voxel_coords_ready"] = _coords_ready(coords)
        if n_dev is None:
            raise ops.CapacityModeUnsupported("frame-by-frame voxelization")
        cam = getattr(self.point_head, "camera_branch", None)
        if cam is not None:  # depends on the frame's inputs only: beside the reader and the backbone
            data["camera_branch"] = cam(example["image_features"], example["points_cuv"], data["points"])
        data["num_active_voxels_dev"] = n_dev
        data["voxel_overflow_dev"] = example.pop("_voxel_overflow_dev", None)
        data["voxel_features"] = self.reader(data["features"], data["num_voxels"], data["voxel_coords"], n_dev=n_dev)
        data = self.backbone(data)
        data.update(points_cuv=example["points_cuv"], image_features=example["image_features"],
                    camera_semantic_embeddings=example["camera_semantic_embeddings"], metadata=example.get("metadata"))
        return data

    @staticmethod
    def _build_camera(build, cfg):
        """the camera CNN (HRNet / FCN head) is not part of this package: an unregistered type degrades to "features
        come in through `example`" with a warning instead of failing the whole config"""
        if cfg is None:
            return None
        try:
            return build(cfg)
        except KeyError as e:
            import warnings
            warnings.warn("camera branch not built (%s); SegMSeg3DNet expects example['image_features'] and "
                          "example['camera_semantic_embeddings']" % (e,))
            return None

    def forward(self, example, return_loss=True, **kwargs):
        return _with_training_kernels(self, example, return_loss)

    def _forward(self, example, return_loss):
        if _capacity_ok(self, example, return_loss) and (self.img_backbone is None or "image_features" in example):
            ret = _capacity_forward(self, example, self._capacity_features)
            if ret is not None:
                return ret
        voxels, coords, num, batch_size, shape, _ = _voxel_inputs(example, self.voxel_generator)
        if self.img_backbone is not None and "image_features" not in example:
            images = example["images"]
            ncam, hi, wi = images.shape[1], images.shape[3], images.shape[4]
            img_data = dict(inputs=self.img_backbone(images.view(-1, 3, hi, wi)), batch_size=batch_size)
            if return_loss:
                img_data["images_sem_labels"] = example["images_sem_labels"].view(-1, hi, wi).unsqueeze(1)
            img_data = self.img_head(batch_dict=img_data, return_loss=return_loss)
            camera_loss = self.img_head.get_loss if return_loss else None
            feats = img_data["image_features"]
            image_features = feats.view(batch_size, ncam, *feats.shape[1:])
            cam_emb = img_data.get("camera_semantic_embeddings")
        else:
            image_features, cam_emb, camera_loss = example["image_features"], example["camera_semantic_embeddings"], None
        data = dict(features=voxels, num_voxels=num, voxel_coords=coords, batch_size=batch_size, input_shape=shape,
                    points=_points_bxyz(example["points"], self.training))
        data["voxel_coords_ready"] = _coords_ready(coords)
        cam = getattr(self.point_head, "camera_branch", None)
        if cam is not None and not return_loss and not self.training:
            data["camera_branch"] = cam(image_features, example["points_cuv"], data["points"])
        data["voxel_features"] = self.reader(data["features"], data["num_voxels"], data["voxel_coords"])
        data = self.backbone(data)
        data.update(points_cuv=example["points_cuv"], image_features=image_features,
                    camera_semantic_embeddings=cam_emb, metadata=example.get("metadata"))
        if return_loss:  # seg_mseg3d_net.py:120-140; the image head's loss joins in when a user-registered camera branch runs
            data["voxel_sem_labels"], data["point_sem_labels"] = example["voxel_sem_labels"], example["point_sem_labels"]
            self.point_head(batch_dict=data, return_loss=True)
            loss, parts = self.point_head.get_loss()
            if camera_loss is not None:
                img_loss, parts = camera_loss(parts)
                loss = loss + img_loss
            ret = dict(loss=[loss])
            ret.update({k: [v] for k, v in parts.items()})
            return ret
        self.point_head(batch_dict=data, return_loss=False)
        return self.point_head.predict(example=example, test_cfg=self.test_cfg)
