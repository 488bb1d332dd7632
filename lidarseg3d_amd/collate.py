"""collate_kitti for the keys of the segmentation path (det3d/torchie/parallel/collate.py:91-170): a list of per-frame
`example` dicts (numpy arrays, what the dataset pipeline emits) -> the batched `example` the detectors consume.  This is the
on-wire format of the hot path: per-frame arrays concatenated along dim 0, `points` / `coordinates` with the frame index
prepended as column 0, images stacked.  Detection-only keys (anchors, gt_boxes, heat maps, calib, ...) are not handled.

`collate_points` is the GPU-native variant: frames already on the device -> one `[sum N, 1 + C]` tensor."""
import collections

import numpy as np
import torch

CONCAT_KEYS = ("voxels", "num_points", "num_voxels", "voxel_sem_labels", "point_sem_labels", "voxel_inst_labels",
               "point_inst_labels", "points_cuv", "points_cp")
FRAME_INDEX_KEYS = ("coordinates", "points", "all_points")
STACK_TENSOR_KEYS = ("images", "images_sem_labels")
UNSUPPORTED_KEYS = ("gt_boxes", "calib", "anchors", "anchors_mask", "reg_targets", "reg_weights", "labels", "hm", "anno_box", "ind",
                    "mask", "cat", "gt_boxes_and_cls")


def collate_kitti(batch_list, samples_per_gpu=1):
    merged = collections.defaultdict(list)
    for example in batch_list:
        for sub in (example if isinstance(example, list) else [example]):  # TTA: a frame may be a list of variants
            for k, v in sub.items():
                merged[k].append(v)
    ret = {}
    for key, elems in merged.items():
        if key in CONCAT_KEYS:
            ret[key] = torch.tensor(np.concatenate(elems, axis=0))
        elif key in FRAME_INDEX_KEYS:
            ret[key] = torch.tensor(np.concatenate(
                [np.concatenate([np.full((e.shape[0], 1), i, dtype=e.dtype), e], axis=1) for i, e in enumerate(elems)], axis=0))
        elif key in STACK_TENSOR_KEYS:
            ret[key] = torch.tensor(np.stack(elems, axis=0))
        elif key == "metadata":
            ret[key] = elems
        elif key in UNSUPPORTED_KEYS:
            raise NotImplementedError("collate_kitti: %r is a detection key; only the segmentation path is covered" % key)
        else:
            ret[key] = np.stack(elems, axis=0)  # e.g. "shape": one grid size per frame
    return ret


def collate_points(frames):
    """frames: list of [N_i, C] tensors on one device -> [sum N_i, 1 + C] with the frame index in column 0"""
    return torch.cat([torch.cat([f.new_full((f.shape[0], 1), float(i)), f], dim=1) for i, f in enumerate(frames)], dim=0)
