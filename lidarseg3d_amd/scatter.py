"""scatter_mean / scatter_max with torch_scatter's call signatures for the cases the reference's dynamic readers use
(det3d/models/readers/voxel_encoder.py:4,366-372,451-456,594-600,682-686: dim 0, an int64 index per row, 1-D or 2-D src), on
ls3d_segment_reduce.  `sys.modules["torch_scatter"] = lidarseg3d_amd.scatter` (lidarseg3d_amd.compat does it when torch_scatter
is not installed) lets the reference's PolarNet / Cylinder3D dynamic readers run unmodified.  Inference only (no autograd)."""
import torch

from . import ops


def _prep(src, index, dim, dim_size):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("lidarseg3d_amd.scatter: reductions over dim 0 only")
    if index.dim() != 1 or index.shape[0] != src.shape[0]:
        raise NotImplementedError("lidarseg3d_amd.scatter: one segment id per row of src")
    n_seg = int(dim_size) if dim_size is not None else (int(index.max().item()) + 1 if index.numel() else 0)
    x = src.reshape(src.shape[0], -1)
    if x.dtype != torch.float32:
        if x.dtype.is_floating_point or (x.numel() and int(x.abs().max().item()) >= 1 << 24):
            raise NotImplementedError("lidarseg3d_amd.scatter: float32 values (or integers below 2^24)")
        x = x.float()
    return x.contiguous(), index.long().contiguous(), n_seg


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert out is None
    x, idx, n_seg = _prep(src, index, dim, dim_size)
    r = ops.segment_reduce(x, idx, n_seg, "mean")
    r = r.reshape((n_seg,) + tuple(src.shape[1:]))
    return r if src.dtype == torch.float32 else r.to(src.dtype)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    """-> (values, argmax); argmax = src.shape[0] for segments without rows, the lowest row index among ties"""
    assert out is None
    x, idx, n_seg = _prep(src, index, dim, dim_size)
    r, arg = ops.segment_reduce(x, idx, n_seg, "max", want_arg=True)
    shape = (n_seg,) + tuple(src.shape[1:])
    r = r.reshape(shape)
    return (r if src.dtype == torch.float32 else r.to(src.dtype)), arg.reshape(shape)
