"""Voxelization / DynamicScatter modules with the signatures of det3d/ops/voxel
(voxelize.py:65-123, scatter_points.py:68-129) on the libls3d kernels."""
import torch
from torch import nn

from . import ops


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size, self.point_cloud_range, self.max_num_points = voxel_size, point_cloud_range, max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else (max_voxels, max_voxels)
        _, g = ops.make_grid(voxel_size, point_cloud_range)
        self.grid_size = torch.tensor(g, dtype=torch.long)
        self.pcd_shape = [g[0], g[1], 1][::-1]

    def forward(self, input):
        """points [N,C] -> coors [N,3] (max_num_points == -1, dynamic) or (voxels, coors, num_points_per_voxel)"""
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        if self.max_num_points == -1 or max_voxels == -1:
            return ops.voxelize_dynamic(input.contiguous(), self.voxel_size, self.point_cloud_range)
        v, c, n, nv = ops.voxelize_hard(input.contiguous(), self.voxel_size, self.point_cloud_range, self.max_num_points,
                                        max_voxels, overflow="break")
        k = int(nv.item())
        return v[:k], c[:k], n[:k]

    def __repr__(self):
        return "%s(voxel_size=%s, point_cloud_range=%s, max_num_points=%s, max_voxels=%s)" % (
            type(self).__name__, self.voxel_size, self.point_cloud_range, self.max_num_points, self.max_voxels)


class _DynamicScatterFn(torch.autograd.Function):
    """the fused DynamicScatter with its backward (the reference: _dynamic_scatter + torch mean/max, scatter_points.py:9-50,85-98)"""

    @staticmethod
    def forward(ctx, points, coors, shape_zyx, mode):
        pts = points.detach().contiguous()
        f, vc, p2v, nv = ops.dynamic_scatter(pts, coors, shape_zyx, mode)
        k = int(nv.item())  # host sync: the number of voxels is a tensor shape
        f, vc = f[:k], vc[:k]
        ctx.save_for_backward(pts, f, p2v)
        ctx.mode = mode
        ctx.mark_non_differentiable(vc)
        return f, vc

    @staticmethod
    def backward(ctx, grad_f, grad_vc=None):
        pts, f, p2v = ctx.saved_tensors
        return ops.dynamic_scatter_backward(grad_f.contiguous(), p2v, pts, f, ctx.mode), None, None, None


class DynamicScatter(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size, self.point_cloud_range, self.average_points = voxel_size, point_cloud_range, average_points
        _, g = ops.make_grid(voxel_size, point_cloud_range)
        self.shape_zyx = [g[2], g[1], g[0]]

    def forward(self, points, coors):
        """points [N,C], coors [N,3] (z,y,x) or [N,4] (batch,z,y,x) -> (features [V,C], coors [V,3|4])"""
        return _DynamicScatterFn.apply(points, coors.int().contiguous(), self.shape_zyx, "mean" if self.average_points else "max")

    def __repr__(self):
        return "%s(voxel_size=%s, point_cloud_range=%s, average_points=%s)" % (
            type(self).__name__, self.voxel_size, self.point_cloud_range, self.average_points)
