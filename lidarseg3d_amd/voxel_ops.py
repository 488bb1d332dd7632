"""Voxelization / DynamicScatter / DynamicScatterWithDistance / HardSimpleVFE / DynamicSimpleVFE with the signatures of det3d/ops/voxel
(voxelize.py:65-123, scatter_points.py:68-213, voxel_encoder.py:13-68) on the libls3d kernels."""
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


class DynamicScatterWithDistance(nn.Module):
    """scatter_points.py:132-213: column 0 of `points` is a per-point weight (the reference calls it distance; it is 1 / distance), the other
    columns are features; pool_method 'max' / 'avg' as DynamicScatter, 'weighted_avg' = sum_i f_i * (d_i / (sum_j d_j + 1e-8)) over a
    voxel's points - here the ordered sums of d and of d * f over the voxel (ls3d_dynamic_scatter mode 2), one division per voxel."""

    def __init__(self, voxel_size, point_cloud_range, pool_method="weighted_avg"):
        super().__init__()
        if pool_method not in ("max", "avg", "weighted_avg"):
            raise ValueError("pool_method %r" % (pool_method,))
        self.voxel_size, self.point_cloud_range, self.pool_method = voxel_size, point_cloud_range, pool_method
        _, g = ops.make_grid(voxel_size, point_cloud_range)
        self.shape_zyx = [g[2], g[1], g[0]]

    @torch.no_grad()
    def forward(self, points, coors):
        """points [N, 1 + C], coors [N,3] (z,y,x) or [N,4] (batch,z,y,x) -> (features [V,C], coors [V,3|4])"""
        coors = coors.int().contiguous()
        d, f = points[:, :1], points[:, 1:]
        if self.pool_method != "weighted_avg":
            out, vc, _, nv = ops.dynamic_scatter(f.contiguous(), coors, self.shape_zyx, "mean" if self.pool_method == "avg" else "max")
            k = int(nv.item())
            return out[:k], vc[:k]
        sums, vc, _, nv = ops.dynamic_scatter(torch.cat([d, f * d], 1).contiguous(), coors, self.shape_zyx, "sum")
        k = int(nv.item())
        return sums[:k, 1:] / (sums[:k, :1] + 1e-8), vc[:k]

    def __repr__(self):
        return "%s(voxel_size=%s, point_cloud_range=%s, pool_method=%s)" % (type(self).__name__, self.voxel_size, self.point_cloud_range,
                                                                             self.pool_method)


class HardSimpleVFE(nn.Module):
    """voxel_encoder.py:13-40: mean of the first four point features over a voxel's points (zero padding adds nothing)"""

    def forward(self, features, num_points, coors=None):
        return ops.vfe_mean(features[:, :, :4].contiguous(), num_points.int().contiguous())


class DynamicSimpleVFE(nn.Module):
    """voxel_encoder.py:43-68: mean of all points of a voxel, dynamic voxelization (coors from Voxelization(max_num_points=-1))"""

    def __init__(self, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.scatter = DynamicScatter(voxel_size, point_cloud_range, True)

    @torch.no_grad()
    def forward(self, features, coors):
        return self.scatter(features, coors)
