"""three_nn / three_interpolate with the call signatures of det3d/ops/pointnet2_batch/pointnet2_utils.py:76-153
(autograd Functions over the pybind module pointnet2_batch_cuda), on the libls3d kernels."""
import torch
from torch.autograd import Function

from . import ops


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,N,3), known (B,M,3) -> dist (B,N,3) L2 distances, idx (B,N,3) int32"""
        assert unknown.is_contiguous() and known.is_contiguous()
        d2, idx = ops.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(d2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)"""
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[2])
        return ops.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return ops.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_nn = ThreeNN.apply
three_interpolate = ThreeInterpolate.apply
