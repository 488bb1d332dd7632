"""Count-weighted synchronised BatchNorm for the data-parallel training step (BASELINE configs[3]).

The reference converts every BatchNorm to torch.nn.SyncBatchNorm before wrapping the model in DistributedDataParallel
(det3d/torchie/apis/train.py:313-321).  Sparse tensors make the weighting matter: the ranks hold DIFFERENT numbers of rows
(active voxels / points of their frames), so the batch statistics are sums over all rows of all ranks divided by the TOTAL row
count - not an average of per-rank means.  This module does exactly that with two small collectives per layer and step (RCCL on
MI355X, gloo in the CPU tests): forward all-gathers the ranks' [mean, sum (x - mean)^2, n]; backward all-reduces [sum dy, sum dy*xhat], which makes
the returned dx the derivative of the SUM of all ranks' losses, the convention DistributedDataParallel's gradient averaging
expects (the same as torch.nn.SyncBatchNorm, which needs CUDA tensors and an NCCL/RCCL group and therefore cannot run under gloo).
With world_size 1 or no process group it reduces to nn.BatchNorm1d."""
import torch
import torch.distributed as dist
from torch import nn


def _active(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class _SyncBNFn(torch.autograd.Function):
    """x: [rows, C] (the 2-d / 3-d variants flatten N, H, W[, D] into rows first)"""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        c = x.shape[1]
        # per-rank (count, mean, sum of squared deviations) merged with the pairwise update of Chan et al.: E[x^2] - mean^2 loses
        # the variance to cancellation on the deep levels (tens of rows, |mean| >> std), which showed as 1 % gradient deviations
        nl = float(x.shape[0])
        ml = x.mean(0) if x.shape[0] > 0 else x.new_zeros(c)
        local = torch.cat([ml, ((x - ml) ** 2).sum(0), x.new_tensor([nl])])
        parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, local, group=group)
        allp = torch.stack(parts)  # [world, 2C+1], the same on every rank: identical statistics everywhere
        cnt = allp[:, -1:]
        n = cnt.sum().clamp_min(1.0)  # every rank empty (e.g. a level without active voxels): statistics 0 / eps, no NaN
        mean = (allp[:, :c] * cnt).sum(0) / n
        var = (allp[:, c:2 * c] + cnt * (allp[:, :c] - mean) ** 2).sum(0) / n
        invstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * invstd
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.n = group, n
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * weight + bias, mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, invstd, weight = ctx.saved_tensors
        c = xhat.shape[1]
        gb, gw = gy.sum(0), (gy * xhat).sum(0)  # local sums: the parameter gradients (DDP averages them over the ranks)
        s = torch.cat([gb, gw])
        dist.all_reduce(s, group=ctx.group)     # sums over ALL rows of all ranks: the statistics are shared
        dx = (gy - s[:c] / ctx.n - xhat * (s[c:] / ctx.n)) * (invstd * weight)
        return dx, gw, gb, None, None


class CountSyncBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d over [rows, C] whose training statistics run over the rows of every rank, weighted by row count"""

    def __init__(self, *a, process_group=None, **k):
        super().__init__(*a, **k)
        self.process_group = process_group

    def _ls3d_sync(self):
        """protocol of ops.batch_norm_train (the HIP BatchNorm kernels): -> (gather, reduce) closures - gather all-gathers the ranks' (mean, M2, n)
        triples [2 c + 1] into [world, 2 c + 1] (merged, with rstd and the running statistics, by ONE ls3d_batch_norm_finalize launch), reduce
        all-reduces a copy of the backward's column sums - or (None, None) without an active group"""
        group = self.process_group
        if not _active(group):
            return None, None

        def gather(local):
            parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, local, group=group)
            return torch.stack(parts)

        def reduce(sums):
            sums = sums.clone()
            dist.all_reduce(sums, group=group)
            return sums
        return gather, reduce

    def forward(self, x):
        if self.training and x.dim() == 2 and torch.is_grad_enabled():
            from . import ops
            y = ops.batch_norm_train(self, x) if (x.is_cuda or ops.sim_mode()) else None  # statistics / normalisation / backward on csrc/norm.hip, the same two collectives
            if y is not None:
                return y
        if not (self.training and _active(self.process_group)):
            return super().forward(x)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp_min(1)), alpha=m)  # unbiased, as nn.BatchNorm does
        return y


class _CountSyncBatchNormNd(object):
    """forward of the 2-d / 3-d variants: channels-first [N, C, ...] flattened to rows, the same statistics code"""

    def forward(self, x):
        if not (self.training and _active(self.process_group)):
            return super().forward(x)
        c = x.shape[1]
        rows = x.movedim(1, -1).reshape(-1, c)
        y, mean, var, n = _SyncBNFn.apply(rows, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp_min(1)), alpha=m)
        return y.reshape(*x.movedim(1, -1).shape).movedim(-1, 1)


class CountSyncBatchNorm2d(_CountSyncBatchNormNd, nn.BatchNorm2d):
    def __init__(self, *a, process_group=None, **k):
        nn.BatchNorm2d.__init__(self, *a, **k)
        self.process_group = process_group


class CountSyncBatchNorm3d(_CountSyncBatchNormNd, nn.BatchNorm3d):
    def __init__(self, *a, process_group=None, **k):
        nn.BatchNorm3d.__init__(self, *a, **k)
        self.process_group = process_group


_SYNC = ((nn.BatchNorm1d, CountSyncBatchNorm1d), (nn.BatchNorm2d, CountSyncBatchNorm2d), (nn.BatchNorm3d, CountSyncBatchNorm3d))


def convert_sync_batchnorm(module, process_group=None):
    """nn.SyncBatchNorm.convert_sync_batchnorm for this path (train.py:313-321): EVERY _BatchNorm - the BatchNorm1d of the sparse
    backbone / readers / point heads, the BatchNorm2d of the camera head's ConvModules (img_heads.py) and of a user's camera
    backbone, BatchNorm3d - becomes its count-weighted synchronised variant with the same parameters / buffers / state_dict keys"""
    out = module
    for plain, synced in _SYNC:
        if isinstance(module, plain) and not isinstance(module, (CountSyncBatchNorm1d, CountSyncBatchNorm2d, CountSyncBatchNorm3d)):
            out = synced(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats, process_group=process_group)
            if module.affine:
                out.weight, out.bias = module.weight, module.bias
            out.running_mean, out.running_var, out.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
            out.training = module.training
            break
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child, process_group))
    return out
