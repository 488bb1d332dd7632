"""Segmentation losses of the point heads: cross entropy with an ignored label + Lovasz-Softmax
(Berman, Triki, Blaschko, CVPR 2018, Alg. 1), as det3d/core/utils/loss_utils.py:217-291 applies them to flat [P, C] predictions
(point_seg_batchloss_head.py:77-121).  seg_loss runs them as the fused kernels of csrc/loss.hip; the torch restatement below is what
those are tested against (and the fallback for anything that is not an f32 [P, C <= 32] tensor on the device)."""
import os

import torch
import torch.nn.functional as F


def lovasz_grad(gt_sorted):
    """gradient of the Lovasz extension of the Jaccard loss w.r.t. the sorted errors (gt_sorted: 0/1 in error order)"""
    gts = gt_sorted.sum()
    inter = gts - gt_sorted.cumsum(0)
    union = gts + (1.0 - gt_sorted).cumsum(0)
    jac = 1.0 - inter / union
    if gt_sorted.numel() > 1:
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
    return jac


def lovasz_softmax(probas, labels, ignore=None):
    """probas [P, C] (after softmax), labels [P]; mean over the classes PRESENT among the non-ignored labels"""
    labels = labels.reshape(-1)
    if ignore is not None:
        keep = labels != ignore
        probas, labels = probas[keep], labels[keep]
    if probas.numel() == 0:
        return probas.sum() * 0.0
    terms = []
    for c in range(probas.shape[1]):
        fg = (labels == c).to(probas.dtype)
        if float(fg.sum()) == 0.0:
            continue
        err = (fg - probas[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        terms.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    if not terms:
        return probas.sum() * 0.0
    return torch.stack(terms).mean()


def seg_loss_torch(logits, labels, ignore):
    """(cross entropy, Lovasz-Softmax) of one prediction level, the torch restatement (a Python loop over the classes with a host
    synchronisation and a sort each): what the fused kernels are tested against"""
    labels = labels.long()
    ce = F.cross_entropy(logits, labels, ignore_index=ignore)
    lv = lovasz_softmax(F.softmax(logits, dim=-1), labels, ignore=ignore)
    return ce, lv


class _FusedSegLoss(torch.autograd.Function):
    """ls3d_seg_loss_forward / _backward (csrc/loss.hip): one softmax pass, one batched radix sort of the C x P class errors, per-class
    scans - no host synchronisation; the backward is one more launch over the forward's workspace"""

    @staticmethod
    def forward(ctx, logits, labels, ignore):
        from . import ops
        logits = logits.contiguous()
        labels = labels.reshape(-1).to(torch.int32).contiguous()
        out, ws = ops.seg_loss_forward(logits, labels, ignore)
        ctx.save_for_backward(labels, ws)
        ctx.shape, ctx.ignore = tuple(logits.shape), ignore
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_ce, g_lv):
        from . import ops
        labels, ws = ctx.saved_tensors
        g_ce = g_ce.reshape(1).float().contiguous() if g_ce is not None else None
        g_lv = g_lv.reshape(1).float().contiguous() if g_lv is not None else None
        return ops.seg_loss_backward(labels, ctx.shape, ctx.ignore, ws, g_ce, g_lv), None, None


FUSED = True  # A/B (module constant; experiments.py)


def seg_loss(logits, labels, ignore):
    """(cross entropy, Lovasz-Softmax) of one prediction level: the fused HIP kernels for f32 logits of <= 32 classes on the device,
    the torch restatement otherwise"""
    from . import ops
    if FUSED and logits.dim() == 2 and logits.dtype == torch.float32 and logits.shape[1] <= 32 and logits.shape[0] > 0 \
            and (logits.is_cuda or ops.sim_mode()):
        return _FusedSegLoss.apply(logits, labels, int(ignore) if ignore is not None else -(1 << 30))
    return seg_loss_torch(logits, labels, ignore)
