"""Segmentation losses of the point heads: cross entropy with an ignored label + Lovasz-Softmax
(Berman, Triki, Blaschko, CVPR 2018, Alg. 1), as det3d/core/utils/loss_utils.py:217-291 applies them to flat [P, C] predictions
(point_seg_batchloss_head.py:77-121).  torch ops on the device (sort / cumsum / dot): plumbing of the training step."""
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


def seg_loss(logits, labels, ignore):
    """(cross entropy, Lovasz-Softmax) of one prediction level"""
    labels = labels.long()
    ce = F.cross_entropy(logits, labels, ignore_index=ignore)
    lv = lovasz_softmax(F.softmax(logits, dim=-1), labels, ignore=ignore)
    return ce, lv
