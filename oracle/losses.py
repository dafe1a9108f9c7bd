"""CPU restatement of the reference's loss arithmetic (TEST INFRASTRUCTURE - see oracle/__init__.py).

Plain torch (autograd-capable) functions; each cites the reference lines it follows.  Pinned by
tests/golden/mil_losses.npz, box_predictor_*.npz, rpn.npz (captured from the reference modules).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import d2


def mil_cross_entropy(x, target, weights=None, avg_positives=False, reduction="mean"):
    """coin/utils/losses.py:13-34.  Softmax WITHOUT max-subtraction (lines 15-18 are commented out upstream)."""
    e = torch.exp(x)
    p = e / e.sum(dim=-1, keepdim=True)
    s = (target * p).sum(dim=-1)
    if avg_positives:
        s = s / (target.sum(dim=-1) + 1e-6)
    loss = -torch.log(s)
    if weights is not None:
        loss = loss * weights
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    return loss.sum()


def mil_focal_loss(x, target, alpha, gamma=1.5, avg_positives=True):
    """coin/utils/losses.py:53-73."""
    e = torch.exp(x)
    p = e / e.sum(dim=-1, keepdim=True)
    tsum = target.sum(dim=-1)
    a = (target * alpha.view(1, -1)).sum(1) / (tsum + 1e-6)
    pt = (target * p).sum(dim=-1)
    if avg_positives:
        pt = pt / (tsum + 1e-6)
    return (-a * torch.pow(1 - pt, gamma) * pt.log()).mean()


def kl_div_mean(p, q, eps=1e-7):
    """nn.KLDivLoss(reduction='mean')(log(p+eps), q): ELEMENT mean (fast_rcnn.py:273,526,538,544; rpn.py:15,335)."""
    logp = torch.log(p + eps)
    pointwise = torch.xlogy(q, q) - q * logp
    return pointwise.mean()


def box_reg_loss(proposal_boxes, gt_boxes, pred_deltas, gt_classes, num_classes, weights=(10.0, 10.0, 5.0, 5.0),
                 normalizer=None):
    """fast_rcnn.py:601-646, class-agnostic, smooth_l1 with beta = 0 (== L1), sum / R."""
    fg = ((gt_classes >= 0) & (gt_classes < num_classes)).nonzero()[:, 0]
    tf = d2.Box2BoxTransform(weights)
    gt_d = tf.get_deltas(proposal_boxes[fg], gt_boxes[fg])
    loss = d2.smooth_l1_loss(pred_deltas[fg], gt_d, 0.0, reduction="sum")
    if normalizer is not None:
        return loss / normalizer
    return loss / max(gt_classes.numel(), 1.0)


def rpn_losses(anchors, logits, labels, deltas, matched_gt, batch_size_per_image, calc_bg=True):
    """rpn.py:289-325.  logits [N,A], labels [N,A] in {-1,0,1}, deltas [N,A,4], matched_gt [N,A,4]."""
    n = labels.shape[0]
    pos = labels == 1
    tf = d2.Box2BoxTransform((1.0, 1.0, 1.0, 1.0))
    gt_d = torch.stack([tf.get_deltas(anchors, k) for k in matched_gt])
    loc = d2.smooth_l1_loss(deltas[pos], gt_d[pos], 0.0, reduction="sum")
    valid = labels >= (0 if calc_bg else 1)
    cls = F.binary_cross_entropy_with_logits(logits[valid], labels[valid].to(logits.dtype), reduction="sum")
    normalizer = batch_size_per_image * n
    return cls / (normalizer if calc_bg else max(int(valid.sum()), 1.0)), loc / normalizer


def rpn_distillation(logits, dist_labels, teacher_probs, eps=1e-7):
    """rpn.py:326-340."""
    valid = dist_labels > 0
    p = torch.sigmoid(logits[valid])
    p = torch.stack((p, 1 - p), dim=1)
    q = teacher_probs[valid]
    q = torch.stack((q, 1 - q), dim=1)
    return kl_div_mean(p, q, eps), int(valid.sum())
