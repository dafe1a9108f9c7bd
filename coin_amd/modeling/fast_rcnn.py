"""FastRCNNOutputLayers on the HIP box-head kernels.

Mirrors coin/modeling/roi_heads/fast_rcnn.py:182-752: same constructor surface, state-dict keys
(``trans.{0,2,4}``, ``cls_score``, ``bbox_pred``, ``logit_scale``, ``text_encoder.*``), loss names and
weights.  Execution: every Linear is ``coin_gemm_nt`` (MFMA) with bias + LeakyReLU fused in the epilogue;
the cosine classifier, MIL-CE, KL, box-regression L1 and text-align L1 are one fused HIP launch each
(forward value + gradient).  Row selection (fg / bg / A / B of each image) is done with index tensors
built once per call instead of per-image split/cat lists.
The CKG terms (`loss_merge_*`) stay torch-composite: they are tens of rows and need double backward
(coin/utils/losses.py:75-96).
"""
from __future__ import annotations

import contextlib
import os

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import layers as L
from .._lib import ACT_LEAKY_RELU, ACT_NONE
from ..box_ops import Box2BoxTransform, batched_nms
from ..structures import Boxes, Instances
from .text_encoder import TEXT_DIMS, text_dim_of


def weight_init(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        nn.init.constant_(m.bias, 0)


def _cat(ts, dim=0):
    return ts[0] if len(ts) == 1 else torch.cat(ts, dim)


def _ranges(starts: List[int], lens: List[int], device) -> torch.Tensor:
    parts = [torch.arange(s, s + l, device=device) for s, l in zip(starts, lens) if l > 0]
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=device)


@torch.no_grad()
def _prototype_ema(proto, feats, one_hot, rate):
    new = proto.clone().float()
    cnt = one_hot.sum(0)
    present = cnt != 0
    mean = one_hot.t() @ feats.float() / cnt.clamp(min=1).unsqueeze(1)
    new = torch.where(present.unsqueeze(1), mean, new)
    return proto * rate + (1 - rate) * new


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:116-175."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    probs = scores.clone()
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    b = Boxes(boxes.reshape(-1, 4))
    b.clip(image_shape)
    boxes = b.tensor.view(-1, nreg, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes = boxes[inds[:, 0], 0] if nreg == 1 else boxes[mask]
    scores = scores[mask]
    probs = probs[inds[:, 0]]
    keep = batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    res = Instances(image_shape)
    res.pred_boxes = Boxes(boxes[keep])
    res.scores = scores[keep]
    res.probs = probs[keep]
    res.pred_classes = inds[keep][:, 1]
    return res, inds[keep][:, 0]


def _gradient_discrepancy(loss_a, loss_b, params):
    """1 - mean over theta of cos(d loss_a / d theta (a constant), d loss_b / d theta (differentiable)) -- coin/utils/losses.py:75-96.
    The reference asks autograd for one parameter at a time (twelve backward walks over the same sub-graph for the six tensors of
    `trans`, each retaining / re-creating the graph); the gradients do not depend on how many are requested per walk, so both sets are
    taken in ONE walk each: the same values, a sixth of the launches (round 6, tools/opsites.py: this loss was ~700 of the targetDET
    step's ~2 400 small launches and 6 ms of its device time)."""
    ga = torch.autograd.grad(loss_a, params, retain_graph=True)
    gb = torch.autograd.grad(loss_b, params, create_graph=True)
    cos = [F.cosine_similarity(a, b, dim=1).mean() if prm.dim() > 1 else F.cosine_similarity(a, b, dim=0) for prm, a, b in zip(params, ga, gb)]
    return (1.0 - torch.stack(cos)).mean()


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, input_shape, *, text_encoder, pooling_type, box2box_transform, text_dim, classes_weight, loss_type,
                 test_score_thresh=0.0, test_nms_thresh=0.5, test_topk_per_image=100, cls_agnostic_bbox_reg=False,
                 smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1", loss_weight=1.0, batch_size_per_image, cls_b_thresh,
                 dataset, prototype_update_rate):
        super().__init__()
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        assert pooling_type in ("attnpool", "meanpool")
        assert cls_agnostic_bbox_reg and box_reg_loss_type == "smooth_l1" and smooth_l1_beta == 0.0, \
            "the fused box-regression kernel implements COIN's configuration (class-agnostic L1)"
        self.text_dim = text_dim
        self.trans = nn.Sequential(nn.Linear(input_size, input_size // 2), nn.LeakyReLU(), nn.Linear(input_size // 2, input_size // 2),
                                   nn.LeakyReLU(), nn.Linear(input_size // 2, input_size))
        self.cls_score = nn.Linear(input_size, text_dim)
        self.logit_scale = nn.Parameter(torch.FloatTensor([0.01]), requires_grad=False)
        self.num_classes = text_encoder.num_classes - 1
        self.bbox_pred = nn.Linear(input_size, 4)
        self.trans.apply(weight_init)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.cls_score.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.box2box_transform = box2box_transform
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image
        self.loss_weight = loss_weight if isinstance(loss_weight, dict) else {}
        self.text_encoder = text_encoder
        self.loss_type, self.classes_weight = loss_type, list(classes_weight)
        self.batch_size_per_image, self.cls_b_thresh = batch_size_per_image, cls_b_thresh
        self.dataset, self.prototype_update_rate = tuple(dataset), prototype_update_rate
        self._inv_scale = 1.0 / 0.01

    @classmethod
    def from_config(cls, cfg, text_encoder, input_shape):
        lw = {"loss_box_reg": cfg.CLOUD.LOSS_BOX_REG_WEIGHT, "loss_box_reg_offline": cfg.CLOUD.LOSS_BOX_REG_OFFLINE_WEIGHT,
              "loss_box_reg_online": cfg.CLOUD.LOSS_BOX_REG_ONLINE_WEIGHT, "loss_cls": cfg.CLOUD.LOSS_CLS_WEIGHT,
              "loss_text_align": cfg.CLOUD.LOSS_TEXT_ALIGN_WEIGHT, "loss_distillation": cfg.CLOUD.LOSS_DISTILLATION_WEIGHT,
              "loss_cls_b": cfg.CLOUD.LOSS_CLS_B_WEIGHT}
        return cls(input_shape, text_encoder=text_encoder, pooling_type=cfg.MODEL.ROI_HEADS.POOLING_TYPE,
                   box2box_transform=Box2BoxTransform(cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
                   text_dim=text_dim_of(cfg), classes_weight=cfg.CLOUD.CLASSES_WEIGHT,
                   loss_type=cfg.CLOUD.LOSS_TYPE, test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                   test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                   cls_agnostic_bbox_reg=cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG, smooth_l1_beta=cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
                   box_reg_loss_type=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE, loss_weight=lw,
                   batch_size_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, cls_b_thresh=cfg.CLOUD.CLS_B_THRESH,
                   dataset=cfg.DATASETS.TRAIN_UNLABEL, prototype_update_rate=cfg.CLOUD.PROTOTYPE_UPDATE_WEIGHT)

    # ------------------------------------------------------------------ forward (fast_rcnn.py:318-353)
    def forward(self, x, branch, return_feats=True):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        t = self.trans
        h = L.linear_act(x, t[0].weight, t[0].bias, ACT_LEAKY_RELU, 0.01)
        h = L.linear_act(h, t[2].weight, t[2].bias, ACT_LEAKY_RELU, 0.01)
        h = L.linear_act(h, t[4].weight, t[4].bias, ACT_NONE)
        class_feats = L.linear_act(h, self.cls_score.weight, self.cls_score.bias, ACT_NONE)
        scores = self.do_classify(class_feats, branch)
        proposal_deltas = L.linear_act(h, self.bbox_pred.weight, self.bbox_pred.bias, ACT_NONE, out_dtype=torch.float32)
        if return_feats and self.training and branch != "test":
            self._last_input = x  # pooled RoI features of the proposal pass: `merge_grad_loss` re-differentiates a few rows
            return scores, proposal_deltas, class_feats
        return scores, proposal_deltas

    _last_input = _last_text = _merge_ctx = None

    def merge_grad_loss(self) -> torch.Tensor:
        """``gradient_discrepancy_loss`` (coin/utils/losses.py:75-96, called at coin/engine/trainer.py:192-197): 1 - mean cosine
        between d loss_merge_a / d theta (detached) and d loss_merge_b / d theta over the parameters theta of `trans`.

        It needs a second derivative (the CKG parameters receive their gradient THROUGH d loss_merge_b / d theta, via the merged
        target m_b), which the fused one-launch kernels do not provide.  Only the A and B rows enter (tens of rows), so the
        sub-graph  trans -> cls_score -> cosine logits -> softmax -> MSE  is replayed for those rows in fp32 with
        differentiable torch ops; everything upstream (pooled features, text embeddings) is a constant of this loss, exactly as
        in the reference where theta are the only differentiation variables.  Call after `losses()` of a step_one / step_two
        forward that produced `loss_merge_a`."""
        assert self._merge_ctx is not None, "merge_grad_loss needs the context of a step_one/step_two losses() call with B boxes"
        if self._merge_ctx[1].dtype == torch.bool:  # packed layout: (all rows, mask A, mask B, text, one-hot A, merged)
            return self._merge_grad_loss_masked(*self._merge_ctx)
        x_a, x_b, text, oh_a, m_b = self._merge_ctx
        na = x_a.shape[0]
        with torch.autocast(x_a.device.type, enabled=False):
            x = torch.cat([x_a, x_b]).detach().float()
            t = self.trans
            h = F.leaky_relu(F.linear(x, t[0].weight, t[0].bias), 0.01)
            h = F.leaky_relu(F.linear(h, t[2].weight, t[2].bias), 0.01)
            h = F.linear(h, t[4].weight, t[4].bias)
            cf = F.linear(h, self.cls_score.weight, self.cls_score.bias)
            scores = F.normalize(cf, dim=1) @ F.normalize(text.detach().float(), dim=1).t() * self._inv_scale
            p = F.softmax(scores, dim=1)
            loss_a = F.mse_loss(p[:na], oh_a.float())
            loss_b = F.mse_loss(p[na:], m_b.float())
            return _gradient_discrepancy(loss_a, loss_b, [prm for prm in t.parameters() if prm.requires_grad])

    def prefetch_text(self):
        """Run the prompt-conditioned text encoder ahead of `forward` (it does not depend on the images): the caller puts it
        on a side stream so that its ~300 small launches overlap the backbone convolutions.  Consumed by the next forward."""
        self._text_prefetch = self.text_encoder(added=True)
        return self._text_prefetch

    def _merge_grad_loss_masked(self, x, m_a, m_b, text, oh_a, merged) -> torch.Tensor:
        """`merge_grad_loss` for the packed layout: the sub-graph is replayed on every row and the two MSE terms are masked
        means (rows outside A / B contribute nothing to either gradient)."""
        kc = self.num_classes + 1
        with torch.autocast(x.device.type, enabled=False):
            x = x.detach().float()
            t = self.trans
            h = F.leaky_relu(F.linear(x, t[0].weight, t[0].bias), 0.01)
            h = F.leaky_relu(F.linear(h, t[2].weight, t[2].bias), 0.01)
            h = F.linear(h, t[4].weight, t[4].bias)
            cf = F.linear(h, self.cls_score.weight, self.cls_score.bias)
            scores = F.normalize(cf, dim=1) @ F.normalize(text.detach().float(), dim=1).t() * self._inv_scale
            p = F.softmax(scores, dim=1)
            fa, fb = m_a.float(), m_b.float()
            loss_a = (((p - oh_a.float()) ** 2).sum(1) * fa).sum() / (fa.sum().clamp(min=1) * kc)
            loss_b = (((p - merged.float()) ** 2).sum(1) * fb).sum() / (fb.sum().clamp(min=1) * kc)
            return _gradient_discrepancy(loss_a, loss_b, [prm for prm in t.parameters() if prm.requires_grad])

    _text_prefetch = None
    _share_text, _text_shared = False, None

    @contextlib.contextmanager
    def shared_text(self):
        """One prompt-encoder pass per detector forward.  The step branches classify twice per forward (the proposal pass and the
        C-box pass, clip_roi_heads.py:193-227) and the reference runs the 12-layer text transformer for each (fast_rcnn.py:339-346);
        its weights do not change inside a forward, so both calls see the same embeddings.  Inside this scope the second call
        reuses the first one's tensor: same values, and the prompt vectors receive the sum of both gradients through ONE backward
        pass of the encoder (≈800 small launches fewer per step)."""
        self._share_text, self._text_shared = os.environ.get("COIN_SHARED_TEXT", "1") != "0", None   # env: A/B measurements only
        try:
            yield
        finally:
            self._share_text, self._text_shared = False, None

    def do_classify(self, image_features, branch):
        if self._text_prefetch is not None:
            text, self._text_prefetch = self._text_prefetch, None
        elif self._share_text and self._text_shared is not None and self._text_shared.requires_grad == (torch.is_grad_enabled() and self.training):
            text = self._text_shared
        else:
            text = self.text_encoder(added=True)
        if self._share_text:
            self._text_shared = text
        self._last_text = text
        # the kernel L2-normalises both operands (the encoder output is already unit-norm: normalising twice, as the
        # reference does at fast_rcnn.py:344, is the identity up to rounding)
        scores = L.cosine_logits(image_features, text, self._inv_scale)
        if self.training and branch != "test":
            fixed = self.text_encoder(added=False).detach()
            fixed = fixed / fixed.norm(dim=1, keepdim=True)
            tn = text / text.norm(dim=1, keepdim=True)
            return scores, L.l1_mean(tn, fixed)
        return scores

    # ------------------------------------------------------------------ helpers
    def _mil(self, scores, labels_or_target, n_fg, n_bg, avg_positives=True, soft=False):
        if self.loss_type == "MILFocalLoss":  # fast_rcnn.py:468-472,581-582,595-596: always avg_positives, no row weights
            a = torch.tensor(self.classes_weight, dtype=torch.float32, device=scores.device)
            kw = {"target": labels_or_target} if soft else {"labels": labels_or_target}
            return L.mil_focal_loss(scores, a, gamma=1.5, avg_positives=True, **kw)
        if self.loss_type != "MILCrossEntropy":
            raise NotImplementedError(self.loss_type)
        w = torch.cat([torch.ones(n_fg, device=scores.device), torch.full((n_bg,), float(self.classes_weight[-1]), device=scores.device)])
        if soft:
            return L.mil_cross_entropy(scores, target=labels_or_target, weights=w, avg_positives=avg_positives)
        return L.mil_cross_entropy(scores, labels=labels_or_target, weights=w, avg_positives=avg_positives)

    def box_reg_loss(self, proposal_boxes, gt_boxes, pred_deltas, gt_classes, normalizer=None):
        norm = float(normalizer) if normalizer is not None else float(max(gt_classes.numel(), 1))
        return L.box_reg_l1(proposal_boxes, gt_boxes, pred_deltas, gt_classes, self.num_classes, self.box2box_transform.weights, norm)

    # ------------------------------------------------------------------ pre_train losses on packed samples (sync-free)
    def losses_packed(self, predictions, ps, update_prototype=False):
        """fast_rcnn.py:366-438 for the fixed-shape sample layout: the same three losses, written with per-row labels /
        weights instead of per-image fg/bg splits (every reduction is a permutation-invariant mean or sum) and with the
        data-dependent branches (`sum(per_image_fg_nums) != 0`) turned into device-side selects."""
        (scores, lta), deltas, feats = predictions
        kc = self.num_classes + 1
        cls = ps.gt_classes
        valid = cls >= 0
        is_fg = valid & (cls < self.num_classes)
        n_valid = valid.sum().clamp(min=1).float()
        any_fg = is_fg.any()
        labels = cls.clamp(min=0)
        w = torch.where(is_fg, 1.0, float(self.classes_weight[-1])) * valid.float()
        if self.loss_type not in ("MILCrossEntropy", "MILFocalLoss"):
            raise NotImplementedError(self.loss_type)
        focal = self.loss_type == "MILFocalLoss"
        if focal:  # no row weights in the reference (fast_rcnn.py:581-582): the mask only removes the filler rows
            w = valid.float()
            alpha = torch.tensor(self.classes_weight, dtype=torch.float32, device=scores.device)
        if self.dataset != ("cliparttrain",):
            total = (L.mil_focal_loss(scores, alpha, labels=labels, weights=w, reduction="sum") if focal else
                     L.mil_cross_entropy(scores, labels=labels, weights=w, avg_positives=True, reduction="sum"))
        else:
            tgt = F.one_hot(labels, kc).float() * torch.where(is_fg, ps.gt_probs.max(1)[0], torch.ones_like(w)).unsqueeze(1)
            total = (L.mil_focal_loss(scores, alpha, target=tgt, weights=w, reduction="sum") if focal else
                     L.mil_cross_entropy(scores, target=tgt, weights=w, avg_positives=False, reduction="sum"))
        losses = {"loss_text_align": lta, "loss_cls": torch.where(any_fg, total / n_valid, torch.zeros_like(total))}
        if update_prototype:
            with torch.no_grad():
                fn = feats.detach().float()
                fn = fn / fn.norm(dim=1, keepdim=True)
                oh = F.one_hot(labels, kc).float() * valid.float().unsqueeze(1)
                te = self.text_encoder
                new = _prototype_ema(te.per_class_feat.data, fn, oh, self.prototype_update_rate)
                te.per_class_feat.data = torch.where(any_fg, new, te.per_class_feat.data)
        # box regression: sum over fg rows / number of sampled rows (fast_rcnn.py:646)
        reg = L.box_reg_l1(ps.boxes, ps.gt_boxes, deltas, torch.where(is_fg, cls, torch.full_like(cls, -1)), self.num_classes,
                           self.box2box_transform.weights, 1.0)
        losses["loss_box_reg"] = reg / n_valid
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    # ------------------------------------------------------------------ step_one / step_two losses on packed samples (sync-free)
    def losses_packed_step(self, predictions, ps, cpred, inst_c, merge_module, branch, update_prototype=False):
        """fast_rcnn.py:440-571 for the fixed-shape sample layout (`PackedStepSamples`: one role per row -- A, B, background,
        filler).  The same terms as `losses(step_*)`, written with row masks and masked means instead of per-image index
        ranges, so that no row count travels to the host.  The CKG module runs on every row (it is a per-row function) and the
        masks select.  `ps.has_b` (host-known: the matcher produced B targets) decides whether the merge terms exist; if B
        targets exist but none was sampled the terms evaluate to 0 (the reference omits them in that case)."""
        assert branch in ("step_one", "step_two")
        (scores, lta), deltas, feats = predictions
        scores_c = cpred[0][0] if cpred is not None else None
        kc, te = self.num_classes + 1, self.text_encoder
        m_a, m_b, m_g = ps.role == 0, ps.role == 1, ps.role == 2
        n_a, n_b, n_g = m_a.sum(), m_b.sum(), m_g.sum()
        fa, fb = m_a.float(), m_b.float()
        ag = m_a | m_g
        losses = {"loss_text_align": lta}
        if self.loss_type not in ("MILCrossEntropy", "MILFocalLoss"):
            raise NotImplementedError(self.loss_type)
        labels = torch.where(ag, ps.gt_classes, torch.zeros_like(ps.gt_classes))
        if self.loss_type == "MILFocalLoss":  # fast_rcnn.py:468-472
            alpha = torch.tensor(self.classes_weight, dtype=torch.float32, device=scores.device)
            total = L.mil_focal_loss(scores, alpha, labels=labels, weights=ag.float(), reduction="sum")
        else:
            w = fa + m_g.float() * float(self.classes_weight[-1])
            total = L.mil_cross_entropy(scores, labels=labels, weights=w, avg_positives=True, reduction="sum")
        losses["loss_cls"] = total / (n_a + n_g).clamp(min=1).float()
        oh_a = F.one_hot(labels, kc).float() * fa.unsqueeze(1)
        if update_prototype:
            with torch.no_grad():
                fn = feats.detach().float()
                fn = fn / fn.norm(dim=1, keepdim=True)
                oh_ag = F.one_hot(labels, kc).float() * ag.float().unsqueeze(1)
                rate = self.prototype_update_rate
                te.per_class_feat.data = _prototype_ema(te.per_class_feat.data, fn, oh_ag, rate)
            if ps.has_b:
                any_b = n_b > 0
                with torch.no_grad():
                    zb = torch.zeros_like(ps.gt_classes_online)
                    oh_on = oh_ag + F.one_hot(torch.where(m_b, ps.gt_classes_online, zb), kc).float() * fb.unsqueeze(1)
                    oh_off = oh_ag + F.one_hot(torch.where(m_b, ps.gt_classes_offline, zb), kc).float() * fb.unsqueeze(1)
                    te.prototype_b_online.data = torch.where(any_b, _prototype_ema(te.prototype_b_online.data, fn, oh_on, rate),
                                                             te.prototype_b_online.data)
                    te.prototype_b_offline.data = torch.where(any_b, _prototype_ema(te.prototype_b_offline.data, fn, oh_off, rate),
                                                              te.prototype_b_offline.data)
                merged = merge_module(fn, te.prototype_b_offline.data, te.prototype_b_online.data, ps.gt_probs_offline, ps.gt_probs_online)
                losses["loss_merge_base"] = L.kl_div_from_probs(merged, oh_a, row_mask=m_a)
                p_all = F.softmax(scores, dim=1)
                losses["loss_merge_b"] = (((p_all - merged) ** 2).sum(1) * fb).sum() / (n_b.clamp(min=1).float() * kc)
                losses["loss_merge_a"] = (((p_all - oh_a) ** 2).sum(1) * fa).sum() / (n_a.clamp(min=1).float() * kc)
                self._merge_ctx = (self._last_input, m_a, m_b, self._last_text, oh_a, merged)
                if branch == "step_two":
                    keep = (merged.max(1)[0] >= self.cls_b_thresh).detach() & m_b
                    losses["loss_cls_b"] = L.kl_div_from_logits(scores, merged.detach(), row_mask=keep)
        if scores_c is not None:
            losses["loss_distillation"] = L.kl_div_from_logits(scores_c, _cat([c.gt_probs for c in inst_c]))
        neg = torch.full_like(ps.gt_classes, -1)
        cls_on = torch.where(m_a, ps.gt_classes, torch.where(m_b, ps.gt_classes_online, torch.where(m_g, ps.gt_classes, neg)))
        reg = L.box_reg_l1(ps.boxes, ps.gt_boxes, deltas, cls_on, self.num_classes, self.box2box_transform.weights, 1.0)
        n_rows = (n_a + n_b + n_g).float()
        norm = torch.where(n_g > 0, n_rows.clamp(min=1), torch.full_like(n_rows, float(self.batch_size_per_image * ps.num_images)))
        losses["loss_box_reg"] = reg / norm
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    # ------------------------------------------------------------------ losses (fast_rcnn.py:355-571)
    def losses(self, predictions, proposals, merge_module, branch, update_prototype=False):
        kc = self.num_classes + 1
        te = self.text_encoder
        if branch == "pre_train":
            (scores, lta), deltas, feats = predictions
            dev = scores.device
            losses = {"loss_text_align": lta}
            nfg = [len(p[0]) for p in proposals]
            nbg = [len(p[1]) for p in proposals]
            assert all(b > 0 or a == 0 for a, b in zip(nfg, nbg)), "image with foreground but no background RoIs (fast_rcnn.py:383-385)"
            starts = [0]
            for a, b in zip(nfg, nbg):
                starts.append(starts[-1] + a + b)
            fg_idx = _ranges(starts[:-1], nfg, dev)
            bg_idx = _ranges([s + a for s, a in zip(starts[:-1], nfg)], nbg, dev)
            cls_fg = _cat([p[0].gt_classes_offline for p in proposals])
            cls_bg = _cat([p[1].gt_classes for p in proposals])
            any_fg = sum(nfg) != 0
            if any_fg:
                s = scores[torch.cat([fg_idx, bg_idx])]
                if self.dataset != ("cliparttrain",):
                    losses["loss_cls"] = self._mil(s, torch.cat([cls_fg, cls_bg]), len(fg_idx), len(bg_idx), True)
                else:
                    probs_fg = _cat([p[0].gt_probs_offline for p in proposals])
                    tgt = torch.cat([F.one_hot(cls_fg, kc) * probs_fg.max(1)[0].unsqueeze(1), F.one_hot(cls_bg, kc)]).float()
                    losses["loss_cls"] = self._mil(s, tgt, len(fg_idx), len(bg_idx), False, soft=True)
            else:
                losses["loss_cls"] = torch.zeros_like(lta)
            if update_prototype and any_fg:
                with torch.no_grad():
                    fn = feats.detach().float()
                    fn = fn / fn.norm(dim=1, keepdim=True)
                    f = fn[torch.cat([fg_idx, bg_idx])]
                    oh = F.one_hot(torch.cat([cls_fg, cls_bg]), kc).float()
                    te.per_class_feat.data = _prototype_ema(te.per_class_feat.data, f, oh, self.prototype_update_rate)
            cls_all = _cat([torch.cat([p[0].gt_classes_offline, p[1].gt_classes]) for p in proposals])
            pboxes = _cat([torch.cat([p[0].proposal_boxes.tensor, p[1].proposal_boxes.tensor]) for p in proposals])
            gboxes = _cat([torch.cat([p[0].gt_boxes.tensor, p[1].proposal_boxes.tensor]) for p in proposals])
            losses["loss_box_reg"] = self.box_reg_loss(pboxes, gboxes, deltas, cls_all)
            return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

        assert branch in ("step_one", "step_two")
        ((scores, lta), deltas, feats), ((scores_c, _), _) = predictions
        proposals, inst_c = proposals
        dev = scores.device
        na = [len(p[0]) for p in proposals]
        nb = [len(p[1]) for p in proposals]
        ng = [len(p[2]) for p in proposals]
        starts = [0]
        for a, b, g in zip(na, nb, ng):
            starts.append(starts[-1] + a + b + g)
        ia = _ranges(starts[:-1], na, dev)
        ib = _ranges([s + a for s, a in zip(starts[:-1], na)], nb, dev)
        ig = _ranges([s + a + b for s, a, b in zip(starts[:-1], na, nb)], ng, dev)
        calc_bg = sum(ng) != 0
        losses = {"loss_text_align": lta}
        cls_a = _cat([p[0].gt_classes for p in proposals])
        cls_g = _cat([p[2].gt_classes for p in proposals])
        s_a = scores[ia]
        losses["loss_cls"] = self._mil(torch.cat([s_a, scores[ig]]), torch.cat([cls_a, cls_g]), len(ia), len(ig), True)
        oh_a, oh_g = F.one_hot(cls_a, kc), F.one_hot(cls_g, kc)
        if update_prototype:
            with torch.no_grad():
                fn = feats.detach().float()
                fn = fn / fn.norm(dim=1, keepdim=True)
                f_a, f_b, f_g = fn[ia], fn[ib], fn[ig]
                rate = self.prototype_update_rate
                te.per_class_feat.data = _prototype_ema(te.per_class_feat.data, torch.cat([f_a, f_g]), torch.cat([oh_a, oh_g]).float(), rate)
            if sum(nb) != 0:
                pb_on = _cat([p[1].gt_probs_online for p in proposals])
                pb_off = _cat([p[1].gt_probs_offline for p in proposals])
                with torch.no_grad():
                    oh_b_on = F.one_hot(_cat([p[1].gt_classes_online for p in proposals]), kc)
                    oh_b_off = F.one_hot(_cat([p[1].gt_classes_offline for p in proposals]), kc)
                    f_abg = torch.cat([f_a, f_b, f_g])
                    te.prototype_b_online.data = _prototype_ema(te.prototype_b_online.data, f_abg, torch.cat([oh_a, oh_b_on, oh_g]).float(), rate)
                    te.prototype_b_offline.data = _prototype_ema(te.prototype_b_offline.data, f_abg, torch.cat([oh_a, oh_b_off, oh_g]).float(), rate)
                pa_on = _cat([p[0].gt_probs_online for p in proposals])
                pa_off = _cat([p[0].gt_probs_offline for p in proposals])
                m_a = merge_module(f_a, te.prototype_b_offline.data, te.prototype_b_online.data, pa_off, pa_on)
                losses["loss_merge_base"] = L.kl_div_from_probs(m_a, oh_a.float())
                m_b = merge_module(f_b, te.prototype_b_offline.data, te.prototype_b_online.data, pb_off, pb_on)
                s_b = scores[ib]
                p_b = F.softmax(s_b, dim=1)
                p_a = F.softmax(s_a, dim=1)
                losses["loss_merge_b"] = F.mse_loss(p_b, m_b)
                losses["loss_merge_a"] = F.mse_loss(p_a, oh_a.float())
                self._merge_ctx = (self._last_input[ia], self._last_input[ib], self._last_text, oh_a, m_b)
                if branch == "step_two":
                    keep = (m_b.max(1)[0] >= self.cls_b_thresh).detach()
                    if bool(keep.any()):
                        losses["loss_cls_b"] = L.kl_div_from_logits(s_b, m_b.detach(), row_mask=keep)
        if scores_c is not None:
            q = _cat([c.gt_probs for c in inst_c])
            losses["loss_distillation"] = L.kl_div_from_logits(scores_c, q)
        cls_on = _cat([torch.cat([p[0].gt_classes, p[1].gt_classes_online, p[2].gt_classes]) for p in proposals])
        pboxes = _cat([torch.cat([p[0].proposal_boxes.tensor, p[1].proposal_boxes.tensor, p[2].proposal_boxes.tensor]) for p in proposals])
        gboxes = _cat([torch.cat([p[0].gt_boxes.tensor, p[1].gt_boxes.tensor, p[2].proposal_boxes.tensor]) for p in proposals])
        norm = None if calc_bg else self.batch_size_per_image * len(proposals)
        losses["loss_box_reg"] = self.box_reg_loss(pboxes, gboxes, deltas, cls_on, normalizer=norm)
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    # ------------------------------------------------------------------ inference (fast_rcnn.py:648-671)
    def inference(self, predictions, proposals: List[Instances]):
        scores, deltas = predictions
        n = [len(p) for p in proposals]
        pb = _cat([p.proposal_boxes.tensor for p in proposals])
        boxes = self.box2box_transform.apply_deltas(deltas.float(), pb).split(n)
        probs = F.softmax(scores.float(), dim=-1).split(n)
        out = [fast_rcnn_inference_single_image(b, s, p.image_size, self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image)
               for b, s, p in zip(boxes, probs, proposals)]
        return [o[0] for o in out], [o[1] for o in out]
