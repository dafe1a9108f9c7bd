"""DualTeacherRPN on the device (registered under the reference's name).

Mirrors coin/modeling/proposal_generator/rpn.py:16-345 on top of a restated detectron2 RPN
(StandardRPNHead, DefaultAnchorGenerator, Matcher, find_top_rpn_proposals).  The 3x3 / 1x1 head
convolutions go through torch (MIOpen); the BCE + L1 losses over all N x 62 250 anchors and their
gradients are ONE fused HIP launch (``coin_rpn_losses_fwd_bwd``), the objectness distillation KL another
(``coin_kl_div_fwd_bwd`` mode 2), proposal NMS is ``coin_nms_batched``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import layers as L
from ..box_ops import Box2BoxTransform, Matcher, cell_anchors, find_top_rpn_proposals, grid_anchors, sample_labels, sample_masks, subsample_labels
from ..registry import PROPOSAL_GENERATOR_REGISTRY
from ..structures import Boxes, ImageList, Instances, pairwise_iou


class StandardRPNHead(nn.Module):
    def __init__(self, in_channels: int, num_anchors: int, box_dim: int = 4):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, 1)
        for l in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def forward(self, features: List[torch.Tensor]):
        lg, dl = [], []
        for x in features:
            t = L.conv_bias_relu(x, self.conv)   # 3x3 + bias + ReLU: hand-written GEMM in the bf16 mode
            lg.append(L.conv2d(t, self.objectness_logits))
            dl.append(L.conv2d(t, self.anchor_deltas))
        return lg, dl


class DefaultAnchorGenerator(nn.Module):
    """detectron2 DefaultAnchorGenerator.  The reference uses one feature level (res4); with several levels (the FPN extension,
    coin_amd/modeling/fpn.py) level i takes sizes[i] / aspect_ratios[i] (a single entry is shared by all levels)."""
    box_dim = 4

    def __init__(self, sizes, aspect_ratios, strides, offset: float = 0.0):
        super().__init__()
        self.strides, self.offset = list(strides), offset
        n = len(self.strides)
        sizes = list(sizes) * n if len(sizes) == 1 else list(sizes)
        ratios = list(aspect_ratios) * n if len(aspect_ratios) == 1 else list(aspect_ratios)
        assert len(sizes) == n and len(ratios) == n, "one anchor size / ratio list per feature level (or one shared list)"
        for i in range(n):
            self.register_buffer(f"cell_anchors_{i}", cell_anchors(sizes[i], ratios[i]), persistent=False)
        self._cache: Dict[Tuple, torch.Tensor] = {}

    @property
    def num_anchors(self):
        return [getattr(self, f"cell_anchors_{i}").shape[0] for i in range(len(self.strides))]

    def forward(self, features: List[torch.Tensor]) -> List[Boxes]:
        return [self.for_hw(tuple(f.shape[-2:]), f.device, i)[0] for i, f in enumerate(features)]

    def for_hw(self, hw: Tuple[int, int], device, level: int = 0) -> List[Boxes]:
        """Anchors of an [h, w] feature map (they depend on nothing else): lets the labelling run before the backbone."""
        key = (tuple(int(v) for v in hw), str(device), level)
        if key not in self._cache:
            self._cache[key] = grid_anchors(getattr(self, f"cell_anchors_{level}"), key[0], self.strides[level], self.offset, device)
        return [Boxes(self._cache[key])]


@PROPOSAL_GENERATOR_REGISTRY.register()
class DualTeacherRPN(nn.Module):
    def __init__(self, *, in_features, head, anchor_generator, anchor_matcher, box2box_transform, batch_size_per_image,
                 positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0, anchor_boundary_thresh=-1.0,
                 loss_weight=None, BG_TRAIN=True):
        super().__init__()
        self.in_features, self.rpn_head, self.anchor_generator = in_features, head, anchor_generator
        self.anchor_matcher, self.box2box_transform = anchor_matcher, box2box_transform
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh, self.min_box_size, self.anchor_boundary_thresh = nms_thresh, float(min_box_size), anchor_boundary_thresh
        self.loss_weight = loss_weight or {"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0, "loss_rpn_distillation": 0.1}
        self.BG_TRAIN = BG_TRAIN
        # sync-free mode (pre_train branch): anchor sampling by random keys + one sort, fixed-shape proposals; the host
        # never waits for the device inside the step.  Off = the reference's randperm stream (used by the golden tests).
        self.sync_free = False
        self.sync_free_step = False  # the same for the step_one / step_two branches (cfg.AMD.SYNC_FREE_STEP)
        self._prefetched = None

    def prefetch_labels(self, feature_hw, device, gt_instances):
        """Anchor labelling + sampling of the sync-free pre_train step ahead of time (it depends only on the anchor grid and
        the teacher boxes, not on the network): the caller runs it on a side stream concurrently with the backbone."""
        self._prefetched = self.label_and_sample_anchors_sync_free(self.anchor_generator.for_hw(feature_hw, device), gt_instances)
        return self._prefetched

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        ch = input_shape[in_features[0]].channels
        ag = DefaultAnchorGenerator(cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                                    [input_shape[f].stride for f in in_features], cfg.MODEL.ANCHOR_GENERATOR.OFFSET)
        assert len(set(ag.num_anchors)) == 1 and len({input_shape[f].channels for f in in_features}) == 1, "one shared RPN head"
        assert tuple(cfg.MODEL.RPN.BBOX_REG_WEIGHTS) == (1.0, 1.0, 1.0, 1.0), "the fused RPN loss kernel uses unit box weights"
        return cls(
            in_features=in_features, head=StandardRPNHead(ch, ag.num_anchors[0]), anchor_generator=ag,
            anchor_matcher=Matcher(cfg.MODEL.RPN.IOU_THRESHOLDS, cfg.MODEL.RPN.IOU_LABELS, allow_low_quality_matches=True),
            box2box_transform=Box2BoxTransform(cfg.MODEL.RPN.BBOX_REG_WEIGHTS), batch_size_per_image=cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            positive_fraction=cfg.MODEL.RPN.POSITIVE_FRACTION,
            pre_nms_topk=(cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST),
            post_nms_topk=(cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST),
            nms_thresh=cfg.MODEL.RPN.NMS_THRESH, min_box_size=cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            anchor_boundary_thresh=cfg.MODEL.RPN.BOUNDARY_THRESH,
            loss_weight={"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                         "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT,
                         "loss_rpn_distillation": cfg.CLOUD.LOSS_DISTILLATION_WEIGHT},
            BG_TRAIN=cfg.CLOUD.BG_TRAIN)

    # ------------------------------------------------------------------ forward
    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], gt_instances=None, branch=None, packed: Optional[bool] = None):
        """packed=True forces the fixed-shape proposal result (PackedProposals: no host round trip) also outside training: the
        inference pass of the EMA teacher is issued without a sync (OpenVocabularyRCNN.inference_begin)."""
        feats = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(feats)
        lg, dl = self.rpn_head(feats)
        # (N, A, H, W) -> (N, H*W*A) ; (N, A*4, H, W) -> (N, H*W*A, 4).  With channels-last activations both are views.
        logits = [s.permute(0, 2, 3, 1).flatten(1) for s in lg]
        deltas = [x.view(x.shape[0], -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2) for x in dl]
        self._level_sizes = None
        if len(logits) > 1:
            # several feature levels (FPN extension): the levels' anchors are concatenated into ONE anchor set, so that labelling and
            # losses below are the single-level code of the reference applied to the union (as detectron2's multi-level RPN does);
            # the proposal selection keeps the levels apart (per-level top-k, NMS inside a level: box_ops.find_top_rpn_proposals)
            self._level_sizes = [int(l.shape[1]) for l in logits]
            anchors, logits, deltas = [Boxes.cat(anchors)], [torch.cat(logits, dim=1)], [torch.cat(deltas, dim=1)]
        losses = {}
        if self.training and branch != "test":
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            if branch == "pre_train" and self.sync_free:
                if self._prefetched is not None:
                    (labels, gt_boxes), self._prefetched = self._prefetched, None
                    assert labels[0].shape[0] == len(anchors[0]), "prefetched anchor labels do not match the feature map"
                else:
                    labels, gt_boxes = self.label_and_sample_anchors_sync_free(anchors, gt_instances)
                losses = self.losses(anchors, logits, labels, deltas, gt_boxes)
            elif branch == "pre_train":
                labels, gt_boxes = self.label_and_sample_anchors(anchors, gt_instances, branch)
                losses = self.losses(anchors, logits, labels, deltas, gt_boxes)
            elif branch in ("step_one", "step_two"):
                ia, ic = [g[0] for g in gt_instances], [g[2] for g in gt_instances]
                if self.sync_free_step:
                    labels, gt_boxes, midx, dlabels = self.label_and_sample_anchors_step_sync_free(anchors, [ia, ic])
                else:
                    labels, gt_boxes, midx, dlabels = self.label_and_sample_anchors(anchors, [ia, ic], branch)
                teacher = [c.gt_probs[:, :-1].sum(1)[m] if len(c) != 0 else torch.zeros_like(m, dtype=torch.float32)
                           for c, m in zip(ic, midx)]
                losses = self.losses(anchors, logits, labels, deltas, gt_boxes, calc_bg=self.BG_TRAIN)
                losses.update(self.losses(anchors, logits, dlabels, None, None, teacher_probs=teacher, only_distillation=True))
            else:
                raise NotImplementedError
        if packed is None:
            packed = self.training and ((self.sync_free and branch == "pre_train") or (self.sync_free_step and branch in ("step_one", "step_two")))
        proposals = self.predict_proposals(anchors, logits, deltas, images.image_sizes, packed=packed)
        return proposals, losses

    # ------------------------------------------------------------------ labels
    def _subsample_labels(self, label: torch.Tensor) -> torch.Tensor:
        pos, neg = subsample_labels(label, self.batch_size_per_image, self.positive_fraction, 0)
        label.fill_(-1)
        label.scatter_(0, pos, 1)
        label.scatter_(0, neg, 0)
        return label

    @torch.no_grad()
    def label_and_sample_anchors_sync_free(self, anchors: List[Boxes], gt_instances):
        """pre_train labelling (rpn.py:139-197) without host round trips: the same matcher (one fused launch sequence for the batch:
        coin_anchor_match), sampling via `sample_labels`.  An image without boxes: every anchor ignored, matched boxes zero (:190-193)."""
        a = Boxes.cat(anchors)
        _, lab, mb = self.anchor_matcher.match_boxes([g.gt_boxes.tensor for g in gt_instances], a.tensor, empty_label=-1)
        out = sample_labels(lab, self.batch_size_per_image, self.positive_fraction, 0)
        return list(out), list(mb)

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors: List[Boxes], gt_instances, branch):
        anchors = Boxes.cat(anchors)
        labels_out, boxes_out = [], []
        if branch == "pre_train":
            _, labs, mbs = self.anchor_matcher.match_boxes([g.gt_boxes.tensor for g in gt_instances], anchors.tensor)
            for g, lab, mb in zip(gt_instances, labs, mbs):
                lab = self._subsample_labels(lab)
                if len(g.gt_boxes) == 0:
                    lab[:] = -1
                labels_out.append(lab)
                boxes_out.append(mb)
            return labels_out, boxes_out
        ga, gc = gt_instances
        midx_out, dist_out = [], []
        idxs, labs, _ = self.anchor_matcher.match_boxes([torch.cat([a.gt_boxes.tensor.reshape(-1, 4), c.gt_boxes.tensor.reshape(-1, 4)])
                                                         for a, c in zip(ga, gc)], anchors.tensor, want_boxes=False)
        for a, c, idx, lab in zip(ga, gc, idxs, labs):
            ba, bc = a.gt_boxes, c.gt_boxes
            both = Boxes.cat([ba, bc])
            in_c = (idx >= len(ba)) & (idx < len(both))
            is_bg = lab == 0
            fg_c = in_c & ~is_bg
            didx = idx - len(ba)
            didx[~fg_c] = 0
            lab[fg_c] = -1
            idx = idx.clone()
            idx[in_c] = 0
            dlab = torch.zeros_like(lab)
            dlab[fg_c] = 1
            lab = self._subsample_labels(lab)
            if len(ba) == 0:
                mb = torch.zeros_like(anchors.tensor)
                lab[~(in_c & is_bg)] = -1
            else:
                mb = ba.tensor[idx]
            labels_out.append(lab)
            boxes_out.append(mb)
            midx_out.append(didx)
            dist_out.append(dlab)
        return labels_out, boxes_out, midx_out, dist_out

    @torch.no_grad()
    def label_and_sample_anchors_step_sync_free(self, anchors: List[Boxes], gt_instances):
        """step_one / step_two labelling (rpn.py:199-254) without host round trips: the same matcher and the same rules written
        with `torch.where` (a boolean-mask assignment synchronises on the device), sampling via `sample_masks`.
        -> (labels, matched A boxes, matched C index, distillation labels), one entry per image."""
        a = Boxes.cat(anchors)
        ga, gc = gt_instances
        labs, boxes, midx, dlabs = [], [], [], []
        idxs, labs_m, _ = self.anchor_matcher.match_boxes([torch.cat([ta.gt_boxes.tensor.reshape(-1, 4), tc.gt_boxes.tensor.reshape(-1, 4)])
                                                           for ta, tc in zip(ga, gc)], a.tensor, want_boxes=False)
        for i, (ta, tc) in enumerate(zip(ga, gc)):
            ba, bc = ta.gt_boxes, tc.gt_boxes
            la, lc = len(ba), len(bc)
            if la + lc == 0:
                z = torch.zeros(len(a), dtype=torch.int64, device=a.tensor.device)
                labs.append(torch.full((len(a),), -1, dtype=torch.int8, device=a.tensor.device))  # `lab[~(in_c & is_bg)] = -1` with in_c empty
                boxes.append(torch.zeros_like(a.tensor))
                midx.append(z)
                dlabs.append(z.to(torch.int8))
                continue
            idx, lab = idxs[i], labs_m[i]
            in_c = (idx >= la) & (idx < la + lc)
            is_bg = lab == 0
            fg_c = in_c & ~is_bg
            didx = torch.where(fg_c, idx - la, torch.zeros_like(idx))
            lab = torch.where(fg_c, torch.full_like(lab, -1), lab)
            idx_a = torch.where(in_c, torch.zeros_like(idx), idx)
            if la == 0:
                mb = torch.zeros_like(a.tensor)
                keep_mask = in_c & is_bg
            else:
                mb = ba.tensor[idx_a]
                keep_mask = None
            labs.append((lab, keep_mask))
            boxes.append(mb)
            midx.append(didx)
            dlabs.append(fg_c.to(lab.dtype))
        # one batched sampling call for the images that have targets
        todo = [i for i, l in enumerate(labs) if isinstance(l, tuple)]
        if todo:
            stack = torch.stack([labs[i][0] for i in todo]).to(torch.int64)
            pos, neg = sample_masks(stack, self.batch_size_per_image, self.positive_fraction, 0)
            out = torch.where(pos, 1, torch.where(neg, 0, -1)).to(torch.int8)
            for j, i in enumerate(todo):
                keep_mask = labs[i][1]
                labs[i] = out[j] if keep_mask is None else torch.where(keep_mask, out[j], torch.full_like(out[j], -1))
        return labs, boxes, midx, dlabs

    # ------------------------------------------------------------------ losses
    def losses(self, anchors, logits, gt_labels, deltas, gt_boxes, teacher_probs=None, only_distillation=False, calc_bg=True):
        num_images = len(gt_labels)
        labels = torch.stack(gt_labels)
        lg = logits[0] if len(logits) == 1 else torch.cat(logits, dim=1)
        if not only_distillation:
            a = Boxes.cat(anchors).tensor
            dl = deltas[0] if len(deltas) == 1 else torch.cat(deltas, dim=1)
            cls_sum, loc_sum = L.rpn_losses(lg, dl, labels, a, torch.stack(gt_boxes), min_label=0 if calc_bg else 1)
            normalizer = self.batch_size_per_image * num_images
            if calc_bg:
                cls = cls_sum / normalizer
            else:
                cls = cls_sum / (labels >= 1).sum().clamp(min=1).float()
            out = {"loss_rpn_cls": cls, "loss_rpn_loc": loc_sum / normalizer}
        else:
            assert teacher_probs is not None, "distillation need teacher probs"
            valid = labels > 0
            out = {}
            if self.sync_free_step or bool(valid.any()):  # rpn.py:336-340: the term is dropped when no anchor matches a private box
                # (sync-free: the masked mean over zero anchors is 0 and the term stays in the dict)
                out["loss_rpn_distillation"] = L.kl_div_binary(lg.reshape(-1), torch.stack(teacher_probs).reshape(-1), valid.reshape(-1))
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    # ------------------------------------------------------------------ proposals
    @torch.no_grad()
    def predict_proposals(self, anchors, logits, deltas, image_sizes, packed: bool = False):
        n = deltas[0].shape[0]
        a = anchors[0].tensor
        d = deltas[0].detach().reshape(-1, 4)
        boxes = self.box2box_transform.apply_deltas(d, a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4)
        return find_top_rpn_proposals(boxes, logits[0].detach(), image_sizes, self.nms_thresh, self.pre_nms_topk[self.training],
                                      self.post_nms_topk[self.training], self.min_box_size, self.training, packed=packed,
                                      level_sizes=getattr(self, "_level_sizes", None))


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name).from_config(cfg, input_shape)
