"""Box arithmetic, matching, sampling and proposal selection on the device.

Restates the detectron2 0.5 behaviour the reference relies on (Matcher, subsample_labels,
Box2BoxTransform, DefaultAnchorGenerator, find_top_rpn_proposals; SURVEY.md §8c lists the call
sites) with device tensors throughout; NMS runs in the HIP kernel ``coin_nms_batched``.
"""
from __future__ import annotations

import functools
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import kernels as K
from .structures import Boxes, Instances

SCALE_CLAMP = math.log(1000.0 / 16)


class Matcher:
    """Column-wise argmax over an IoU matrix [num_gt, num_pred] + threshold bands -> labels in {-1,0,1}."""

    def __init__(self, thresholds: Sequence[float], labels: Sequence[int], allow_low_quality_matches: bool = False):
        th = [-float("inf")] + list(thresholds) + [float("inf")]
        assert all(lo <= hi for lo, hi in zip(th[:-1], th[1:])) and len(labels) == len(th) - 1
        self.thresholds, self.labels, self.allow_low_quality_matches = th, list(labels), allow_low_quality_matches

    def __call__(self, iou: torch.Tensor):
        """-> (matched gt index [P] int64, label [P] int8).  Pure elementwise / reduction ops: no host sync."""
        if iou.numel() == 0:
            n = iou.size(1)
            return iou.new_zeros((n,), dtype=torch.int64), iou.new_full((n,), self.labels[0], dtype=torch.int8)
        vals, matches = iou.max(dim=0)
        labels = torch.ones_like(matches, dtype=torch.int8)
        for l, lo, hi in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            labels = torch.where((vals >= lo) & (vals < hi), torch.full_like(labels, l), labels)
        if self.allow_low_quality_matches:
            best_per_gt = iou.max(dim=1, keepdim=True).values
            labels = torch.where((iou == best_per_gt).any(dim=0), torch.ones_like(labels), labels)
        return matches, labels

    def match_boxes(self, gt_boxes: Sequence[torch.Tensor], pred_boxes: torch.Tensor, empty_label=None, want_boxes: bool = True):
        """`self(pairwise_iou(gt_i, pred))` for a batch of images against one shared set of boxes, as one HIP launch sequence
        (coin_anchor_match: IoU, arg-max, threshold bands and the low-quality rule fused; bit-exact with the composed ops).
        -> (matched index [N, P] int64, label [N, P] int8, matched gt box [N, P, 4] or None).  An image without boxes gets index 0 and
        `empty_label` (default: the Matcher's own answer for an empty IoU matrix, its lowest band's label)."""
        th, lb = self.thresholds[1:-1], self.labels
        if len(th) == 1:
            lo = hi = th[0]
            bands = (lb[0], lb[0], lb[1])
        elif len(th) == 2:
            lo, hi = th
            bands = (lb[0], lb[1], lb[2])
        else:
            raise NotImplementedError("Matcher.match_boxes: one or two IoU thresholds")
        return K.anchor_match(list(gt_boxes), pred_boxes, lo, hi, bands, lb[0] if empty_label is None else empty_label,
                              self.allow_low_quality_matches, want_boxes)


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """Random fg / bg index subsets: two randperm draws, positives first (detectron2 sampling.py)."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero()[:, 0]
    negative = (labels == bg_label).nonzero()[:, 0]
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    p1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    p2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[p1], negative[p2]


def sample_labels(cls: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int) -> torch.Tensor:
    """Sync-free form of `subsample_labels` for a batch: cls [N, M] int8 / int64 (-1 = ignore, bg_label = negative, else positive)
    -> int8 [N, M]: 1 = chosen positive, 0 = chosen negative, -1 = not chosen; each row a uniformly random subset with
    |pos| = min(#pos, int(num_samples*positive_fraction)), |neg| = min(#neg, num_samples - |pos|).
    One uniform key per element (torch.rand) and a radix select per class in ONE launch (coin_sample_labels: the elements an
    ascending stable sort by key would rank first); no data-dependent shapes, so nothing is read back to the host.  The random
    stream differs from torch.randperm's (the reference's), the distribution does not."""
    keys = torch.rand(cls.shape, device=cls.device)
    return K.sample_labels(cls.contiguous(), keys, bg_label, num_samples, int(num_samples * positive_fraction))


def sample_masks(cls: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """`sample_labels` as boolean masks (chosen_pos, chosen_neg)."""
    out = sample_labels(cls, num_samples, positive_fraction, bg_label)
    return out == 1, out == 0


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp: float = SCALE_CLAMP):
        self.weights, self.scale_clamp = tuple(float(w) for w in weights), scale_clamp

    def get_deltas(self, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
        sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
        tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
        tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
        """detectron2 Box2BoxTransform.apply_deltas on (x, y) / (w, h) pairs, a third of the launches of the column-by-column form.
        Every multiplication by 0.5 is exact and the products / sums are separate roundings as there.  The division by the box weights
        is a TRUE division by a device tensor; detectron2's `deltas[:, 0::4] / wx` with a Python float is evaluated by torch as a
        multiplication by the reciprocal on the GPU, so dx / dy / dw / dh may differ from it by one ulp for weights whose reciprocal is
        inexact (10, 5) -- within the 1e-3 px the inference golden allows, not bit-identical (round-4 ADVICE)."""
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        wh = boxes[:, 2:] - boxes[:, :2]                                   # [R, 2] (w, h)
        ctr = torch.add(boxes[:, :2], wh, alpha=0.5)                       # (cx, cy)
        d = deltas.reshape(deltas.shape[0], -1, 4) / self._weights_on(deltas.device)   # [R, k, 4] / (wx, wy, ww, wh)
        dwh = torch.clamp(d[..., 2:], max=self.scale_clamp)
        pc = d[..., :2] * wh[:, None, :]
        pc = pc + ctr[:, None, :]
        pwh = torch.exp(dwh) * wh[:, None, :]
        return torch.cat((torch.add(pc, pwh, alpha=-0.5), torch.add(pc, pwh, alpha=0.5)), dim=-1).reshape(deltas.shape)

    _wdev = None

    def _weights_on(self, device):
        """(wx, wy, ww, wh) as a device tensor, uploaded once per device (one blocking pageable copy on first use, none per call)."""
        if self._wdev is None:
            self._wdev = {}
        t = self._wdev.get(device)
        if t is None:
            t = self._wdev[device] = torch.tensor(self.weights, dtype=torch.float32).to(device)
        return t


def cell_anchors(sizes: Sequence[float], aspect_ratios: Sequence[float]) -> torch.Tensor:
    out = []
    for s in sizes:
        area = s ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(base: torch.Tensor, grid_hw: Tuple[int, int], stride: int, offset: float, device) -> torch.Tensor:
    """[H*W*A, 4] anchors ordered (y, x, anchor), matching logits.permute(0,2,3,1).flatten(1)."""
    gh, gw = grid_hw
    sx = torch.arange(offset * stride, gw * stride, step=stride, dtype=torch.float32, device=device)
    sy = torch.arange(offset * stride, gh * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1)
    return (shifts.view(-1, 1, 4) + base.to(device).view(1, -1, 4)).reshape(-1, 4)


def add_ground_truth_to_proposals(gt, proposals: List[Instances]) -> List[Instances]:
    logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    out = []
    for g, p in zip(gt, proposals):
        gb = g if isinstance(g, Boxes) else g.gt_boxes
        gp = Instances(p.image_size)
        gp.proposal_boxes = gb
        gp.objectness_logits = torch.full((len(gb),), logit, device=p.objectness_logits.device)
        out.append(Instances.cat([p, gp]))
    return out


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float, max_keep: int = -1) -> torch.Tensor:
    """Indices kept by greedy NMS in descending-score order (HIP kernel; device only)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes.float()[order].contiguous().unsqueeze(0)
    counts = torch.tensor([n], dtype=torch.int32, device=boxes.device)
    keep, num = K.nms_batched(b, counts, iou_threshold, n if max_keep < 0 else max_keep)
    return order[keep[0, : int(num[0])].long()]


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    offsets = idxs.to(boxes) * (boxes.max() + 1.0)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


class PackedProposals:
    """Fixed-shape RPN output of the sync-free path: boxes [N,P,4], logits [N,P], valid [N,P] (bool)."""

    def __init__(self, boxes, logits, valid, image_sizes):
        self.boxes, self.logits, self.valid, self.image_sizes = boxes, logits, valid, image_sizes

    def __len__(self):
        return self.boxes.shape[0]


@functools.lru_cache(maxsize=64)
def _sizes_on_device(image_sizes, device):
    """Image sizes as a device tensor, cached: a pageable host->device copy is a blocking, stream-draining call, and the
    training batches repeat a handful of size tuples."""
    return torch.tensor(image_sizes, dtype=torch.float32, device=device)


NMS_MAX_CANDIDATES = 16384   # coin_nms_batched (include/coin_hip.h): boxes per image


@functools.lru_cache(maxsize=64)
def _clip_limits(image_sizes, device):
    return torch.tensor([[w, h, w, h] for h, w in image_sizes], dtype=torch.float32, device=device).view(len(image_sizes), 1, 4)


@torch.no_grad()
def find_top_rpn_proposals(proposals: torch.Tensor, logits: torch.Tensor, image_sizes: List[Tuple[int, int]], nms_thresh: float,
                           pre_nms_topk: int, post_nms_topk: int, min_box_size: float, training: bool, packed: bool = False,
                           level_sizes: Optional[List[int]] = None):
    """detectron2 find_top_rpn_proposals, batched: per-level top-k -> clip -> drop empties -> NMS inside each level -> top-k over all.
    proposals [N, A, 4], logits [N, A]; `level_sizes` = anchors per feature level when A is the concatenation of several levels
    (FPN extension): `pre_nms_topk` then applies to EVERY level and boxes of different levels never suppress each other (batched_nms
    with the level as the group id -- the coordinate-offset form), as in detectron2's multi-level RPN.  One host sync (the keep counts)."""
    n, a = logits.shape
    lvl = None
    if level_sizes is not None and len(level_sizes) > 1:
        assert sum(level_sizes) == a
        bs, ls, ids, off = [], [], [], 0
        for li, al in enumerate(level_sizes):
            kl = min(al, pre_nms_topk)
            tl, idx = logits[:, off:off + al].float().sort(descending=True, dim=1)
            tl, idx = tl[:, :kl], idx[:, :kl]
            bs.append(torch.gather(proposals[:, off:off + al].float(), 1, idx.unsqueeze(-1).expand(-1, -1, 4)))
            ls.append(tl)
            ids.append(torch.full((n, kl), li, dtype=torch.float32, device=logits.device))
            off += al
        # the candidates of all levels in ONE descending score order (ties: lower level first): the order NMS visits them in and the
        # order the final top-k is taken in
        if sum(l.shape[1] for l in ls) > NMS_MAX_CANDIDATES:
            raise ValueError(f"find_top_rpn_proposals: {sum(l.shape[1] for l in ls)} candidates over {len(ls)} levels exceed coin_nms_batched's "
                             f"{NMS_MAX_CANDIDATES}; lower MODEL.RPN.PRE_NMS_TOPK_TRAIN / _TEST (it applies per level: detectron2's FPN configs use 2000 / 1000)")
        top_logits, order = torch.cat(ls, dim=1).sort(descending=True, dim=1, stable=True)
        boxes = torch.gather(torch.cat(bs, dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 4))
        lvl = torch.gather(torch.cat(ids, dim=1), 1, order)
    else:
        k = min(a, pre_nms_topk)
        top_logits, idx = logits.float().sort(descending=True, dim=1)
        top_logits, idx = top_logits[:, :k], idx[:, :k]
        boxes = torch.gather(proposals.float(), 1, idx.unsqueeze(-1).expand(-1, -1, 4))
    if training and not packed and not bool(torch.isfinite(boxes).all() & torch.isfinite(top_logits).all()):
        raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
    hw = _sizes_on_device(tuple(tuple(int(v) for v in sz) for sz in image_sizes), boxes.device)  # [N, 2] (h, w)
    lim = _clip_limits(tuple(tuple(int(v) for v in sz) for sz in image_sizes), boxes.device)   # [N, 1, 4]: (w, h, w, h) per image
    boxes = torch.minimum(boxes.clamp(min=0), lim)                         # clip: the same two operations per coordinate as before
    valid = ((boxes[..., 2:] - boxes[..., :2]) > min_box_size).all(dim=-1)
    valid &= torch.isfinite(boxes).all(dim=-1) & torch.isfinite(top_logits)
    # stable partition: valid boxes first, still in descending score order.  By prefix sums instead of a stable argsort of the flags (the
    # partition is unique, so the order is the same): element i goes to slot (number of valid before it) if valid, else to
    # (number of valid in the row) + (number of invalid before it); `order` is the inverse of that permutation.  On the GPU the segmented
    # sort of [4, 12 000] flags took 80 us of the proposal chain per step, the scan + scatter take ~25.
    kk = valid.shape[1]
    cv = torch.cumsum(valid.to(torch.int64), dim=1)
    nv = cv[:, -1:] if kk else cv.new_zeros((valid.shape[0], 1))
    ar = torch.arange(kk, device=valid.device, dtype=torch.int64).unsqueeze(0)
    dest = torch.where(valid, cv - 1, nv + ar - cv)          # (ar + 1 - cv) invalid up to and including i, minus one
    order = torch.empty_like(dest).scatter_(1, dest, ar.expand_as(dest))
    boxes = torch.gather(boxes, 1, order.unsqueeze(-1).expand(-1, -1, 4)).contiguous()
    top_logits = torch.gather(top_logits, 1, order)
    counts = nv.squeeze(1).to(torch.int32)
    if lvl is None:
        keep, num = K.nms_batched(boxes, counts, nms_thresh, post_nms_topk)
    else:   # level-wise NMS: boxes of level l are moved by l * (largest image extent + 1), so that levels cannot overlap
        shift = torch.gather(lvl, 1, order) * (float(max(max(int(v) for v in sz) for sz in image_sizes)) + 1.0)
        keep, num = K.nms_batched((boxes + shift.unsqueeze(-1)).contiguous(), counts, nms_thresh, post_nms_topk)
    if packed:
        # fixed-shape result, nothing read back: [N, post_nms_topk] rows + a validity mask
        p = min(post_nms_topk, keep.shape[1])
        ki = keep[:, :p].long().clamp_(min=0, max=boxes.shape[1] - 1)
        ok = torch.arange(p, device=boxes.device).unsqueeze(0) < num.unsqueeze(1)
        pb = torch.gather(boxes, 1, ki.unsqueeze(-1).expand(-1, -1, 4))
        pl = torch.gather(top_logits, 1, ki)
        return PackedProposals(torch.where(ok.unsqueeze(-1), pb, torch.zeros_like(pb)), torch.where(ok, pl, torch.full_like(pl, -1e4)), ok, image_sizes)
    num_host = num.tolist()
    out = []
    for i, size in enumerate(image_sizes):
        ki = keep[i, : num_host[i]].long()
        res = Instances(size)
        res.proposal_boxes = Boxes(boxes[i][ki])
        res.objectness_logits = top_logits[i][ki]
        out.append(res)
    return out


def detector_postprocess(results: Instances, output_height: int, output_width: int) -> Instances:
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    boxes = results.pred_boxes if results.has("pred_boxes") else results.proposal_boxes
    boxes.scale(sx, sy)
    boxes.clip(results.image_size)
    return results[boxes.nonempty()]
