"""Dual-teacher box matching of the target-detector step: the (A, B, C) split.

Behaviour of coin/engine/trainer.py:338-485 (``match_dual_teacher``, ``match_boxes``, ``merge_boxes``) and
coin/utils/util.py:434-507 (``delete_duplicate_boxes``, ``filter_result``, ``online_boxes_merging``):

    A  consistent   a cloud ("online") box and a CLIP-teacher ("offline") box overlap (IoU >= MATCHER.IOU_THRESHOLDS)
                    and carry the same label (tag 'RPN': any label)
    B  inconsistent they overlap but the labels differ
    C  private      seen by only one of the two teachers

The reference walks ``Instances`` objects, Python sets and ``random.randint`` tie-breaks on the CPU.  Here the same
decisions are taken on plain index lists into the two detection sets (a few dozen boxes per image), the tie-breaks draw
from the same ``random`` stream in the same order, and the result is materialised once at the end.  It runs on CPU tensors:
the only device -> host traffic of the training step is the teacher's <= 100 detections per image.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..structures import Boxes, Instances, pairwise_iou


def weighted_box_fusion_split(box_a, box_b, score_a, score_b):
    """coin/layers/nms.py:24-31: per-pair score-weighted average of two boxes."""
    w = torch.stack((score_a, score_b), dim=1)
    w = w / w.sum(dim=1, keepdim=True)
    return box_a * w[:, 0:1] + box_b * w[:, 1:]


def _iou_pairs(a: torch.Tensor, b: torch.Tensor, thr: float) -> torch.Tensor:
    return (pairwise_iou(Boxes(a), Boxes(b)) >= thr).nonzero()


def _identical_groups(boxes: torch.Tensor) -> Tuple[List[int], List[List[int]]]:
    """Rows that share their coordinates with another row.  Candidates are rows with an equal coordinate sum (visited in
    ascending order of that sum); a candidate set is a group iff the differences to its first member sum to zero
    (util.py:436-451).  -> (rows outside every group, ascending; groups).
    One fp32 row sum on the tensor, then plain Python on the few dozen values (a torch call per candidate costs more than the
    whole bookkeeping)."""
    sums = boxes.sum(1).tolist()
    by_sum: Dict[float, List[int]] = {}
    for i, v in enumerate(sums):
        by_sum.setdefault(v, []).append(i)
    in_group = [False] * len(sums)
    groups = []
    for v in sorted(by_sum):
        members = by_sum[v]
        if len(members) > 1:
            idx = torch.tensor(members, dtype=torch.long)
            if float((boxes[idx] - boxes[members[0]]).sum()) == 0.0:
                groups.append(members)
                for m in members:
                    in_group[m] = True
    return [i for i, g in enumerate(in_group) if not g], groups


def _pick(group: Sequence[int]) -> int:
    return group[random.randint(0, len(group) - 1)]


def _overlap_groups(boxes: torch.Tensor, thr: float) -> List[List[int]]:
    """Groups (size > 1) of mutually overlapping boxes, merged transitively the way util.py:459-483 does it."""
    iou = (pairwise_iou(Boxes(boxes), Boxes(boxes)) >= thr).tolist()
    sets = [{j for j, hit in enumerate(row) if hit} for row in iou]
    if all(len(s_) <= 1 for s_ in sets):
        return []

    def absorb(i: int, visited: List[int]) -> set:
        for j in list(sets[i]):
            if j != i and j not in visited and sets[j] - sets[i]:
                sets[i] = sets[i] | absorb(j, visited + [i])
        return sets[i]

    for i in range(len(sets)):
        for j in list(sets[i]):
            if j != i:
                sets[i] = sets[i] | absorb(j, [i])
        for j in sets[i]:
            if j != i:
                sets[j] = set()
    return [sorted(s_) for s_ in sets if len(s_) > 1]


class _Side:
    """One teacher's detections as plain CPU tensors."""

    def __init__(self, inst: Instances):
        self.size = inst.image_size
        self.boxes = inst.gt_boxes.tensor.detach().float().cpu().reshape(-1, 4)
        self.classes = inst.gt_classes.detach().cpu().long()
        self.scores = inst.scores.detach().float().cpu()
        self.probs = inst.probs.detach().float().cpu()

    def __len__(self):
        return self.boxes.shape[0]


def _dedupe_rows(boxes: torch.Tensor) -> List[int]:
    """delete_duplicate_boxes (util.py:434-457) as a row order: rows outside duplicate groups first, then one random member
    per group."""
    keep, groups = _identical_groups(boxes)
    return keep + [_pick(g) for g in groups]


def match_dual_teacher(online_result: Dict[str, Instances], offline_result: Instances, tag: str, iou_threshold: float = 0.5,
                       weight_for_box_a: float = 1.0, device=None):
    """-> (A, B, C) Instances for tag 'RCNN', (A, None, C) for 'RPN' (field names as trainer.py:390-455)."""
    on, off = _Side(online_result[tag]), _Side(offline_result)
    size = online_result[tag].image_size
    # common: list of (source of the "online" row, source of the "offline" row); a source is ('on'|'off', index)
    if len(on) == 0 and len(off) == 0:
        common, off_only, on_only = [], [], []
    elif len(on) == 0:
        fg = (off.scores > 0.8).tolist()
        common = [(("off", i), ("off", i)) for i, f in enumerate(fg) if f]
        off_only, on_only = [i for i, f in enumerate(fg) if not f], []
    elif len(off) == 0:
        common = [(("on", i), ("on", i)) for i in range(len(on))]
        off_only, on_only = [], []
    else:
        uniq, groups = _identical_groups(off.boxes)
        uniq_t = torch.tensor(uniq, dtype=torch.long)
        pairs = _iou_pairs(on.boxes, off.boxes[uniq_t], iou_threshold)
        common = [(("on", int(i)), ("off", uniq[int(j)])) for i, j in pairs.tolist()]
        matched_off = {int(j) for j in pairs[:, 1].tolist()}
        off_only = [uniq[j] for j in range(len(uniq)) if j not in matched_off]
        used_on = [int(i) for i in pairs[:, 0].tolist()]
        for grp in groups:  # one box reported under several labels by the offline teacher's class-wise NMS
            hits = _iou_pairs(on.boxes, off.boxes[torch.tensor(grp)], iou_threshold)
            if hits.shape[0] == 0:
                off_only.append(_pick(grp))
                continue
            first = int(hits[0, 0])
            agree = [g for g in grp if int(off.classes[g]) == int(on.classes[first])]
            used_on.append(first)
            if agree:
                assert len(agree) == 1, "identical offline boxes with identical labels"
                common.append((("on", first), ("off", agree[0])))
            else:
                common.append((("on", first), ("off", _pick(grp))))
        common = _resolve_overlapping_online(on, off, common)
        used = set(used_on)
        on_only = [i for i in range(len(on)) if i not in used]

    sides = {"on": on, "off": off}
    both = {attr: torch.cat([getattr(on, attr), getattr(off, attr)]) if len(on) and len(off) else getattr(on if len(on) else off, attr)
            for attr in ("boxes", "classes", "scores", "probs")}
    base = {"on": 0, "off": len(on) if len(on) and len(off) else 0}

    def take(attr, srcs):  # rows of the two detection sets in the given order: one gather instead of a tensor index per row
        if not srcs:
            return getattr(off if len(off) else on, attr)[:0]
        return both[attr][torch.tensor([base[s_] + i for s_, i in srcs], dtype=torch.long)]

    # ---- C: private boxes (offline-only first, then online-only)
    c_src = [("off", i) for i in off_only] + [("on", i) for i in on_only]
    c = Instances(size)
    c.gt_boxes = Boxes(take("boxes", c_src).reshape(-1, 4))
    c.gt_classes = take("classes", c_src)
    c.gt_scores = take("scores", c_src)
    c.gt_probs = take("probs", c_src)

    def build(rows: List[Tuple], with_b_labels: bool) -> Instances:
        s_on, s_off = [r[0] for r in rows], [r[1] for r in rows]
        box_on, box_off = take("boxes", s_on).reshape(-1, 4), take("boxes", s_off).reshape(-1, 4)
        sc_on, sc_off = take("scores", s_on), take("scores", s_off)
        fused = weighted_box_fusion_split(box_on, box_off, sc_on, sc_off) if weight_for_box_a != 1.0 else box_on
        order = _dedupe_rows(fused)
        sel = torch.tensor(order, dtype=torch.long)
        inst = Instances(size)
        inst.gt_boxes = Boxes(fused[sel].reshape(-1, 4))
        if with_b_labels:
            inst.gt_classes_offline = take("classes", s_off)[sel]
            inst.gt_classes_online = take("classes", s_on)[sel]
        else:
            inst.gt_classes = take("classes", s_off)[sel]
        inst.gt_scores_online, inst.gt_scores_offline = sc_on[sel], sc_off[sel]
        inst.gt_probs_online, inst.gt_probs_offline = take("probs", s_on)[sel], take("probs", s_off)[sel]
        return inst

    cls_list = {"on": on.classes.tolist(), "off": off.classes.tolist()}
    cls_of = lambda src: cls_list[src[0]][src[1]]
    if tag == "RCNN":
        a = build([r for r in common if cls_of(r[0]) == cls_of(r[1])], False)
        b = build([r for r in common if cls_of(r[0]) != cls_of(r[1])], True)
        if len(b) and len(a):  # a B box that coincides with an A box is dropped (trainer.py:431-436)
            coincide = (b.gt_boxes.tensor.unsqueeze(1) == a.gt_boxes.tensor).sum(-1) == 4
            b = b[coincide.sum(1) == 0]
    elif tag == "RPN":
        a, b = build(common, False), None
    else:
        raise ValueError(tag)
    if device is not None:
        a, c = a.to(device), c.to(device)
        b = b.to(device) if b is not None else None
    return a, b, c


def _resolve_overlapping_online(on: _Side, off: _Side, common: List[Tuple]) -> List[Tuple]:
    """online_boxes_merging (util.py:485-507): the cloud detector may report one region under several labels
    (self-IoU >= 0.95).  Of the matched pairs of such a group keep those agreeing with the offline vote; if the offline boxes
    disagree among themselves keep the pairs whose labels differ (they become B boxes).  Kept pairs move to the end."""
    for grp in _overlap_groups(on.boxes, 0.95):
        assert len({int(on.classes[i]) for i in grp}) != 1
        on_box = lambda r: (on if r[0][0] == "on" else off).boxes[r[0][1]]
        on_cls = lambda r: int((on if r[0][0] == "on" else off).classes[r[0][1]])
        off_cls = lambda r: int((on if r[1][0] == "on" else off).classes[r[1][1]])
        hit = [k for k, r in enumerate(common) if any(bool((on_box(r) == on.boxes[g]).all()) for g in grp)]
        rest = [k for k in range(len(common)) if k not in set(hit)]
        first = [k for k, r in enumerate(common) if bool((on_box(r) == on.boxes[grp[0]]).all())]
        votes = {off_cls(common[k]) for k in first}
        if len(votes) == 1:
            agree = [k for k in hit if on_cls(common[k]) == next(iter(votes))]
            hit = agree if agree else hit
        else:
            hit = [k for k in hit if on_cls(common[k]) != off_cls(common[k])]
        common = [common[k] for k in rest] + [common[k] for k in hit]
    return common
