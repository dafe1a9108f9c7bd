"""CPU restatement of COIN's target-detector training step (``CoinTrainer``) -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/coin/engine/trainer.py (run_step :160-218, match_dual_teacher :338-461, match_boxes :463-478,
merge_boxes :480-485), coin/engine/base.py (process :80-126, preprocess_results :128-135) and coin/utils/util.py
(delete_duplicate_boxes :434-457, find_same :459-464, filter_result :466-483, online_boxes_merging :485-507).
Pinned by tests/test_oracle_golden.py against tests/golden/match_dual_teacher.npz, captured by running the reference's own
``CoinTrainer.match_dual_teacher`` (tests/golden/gen_golden.py:case_match_dual_teacher).

Notation of the reference: A = consistent boxes (both teachers agree on the label), B = inconsistent (same place, different
label), C = private (seen by one teacher only).  "online" = cloud detector (cached results), "offline" = CLIP-detector teacher.
Tie-breaks draw from Python's ``random`` module in the same order as the reference, so a seeded run reproduces it.
"""
from __future__ import annotations

import copy
import random
from typing import Dict, List, Optional, Tuple

import torch

from . import d2
from .coin import weighted_box_fusion_split


class MyInstances(d2.Instances):
    """coin/utils/util.py:188-214: Instances whose ``set`` can skip the length check."""

    def set(self, name, value, check_len=True):
        if check_len and len(self._fields):
            assert len(self) == len(value), "Adding a field of length {} to a Instances of length {}".format(len(value), len(self))
        self._fields[name] = value

    def to(self, *args, **kwargs):
        ret = MyInstances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v, check_len=False)
        return ret


# ------------------------------------------------------------------------------------------ base.py:80-135
def process(instances: d2.Instances, old_size, new_size, random_flip: str, thresh=None, keep_name=False) -> d2.Instances:
    """Rescale cached / predicted boxes from the stored image size to the network input size, apply the view's flip, rename
    pred_* -> gt_* (base.py:80-126)."""
    img_h, img_w = old_size
    net_h, net_w = new_size
    new = MyInstances((net_h, net_w))
    new._fields = copy.deepcopy(instances.get_fields())
    boxes = new.pred_boxes if new.has("pred_boxes") else new.gt_boxes
    boxes.scale(net_w / img_w, net_h / img_h)
    if random_flip == "horizontal":
        t = boxes.tensor.clone()
        t[:, 0] = net_w - boxes.tensor[:, 2]
        t[:, 2] = net_w - boxes.tensor[:, 0]
        boxes = d2.Boxes(t)
    elif random_flip == "vertical":
        t = boxes.tensor.clone()
        t[:, 1] = net_h - boxes.tensor[:, 3]
        t[:, 3] = net_h - boxes.tensor[:, 1]
        boxes = d2.Boxes(t)
    elif random_flip != "no":
        raise NotImplementedError
    if new.has("pred_boxes"):
        if keep_name:
            new.set("pred_boxes", boxes)
        else:
            new.remove("pred_boxes")
            new.set("gt_boxes", boxes)
    else:
        new.set("gt_boxes", boxes)
    if not keep_name:
        new.set("gt_classes", new.get("pred_classes"))
        new.remove("pred_classes")
    if thresh is not None:
        return new[instances.scores >= thresh]  # the ORIGINAL scores select (base.py:119)
    return new


def preprocess_results(results: Dict, new_image_size, random_flip, thresh=None) -> Dict:
    size = (results["height"], results["width"])
    results["RCNN"] = process(results["RCNN"]["instances"], size, new_image_size, random_flip, thresh)
    key = "RPN_AUG" if "RPN_AUG" in results else "RPN"
    results["RPN"] = process(results[key]["instances"], size, new_image_size, random_flip, thresh)
    results.pop("RPN_AUG", None)
    return results


# ------------------------------------------------------------------------------------------ util.py:434-507
def delete_duplicate_boxes(instances: d2.Instances, return_split: bool = False):
    """Groups of boxes with identical coordinates (candidates: equal coordinate SUM, confirmed by a zero difference sum)
    are either returned separately (``return_split``) or reduced to one random member."""
    sums = instances.gt_boxes.tensor.sum(1)
    uniq = torch.unique(sums)
    member = torch.eq(uniq.unsqueeze(1), sums)          # [n_unique, n]
    groups = member[member.sum(1) != 1]                  # rows with more than one member
    outs = []
    for i in range(groups.size(0)):
        m = groups[i]
        g = instances[m]
        if (g.gt_boxes.tensor - g.gt_boxes.tensor[0]).sum() == 0:
            outs.append(g if return_split else g[random.randint(0, len(g) - 1)])
        else:  # same sum, different boxes: not a duplicate group after all
            groups[i][groups[i].nonzero()[:, 0]] = False
    keep = (groups.sum(0) == 0).nonzero()[:, 0]
    if return_split:
        return instances[keep], outs
    return d2.Instances.cat([instances[keep]] + outs)


def _find_same(sets: List[set], ups: List[int], i: int) -> set:
    for j in sets[i]:
        if j != i and j not in ups:
            if sets[j] - sets[i] == set():
                pass
            else:
                sets[i] = sets[i] | _find_same(sets, ups + [i], j)
    return sets[i]


def filter_result(result: d2.Instances, thresh: float) -> List[d2.Instances]:
    """Connected groups (size > 1) of boxes whose mutual IoU >= thresh (util.py:466-483)."""
    boxes = result.gt_boxes
    iou = d2.pairwise_iou(boxes, boxes) >= thresh
    sets = [set(iou[i].nonzero()[:, 0].tolist()) for i in range(len(boxes))]
    for i in range(len(sets)):
        for j in sets[i]:
            if j != i:
                sets[i] = sets[i] | _find_same(sets, [i], j)
        for j in sets[i]:
            if j != i:
                sets[j] = set()
    sets = [s for s in sets if len(s) != 0]
    return [result[list(s)] for s in sets if len(s) != 1]


def online_boxes_merging(instances, common_offline, common_online):
    """The cloud detector can emit the same region under several labels (self-IoU >= 0.95); keep, among the matched pairs of
    such a group, those that agree with the offline vote -- or, if the offline boxes disagree among themselves, those whose
    labels differ (they become B boxes) (util.py:485-507)."""
    for group in filter_result(instances, 0.95):
        assert group.gt_classes.unique().size(0) != 1
        same = torch.eq(group.gt_boxes.tensor.unsqueeze(1), common_online.gt_boxes.tensor).sum(-1) == 4
        idx = torch.unique(same.nonzero()[:, 1])
        flag = torch.ones(len(common_online))
        flag[idx] = 0
        other = flag.nonzero()[:, 0]
        s = same[0].nonzero()[:, 0]
        if common_offline.gt_classes[s].unique().size(0) == 1:
            mask = common_online[idx].gt_classes == common_offline.gt_classes[s].unique()
            if mask.sum() != 0:
                idx = idx[mask]
        else:
            idx = idx[common_online[idx].gt_classes != common_offline.gt_classes[idx]]
        common_online = d2.Instances.cat([common_online[other], common_online[idx]])
        common_offline = d2.Instances.cat([common_offline[other], common_offline[idx]])
    return common_offline, common_online


# ------------------------------------------------------------------------------------------ trainer.py:338-485
def merge_boxes(online_box, offline_box, online_scores, offline_scores, weight_for_box_a: float):
    """trainer.py:480-485: the cloud box before BURN_UP_STEP (weight 1.0), score-weighted fusion afterwards."""
    if weight_for_box_a != 1.0:
        return weighted_box_fusion_split(online_box, offline_box, online_scores, offline_scores)
    return online_box


def _complement(n: int, used) -> torch.Tensor:
    # list(set(range(n)) - set(used)): CPython iterates small-int sets in ascending order
    return torch.LongTensor(sorted(set(range(n)) - set(used)))


def match_dual_teacher(online_result: Dict[str, d2.Instances], offline_result: d2.Instances, tag: str, iou_threshold: float = 0.5,
                       weight_for_box_a: float = 1.0) -> Tuple[d2.Instances, Optional[d2.Instances], d2.Instances]:
    """-> (A, B, C) for tag 'RCNN', (A, None, C) for 'RPN' (trainer.py:338-461)."""
    on = online_result[tag]
    off = offline_result
    if len(on) == 0 and len(off) == 0:
        com_on, com_off, off_only, on_only = on, off, off, on
    elif len(on) == 0:      # nothing from the cloud detector: confident offline boxes count as consistent
        fg = off.scores > 0.8
        com_on, com_off, off_only, on_only = off[fg], off[fg], off[~fg], on
    elif len(off) == 0:     # nothing from the offline teacher: every cloud box is consistent
        com_on, com_off, off_only, on_only = on, on, off, off
    else:
        uniq, dup_groups = delete_duplicate_boxes(off, return_split=True)
        pairs = (d2.pairwise_iou(on.gt_boxes, uniq.gt_boxes) >= iou_threshold).nonzero()
        com_on_l, com_off_l = [on[pairs[:, 0]]], [uniq[pairs[:, 1]]]
        off_only = [uniq[_complement(len(uniq), pairs[:, 1].tolist())]]
        used_on = pairs[:, 0].tolist()
        for grp in dup_groups:  # one physical box reported under several labels by the offline teacher
            hits = (d2.pairwise_iou(on.gt_boxes, grp.gt_boxes) >= iou_threshold).nonzero()
            if hits.size(0) != 0:
                first = hits[0, 0].item()
                same_label = grp.gt_classes == on.gt_classes[hits[0, 0]]
                com_on_l.append(on[first])
                used_on.append(first)
                if same_label.sum() >= 1:
                    com_off_l.append(grp[same_label])
                else:
                    com_off_l.append(grp[random.randint(0, len(grp) - 1)])
            else:
                off_only.append(grp[random.randint(0, len(grp) - 1)])
        com_off, com_on = d2.Instances.cat(com_off_l), d2.Instances.cat(com_on_l)
        com_off, com_on = online_boxes_merging(on, com_off, com_on)
        on_only = on[_complement(len(on), used_on)]
    off_only = off_only if isinstance(off_only, list) else [off_only]

    c = d2.Instances.cat(off_only + [on_only])
    c.gt_scores, c.gt_probs = c.scores, c.probs
    c.remove("scores")
    c.remove("probs")

    def fuse(sel_on, sel_off, inst):
        inst.gt_scores_online, inst.gt_scores_offline = sel_on.scores, sel_off.scores
        inst.remove("scores")
        inst.gt_probs_online, inst.gt_probs_offline = sel_on.probs, sel_off.probs
        inst.remove("probs")
        inst.gt_boxes.tensor = merge_boxes(sel_on.gt_boxes.tensor, sel_off.gt_boxes.tensor, inst.gt_scores_online, inst.gt_scores_offline,
                                           weight_for_box_a)
        return inst

    if tag == "RCNN":
        same = com_off.gt_classes == com_on.gt_classes
        a = delete_duplicate_boxes(fuse(com_on[same], com_off[same], com_off[same]))
        b = com_off[~same]
        b.gt_classes_offline = b.gt_classes
        b.gt_classes_online = com_on[~same].gt_classes
        b.remove("gt_classes")
        b = delete_duplicate_boxes(fuse(com_on[~same], com_off[~same], b))
        # a B box that coincides with an A box is dropped (trainer.py:431-436)
        coincide = torch.eq(b.gt_boxes.tensor.unsqueeze(1), a.gt_boxes.tensor).sum(-1) == 4
        b = b[coincide.sum(1) == 0]
    elif tag == "RPN":
        a = delete_duplicate_boxes(fuse(com_on, com_off, copy.deepcopy(com_off)))
        b = None
    else:
        raise ValueError(tag)
    return a, b, c


def match_boxes(batched_input: List[Dict], offline_results: List[Dict], cloud_results, iou_threshold=0.5, weight_for_box_a=1.0):
    """trainer.py:463-478: per image, bring the teacher's detections (output-image coordinates) and the cached cloud detections
    (stored-image coordinates) to the weak view's network coordinates, then split into (A, B, C) for the RoI head and the RPN."""
    rcnn, rpn = [], []
    for data, off in zip(batched_input, offline_results):
        online = cloud_results(data["file_name"])
        net = tuple(data["image"].shape[1:])
        off_i = process(off["instances"].to("cpu"), (data["height"], data["width"]), net, "no")
        assert online["height"] == data["height"] and online["width"] == data["width"] and online["image_id"] == data["image_id"]
        online = preprocess_results(online, net, data["random_flip"], thresh=None)
        rcnn.append(match_dual_teacher(online, off_i, "RCNN", iou_threshold, weight_for_box_a))
        rpn.append(match_dual_teacher(online, off_i, "RPN", iou_threshold, weight_for_box_a))
    return rcnn, rpn


# ------------------------------------------------------------------------------------------ trainer.py:160-218
STUDENT_SKIP_STEP_ONE = ("loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base", "loss_cls_b")
STUDENT_SKIP_STEP_TWO = ("loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base")


def student_loss(record: Dict[str, torch.Tensor], burned_up: bool) -> torch.Tensor:
    """Sum of the loss terms that train the student (trainer.py:199-205): the CKG terms never do, `loss_cls_b` only once the
    burn-up phase is over."""
    skip = STUDENT_SKIP_STEP_TWO if burned_up else STUDENT_SKIP_STEP_ONE
    return sum(v for k, v in record.items() if k not in skip)
