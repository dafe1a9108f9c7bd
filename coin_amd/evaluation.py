"""Pascal-VOC style detection AP as COIN's evaluator computes it.  Host code only.

Behaviour of coin/evaluation/cloud_pascal_voc_evaluation.py: ``Cloud_PascalVOCDetectionEvaluator.process/evaluate`` (:54-130),
``parse_rec`` (:147-173), ``voc_ap`` (:176-204), ``voc_eval`` (:207-319):

  * detections go through the text round trip of the reference (score ``%.3f``, coordinates ``%.1f``; no +1 shift), which fixes
    both their values and -- through ties -- their ranking;
  * per class, detections are ranked by ``numpy.argsort(-score)``; a detection is a true positive if its best-overlapping
    ground-truth box of that class (inclusive-pixel IoU, +1 on widths/heights) exceeds the threshold strictly, is not "difficult"
    and has not been claimed yet; matches to difficult boxes are ignored, everything else is a false positive;
  * AP per class with the VOC07 11-point rule (``year == 2007``) or the area under the monotone precision envelope, for IoU
    thresholds 0.50 ... 0.95; result ``{"bbox": {"AP", "AP50", "AP75", "AP50-<class>" ...}}`` in percent.

Here the ground truth is parsed once per image, IoUs are computed per image for all of its detections of a class at once, and the
ten thresholds share that work; only the claim bookkeeping walks the ranking.  Pinned against values produced by the reference's own
evaluator on a synthetic VOC tree (tests/golden/voc_eval.npz, gen_golden.py:case_voc_eval).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from collections import OrderedDict, defaultdict
from typing import Dict, List, Sequence

import numpy as np


def parse_voc_xml(path: str) -> List[Dict]:
    """Objects of one annotation file: name, difficult flag (0 when the tag is absent or malformed), integer box."""
    objs = []
    for obj in ET.parse(path).findall("object"):
        def as_int(tag, default=0):
            node = obj.find(tag)
            try:
                return int(node.text)
            except (AttributeError, TypeError, ValueError):
                return default

        bb = obj.find("bndbox")
        objs.append({"name": obj.find("name").text, "difficult": as_int("difficult"),
                     "bbox": [int(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]})
    return objs


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            sel = rec >= t
            ap += (float(prec[sel].max()) if sel.any() else 0.0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]          # monotone precision envelope
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def _iou_inclusive(bb: np.ndarray, gt: np.ndarray) -> np.ndarray:
    iw = np.maximum(np.minimum(gt[:, 2], bb[2]) - np.maximum(gt[:, 0], bb[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gt[:, 3], bb[3]) - np.maximum(gt[:, 1], bb[1]) + 1.0, 0.0)
    inter = iw * ih
    union = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gt[:, 2] - gt[:, 0] + 1.0) * (gt[:, 3] - gt[:, 1] + 1.0) - inter
    return inter / union


class PascalVOCEvaluator:
    """``reset() / process(inputs, outputs) / evaluate()`` like the reference's DatasetEvaluator.

    dirname: VOC tree (``Annotations/<id>.xml``, ``ImageSets/Main/<split>.txt``); class_names: ``thing_classes`` of the dataset."""

    THRESHOLDS = tuple(range(50, 100, 5))

    def __init__(self, dirname: str, split: str, class_names: Sequence[str], year: int = 2007):
        assert year in (2007, 2012), year
        self._anno = os.path.join(dirname, "Annotations", "{}.xml")
        with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
            self._image_ids = [line.strip() for line in f.readlines()]
        self._class_names, self._is_2007 = list(class_names), year == 2007
        self._gt_cache: Dict[str, List[Dict]] = {}
        self.reset()

    def reset(self):
        self._lines = defaultdict(list)  # class id -> "image score x0 y0 x1 y1" rows, as the reference stores them

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            inst = out["instances"].to("cpu")
            boxes = inst.pred_boxes.tensor.numpy()
            for box, score, cls in zip(boxes, inst.scores.tolist(), inst.pred_classes.tolist()):
                x0, y0, x1, y1 = box
                self._lines[cls].append(f"{inp['image_id']} {score:.3f} {x0:.1f} {y0:.1f} {x1:.1f} {y1:.1f}")

    # ------------------------------------------------------------------
    def _ground_truth(self, image_id: str) -> List[Dict]:
        if image_id not in self._gt_cache:
            self._gt_cache[image_id] = parse_voc_xml(self._anno.format(image_id))
        return self._gt_cache[image_id]

    def _class_ap(self, cls_id: int) -> Dict[int, float]:
        name = self._class_names[cls_id]
        gt, npos = {}, 0
        for image_id in self._image_ids:
            objs = [o for o in self._ground_truth(image_id) if o["name"] == name]
            boxes = np.array([o["bbox"] for o in objs], dtype=float).reshape(-1, 4)
            difficult = np.array([o["difficult"] for o in objs]).astype(bool)
            npos += int((~difficult).sum())
            gt[image_id] = (boxes, difficult)
        rows = [ln.strip().split(" ") for ln in self._lines.get(cls_id, [])]
        ids = [r[0] for r in rows]
        conf = np.array([float(r[1]) for r in rows])
        bb = np.array([[float(v) for v in r[2:]] for r in rows]).reshape(-1, 4)
        order = np.argsort(-conf)
        # best overlap and its ground-truth index per detection: shared by the ten thresholds
        best_ov = np.full(len(rows), -np.inf)
        best_j = np.zeros(len(rows), dtype=int)
        for d in order:
            boxes, _ = gt[ids[d]]
            if boxes.size:
                ov = _iou_inclusive(bb[d], boxes)
                best_j[d] = int(np.argmax(ov))
                best_ov[d] = ov[best_j[d]]
        out = {}
        for thr in self.THRESHOLDS:
            claimed = {k: np.zeros(len(v[0]), dtype=bool) for k, v in gt.items()}
            tp, fp = np.zeros(len(rows)), np.zeros(len(rows))
            for rank, d in enumerate(order):
                if best_ov[d] > thr / 100.0:
                    _, difficult = gt[ids[d]]
                    j = best_j[d]
                    if difficult[j]:
                        continue                      # neither a hit nor a miss
                    if claimed[ids[d]][j]:
                        fp[rank] = 1.0
                    else:
                        tp[rank] = 1.0
                        claimed[ids[d]][j] = True
                else:
                    fp[rank] = 1.0
            tp, fp = np.cumsum(tp), np.cumsum(fp)
            with np.errstate(divide="ignore", invalid="ignore"):
                rec = tp / float(npos)
            prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
            out[thr] = voc_ap(rec, prec, self._is_2007) * 100
        return out

    def _gather(self):
        """The test loader shards the images over the ranks (InferenceSampler): every rank evaluates on the union of the
        predictions (the reference gathers them on rank 0, pascal_voc_evaluation.py `comm.gather`)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, dict(self._lines))
        merged = defaultdict(list)
        for part in parts:            # rank order: the same rows in the same order on every rank
            for cls, rows in part.items():
                merged[cls].extend(rows)
        self._lines = merged

    def evaluate(self) -> "OrderedDict[str, Dict[str, float]]":
        self._gather()
        per_class = [self._class_ap(c) for c in range(len(self._class_names))]
        mean = {thr: float(np.mean([pc[thr] for pc in per_class])) for thr in self.THRESHOLDS}
        res = OrderedDict()
        res["bbox"] = {"AP": float(np.mean(list(mean.values()))), "AP50": mean[50], "AP75": mean[75]}
        for name, pc in zip(self._class_names, per_class):
            res["bbox"]["AP50-" + name] = pc[50]
        return res
