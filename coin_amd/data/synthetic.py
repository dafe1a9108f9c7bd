"""Synthetic VOC-shaped inputs and cached teacher results for the hot path (SURVEY.md §8d recipe).

The reference's input side (coin/data/**, the GroundingDINO / CLIP collectors) is out of scope; the hot path
consumes (a) two augmented views per image as ``{"image": uint8 [3,H,W], "height", "width", "file_name",
"image_id", "random_flip"}`` dicts (coin/data/dataset_mapper.py:363-450) and (b) one cached teacher result
per file name in the collectors' format (coin/modeling/meta_arch/gdino_collector.py:51-75, Appendix A.1):
``{"file_name","image_id","height","width","RCNN": {"instances"}, "RPN": {"instances"}}`` with
``pred_boxes`` (original-image pixels), ``scores``, ``pred_classes``, ``probs [G, K+1]``.
Everything is generated from a seeded generator and kept resident on the device (the metric times the
hot path with inputs already in HBM).
"""
from __future__ import annotations

import copy
from typing import Dict, List, Tuple

import torch

from ..structures import Boxes, Instances


def _boxes(n, h, w, g, lo=32.0, hi=400.0):
    bw = torch.rand(n, generator=g) * (hi - lo) + lo
    bh = torch.rand(n, generator=g) * (hi - lo) + lo
    bw, bh = bw.clamp(max=w - 2.0), bh.clamp(max=h - 2.0)
    x0 = torch.rand(n, generator=g) * (w - bw)
    y0 = torch.rand(n, generator=g) * (h - bh)
    return torch.stack([x0, y0, x0 + bw, y0 + bh], dim=1)


def synthetic_teacher_result(file_name: str, image_id: str, h: int, w: int, num_boxes: int, num_classes: int, g: torch.Generator,
                             device="cpu") -> Dict:
    boxes = _boxes(num_boxes, h, w, g, hi=min(400.0, min(h, w) * 0.6))
    probs = torch.softmax(3.0 * torch.randn(num_boxes, num_classes + 1, generator=g), dim=1)
    probs[:, -1] = probs.min(dim=1).values * 0.5          # background column forced smallest
    probs = probs / probs.sum(dim=1, keepdim=True)

    def inst():
        i = Instances((h, w))
        i.pred_boxes = Boxes(boxes.clone().to(device))
        i.scores = probs[:, :-1].max(dim=1).values.to(device)
        i.pred_classes = probs[:, :-1].argmax(dim=1).to(device)
        i.probs = probs.clone().to(device)
        return i

    return {"file_name": file_name, "image_id": image_id, "height": h, "width": w, "RCNN": {"instances": inst()}, "RPN": {"instances": inst()}}


class SyntheticTeacherCache:
    """Stands in for CLIP_COLLECTOR.__call__ (coin/modeling/meta_arch/clip_collector.py:69-74): file name -> deep copy.
    The store has the reference's layout ``{dataset_name: {file_name: result}}`` (gdino_collector.py:51-75), which is what the
    trainers write into their checkpoints ("results" / "online_results") and ``CloudResults`` reads back."""

    def __init__(self, dataset_name: str = "synthetic_voc_train"):
        self.dataset_name = dataset_name
        self._results: Dict[str, Dict[str, Dict]] = {dataset_name: {}}

    def add(self, result: Dict):
        self._results.setdefault(self.dataset_name, {})[result["file_name"]] = result

    def entry(self, file_name: str) -> Dict:
        """The stored record itself (no copy)."""
        for per_dataset in self._results.values():
            if file_name in per_dataset:
                return per_dataset[file_name]
        raise KeyError(f"no cached teacher result for {file_name!r}")

    def __call__(self, file_name: str) -> Dict:
        r = self.entry(file_name)
        out = {k: v for k, v in r.items() if k not in ("RCNN", "RPN")}
        for tag in ("RCNN", "RPN"):
            src = r[tag]["instances"]
            dst = Instances(src.image_size)
            for name, v in src.get_fields().items():
                dst.set(name, Boxes(v.tensor.clone()) if isinstance(v, Boxes) else v.clone())
            out[tag] = {"instances": dst}
        return out

    def get_results(self):
        return self._results

    def set_results(self, results):
        self._results = results


class SyntheticTwoViewLoader:
    """Infinite iterator of (strong_views, weak_views): lists of `images_per_batch` dicts each."""

    def __init__(self, images_per_batch: int, height: int, width: int, num_classes: int, boxes_per_image: int, seed: int, device,
                 num_images: int = None, flip: bool = True):
        g = torch.Generator().manual_seed(seed)
        self.n = num_images or images_per_batch
        self.per_batch = images_per_batch
        self.cache = SyntheticTeacherCache()
        self.items: List[Tuple[Dict, Dict]] = []
        for i in range(self.n):
            name = f"synthetic/JPEGImages/{seed}_{i:06d}.png"
            weak = torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8)
            noise = torch.randint(-24, 25, (3, height, width), generator=g, dtype=torch.int16)
            strong = (weak.to(torch.int16) + noise).clamp(0, 255).to(torch.uint8)       # colour-jitter stand-in
            flipped = flip and bool(torch.rand(1, generator=g) < 0.5)
            if flipped:
                weak, strong = weak.flip(-1), strong.flip(-1)
            base = {"file_name": name, "image_id": f"{seed}_{i:06d}", "height": height, "width": width,
                    "random_flip": "horizontal" if flipped else "no"}
            self.items.append(({**base, "image": strong.contiguous().to(device)}, {**base, "image": weak.contiguous().to(device)}))
            self.cache.add(synthetic_teacher_result(name, base["image_id"], height, width, boxes_per_image, num_classes, g, device))
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        s, w = [], []
        for _ in range(self.per_batch):
            a, b = self.items[self._pos % self.n]
            self._pos += 1
            s.append(dict(a))
            w.append(dict(b))
        return s, w


def synthetic_offline_detections(cloud_result: Dict, g: torch.Generator, jitter_px: float = 4.0, relabel_frac: float = 0.25,
                                 extra: int = 8, device="cpu") -> Dict:
    """A CLIP-detector-like detection set for one image: the cloud boxes jittered by N(0, jitter_px), a quarter
    of the labels permuted (-> inconsistent 'B' boxes) plus `extra` private boxes (-> 'C').  Same schema as
    OpenVocabularyRCNN.inference output: {"instances": Instances(pred_boxes, scores, pred_classes, probs)} in image pixels."""
    src = cloud_result["RCNN"]["instances"]
    h, w = cloud_result["height"], cloud_result["width"]
    boxes = src.pred_boxes.tensor.detach().cpu() + jitter_px * torch.randn(len(src), 4, generator=g)
    probs = src.probs.detach().cpu().clone()
    k = probs.shape[1] - 1
    n = len(src)
    flip = torch.rand(n, generator=g) < relabel_frac
    for i in flip.nonzero()[:, 0].tolist():
        perm = torch.randperm(k, generator=g)
        probs[i, :k] = probs[i, :k][perm]
    if extra:
        boxes = torch.cat([boxes, _boxes(extra, h, w, g, hi=min(400.0, min(h, w) * 0.6))])
        p = torch.softmax(3.0 * torch.randn(extra, k + 1, generator=g), dim=1)
        p[:, -1] = p.min(dim=1).values * 0.5
        probs = torch.cat([probs, p / p.sum(dim=1, keepdim=True)])
    boxes[:, 0::2] = boxes[:, 0::2].clamp(0, w)
    boxes[:, 1::2] = boxes[:, 1::2].clamp(0, h)
    inst = Instances((h, w))
    inst.pred_boxes = Boxes(boxes.to(device))
    inst.scores = probs[:, :-1].max(dim=1).values.to(device)
    inst.pred_classes = probs[:, :-1].argmax(dim=1).to(device)
    inst.probs = probs.to(device)
    return {"instances": inst}
