"""Two-view (strong / weak) input mapper of the unsupervised target-domain loader, with the image work on the GPU
(SURVEY.md §8(f)-3).

Behaviour of ``DatasetMapperUnsupervised`` (coin/data/dataset_mapper.py:313-450):

  1. read the image (``utils.read_image``, ``INPUT.FORMAT``), check its size against the dataset dict;
  2. weak view: detectron2 ``ResizeShortestEdge(MIN_SIZE_TRAIN, MAX_SIZE_TRAIN, MIN_SIZE_TRAIN_SAMPLING)`` + ``RandomFlip``
     (``utils.build_augmentation``); ``dataset_dict["random_flip"]`` records the flip (the trainers mirror the cached teacher
     boxes with it, coin/engine/base.py:95-112); annotations, when present, follow the same transforms;
  3. strong view = the weak image through ``build_strong_augmentation`` (coin/data/detection_utils.py:22-45): ColorJitter(0.4, 0.4,
     0.4, 0.1) with p 0.8, RandomGrayscale p 0.2, GaussianBlur([0.1, 2.0]) p 0.5, Solarize(0.5) p 0.2;
  4. -> ``(strong_dict, weak_dict)`` with ``"image"`` = uint8 [3, h, w] tensors of equal size.

The reference does steps 2-3 with Pillow on two loader workers.  Here the decoded image is uploaded once and every pixel
operation is a HIP kernel that reproduces Pillow's arithmetic bit for bit (``coin_aug_*``, csrc/augment.hip); the random decisions
are drawn on the host from the SAME three generators in the SAME order as the reference's libraries (numpy for detectron2's
augmentations, torch for torchvision's transforms, ``random`` for the blur radius), so a seeded run takes the same decisions.
The images stay on the device: ``OpenVocabularyRCNN.preprocess_image`` consumes them without a copy.
"""
from __future__ import annotations

import copy
import random
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import kernels as K
from ..structures import Boxes, Instances

_JITTER = (("brightness", 0.6, 1.4), ("contrast", 0.6, 1.4), ("saturation", 0.6, 1.4), ("hue", -0.1, 0.1))
_POINT_OPS = {"brightness": K.AUG_BRIGHTNESS, "contrast": K.AUG_CONTRAST, "saturation": K.AUG_SATURATION, "hue": K.AUG_HUE,
              "grayscale": K.AUG_GRAYSCALE, "solarize": K.AUG_SOLARIZE}


def read_image(file_name: str, fmt: str = "RGB") -> np.ndarray:
    """detectron2 ``utils.read_image``: PIL open (EXIF orientation applied), convert, HWC uint8; "BGR" flips the channels."""
    from PIL import Image, ImageOps

    with open(file_name, "rb") as f:
        img = ImageOps.exif_transpose(Image.open(f))
        img = img.convert("RGB" if fmt in ("RGB", "BGR") else fmt)
        a = np.asarray(img)
    return np.ascontiguousarray(a[:, :, ::-1]) if fmt == "BGR" else a


def shortest_edge_size(h: int, w: int, size: int, max_size: int) -> Tuple[int, int]:
    """detectron2 ``ResizeShortestEdge``: scale the short edge to `size`, cap the long edge at `max_size`, round half up."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class DatasetMapperUnsupervised:
    def __init__(self, cfg, is_train: bool = True, device=None, np_rng=None, torch_generator: Optional[torch.Generator] = None,
                 py_rng=None):
        self.is_train = is_train
        self.img_format = cfg.INPUT.FORMAT
        if is_train:
            self.min_size, self.max_size = tuple(cfg.INPUT.MIN_SIZE_TRAIN), cfg.INPUT.MAX_SIZE_TRAIN
            self.sample_style = cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING
            self.flip = cfg.INPUT.RANDOM_FLIP
        else:
            self.min_size, self.max_size, self.sample_style, self.flip = (cfg.INPUT.MIN_SIZE_TEST,), cfg.INPUT.MAX_SIZE_TEST, "choice", "none"
        if self.sample_style == "range":
            assert len(self.min_size) == 2, "short_edge_length must be two values using 'range' sample style"
        assert self.flip in ("horizontal", "none"), "COIN's configs flip horizontally or not at all"
        assert not cfg.INPUT.CROP.ENABLED, "INPUT.CROP is off in every COIN config"
        self.device = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        # None -> the libraries' global generators, as in the reference
        self.np_rng, self.torch_generator, self.py_rng = np_rng if np_rng is not None else np.random, torch_generator, py_rng or random

    # ------------------------------------------------------------------ random decisions, in the reference's order of draws
    def draw_params(self, h: int, w: int) -> Dict:
        size = (int(self.np_rng.randint(self.min_size[0], self.min_size[1] + 1)) if self.sample_style == "range"
                else int(self.np_rng.choice(self.min_size)))
        out = {"size": shortest_edge_size(h, w, size, self.max_size), "strong_ops": []}
        out["flip"] = bool(self.np_rng.uniform() < 0.5) if self.flip == "horizontal" else False
        if not self.is_train:
            return out
        g = self.torch_generator
        u = lambda: float(torch.rand(1, generator=g))
        ops: List[Tuple[str, float]] = []
        if not (0.8 < u()):                       # RandomApply([ColorJitter]).forward: `if self.p < torch.rand(1): return img`
            order = torch.randperm(4, generator=g).tolist()                     # ColorJitter.get_params
            fac = [float(torch.empty(1).uniform_(lo, hi, generator=g)) for _, lo, hi in _JITTER]
            ops += [(_JITTER[i][0], fac[i]) for i in order]
        if u() < 0.2:                             # RandomGrayscale.forward
            ops.append(("grayscale", 0.0))
        if not (0.5 < u()):                       # RandomApply([GaussianBlur]); the radius comes from python's `random`
            ops.append(("blur", self.py_rng.uniform(0.1, 2.0)))
        if not (0.2 < u()):                       # RandomApply([Solarize(0.5)]): threshold round(0.5 * 256)
            ops.append(("solarize", 128.0))
        out["strong_ops"] = ops
        return out

    # ------------------------------------------------------------------ pixels (device)
    def weak_view(self, img_hwc: torch.Tensor, params: Dict) -> torch.Tensor:
        oh, ow = params["size"]
        return K.aug_resize_bilinear(img_hwc, oh, ow, params["flip"])

    def strong_view(self, weak_hwc: torch.Tensor, ops) -> torch.Tensor:
        x = weak_hwc
        for name, p in ops:
            if name == "blur":
                x = K.aug_gaussian_blur(x, p)
            elif name == "hue":
                x = K.aug_point_op(x, K.AUG_HUE, iparam=int(p * 255) % 256)    # F_pil.adjust_hue: np_h += np.uint8(hue_factor * 255)
            elif name == "solarize":
                x = K.aug_point_op(x, K.AUG_SOLARIZE, iparam=int(p))
            else:
                x = K.aug_point_op(x, _POINT_OPS[name], fparam=p)
        return x

    @staticmethod
    def to_chw(img_hwc: torch.Tensor) -> torch.Tensor:
        return K.aug_point_op(img_hwc, K.AUG_COPY, out_chw=True)

    # ------------------------------------------------------------------ annotations (host; the target domain usually has none)
    @staticmethod
    def transform_annotations(annos: List[Dict], h: int, w: int, params: Dict) -> Instances:
        """detectron2 transform_instance_annotations + annotations_to_instances + filter_empty_instances for XYXY_ABS boxes."""
        oh, ow = params["size"]
        keep = [a for a in annos if a.get("iscrowd", 0) == 0]
        boxes = np.array([a["bbox"] for a in keep], dtype=np.float64).reshape(-1, 4)
        boxes[:, 0::2] *= ow * 1.0 / w
        boxes[:, 1::2] *= oh * 1.0 / h
        if params["flip"]:
            x0, x1 = ow - boxes[:, 2], ow - boxes[:, 0]
            boxes[:, 0], boxes[:, 2] = x0, x1
        boxes = np.minimum(boxes.clip(min=0), np.array([ow, oh, ow, oh], dtype=np.float64))
        inst = Instances((oh, ow))
        inst.gt_boxes = Boxes(torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4))
        inst.gt_classes = torch.tensor([int(a["category_id"]) for a in keep], dtype=torch.int64)
        t = inst.gt_boxes.tensor
        nonempty = ((t[:, 2] - t[:, 0]) > 1e-5) & ((t[:, 3] - t[:, 1]) > 1e-5)
        return inst[nonempty]

    # ------------------------------------------------------------------ the mapper
    def __call__(self, dataset_dict: Dict, image: Optional[np.ndarray] = None):
        d = copy.deepcopy(dataset_dict)
        img = image if image is not None else read_image(d["file_name"], self.img_format)
        h, w = img.shape[:2]
        if "width" in d or "height" in d:  # check_image_size
            if (d.get("width", w), d.get("height", h)) != (w, h):
                raise ValueError(f"Mismatched image shape for {d.get('file_name')}: got {(w, h)}, expect {(d.get('width'), d.get('height'))}")
        d.setdefault("width", w)
        d.setdefault("height", h)
        params = self.draw_params(h, w)
        d["random_flip"] = "horizontal" if params["flip"] else "no"
        dev_img = torch.from_numpy(np.array(img, dtype=np.uint8, order="C")).to(self.device, non_blocking=True)   # own, writable copy (PIL hands out read-only views)
        weak = self.weak_view(dev_img, params)
        if not self.is_train:
            d.pop("annotations", None)
            d["image"] = self.to_chw(weak)
            return d
        if "annotations" in d:
            d["instances"] = self.transform_annotations(d.pop("annotations"), h, w, params)
        strong = self.strong_view(weak, params["strong_ops"])
        weak_d = dict(d)
        d["image"] = self.to_chw(strong)
        weak_d["image"] = self.to_chw(weak)
        assert d["image"].shape == weak_d["image"].shape
        return d, weak_d


class TESTMapper(DatasetMapperUnsupervised):
    """The evaluation mapper (coin/data/dataset_mapper.py:59-121 = detectron2's DatasetMapper in test mode): shortest edge to
    ``INPUT.MIN_SIZE_TEST`` (cap ``MAX_SIZE_TEST``), no flip, annotations dropped; one dict with ``image`` uint8 [3, h, w] on the device."""

    def __init__(self, cfg, is_train: bool = False, **kw):
        super().__init__(cfg, False, **kw)
