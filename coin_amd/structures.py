"""Boxes / Instances / ImageList with the detectron2 0.5 interface the reference modules use
(coin/modeling/**: `.tensor`, `Boxes.cat`, `Instances.get_fields()/set/has/remove/cat/[]`, `image_size`),
plus COIN's ``MyInstances.set(check_len=False)`` extension (coin/utils/util.py:188-267).
Tensors may live on any device; nothing here syncs with the host except `len()` on an empty container.
"""
from __future__ import annotations

import itertools
from typing import Any, Dict, List, Tuple

import torch


class Boxes:
    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if not tensor.is_floating_point():
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *a, **k):
        return Boxes(self.tensor.to(*a, **k))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]):
        h, w = box_size
        b = self.tensor
        self.tensor = torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, sx: float, sy: float):
        # (in-place on the strided views: `t[:, 0::2] *= sx` is getitem + mul_ + a setitem copy of the view onto itself -- two launches)
        self.tensor[:, 0::2].mul_(sx)
        self.tensor[:, 1::2].mul_(sy)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list: List["Boxes"]):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __repr__(self):
        return f"Boxes({self.tensor})"


def pairwise_iou(b1: Boxes, b2: Boxes) -> torch.Tensor:
    a, b = b1.tensor, b2.tensor
    area1, area2 = b1.area(), b2.area()
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp_(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros((), dtype=inter.dtype, device=inter.device))


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name: str, value: Any, check_len: bool = True):
        if check_len and len(self._fields):
            assert len(self) == len(value), f"Adding a field of length {len(value)} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        ret = type(self)(self._image_size)
        for name, v in self._fields.items():
            ret._fields[name] = v.to(*a, **k) if hasattr(v, "to") else v
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = type(self)(self._image_size)
        for name, v in self._fields.items():
            ret._fields[name] = v[item]
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(lst: List["Instances"]):
        assert len(lst) > 0
        if len(lst) == 1:
            return lst[0]
        ret = type(lst[0])(lst[0].image_size)
        for k in lst[0]._fields.keys():
            vals = [i.get(k) for i in lst]
            v0 = vals[0]
            if isinstance(v0, torch.Tensor):
                vals = torch.cat(vals, dim=0)
            elif isinstance(v0, list):
                vals = list(itertools.chain(*vals))
            elif hasattr(type(v0), "cat"):
                vals = type(v0).cat(vals)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret._fields[k] = vals
        return ret

    def __repr__(self):
        return f"Instances(num={len(self) if self._fields else 0}, size={self._image_size}, fields={list(self._fields)})"


class MyInstances(Instances):
    """coin/utils/util.py:188-267.  Its extensions (`set(check_len=False)`, `to` / `__getitem__` / `cat` returning the subclass) are already
    in the base class above; the subclass exists so that a cached result keeps its class across a save / load round trip
    (coin_amd/checkpoint.py writes it under the reference's class path coin.utils.util.MyInstances)."""


class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride
