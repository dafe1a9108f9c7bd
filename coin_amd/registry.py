"""Name -> class registries with the reference's registry names (detectron2.utils.registry.Registry
semantics: ``@REG.register()`` on a class or function, ``REG.get(name)``).

Reference: META_ARCH_REGISTRY / BACKBONE_REGISTRY / PROPOSAL_GENERATOR_REGISTRY / ROI_HEADS_REGISTRY come
from detectron2; TEXT_ENCODER_REGISTRY coin/modeling/text_encoder/build.py:7, MERGE_REGISTRY
coin/modeling/merge/build.py.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, Optional


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def _do_register(self, name: str, obj: Any) -> None:
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj: Any = None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name: str) -> Any:
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator:
        return iter(self._obj_map.items())


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
TEXT_ENCODER_REGISTRY = Registry("TEXT_ENCODER")
MERGE_REGISTRY = Registry("MERGE")
