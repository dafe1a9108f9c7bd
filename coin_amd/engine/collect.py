"""Collection of the CLIP-relabelled pre-training targets (coin/engine/pre_train.py:148-161,
coin/modeling/meta_arch/clip_collector.py:46-63).

Before CLIPDET pre-training the reference runs every training image once through the CLIP teacher: the boxes cached from the cloud
detector (``GDINO_collect.pth``) are re-scored by CLIP and the ones it calls background are dropped; the result -- same nested layout
``{dataset: {file_name: result}}`` -- is what ``PRETrainer.set_boxes`` reads during training.  Batch size is 1 as in the reference.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable

import torch

from ..checkpoint import CloudResults


@torch.no_grad()
def collect_clip_results(clip_model, items: Iterable[Dict], cloud_results: Callable[[str], Dict], dataset_name: str = "train",
                         device="cpu") -> CloudResults:
    """items: dicts with ``image`` (uint8 [3,H,W]), ``file_name``, ``image_id``, ``height``, ``width`` (one per training image);
    cloud_results: file name -> cached cloud-detector result.  -> the CLIP-relabelled cache."""
    was_training = clip_model.training
    clip_model.eval()
    out: Dict[str, Dict] = {}
    for item in items:
        if item["file_name"] in out:
            continue
        pre = cloud_results(item["file_name"])
        assert pre is not None, f"no cached cloud result for {item['file_name']}"
        res = clip_model([item], pre)
        for v in res.values():
            if isinstance(v, dict) and "instances" in v:
                v["instances"] = v["instances"].to("cpu")
        out[res["file_name"]] = res
    clip_model.train(was_training)
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # clip_collector.py:60-63: every rank relabels its shard of the loader, then all ranks hold the union
        # (`comm.all_gather(self._results)`: a pickled-object gather, once per run, off the training path)
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, out)
        for part in parts:
            out.update(part)
    return CloudResults({dataset_name: out}, device=device)
