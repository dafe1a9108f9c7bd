"""Trainer base: the per-view target preparation of coin/engine/base.py:80-136 and a sync-free metric sink.

``process`` rescales cached teacher boxes (original-image pixels) to the network input size and mirrors them
when the view was flipped; ``preprocess_results`` applies it to the RCNN and RPN entries.  The reference's
``_write_metrics`` (base.py:206-243) does ~10 ``.item()`` host syncs + a gloo gather EVERY step; here losses
stay on the device and are read back only every ``log_period`` steps.
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch

from ..structures import Boxes, Instances, MyInstances


class BASE_Trainer:
    log_period = 20

    def process(self, instances: Instances, old_size, new_size, random_flip, thresh=None, keep_name=False) -> Instances:
        img_h, img_w = old_size
        net_h, net_w = new_size
        new = MyInstances((net_h, net_w))
        for k, v in instances.get_fields().items():
            new.set(k, Boxes(v.tensor.clone()) if isinstance(v, Boxes) else v, check_len=False)
        boxes = new.pred_boxes if new.has("pred_boxes") else new.gt_boxes
        boxes.scale(net_w / img_w, net_h / img_h)
        t = boxes.tensor
        if random_flip == "horizontal":
            boxes = Boxes(torch.stack((net_w - t[:, 2], t[:, 1], net_w - t[:, 0], t[:, 3]), dim=1))
        elif random_flip == "vertical":
            boxes = Boxes(torch.stack((t[:, 0], net_h - t[:, 3], t[:, 2], net_h - t[:, 1]), dim=1))
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
            return new[instances.scores >= thresh]
        return new

    def preprocess_results(self, results: Dict, new_image_size, random_flip, thresh=None) -> Dict:
        size = (results["height"], results["width"])
        results["RCNN"] = self.process(results["RCNN"]["instances"], size, new_image_size, random_flip, thresh)
        key = "RPN_AUG" if "RPN_AUG" in results else "RPN"
        results["RPN"] = self.process(results[key]["instances"], size, new_image_size, random_flip, thresh)
        results.pop("RPN_AUG", None)
        return results

    # ---- evaluation (base.py:176-204 -> detectron2 inference_on_dataset)
    @staticmethod
    @torch.no_grad()
    def test(model, items, evaluator, batch_size: int = 1) -> Dict:
        """Run `model` in inference mode over `items` (dicts with image / height / width / image_id) and hand inputs + outputs to
        `evaluator` (reset / process / evaluate, e.g. coin_amd.evaluation.PascalVOCEvaluator).  -> evaluator.evaluate()."""
        was_training = model.training
        model.eval()
        evaluator.reset()
        batch = []
        for item in items:   # `items` may be lazy (coin_amd.data.LazyTestSet): an image is mapped when its batch is due and dropped afterwards
            batch.append(item)
            if len(batch) == batch_size:
                evaluator.process(batch, model(batch, branch="test"))
                batch = []
        if batch:
            evaluator.process(batch, model(batch, branch="test"))
        model.train(was_training)
        return evaluator.evaluate()

    # ---- periodic evaluation (MyEvalHook, coin/engine/hooks.py:144-190; build_hooks: pre_train.py:300-310, trainer.py:296-318)
    _eval_items = _eval_factory = None

    def set_evaluation(self, items, evaluator_factory, batch_size: int = 1):
        """`items`: the test set as dataset-mapper outputs (image / height / width / image_id ...), or a lazy re-iterable that maps an image
        when it is reached (`coin_amd.data.LazyTestSet`, this rank's shard); `evaluator_factory()` builds a fresh
        evaluator (reset / process / evaluate, e.g. coin_amd.evaluation.PascalVOCEvaluator).  With `TEST.EVAL_PERIOD > 0` `train()` then
        evaluates after every EVAL_PERIOD-th iteration (BEFORE that iteration's checkpoint, so the file carries the new AP50) and
        after the last one, exactly like the reference's hook order."""
        # a re-iterable (list, LazyTestSet) is kept as it is; a one-shot generator is materialised
        self._eval_items = items if hasattr(items, "__len__") else list(items)
        self._eval_factory, self._eval_batch = evaluator_factory, batch_size

    def _eval_due(self, start: int = -1) -> bool:
        """detectron2 EvalHook.after_step / after_train with MyEvalHook's start iteration; `self.iter` = index of the step just done."""
        if self._eval_factory is None:
            return False
        period, nxt = self.cfg.TEST.EVAL_PERIOD, self.iter + 1
        periodic = period > 0 and nxt % period == 0 and nxt > start and nxt != self.max_iter
        return periodic or nxt >= self.max_iter

    def _evaluate(self, model) -> Dict:
        return self.test(model, self._eval_items, self._eval_factory(), batch_size=self._eval_batch)

    # ---- metrics without a per-step host sync
    def _write_metrics(self, metrics_dict: Dict[str, torch.Tensor], iteration: int) -> Optional[Dict[str, float]]:
        self._last_metrics = metrics_dict
        if iteration % self.log_period != 0:
            return None
        keys = [k for k, v in metrics_dict.items() if isinstance(v, torch.Tensor)]
        vals = torch.stack([metrics_dict[k].detach().float() for k in keys]).cpu().tolist() if keys else []
        out = dict(zip(keys, vals))
        out["total_loss"] = sum(v for k, v in out.items() if k.startswith("loss"))
        return out
