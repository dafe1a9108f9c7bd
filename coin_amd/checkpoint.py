"""On-disk formats of the reference that cross the hot-path boundary.  Host code only.

1. Cached teacher results -- ``GDINO_collect.pth`` and the ``results`` / ``online_results`` entries of the trainers' checkpoints
   (coin/engine/pre_train.py:138-161, coin/engine/trainer.py:128-137,226-231;
   coin/modeling/meta_arch/gdino_collector.py:51-75):
       {"results": {dataset_name: {file_name: {"file_name", "image_id", "height", "width",
                                               "RCNN": {"instances": Instances}, "RPN": {"instances": Instances}[, "RPN_AUG": ...]}}}}
   with ``detectron2.structures.Instances`` (fields pred_boxes: Boxes, scores, pred_classes, probs) pickled by ``torch.save``.
   detectron2 is not a dependency of this build: the pickles are read with a class map onto ``coin_amd.structures`` (an
   ``Instances`` is {_image_size, _fields}, a ``Boxes`` is {tensor}) and written under detectron2's module paths, so files travel in
   both directions.

2. Checkpoints -- fvcore ``Checkpointer.save`` layout used by DetectionTSCheckpointer (coin/checkpoint/detection_checkpoint.py,
   trainer.py:128-137): {"model": state_dict of EnsembleTSModel (keys prefixed ``offline_teacher.``, ``model_student.``,
   ``merge_model.``; ts_ensemble.py:24-37), "optimizer", "optimizer_merge", "scheduler", "scheduler_merge", "iteration",
   "ap_50_student", "ap_50_offline_teacher", "online_results"}; pre-train checkpoints hold the bare detector under "model"
   (pre_train.py:138-146) and model-zoo files are a bare state dict (trainer.py:229-231).

Pinned (round 5) against artefacts written by the reference's OWN save code -- `PRETrainer.save` / `collect_results`, `CoinTrainer.save` over
`EnsembleTSModel` + `DetectionTSCheckpointer`, with `MyInstances` and detectron2-path containers inside -- committed under
tests/golden/ckpt/ (generator: tests/golden/gen_golden.py::case_checkpoint_formats; readers: tests/test_checkpoint_formats.py), and in the
other direction by the reference's `resume_or_load` reading what this module writes (tests/golden/live_checkpoint_roundtrip.py, run by
tests/test_reference_live.py where the reference checkout exists).  fvcore's `Checkpointer` / detectron2's `DetectionCheckpointer` are not
installed here: they are restated from their published behaviour as scaffolding in tests/golden/_ref_shim.py.
"""
from __future__ import annotations

import io
import pickle
import sys
import types
from typing import Any, Callable, Dict, Optional

import torch

from .structures import Boxes, Instances, MyInstances

# reference-side class path -> our class (both are plain attribute bags with the same attribute names)
_CLASS_MAP = {
    ("detectron2.structures.instances", "Instances"): Instances,
    ("detectron2.structures", "Instances"): Instances,
    ("detectron2.structures.boxes", "Boxes"): Boxes,
    ("detectron2.structures", "Boxes"): Boxes,
    ("coin.utils.util", "MyInstances"): MyInstances,
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        hit = _CLASS_MAP.get((module, name))
        if hit is not None:
            return hit
        return super().find_class(module, name)


class _PickleModule:
    """`pickle_module` for torch.load: the stock pickle with the class map above."""

    __name__ = "coin_amd_compat_pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _Unpickler(io.BytesIO(b), **kw).load())
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


def load_file(path: str, map_location="cpu") -> Any:
    """torch.load for files written by the reference (detectron2 / coin classes inside are mapped onto coin_amd.structures)."""
    return torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)


_MISSING = object()


class _as_detectron2:
    """While active, our Instances / Boxes pickle under detectron2's class paths (what the reference's torch.load expects)."""

    _PATHS = ((Instances, "detectron2.structures.instances", "Instances"), (Boxes, "detectron2.structures.boxes", "Boxes"),
              (MyInstances, "coin.utils.util", "MyInstances"))

    def __enter__(self):
        self._saved_mods, self._saved_names, self._saved_attrs = {}, [], []
        for cls, mod, name in self._PATHS:
            for m in (mod.rpartition(".")[0].rpartition(".")[0], mod.rpartition(".")[0], mod):
                if m and m not in sys.modules:
                    self._saved_mods[m] = None
                    sys.modules[m] = types.ModuleType(m)
            self._saved_names.append((cls, cls.__module__, cls.__qualname__, cls.__name__))
            self._saved_attrs.append((sys.modules[mod], name, sys.modules[mod].__dict__.get(name, _MISSING)))   # a real module may be loaded
            setattr(sys.modules[mod], name, cls)
            cls.__module__, cls.__qualname__, cls.__name__ = mod, name, name
        return self

    def __exit__(self, *a):
        for cls, mod, qual, name in self._saved_names:
            cls.__module__, cls.__qualname__, cls.__name__ = mod, qual, name
        for module, name, old in self._saved_attrs:
            if old is _MISSING:
                module.__dict__.pop(name, None)
            else:
                setattr(module, name, old)
        for m in self._saved_mods:
            sys.modules.pop(m, None)
        return False


def save_file(obj: Any, path: str) -> None:
    """torch.save such that the reference (with detectron2 installed) can torch.load the result."""
    with _as_detectron2():
        torch.save(obj, path)


# ------------------------------------------------------------------------------------------ cached teacher results
def _instances_to(inst: Instances, device) -> Instances:
    return inst.to(device)


class CloudResults:
    """file name -> deep copy of the cached cloud-detector result (GDINO_COLLECTOR.forward, gdino_collector.py:83-88), for
    ``CoinTrainer(cloud_results=...)`` / ``PRETrainer(collect_model=...)``."""

    def __init__(self, results: Dict[str, Dict[str, Dict]], device="cpu"):
        self._results, self.device = results, device

    @classmethod
    def load(cls, path: str, device="cpu") -> "CloudResults":
        """Reads ``GDINO_collect.pth`` ({"results": ...}), a pre-train checkpoint ("results") or a CoinTrainer checkpoint
        ("online_results") -- pre_train.py:152-153, trainer.py:226-231,246-247."""
        blob = load_file(path)
        for key in ("results", "online_results"):
            if isinstance(blob, dict) and key in blob:
                return cls(blob[key], device)
        raise KeyError(f"{path}: neither 'results' nor 'online_results' found")

    def save(self, path: str) -> None:
        save_file({"results": self._results}, path)

    def get_results(self):
        return self._results

    def set_results(self, results):
        self._results = results

    def entry(self, file_name: str) -> Dict:
        """The stored record itself (no copy); KeyError names the file when no dataset holds it."""
        for results in self._results.values():
            if isinstance(results, dict) and file_name in results:
                return results[file_name]
        raise KeyError(f"no cached teacher result for {file_name!r} (datasets: {list(self._results)[:4]})")

    def __call__(self, file_name: str) -> Dict:
        src = self.entry(file_name)
        out = {k: v for k, v in src.items() if not isinstance(v, dict)}
        for tag, entry in src.items():
            if isinstance(entry, dict):
                out[tag] = {k: (_copy_instances(v, self.device) if isinstance(v, Instances) else v) for k, v in entry.items()}
        return out


def _copy_instances(inst: Instances, device) -> Instances:
    new = type(inst)(inst.image_size)
    for k, v in inst.get_fields().items():
        new.set(k, Boxes(v.tensor.clone().to(device)) if isinstance(v, Boxes) else (v.clone().to(device) if torch.is_tensor(v) else v),
                check_len=False)
    return new


# ------------------------------------------------------------------------------------------ checkpoints
_PREFIXES = {"offline_teacher": "offline_teacher.", "student": "model_student.", "merge": "merge_model."}


def split_ensemble_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """EnsembleTSModel.state_dict() (ts_ensemble.py:24-37) -> {"offline_teacher": ..., "student": ..., "merge": ...}."""
    out = {k: {} for k in _PREFIXES}
    for key, v in sd.items():
        for part, prefix in _PREFIXES.items():
            if key.startswith(prefix):
                out[part][key[len(prefix):]] = v
                break
    return out


def detector_state_dict(blob: Any) -> Dict[str, torch.Tensor]:
    """The detector weights of a pre-train checkpoint ({"model": sd, ...}) or of a model-zoo file (bare sd) -- trainer.py:224-231."""
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob and isinstance(blob["model"], dict) else blob
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}


def optimizer_state(opt) -> Dict[str, Any]:
    """``torch.optim.SGD.state_dict()`` layout of a FusedSGD (one parameter per group, so parameter index == group index): what the
    reference's checkpointer writes under "optimizer" / "optimizer_merge" (fvcore Checkpointer.save: every checkpointable's
    state_dict()) and what its ``optimizer.load_state_dict`` accepts."""
    sd = opt.state_dict()
    bufs = sd.get("momentum_buffers")
    state = {}
    if bufs is not None and not sd.get("first", True):
        state = {i: {"momentum_buffer": b.detach().cpu().clone()} for i, b in enumerate(bufs)}
    groups = []
    for i, g in enumerate(opt.param_groups):
        groups.append({"lr": g["lr"], "momentum": g["momentum"], "dampening": 0, "weight_decay": g["weight_decay"], "nesterov": False,
                       "initial_lr": g["base_lr"], "name": g.get("name"), "params": [i]})
    return {"state": state, "param_groups": groups}


def load_optimizer_state(opt, sd: Dict[str, Any]) -> None:
    """Inverse of `optimizer_state` (also takes a reference-written torch SGD state dict with one parameter per group)."""
    groups = sd["param_groups"]
    assert len(groups) == len(opt.param_groups), "optimizer state has a different number of parameter groups"
    for g, s in zip(opt.param_groups, groups):
        g["lr"], g["weight_decay"] = s["lr"], s["weight_decay"]
        g["base_lr"] = s.get("initial_lr", g["base_lr"])
    state = sd.get("state", {})
    bufs = [state[i]["momentum_buffer"] if i in state and state[i].get("momentum_buffer") is not None else None for i in range(len(groups))]
    opt.load_momentum(bufs)


def scheduler_state(sched) -> Dict[str, Any]:
    """torch ``_LRScheduler.state_dict()`` of WarmupTwoStageMultiStepLR (lr_scheduler.py:22-66): every attribute but the optimizer."""
    return {"milestones": list(sched.milestones), "factor_list": list(sched.factor_list), "gamma": sched.gamma, "warmup_factor": sched.warmup_factor,
            "warmup_iters": sched.warmup_iters, "warmup_method": sched.warmup_method, "base_lrs": list(sched.base_lrs),
            "last_epoch": sched.last_epoch, "_step_count": sched.last_epoch + 1, "_last_lr": [g["lr"] for g in sched.optimizer.param_groups]}


def load_cointrainer_weights(trainer, weights: str, resume: bool = False) -> None:
    """``CoinTrainer.resume_or_load`` (trainer.py:220-262) for ``MODEL.WEIGHTS``:
    "offline.pth+cloud_results.pth" -> teacher weights + cached cloud results; a single path -> a CoinTrainer checkpoint, with
    ``resume`` also both optimizers and both schedulers (trainer.py:237-241), without it only `scheduler(.merge).last_epoch` is
    set to the stored iteration (trainer.py:246-247: the rate itself is re-derived at the next scheduler step, as there)."""
    paths = weights.split("+")
    if len(paths) == 2:
        trainer.offline_teacher.load_state_dict(detector_state_dict(load_file(paths[0])), strict=False)
        trainer.model_CLOUD = CloudResults.load(paths[1], device=trainer.device)
    elif len(paths) == 1:
        blob = load_file(paths[0])
        parts = split_ensemble_state_dict(blob["model"])
        trainer.offline_teacher.load_state_dict(parts["offline_teacher"], strict=False)
        trainer.model.load_state_dict(parts["student"], strict=False)
        trainer.merge.load_state_dict(parts["merge"], strict=False)
        if resume:
            for name in ("optimizer", "optimizer_merge"):
                if blob.get(name) is not None:
                    load_optimizer_state(getattr(trainer, name), blob[name])
            for name in ("scheduler", "scheduler_merge"):
                if blob.get(name) is not None:
                    getattr(trainer, name).load_state_dict(blob[name])
        else:
            trainer.scheduler.last_epoch = trainer.scheduler_merge.last_epoch = blob.get("iteration", -1)
        trainer.start_iter = trainer.iter = blob.get("iteration", -1) + 1
        if resume:  # the AP50 histories travel with the checkpoint (hooks.py:60-84)
            trainer.ap_50_student = dict(blob.get("ap_50_student") or {}) or None
            trainer.ap_50_offline_teacher = dict(blob.get("ap_50_offline_teacher") or {}) or None
        if blob.get("online_results") is not None:
            trainer.model_CLOUD = CloudResults(blob["online_results"], device=trainer.device)
    else:
        raise AssertionError("pretrain models path should be two paths split by '+'. ")


def save_cointrainer_checkpoint(trainer, path: str, iteration: Optional[int] = None) -> None:
    """DetectionTSCheckpointer.save layout (trainer.py:128-137)."""
    model = {}
    for part, module in (("offline_teacher", trainer.offline_teacher), ("student", trainer.model), ("merge", trainer.merge)):
        model.update({_PREFIXES[part] + k: v.detach().cpu() for k, v in module.state_dict().items()})
    blob = {"model": model, "optimizer": optimizer_state(trainer.optimizer), "optimizer_merge": optimizer_state(trainer.optimizer_merge),
            "scheduler": scheduler_state(trainer.scheduler), "scheduler_merge": scheduler_state(trainer.scheduler_merge),
            "iteration": trainer.iter - 1 if iteration is None else iteration, "ap_50_student": dict(getattr(trainer, "ap_50_student", None) or {}),
            "ap_50_offline_teacher": dict(getattr(trainer, "ap_50_offline_teacher", None) or {}),
            "online_results": trainer.model_CLOUD.get_results() if hasattr(trainer.model_CLOUD, "get_results") else None}
    save_file(blob, path)
