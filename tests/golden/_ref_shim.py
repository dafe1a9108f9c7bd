"""Import shim that lets the reference's own hot-path modules run in the build container.

Used ONLY by ``tests/golden/gen_golden.py`` (golden-vector capture) and by the
optional ``tests/test_reference_live.py`` when ``/root/reference`` exists.  It
never runs on the GPU box and nothing here is copied from the reference: the
reference files are imported from where they lie.

What it does (recipe of SURVEY.md Appendix D):
  * registers ``coin`` and the sub-packages whose ``__init__`` would pull in
    GroundingDINO / GLIP / data loaders as *namespace* modules, so that only
    the requested files execute;
  * injects stand-ins for ``detectron2``, ``fvcore``, ``torchvision.transforms``,
    ``cv2``, ``supervision``, ``ftfy``, ``groundingdino``.  Non-arithmetic
    scaffolding (``configurable``, ``Registry``, loggers, event storage) is a
    no-op; every ARITHMETIC member comes from ``oracle/d2.py`` (the restated,
    known-answer-tested third-party semantics), so the reference's COIN-level
    logic is executed unmodified on top of it;
  * neutralises the two lines that cannot run in CPU fp32
    (``clip_text.py:137`` fp16 weights -> caller uses ``.float()``;
    ``clip_text.py:204`` in-place ``x /= norm`` -> out-of-place, mathematically
    identical) by patching the module source in memory at import time.
"""
from __future__ import annotations

import importlib
import importlib.util
import logging
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("COIN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "coin"))


class _Lenient(types.ModuleType):
    """Module whose unknown attributes resolve to inert placeholders (classes usable as bases)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        placeholder = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, placeholder)
        return placeholder


class Registry(dict):
    def __init__(self, name="registry"):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def configurable(init_func=None, *, from_config=None):
    """Identity: reference classes are built with explicit kwargs, ``from_config`` is bypassed."""
    if init_func is not None:
        return init_func
    return lambda f: f


class _Storage:
    iter = 0

    def put_scalar(self, *a, **k):
        pass

    def put_scalars(self, *a, **k):
        pass

    def put_image(self, *a, **k):
        pass


_STORAGE = _Storage()


def _mod(name: str, **attrs) -> types.ModuleType:
    m = _Lenient(name)
    m.__path__ = []  # behave as a package so that ``import a.b.c`` resolves
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def install() -> None:
    """Install stand-in third-party modules + namespace ``coin`` packages (idempotent)."""
    if "coin" in sys.modules and getattr(sys.modules["coin"], "_shim", False):
        return
    assert reference_available(), f"reference not found at {REFERENCE_ROOT}"
    repo_root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    from oracle import d2

    class Backbone(nn.Module):
        size_divisibility = 0

    class GeneralizedRCNN:
        @staticmethod
        def _postprocess(instances, batched_inputs, image_sizes):
            out = []
            for res, inp, size in zip(instances, batched_inputs, image_sizes):
                h, w = inp.get("height", size[0]), inp.get("width", size[1])
                out.append({"instances": d2.detector_postprocess(res, h, w)})
            return out

    class ToTensor:
        def __call__(self, pic):  # HWC uint8 ndarray -> CHW float in [0,1]
            t = torch.from_numpy(pic).permute(2, 0, 1).contiguous()
            return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return (t - mean) / std

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def setup_logger(*a, **k):
        return logging.getLogger("coin_ref")

    def retry_if_cuda_oom(f):
        return f

    _mod("detectron2")
    _mod("detectron2.config", configurable=configurable, CfgNode=dict)
    _mod("detectron2.layers", ShapeSpec=d2.ShapeSpec, batched_nms=d2.batched_nms, nonzero_tuple=d2.nonzero_tuple, cat=d2.cat)
    _mod("detectron2.layers.batch_norm", FrozenBatchNorm2d=d2.FrozenBatchNorm2d)
    _mod("detectron2.structures", Boxes=d2.Boxes, Instances=d2.Instances, ImageList=d2.ImageList, pairwise_iou=d2.pairwise_iou)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=Registry)
    _mod("detectron2.utils.logger", setup_logger=setup_logger)
    _mod("detectron2.utils.comm", get_rank=lambda: 0, get_world_size=lambda: 1, is_main_process=lambda: True,
         get_local_rank=lambda: 0)
    _mod("detectron2.utils.events", get_event_storage=lambda: _STORAGE)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=retry_if_cuda_oom)
    _mod("detectron2.utils.env", TORCH_VERSION=(2, 10))
    _mod("detectron2.utils.collect_env")
    _mod("detectron2.utils.file_io")
    _mod("detectron2.utils.visualizer")
    _mod("detectron2.data", MetadataCatalog=None)
    _mod("detectron2.data.detection_utils")
    _mod("detectron2.engine")
    _mod("detectron2.engine.defaults")
    _mod("detectron2.engine.train_loop")
    _mod("detectron2.evaluation")
    _mod("detectron2.solver")
    _mod("detectron2.solver.build", maybe_add_gradient_clipping=lambda cfg, opt: opt)
    _mod("detectron2.solver.lr_scheduler", _get_warmup_factor_at_iter=d2.get_warmup_factor_at_iter)
    _mod("detectron2.modeling", META_ARCH_REGISTRY=Registry("META_ARCH"))
    _mod("detectron2.modeling.meta_arch")
    _mod("detectron2.modeling.meta_arch.build", META_ARCH_REGISTRY=sys.modules["detectron2.modeling"].META_ARCH_REGISTRY)
    _mod("detectron2.modeling.meta_arch.rcnn", GeneralizedRCNN=GeneralizedRCNN)
    _mod("detectron2.modeling.backbone", Backbone=Backbone, BACKBONE_REGISTRY=Registry("BACKBONE"), build_backbone=None)
    _mod("detectron2.modeling.matcher", Matcher=d2.Matcher)
    _mod("detectron2.modeling.poolers", ROIPooler=d2.ROIPooler)
    _mod("detectron2.modeling.box_regression", Box2BoxTransform=d2.Box2BoxTransform,
         _dense_box_regression_loss=d2.dense_box_regression_loss)
    _mod("detectron2.modeling.roi_heads", ROI_HEADS_REGISTRY=Registry("ROI_HEADS"), ROIHeads=d2.ROIHeads,
         select_foreground_proposals=None)
    _mod("detectron2.modeling.roi_heads.mask_head", build_mask_head=None)
    _mod("detectron2.modeling.proposal_generator", RPN=d2.RPN, build_proposal_generator=None)
    _mod("detectron2.modeling.proposal_generator.build", PROPOSAL_GENERATOR_REGISTRY=Registry("PROPOSAL_GENERATOR"))
    _mod("detectron2.modeling.proposal_generator.proposal_utils", add_ground_truth_to_proposals=d2.add_ground_truth_to_proposals)
    _mod("fvcore")
    _mod("fvcore.nn", smooth_l1_loss=d2.smooth_l1_loss, giou_loss=None)
    _mod("fvcore.nn.precise_bn")
    if "torchvision" not in sys.modules:
        _mod("torchvision")
        _mod("torchvision.transforms", ToTensor=ToTensor, Normalize=Normalize, Compose=Compose)
    for name in ("cv2", "supervision", "groundingdino", "groundingdino.util", "groundingdino.util.misc", "tensorboardX"):
        if name not in sys.modules:
            _mod(name)
    _mod("ftfy", fix_text=lambda s: s)

    # namespace packages: nothing in coin/__init__.py or the heavy sub-package __init__s executes
    def ns(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
        m._shim = True
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], child, m)
        return m

    ns("coin", "coin")
    ns("coin.modeling", "coin/modeling")
    ns("coin.modeling.meta_arch", "coin/modeling/meta_arch")
    ns("coin.modeling.roi_heads", "coin/modeling/roi_heads")
    ns("coin.modeling.backbone", "coin/modeling/backbone")
    ns("coin.modeling.proposal_generator", "coin/modeling/proposal_generator")
    ns("coin.utils", "coin/utils")
    ns("coin.layers", "coin/layers")
    ns("coin.solver", "coin/solver")
    ns("coin.engine", "coin/engine")
    # light packages whose __init__ is safe under the stand-ins are imported for real (text_encoder, merge)
    _import_patched(
        "coin.modeling.text_encoder.clip_text",
        "coin/modeling/text_encoder/clip_text.py",
        [("x /= torch.norm(x, dim=-1, keepdim=True)", "x = x / torch.norm(x, dim=-1, keepdim=True)")],
        package_init="coin/modeling/text_encoder",
    )
    # coin.modeling.roi_heads.build is needed by clip_rcnn.py's `from coin.modeling.roi_heads import build_roi_heads`
    rb = importlib.import_module("coin.modeling.roi_heads.build")
    sys.modules["coin.modeling.roi_heads"].build_roi_heads = rb.build_roi_heads


def _import_patched(modname: str, rel: str, replacements, package_init: str | None = None):
    """Import a reference file with in-memory source substitutions (nothing is written to disk)."""
    if package_init is not None:
        pkg = modname.rpartition(".")[0]
        if pkg not in sys.modules or getattr(sys.modules[pkg], "_shim", False) is False:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, package_init)]
            m._shim = True
            sys.modules[pkg] = m
            parent, _, child = pkg.rpartition(".")
            setattr(sys.modules[parent], child, m)
            b = importlib.import_module(pkg + ".build")
            for k in dir(b):
                if not k.startswith("_"):
                    setattr(m, k, getattr(b, k))
    path = os.path.join(REFERENCE_ROOT, rel)
    src = open(path).read()
    for old, new in replacements:
        assert old in src, f"patch anchor not found in {rel}: {old!r}"
        src = src.replace(old, new)
    spec = importlib.util.spec_from_loader(modname, loader=None, origin=path)
    mod = importlib.util.module_from_spec(spec)
    mod.__file__ = path
    mod.__package__ = modname.rpartition(".")[0]
    sys.modules[modname] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    setattr(sys.modules[mod.__package__], modname.rpartition(".")[2], mod)
    return mod


def ref(modname: str):
    """Import a reference module (``coin.…``) under the shim."""
    install()
    return importlib.import_module(modname)


# --------------------------------------------------------------------------- #
# checkpoint stack: scaffolding so that the reference's OWN save / load code runs
# --------------------------------------------------------------------------- #
def install_checkpoint_stack() -> None:
    """Lets the reference's checkpoint code (coin/checkpoint/detection_checkpoint.py, PRETrainer.save / resume_or_load,
    CoinTrainer.save / resume_or_load) run here, so that golden ARTEFACTS are written by the reference itself and the product's files
    are read back by the reference's loaders (tests/golden/gen_golden.py::case_checkpoint_formats, tests/test_reference_live.py).

    What is stood in for, and only that -- non-arithmetic third-party scaffolding, restated from its published behaviour:
      * ``fvcore.common.checkpoint`` (fvcore 0.1.5): ``Checkpointer.save`` = torch.save({"model": model.state_dict(),
        **{name: obj.state_dict() for the checkpointables}, **kwargs}) + the ``last_checkpoint`` tag file; ``Checkpointer.load`` =
        torch.load -> ``_load_model`` ("module." prefix stripped, shape-mismatched keys dropped, ``load_state_dict(strict=False)``)
        -> ``load_state_dict`` of the requested checkpointables -> the remaining dict; ``_IncompatibleKeys``;
        ``_strip_prefix_if_present``;
      * ``detectron2.checkpoint.DetectionCheckpointer`` (detectron2 0.5): save_to_disk on the main process, a bare state dict is
        wrapped as {"model": ...}, ``_load_model`` = the parent's;
      * the class paths detectron2 pickles its containers under: ``detectron2.structures.instances.Instances`` /
        ``detectron2.structures.boxes.Boxes`` (the stand-ins' ``__module__`` is set accordingly), so that the bytes written here are
        what a real installation reads and writes.
    ``torch.load`` is called with ``weights_only=False`` (torch >= 2.6 changed the default; the reference targets torch 1.9)."""
    import collections
    import copy

    from torch.nn.parallel import DataParallel, DistributedDataParallel

    install()
    from oracle import d2

    class _IncompatibleKeys(collections.namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys", "incorrect_shapes"])):
        pass

    def _strip_prefix_if_present(state_dict, prefix):
        keys = sorted(state_dict.keys())
        if not all(len(k) == 0 or k.startswith(prefix) for k in keys):
            return
        for k in keys:
            state_dict[k[len(prefix):]] = state_dict.pop(k)
        md = getattr(state_dict, "_metadata", None)
        if md is not None:
            for k in list(md.keys()):
                if len(k) == 0:
                    continue
                md[k[len(prefix):]] = md.pop(k)

    class Checkpointer:
        def __init__(self, model, save_dir="", *, save_to_disk=True, **checkpointables):
            if isinstance(model, (DistributedDataParallel, DataParallel)):
                model = model.module
            self.model = model
            self.checkpointables = copy.copy(checkpointables)
            self.logger = logging.getLogger("fvcore.checkpoint")
            self.save_dir, self.save_to_disk = save_dir, save_to_disk

        def add_checkpointable(self, key, checkpointable):
            self.checkpointables[key] = checkpointable

        def save(self, name, **kwargs):
            if not self.save_dir or not self.save_to_disk:
                return
            data = {"model": self.model.state_dict()}
            for key, obj in self.checkpointables.items():
                data[key] = obj.state_dict()
            data.update(kwargs)
            basename = "{}.pth".format(name)
            save_file = os.path.join(self.save_dir, basename)
            assert os.path.basename(save_file) == basename, basename
            with open(save_file, "wb") as f:
                torch.save(data, f)
            self.tag_last_checkpoint(basename)

        def load(self, path, checkpointables=None):
            if not path:
                return {}
            assert os.path.isfile(path), "Checkpoint {} not found!".format(path)
            checkpoint = self._load_file(path)
            self._load_model(checkpoint)
            for key in self.checkpointables if checkpointables is None else checkpointables:
                if key in checkpoint:
                    self.checkpointables[key].load_state_dict(checkpoint.pop(key))
            return checkpoint

        def has_checkpoint(self):
            return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

        def get_checkpoint_file(self):
            try:
                with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                    last = f.read().strip()
            except IOError:
                return ""
            return os.path.join(self.save_dir, last)

        def tag_last_checkpoint(self, last_filename_basename):
            with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
                f.write(last_filename_basename)

        def _load_file(self, f):
            return torch.load(f, map_location=torch.device("cpu"), weights_only=False)

        def _convert_ndarray_to_tensor(self, state_dict):
            import numpy as np

            for k in list(state_dict.keys()):
                v = state_dict[k]
                if isinstance(v, np.ndarray):
                    state_dict[k] = torch.from_numpy(v)

        def _load_model(self, checkpoint):
            sd = checkpoint.pop("model")
            self._convert_ndarray_to_tensor(sd)
            _strip_prefix_if_present(sd, "module.")
            model_sd = self.model.state_dict()
            incorrect = []
            for k in list(sd.keys()):
                if k in model_sd and tuple(model_sd[k].shape) != tuple(sd[k].shape):
                    incorrect.append((k, tuple(sd[k].shape), tuple(model_sd[k].shape)))
                    sd.pop(k)
            inc = self.model.load_state_dict(sd, strict=False)
            return _IncompatibleKeys(missing_keys=inc.missing_keys, unexpected_keys=inc.unexpected_keys, incorrect_shapes=incorrect)

    class DetectionCheckpointer(Checkpointer):
        def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
            super().__init__(model, save_dir, save_to_disk=True if save_to_disk is None else save_to_disk, **checkpointables)

        def _load_file(self, filename):
            loaded = super()._load_file(filename)
            if "model" not in loaded:
                loaded = {"model": loaded}
            return loaded

        def _load_model(self, checkpoint):
            return super()._load_model(checkpoint)

    for name in ("fvcore.common", "fvcore.common.checkpoint", "detectron2.checkpoint", "detectron2.checkpoint.c2_model_loading",
                 "coin.checkpoint", "coin.checkpoint.detection_checkpoint"):
        sys.modules.pop(name, None)    # lenient placeholders an earlier case may have registered
    _mod("fvcore.common")
    _mod("fvcore.common.checkpoint", Checkpointer=Checkpointer, _IncompatibleKeys=_IncompatibleKeys, _strip_prefix_if_present=_strip_prefix_if_present)
    _mod("detectron2.checkpoint", DetectionCheckpointer=DetectionCheckpointer)
    _mod("detectron2.checkpoint.c2_model_loading", align_and_update_state_dicts=None)
    # pickle class paths of detectron2's containers
    _mod("detectron2.structures.instances", Instances=d2.Instances)
    _mod("detectron2.structures.boxes", Boxes=d2.Boxes)
    d2.Instances.__module__, d2.Boxes.__module__ = "detectron2.structures.instances", "detectron2.structures.boxes"
    m = types.ModuleType("coin.checkpoint")
    m.__path__ = [os.path.join(REFERENCE_ROOT, "coin", "checkpoint")]
    m._shim = True
    sys.modules["coin.checkpoint"] = m
    sys.modules["coin"].checkpoint = m
