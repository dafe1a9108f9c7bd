"""Config surface of the hot path: a yacs-style ``CfgNode`` (attribute access, ``_BASE_`` YAML
inheritance, ``merge_from_list`` for trailing ``KEY VALUE`` options), the subset of detectron2 0.5
defaults that COIN's code reads and ``add_config`` with the keys of
/root/reference/coin/config.py:17-143 (same names, same defaults), so the reference's YAML files under
configs/coin/ load unchanged.
"""
from __future__ import annotations

import copy
import os
from ast import literal_eval
from typing import Any, List

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    """dict with attribute access + freeze; nested dicts become CfgNodes."""

    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    # ---- merging
    @staticmethod
    def load_yaml_with_base(filename: str) -> dict:
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not base.startswith("/"):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode.load_yaml_with_base(base)
            _merge_dicts(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, filename: str):
        self._merge(CfgNode.load_yaml_with_base(filename), self, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, self, [])

    def merge_from_list(self, opts: List[Any]):
        assert len(opts) % 2 == 0, f"Override list has odd length: {opts}"
        for full_key, v in zip(opts[0::2], opts[1::2]):
            d = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                assert p in d, f"Non-existent key: {full_key}"
                d = d[p]
            assert parts[-1] in d, f"Non-existent key: {full_key}"
            d[parts[-1]] = _coerce(_decode(v), d[parts[-1]], full_key)

    @staticmethod
    def _merge(src: dict, dst: "CfgNode", stack: List[str]):
        for k, v in src.items():
            full = ".".join(stack + [k])
            if k not in dst:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(dst[k], CfgNode):
                assert isinstance(v, dict), f"{full}: expected a mapping"
                CfgNode._merge(v, dst[k], stack + [k])
            else:
                dst[k] = _coerce(_decode(v), dst[k], full)

    def dump(self) -> str:
        return yaml.safe_dump(_to_plain(self), sort_keys=True)


def _to_plain(n):
    if isinstance(n, CfgNode):
        return {k: _to_plain(v) for k, v in n.items()}
    if isinstance(n, tuple):
        return [_to_plain(v) for v in n]
    if isinstance(n, list):
        return [_to_plain(v) for v in n]
    return n


def _merge_dicts(a: dict, b: dict):
    for k, v in a.items():
        if isinstance(v, dict) and isinstance(b.get(k), dict):
            _merge_dicts(v, b[k])
        else:
            b[k] = v


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    """yacs type check with the tuple<->list and int->float conversions it allows."""
    if old is None or new is None or type(new) == type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, str) and not isinstance(new, str):
        return str(new) if not isinstance(new, (list, tuple, dict)) else new
    if isinstance(old, (tuple, list)) and isinstance(new, str):
        return new
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")


CN = CfgNode


def _d2_defaults() -> CfgNode:
    """detectron2 0.5 `config/defaults.py` values for the keys the COIN hot path reads."""
    C = CN()
    C.VERSION = 2
    C.MODEL = CN()
    C.MODEL.LOAD_PROPOSALS = False
    C.MODEL.MASK_ON = False
    C.MODEL.KEYPOINT_ON = False
    C.MODEL.DEVICE = "cuda"
    C.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    C.MODEL.WEIGHTS = ""
    C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    C.INPUT = CN()
    C.INPUT.MIN_SIZE_TRAIN = (800,)
    C.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
    C.INPUT.MAX_SIZE_TRAIN = 1333
    C.INPUT.MIN_SIZE_TEST = 800
    C.INPUT.MAX_SIZE_TEST = 1333
    C.INPUT.RANDOM_FLIP = "horizontal"
    C.INPUT.CROP = CN({"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]})
    C.INPUT.FORMAT = "BGR"
    C.INPUT.MASK_FORMAT = "polygon"
    C.DATASETS = CN()
    C.DATASETS.TRAIN = ()
    C.DATASETS.PROPOSAL_FILES_TRAIN = ()
    C.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN = 2000
    C.DATASETS.TEST = ()
    C.DATASETS.PROPOSAL_FILES_TEST = ()
    C.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST = 1000
    C.DATALOADER = CN()
    C.DATALOADER.NUM_WORKERS = 4
    C.DATALOADER.ASPECT_RATIO_GROUPING = True
    C.DATALOADER.SAMPLER_TRAIN = "TrainingSampler"
    C.DATALOADER.REPEAT_THRESHOLD = 0.0
    C.DATALOADER.FILTER_EMPTY_ANNOTATIONS = True
    C.MODEL.BACKBONE = CN({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    C.MODEL.PROPOSAL_GENERATOR = CN({"NAME": "RPN", "MIN_SIZE": 0})
    # FPN extension (coin_amd/modeling/fpn.py, swin.py; no counterpart in the reference's config tree)
    C.MODEL.FPN = CN({"OUT_CHANNELS": 256})
    C.MODEL.SWIN = CN({"EMBED_DIM": 96, "DEPTHS": [2, 2, 6, 2], "NUM_HEADS": [3, 6, 12, 24], "WINDOW_SIZE": 7})
    C.MODEL.ANCHOR_GENERATOR = CN()
    C.MODEL.ANCHOR_GENERATOR.NAME = "DefaultAnchorGenerator"
    C.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64, 128, 256, 512]]
    C.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    C.MODEL.ANCHOR_GENERATOR.ANGLES = [[-90, 0, 90]]
    C.MODEL.ANCHOR_GENERATOR.OFFSET = 0.0
    C.MODEL.RPN = CN()
    C.MODEL.RPN.HEAD_NAME = "StandardRPNHead"
    C.MODEL.RPN.IN_FEATURES = ["res4"]
    C.MODEL.RPN.BOUNDARY_THRESH = -1
    C.MODEL.RPN.IOU_THRESHOLDS = [0.3, 0.7]
    C.MODEL.RPN.IOU_LABELS = [0, -1, 1]
    C.MODEL.RPN.BATCH_SIZE_PER_IMAGE = 256
    C.MODEL.RPN.POSITIVE_FRACTION = 0.5
    C.MODEL.RPN.BBOX_REG_LOSS_TYPE = "smooth_l1"
    C.MODEL.RPN.BBOX_REG_LOSS_WEIGHT = 1.0
    C.MODEL.RPN.BBOX_REG_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
    C.MODEL.RPN.SMOOTH_L1_BETA = 0.0
    C.MODEL.RPN.LOSS_WEIGHT = 1.0
    C.MODEL.RPN.PRE_NMS_TOPK_TRAIN = 12000
    C.MODEL.RPN.PRE_NMS_TOPK_TEST = 6000
    C.MODEL.RPN.POST_NMS_TOPK_TRAIN = 2000
    C.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    C.MODEL.RPN.NMS_THRESH = 0.7
    C.MODEL.ROI_HEADS = CN()
    C.MODEL.ROI_HEADS.NAME = "Res5ROIHeads"
    C.MODEL.ROI_HEADS.NUM_CLASSES = 80
    C.MODEL.ROI_HEADS.IN_FEATURES = ["res4"]
    C.MODEL.ROI_HEADS.IOU_THRESHOLDS = [0.5]
    C.MODEL.ROI_HEADS.IOU_LABELS = [0, 1]
    C.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 512
    C.MODEL.ROI_HEADS.POSITIVE_FRACTION = 0.25
    C.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.05
    C.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.5
    C.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT = True
    C.MODEL.ROI_BOX_HEAD = CN()
    C.MODEL.ROI_BOX_HEAD.NAME = ""
    C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE = "smooth_l1"
    C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT = 1.0
    C.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
    C.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA = 0.0
    C.MODEL.ROI_BOX_HEAD.FC_DIM = 1024  # 2-FC head of the FPN extension
    C.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 14
    C.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    C.MODEL.ROI_BOX_HEAD.POOLER_TYPE = "ROIAlignV2"
    C.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = False
    C.MODEL.RESNETS = CN({"DEPTH": 50, "OUT_FEATURES": ["res4"], "NORM": "FrozenBN", "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64,
                          "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64})
    C.SOLVER = CN()
    C.SOLVER.LR_SCHEDULER_NAME = "WarmupMultiStepLR"
    C.SOLVER.MAX_ITER = 40000
    C.SOLVER.BASE_LR = 0.001
    C.SOLVER.MOMENTUM = 0.9
    C.SOLVER.NESTEROV = False
    C.SOLVER.WEIGHT_DECAY = 0.0001
    C.SOLVER.WEIGHT_DECAY_NORM = 0.0
    C.SOLVER.GAMMA = 0.1
    C.SOLVER.STEPS = (30000,)
    C.SOLVER.WARMUP_FACTOR = 1.0 / 1000
    C.SOLVER.WARMUP_ITERS = 1000
    C.SOLVER.WARMUP_METHOD = "linear"
    C.SOLVER.CHECKPOINT_PERIOD = 5000
    C.SOLVER.IMS_PER_BATCH = 16
    C.SOLVER.REFERENCE_WORLD_SIZE = 0
    C.SOLVER.BIAS_LR_FACTOR = 1.0
    C.SOLVER.WEIGHT_DECAY_BIAS = C.SOLVER.WEIGHT_DECAY
    C.SOLVER.CLIP_GRADIENTS = CN({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0})
    C.SOLVER.AMP = CN({"ENABLED": False})
    C.TEST = CN()
    C.TEST.EXPECTED_RESULTS = []
    C.TEST.EVAL_PERIOD = 0
    C.TEST.DETECTIONS_PER_IMAGE = 100
    C.TEST.AUG = CN({"ENABLED": False})
    C.TEST.PRECISE_BN = CN({"ENABLED": False, "NUM_ITER": 200})
    C.OUTPUT_DIR = "./output"
    C.SEED = -1
    C.CUDNN_BENCHMARK = False
    C.VIS_PERIOD = 0
    return C


def add_config(cfg: CfgNode) -> None:
    """Keys of /root/reference/coin/config.py:17-143 (same names and defaults)."""
    _C = cfg
    _C.RESUME = False
    _C.SOLVER.IMG_PER_BATCH_UNLABEL = 3
    _C.SOLVER.FACTOR_LIST = (1,)
    _C.SOLVER.REFERENCE_WORLD_SIZE = 0
    _C.SOLVER.PER_MODULE_PARAM_WEIGHT = []
    _C.DATASETS.TRAIN_UNLABEL = ("",)
    _C.DATASETS.STYLE_NAME = ""
    _C.TEST.EVALUATOR = "VOCeval"
    _C.TEST.DETECTIONS_PER_IMAGE = 100
    _C.TEST.SAVE_DETECTION_PKLS = False
    _C.INPUT.TEACHER_CLOUD = CN()
    _C.INPUT.TEACHER_CLOUD.MIN_SIZE_TEST = 600
    _C.INPUT.TEACHER_CLOUD.MAX_SIZE_TEST = 1333
    _C.INPUT.TEACHER_CLOUD.FORMAT = "RGB"
    _C.INPUT.TEACHER_CLOUD.NORM = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    _C.INPUT.TEACHER_CLOUD.COLLECT_AUG = ""
    _C.INPUT.TEACHER_CLOUD.MIN_CENTER_ZOOM_SIZE = 320
    _C.INPUT.TEACHER_OFFLINE = CN()
    _C.INPUT.TEACHER_OFFLINE.PIXEL_MEAN = [0.48145466, 0.4578275, 0.40821073]
    _C.INPUT.TEACHER_OFFLINE.PIXEL_STD = [0.26862954, 0.26130258, 0.27577711]
    _C.INPUT.MIN_SIZE_TRAIN = (600,)
    _C.INPUT.MIN_SIZE_TEST = 600
    _C.MODEL.RESNETS.DEPTH = 50
    _C.MODEL.RESNETS.OUT_FEATURES = ["res4"]
    _C.MODEL.RESNETS.NORM = "FrozenBN"
    _C.MODEL.ROI_HEADS.TEACHER_OFFLINE = "CLIPRes5ROIHeads"
    _C.MODEL.TEACHER_CLOUD = CN()
    _C.MODEL.TEACHER_CLOUD.META_ARCHITECTURE = ""
    _C.MODEL.TEACHER_CLOUD.USE_DINO_TYPE_FILTER = False
    _C.MODEL.TEACHER_CLOUD.PROCESSOR_ARCHITECTURE = ""
    _C.MODEL.TEACHER_CLOUD.COLLECT_ARCHITECTURE = ""
    _C.MODEL.TEACHER_CLOUD.TYPE = ""
    _C.MODEL.TEACHER_CLOUD.CONFIG_PATH = ""
    _C.MODEL.TEACHER_CLOUD.WEIGHT = ""
    _C.MODEL.TEACHER_CLOUD.TEST_THRESHOLD = 0.25
    _C.MODEL.TEACHER_CLOUD.PER_CLASS_TEST = False
    _C.MODEL.TEACHER_CLOUD.TOKEN = ""
    _C.MODEL.TEACHER_OFFLINE = CN()
    _C.MODEL.TEACHER_OFFLINE.META_ARCHITECTURE = "CLIP"
    _C.MODEL.TEACHER_OFFLINE.COLLECT_ARCHITECTURE = "CLIP_COLLECTOR"
    _C.MODEL.TEACHER_OFFLINE.TYPE = ""
    _C.MODEL.TEACHER_OFFLINE.TEXT_ENCODER = "CLIP_TEXT"
    _C.MODEL.ROI_HEADS.POOLING_TYPE = "meanpool"
    _C.MODEL.MERGE = "CKGNet"
    _C.MODEL.MERGE_DIM = 1024
    _C.MODEL.REGION_CLIP = False
    _C.CLOUD = CN()
    _C.CLOUD.Trainer = ""
    _C.CLOUD.PRE_TRAIN_NAME = ""
    _C.CLOUD.BURN_UP_STEP = 45000
    _C.CLOUD.PROTOTYPE_UPDATE_START = 5000
    _C.CLOUD.OFFLINE_TEACHER_UPDATE_ITER = 1
    _C.CLOUD.EMA_KEEP_RATE_OFFLINE = 0.9996
    _C.CLOUD.UPDATE_BACKBONE = False
    _C.CLOUD.ADD_PROMPT_NUM = 4
    _C.CLOUD.CLS_B_THRESH = 0.7
    _C.CLOUD.PROTOTYPE_UPDATE_WEIGHT = 0.9996
    _C.CLOUD.NMS_METHOD = "ms"
    _C.CLOUD.LOSS_TYPE = "MILCrossEntropy"
    _C.CLOUD.BG_TRAIN = True
    _C.CLOUD.CLASSES_WEIGHT = []
    _C.CLOUD.LOSS_BOX_REG_WEIGHT = 1.0
    _C.CLOUD.LOSS_BOX_REG_OFFLINE_WEIGHT = 1.0
    _C.CLOUD.LOSS_BOX_REG_ONLINE_WEIGHT = 1.0
    _C.CLOUD.LOSS_CLS_WEIGHT = 1.0
    _C.CLOUD.LOSS_TEXT_ALIGN_WEIGHT = 10.0
    _C.CLOUD.LOSS_CLS_B_WEIGHT = 0.1
    _C.CLOUD.LOSS_DISTILLATION_WEIGHT = 0.1
    _C.CLOUD.TEACHER_CLOUD = CN()
    _C.CLOUD.TEACHER_CLOUD.RPN_SEPARATE_COLLECT = False
    _C.CLOUD.TEACHER_CLOUD.RPN_THRESH = 0.25
    _C.CLOUD.TEACHER_CLOUD.RCNN_THRESH = 0.25
    _C.CLOUD.TEACHER_CLOUD.ZOOM_MATCHER_THRESH = 0.6
    _C.CLOUD.TEACHER_CLOUD.COLLECT_NMS_THRESH = 0.6
    _C.CLOUD.MATCHER = CN()
    _C.CLOUD.MATCHER.IOU_THRESHOLDS = 0.5
    # ---- additions of this build (not in the reference): synthetic data + MI355X execution knobs
    _C.AMD = CN()
    _C.AMD.COMPUTE_DTYPE = "bf16"      # "bf16" (throughput) or "fp32" (1e-4 parity)
    _C.AMD.SYNTHETIC = CN({"ENABLED": False, "NUM_IMAGES": 2, "HEIGHT": 800, "WIDTH": 1333, "BOXES_PER_IMAGE": 32})
    _C.AMD.CLASS_NAMES = []            # thing classes when no dataset registry is available
    _C.AMD.TEXT_TEMPLATES = 81         # templates averaged into per_class_feat (clip_text.py:262-279)
    # architecture overrides for small-scale tests (0 / [] = take the CLIP architecture of MODEL.TEACHER_OFFLINE.TYPE)
    _C.AMD.ARCH = CN({"LAYERS": [], "WIDTH": 0, "TEXT_WIDTH": 0, "TEXT_LAYERS": 0, "TEXT_HEADS": 0, "TEXT_DIM": 0,
                      "CONTEXT_LENGTH": 77, "VOCAB_SIZE": 49408})
    _C.AMD.SYNC_FREE = True            # pre_train step without host<->device round trips (random-key sampling)
    _C.AMD.SYNC_FREE_STEP = True       # the same for the step_one / step_two branches of CoinTrainer (losses pinned to the goldens on CPU and GPU; measured faster and steadier: DESIGN.md section 7)
    _C.AMD.TEXT_GRAPH = True           # prompt-conditioned text transformer: forward + backward replayed from two captured HIP graphs
    _C.AMD.STEP_GRAPHS = True          # training forward + backward of the backbone's trainable stages and of RoIAlign -> res5 replayed as HIP graphs per shape (coin_amd/graphs.py)
    _C.AMD.TEACHER_GRAPH = True        # CoinTrainer: EMA-due iterations replay the teacher's fixed-shape inference half as one HIP graph (default stream)
    _C.AMD.TEACHER_PREFETCH = True     # CoinTrainer: while the teacher is frozen (no EMA due), enqueue the next iteration's teacher pass ahead of this step
    _C.AMD.TEACHER_STREAM = True       # CoinTrainer: teacher EMA / inference / detection read-back on their own HIP stream
    _C.AMD.FIXED_ROI_SAMPLER = False   # benchmark mode of SURVEY §8d: exactly 128 fg + 384 bg per view
    _C.AMD.CONV_GEMM = True            # bf16 mode: res5 convolutions forward / dgrad on coin_conv_gemm_bf16 (BN statistics in its epilogue)
    _C.AMD.GRAD_ARENA = False          # one GPU: also pack the gradients into the flat arena of coin_amd.parallel.GradReducer (measured 0.8 ms/step slower than handing autograd's tensors to the optimizer; the arena is always used when world_size > 1)
    _C.AMD.CONV_GEMM_WGRAD = True      # ... and their weight gradients on coin_conv_wgrad_bf16 (round 3: 1.3-1.7x the library's wrw kernels)


def get_cfg() -> CfgNode:
    cfg = _d2_defaults()
    add_config(cfg)
    return cfg
