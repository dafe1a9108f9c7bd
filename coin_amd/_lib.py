"""ctypes binding of ``libcoin_hip.so`` (the C ABI declared in ``include/coin_hip.h``).

The product path has NO fallback: if the shared object is missing or an entry point
returns non-zero, a :class:`CoinHipError` is raised.  PyTorch is used only as the owner of
device memory and streams: every call passes ``tensor.data_ptr()`` and the current HIP
stream handle.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libcoin_hip.so")
HEADER_PATH = os.path.abspath(os.path.join(_HERE, "..", "include", "coin_hip.h"))

COIN_F32, COIN_BF16 = 0, 1
COIN_NCHW, COIN_NHWC = 0, 1
ACT_NONE, ACT_LEAKY_RELU, ACT_RELU = 0, 1, 2

_ERRORS = {-1: "COIN_EINVAL (bad argument)", -2: "COIN_ESHAPE (unsupported shape)", -3: "COIN_EALIGN (misaligned pointer / ld)"}


class CoinHipError(RuntimeError):
    pass


class RoiLevel(ctypes.Structure):
    """coin_roi_level (include/coin_hip.h): one pyramid level of the multi-level pooler."""
    _fields_ = [("feat", c_void_p), ("H", c_int), ("W", c_int), ("spatial_scale", c_float)]


class WdTensor(ctypes.Structure):
    """coin_wd_tensor (include/coin_hip.h): one weight of the one-launch data-gradient re-layout."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("cout", ctypes.c_int32), ("cin", ctypes.c_int32), ("ks", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class SgdTensor(ctypes.Structure):
    _fields_ = [
        ("param", c_void_p), ("grad", c_void_p), ("momentum_buf", c_void_p), ("bf16_shadow", c_void_p),
        ("numel", c_int64), ("lr", c_float), ("weight_decay", c_float),
    ]


class EmaTensor(ctypes.Structure):
    _fields_ = [("teacher", c_void_p), ("student", c_void_p), ("numel", c_int64)]


_P, _I, _F, _L, _Z = c_void_p, c_int, c_float, c_int64, ctypes.c_size_t
# name -> argtypes ; every entry point returns int.  Mirrors include/coin_hip.h one to one
# (tests/test_abi.py parses the header and checks this table against it).
SIGNATURES = {
    "coin_roi_align_fwd": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _I, _P, _I, _P],
    "coin_roi_align_bwd": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _I, _P, _I, _P],
    "coin_roi_align_fwd_levels": [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    "coin_roi_align_bwd_level": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _I, _P],
    "coin_gemm_nt": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _F, _I, _I, _P],
    "coin_conv_gemm_bf16": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P],
    "coin_conv_gemm_bf16_ws": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P, _Z, _P],
    "coin_conv_gemm_bf16_rpool": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P],
    "coin_conv_gemm_stats_finalize": [_P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "coin_conv_gemm_stats_tile_rows": [_I, _I, _I, _I, _I, _I, _I],
    "coin_conv_wgrad_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "coin_window_attn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "coin_window_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "coin_anchor_match": [_P, _P, _I, _P, _I, _I, _F, _F, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "coin_sample_labels": [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P],
    "coin_aug_resize_bilinear_u8": [_P, _I, _I, _P, _I, _I, _I, _P, _P],
    "coin_aug_point_op_u8": [_P, _P, _I, _I, _I, _F, _I, _P, _I, _P],
    "coin_aug_gaussian_blur_u8": [_P, _P, _I, _I, _F, _P, _P],
    "coin_transpose2d": [_P, _P, _I, _I, _I, _P],
    "coin_bias_act_bwd": [_P, _P, _P, _I, _I, _I, _P, _I, _F, _I, _P, _P],
    "coin_cosine_logits_fwd": [_P, _I, _P, _I, _I, _I, _F, _P, _P, _I, _P],
    "coin_cosine_logits_bwd": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _I, _P, _P],
    "coin_mil_ce_fwd_bwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "coin_mil_focal_fwd_bwd": [_P, _I, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, _P, _P],
    "coin_kl_div_fwd_bwd": [_P, _I, _P, _I, _P, _I, _I, _I, _F, _P, _P, _P],
    "coin_box_reg_l1_fwd_bwd": [_P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P],
    "coin_l1_mean_fwd_bwd": [_P, _P, _L, _P, _P, _P],
    "coin_rpn_losses_fwd_bwd": [_P, _P, _P, _P, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P],
    "coin_normalize_pad": [_P, _I, _I, ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P, _I, _I, _I, _I, _I, _P],
    "coin_bn_stats": [_P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P],
    "coin_bn_apply_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "coin_bn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "coin_avgpool2_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "coin_avgpool2_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "coin_nms_batched": [_P, _P, _I, _I, _F, _I, _P, _P, _P, _P],
    "coin_sgd_step": [_P, _I, _L, _F, _F, _F, _I, _P, _P],
    "coin_weight_dgrad_layout": [_P, _I, _I, _P],
    "coin_ema_update": [_P, _I, _L, _F, _P],
}

ABI_VERSION = 3   # == COIN_ABI_VERSION of include/coin_hip.h (tests/test_abi.py); lib() refuses any other library

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (``make -C coin_amd/csrc``)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise CoinHipError(f"building libcoin_hip.so failed (exit {res.returncode})")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the shared object (once).  Raises loudly if it is absent: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch owns the device memory and streams these kernels run on: make sure ITS HIP runtime (the libamdhip64 bundled
    # with the wheel) is the one already mapped before libcoin_hip.so resolves its libamdhip64 dependency, otherwise a
    # second runtime instance from /opt/rocm is loaded and every launch fails with hipErrorNoDevice.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise CoinHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C coin_amd/csrc`). coin_amd has no fallback path without its HIP kernels."
        )
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise CoinHipError(f"cannot load {LIB_PATH}: {e}") from e
    # first of all: a stale build lacks the symbols newer ABI versions added (a bare AttributeError from ctypes) and would be passed
    # shifted arguments by the prototypes that changed
    l.coin_abi_version.restype = c_int
    if l.coin_abi_version() != ABI_VERSION:
        raise CoinHipError(f"{LIB_PATH} implements C-ABI version {l.coin_abi_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild it (`make -C coin_amd/csrc`)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    l.coin_nms_workspace_bytes.argtypes = [c_int, c_int]
    l.coin_nms_workspace_bytes.restype = ctypes.c_size_t
    l.coin_conv_gemm_stats_bytes.argtypes = [c_int, c_int]
    l.coin_conv_gemm_stats_bytes.restype = ctypes.c_size_t
    l.coin_conv_wgrad_workspace_bytes.argtypes = [c_int, c_int, c_int]
    l.coin_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    l.coin_conv_gemm_workspace_bytes.argtypes = [c_int, c_int, c_int]
    l.coin_conv_gemm_workspace_bytes.restype = ctypes.c_size_t
    l.coin_window_attn_bwd_workspace_bytes.argtypes = [c_int, c_int]
    l.coin_window_attn_bwd_workspace_bytes.restype = ctypes.c_size_t
    l.coin_clear_last_error.restype = c_int
    l.coin_build_arch.restype = c_char_p
    _lib = l
    return l


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise CoinHipError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise CoinHipError(f"{what}: HIP launch failed with hipError_t {rc}")
