"""Tensor-level wrappers over the C ABI (``include/coin_hip.h``).

Each function checks device / dtype / contiguity, passes raw device pointers and the current
HIP stream to ``libcoin_hip.so`` and raises :class:`coin_amd._lib.CoinHipError` on failure.
Nothing here computes on the CPU and nothing falls back to torch ops.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, COIN_BF16, COIN_F32, COIN_NCHW, COIN_NHWC, CoinHipError, check

__all__ = [
    "roi_align_fwd", "roi_align_bwd", "gemm_nt", "conv_gemm", "conv_wgrad", "window_attn_fwd", "window_attn_bwd", "roi_align_fwd_levels", "roi_align_bwd_level", "conv_stats_finalize", "transpose2d", "bias_act_bwd", "cosine_logits_fwd",
    "cosine_logits_bwd", "bn_stats", "bn_apply_fwd", "bn_bwd", "avgpool2_fwd", "avgpool2_bwd", "nms_batched", "mil_ce", "mil_focal", "kl_div", "box_reg_l1", "l1_mean", "rpn_losses", "normalize_pad",
    "SgdTable", "EmaTable",
]


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return COIN_F32
    if t.dtype == torch.bfloat16:
        return COIN_BF16
    raise CoinHipError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _dev(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise CoinHipError("coin_amd kernels need device tensors (got a CPU tensor); there is no CPU path")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


try:  # `torch.cuda.current_stream().cuda_stream` builds two Python objects per kernel launch (10 us; ~900 launches per targetDET step)
    _raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:  # pragma: no cover
    _raw_stream = _cur_dev = None


def _stream():
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_cur_dev()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-entry-point device timing (HIP events on the launch stream); used by bench.py's roofline line
_TIMING = None


def timing_begin(names: Sequence[str]) -> None:
    global _TIMING, _PAUSED
    _TIMING = {n: [] for n in names}
    _PAUSED = False


def timing_pause() -> None:
    """Stop recording (what was recorded stays for timing_end): a long run must not keep creating two events per launch."""
    global _PAUSED
    _PAUSED = True


_PAUSED = False


def timing_active() -> bool:
    """HIP events are being recorded around the timed entry points (bench.py's first timed steps): the step graphs stay off meanwhile,
    a replayed graph launches none of them from the host."""
    return _TIMING is not None and not _PAUSED


SHAPES: dict = {}   # filled by timing_end: {entry point: {shape tag: [launches, total ms, algorithmic units per launch]}} for the tagged calls


def timing_end() -> dict:
    """-> {name: (launches, mean_ms, algorithmic_bytes_per_launch)}; call after a device synchronize.  The per-shape breakdown of the
    entry points whose calls carry a tag (the GEMM convolutions: (M, N, K, kernel size)) is left in `SHAPES`."""
    global _TIMING
    out = {}
    SHAPES.clear()
    for n, evs in (_TIMING or {}).items():
        if evs:
            ms = [ev[0].elapsed_time(ev[1]) for ev in evs]
            out[n] = (len(ms), sum(ms) / len(ms), sum(ev[2] for ev in evs) / len(evs))
            for t, ev in zip(ms, evs):
                if ev[3] is not None:
                    rec = SHAPES.setdefault(n, {}).setdefault(ev[3], [0, 0.0, ev[2]])
                    rec[0] += 1
                    rec[1] += t
    _TIMING = None
    return out


class _timed:
    def __init__(self, name: str, alg_bytes: int = 0, tag=None):
        self.on = _TIMING is not None and not _PAUSED and name in _TIMING
        self.name, self.bytes, self.tag = name, alg_bytes, tag

    def __enter__(self):
        if self.on:
            self.s, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.s.record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if self.on:
            self.e.record(torch.cuda.current_stream())
            _TIMING[self.name].append((self.s, self.e, self.bytes, self.tag))
        return False


def _nbytes(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.numel() * t.element_size()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise CoinHipError(f"{name} must be contiguous float32")
    return t


# --------------------------------------------------------------------------- RoIAlign
def _feat_dims(feat: torch.Tensor, layout: int) -> Tuple[int, int, int, int]:
    if layout == COIN_NCHW:
        n, c, h, w = feat.shape
    else:
        n, h, w, c = feat.shape
    return n, c, h, w


def roi_align_fwd(feat: torch.Tensor, rois: torch.Tensor, output_size: Tuple[int, int], spatial_scale: float,
                  sampling_ratio: int = 0, aligned: bool = True, layout: int = COIN_NHWC) -> torch.Tensor:
    """feat: contiguous [N,C,H,W] (NCHW) or [N,H,W,C] (NHWC); rois [R,5] f32 -> [R,C,ph,pw] / [R,ph,pw,C]."""
    _dev(feat, rois)
    if not feat.is_contiguous():
        raise CoinHipError("feat must be contiguous in the declared layout")
    rois = _f32c(rois, "rois")
    n, c, h, w = _feat_dims(feat, layout)
    ph, pw = output_size
    r = rois.shape[0]
    shape = (r, c, ph, pw) if layout == COIN_NCHW else (r, ph, pw, c)
    out = torch.empty(shape, dtype=feat.dtype, device=feat.device)
    # algorithmic bytes: read the map once + rois + write the pooled tiles
    with _timed("coin_roi_align_fwd", feat.numel() * feat.element_size() + rois.numel() * 4 + out.numel() * out.element_size()):
        check(_lib.lib().coin_roi_align_fwd(_p(feat), n, c, h, w, layout, _p(rois), r, ph, pw, float(spatial_scale),
                                            int(sampling_ratio), int(aligned), _p(out), _dt(feat), _stream()),
              "coin_roi_align_fwd")
    return out


def roi_align_bwd(grad_out: torch.Tensor, rois: torch.Tensor, feat_shape: Sequence[int], spatial_scale: float,
                  sampling_ratio: int = 0, aligned: bool = True, layout: int = COIN_NHWC,
                  grad_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns the float32 gradient map of shape ``feat_shape`` (`grad_feat`, if given, is overwritten)."""
    _dev(grad_out, rois, grad_feat)
    if not grad_out.is_contiguous():
        raise CoinHipError("grad_out must be contiguous")
    rois = _f32c(rois, "rois")
    if grad_feat is None:
        grad_feat = torch.empty(tuple(feat_shape), dtype=torch.float32, device=grad_out.device)
    else:
        _f32c(grad_feat, "grad_feat")
    if layout == COIN_NCHW:
        n, c, h, w = feat_shape
        ph, pw = grad_out.shape[2:]
    else:
        n, h, w, c = feat_shape
        ph, pw = grad_out.shape[1:3]
    r = rois.shape[0]
    with _timed("coin_roi_align_bwd", grad_out.numel() * grad_out.element_size() + rois.numel() * 4 + grad_feat.numel() * 4):
        check(_lib.lib().coin_roi_align_bwd(_p(grad_out), n, c, h, w, layout, _p(rois), r, ph, pw, float(spatial_scale),
                                            int(sampling_ratio), int(aligned), _p(grad_feat), _dt(grad_out), _stream()),
              "coin_roi_align_bwd")
    return grad_feat


# --------------------------------------------------------------------------- box head
def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
            act_alpha: float = 0.01, out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None
            ) -> torch.Tensor:
    """C[M,N] = act(A[M,K] @ B[N,K]^T + bias).  A, B row-major (last dim contiguous), same dtype."""
    _dev(a, b, bias, out)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise CoinHipError(f"gemm_nt shape mismatch {tuple(a.shape)} x {tuple(b.shape)}^T")
    if a.dtype != b.dtype or a.stride(1) != 1 or b.stride(1) != 1:
        raise CoinHipError("gemm_nt operands must share dtype and be K-contiguous")
    m, k = a.shape
    n = b.shape[0]
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif out.stride(1) != 1 or out.shape != (m, n):
        raise CoinHipError("gemm_nt: bad `out`")
    if bias is not None:
        _f32c(bias, "bias")
    with _timed("coin_gemm_nt", 2 * m * n * k):  # "bytes" slot carries FLOPs for the MFMA-bound entry point
        check(_lib.lib().coin_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), m, n, k, _p(bias),
                                      act, float(act_alpha), _dt(a), _dt(out), _stream()), "coin_gemm_nt")
    return out


def conv_gemm(a: torch.Tensor, w: torch.Tensor, spatial: Optional[Tuple[int, int, int]] = None, stats_rows: Optional[int] = None,
              out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, residual_pool: Optional[Tuple[int, int]] = None):
    """bf16 GEMM / implicit-GEMM convolution (coin_conv_gemm_bf16).
    a: [M, K] row-major (1x1 convolution on NHWC rows, nn.Linear) or, with ``spatial=(H, W, Cin)``, the NHWC activation flattened
    to [M, Cin] for the implicit 3x3 / pad 1 convolution (M = NB*H*W);  w: [N, K] with K = Cin or 9*Cin (ky, kx, ci).
    -> C [M, N] bf16, and -- when ``stats_rows`` is given -- the per-row-tile statistics partials of the stored outputs over the
    first ``stats_rows`` rows (input of `conv_stats_finalize`).  ``residual`` [M, N] bf16 is added to the (bf16-rounded) product in
    the epilogue.  ``residual_pool=(out_h, out_w)``: `residual` is instead the gradient of a 2x2 average pool of the output grid
    ([M / (out_h out_w) * (out_h // 2) * (out_w // 2), N]) and every output pixel adds a quarter of its pooled pixel
    (coin_conv_gemm_bf16_rpool) -> (C, None), or None when the persistent kernel does not serve the shape (the caller materialises
    the pool gradient instead)."""
    _dev(a, w, out, residual)
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or a.dim() != 2 or w.dim() != 2 or a.stride(1) != 1 or w.stride(1) != 1:
        raise CoinHipError("conv_gemm needs 2-D K-contiguous bf16 operands")
    m, n, k = a.shape[0], w.shape[0], w.shape[1]
    if spatial is None:
        if a.shape[1] != k:
            raise CoinHipError(f"conv_gemm shape mismatch {tuple(a.shape)} x {tuple(w.shape)}^T")
        mode, h, wd, cin = 0, 0, 0, 0
    else:
        h, wd, cin = spatial
        if a.shape[1] != cin or k != 9 * cin or not a.is_contiguous() or m % (h * wd):
            raise CoinHipError("conv_gemm (3x3): a must be the contiguous [NB*H*W, Cin] activation and w [Cout, 9*Cin]")
        mode = 1
    if out is None:
        out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    elif out.shape != (m, n) or out.stride(1) != 1 or out.dtype != torch.bfloat16:
        raise CoinHipError("conv_gemm: bad `out`")
    if residual_pool is not None:
        oh, ow = residual_pool
        if residual is None or stats_rows is not None or oh < 2 or ow < 2 or m % (oh * ow):
            raise CoinHipError("conv_gemm: residual_pool needs a residual, no statistics and M a multiple of out_h * out_w")
        if residual.shape != (m // (oh * ow) * (oh // 2) * (ow // 2), n) or residual.stride(1) != 1 or residual.dtype != torch.bfloat16:
            raise CoinHipError("conv_gemm: the pooled `residual` must be a bf16 [M / (h w) * (h // 2) * (w // 2), N] matrix")
    elif residual is not None and (residual.shape != (m, n) or residual.stride(1) != 1 or residual.dtype != torch.bfloat16):
        raise CoinHipError("conv_gemm: `residual` must be a bf16 [M, N] matrix with contiguous rows")
    part = None
    if stats_rows is not None:
        # row tiles of 256 (the persistent core) or 128 rows (the small-map core): the partials tensor is sized for the kernel that
        # will run -- the query IS the library's dispatch function -- and `conv_stats_finalize` reads the tile height off its length
        tr = _lib.lib().coin_conv_gemm_stats_tile_rows(a.stride(0), mode, cin, w.stride(0), m, n, k)
        part = torch.empty(((m + tr - 1) // tr) * 3 * n, dtype=torch.float32, device=a.device)
    wsb = _lib.lib().coin_conv_gemm_workspace_bytes(m, n, k)   # fp32 partial tiles of the split-K tail round (0: not needed for this shape)
    ws = None
    if wsb:
        # one buffer per (device, stream), grown to the largest request: calls on one stream are ordered, but the EMA teacher's pass
        # runs on its own stream beside the student's step and must not share the slabs with it
        key = (a.device, _stream().value)
        ws = _GEMM_WS.get(key)
        if ws is None or ws.numel() < wsb:
            ws = _GEMM_WS[key] = torch.empty(wsb, dtype=torch.uint8, device=a.device)
    if residual_pool is not None:
        shape_key = (m, n, k, mode, a.stride(0), w.stride(0))
        if shape_key in _RPOOL_UNSERVED:
            return None
        with _timed("coin_conv_gemm_bf16", 2 * m * n * k, (m, n, k, 3 if mode else 1)):
            rc = _lib.lib().coin_conv_gemm_bf16_rpool(_p(a), a.stride(0), mode, h, wd, cin, _p(w), w.stride(0), _p(out), out.stride(0), _p(residual),
                                                      residual.stride(0), int(residual_pool[0]), int(residual_pool[1]), m, n, k, _p(ws),
                                                      wsb if ws is not None else 0, _stream())
        if rc == -2:   # COIN_ESHAPE: not a shape of the persistent kernel (remembered: the failed call launched nothing)
            _RPOOL_UNSERVED.add(shape_key)
            return None
        check(rc, "coin_conv_gemm_bf16_rpool")
        return out, None
    with _timed("coin_conv_gemm_bf16", 2 * m * n * k, (m, n, k, 3 if mode else 1)):  # "bytes" slot carries FLOPs for the MFMA entry point
        check(_lib.lib().coin_conv_gemm_bf16_ws(_p(a), a.stride(0), mode, h, wd, cin, _p(w), w.stride(0), _p(out), out.stride(0),
                                                _p(residual), residual.stride(0) if residual is not None else 0, m, n, k,
                                                _p(part), int(stats_rows or 0), _p(ws), wsb if ws is not None else 0, _stream()), "coin_conv_gemm_bf16_ws")
    return out, part


_GEMM_WS: dict = {}
_RPOOL_UNSERVED: set = set()


def capture_stream_value() -> int:
    """Raw handle of the stream torch.cuda.graph captures on (created here if no capture has happened yet)."""
    if torch.cuda.graph.default_capture_stream is None:
        from . import streams

        torch.cuda.graph.default_capture_stream = streams.role_stream(torch.cuda.current_device(), "capture")
    return int(torch.cuda.graph.default_capture_stream.cuda_stream)


def take_stream_workspaces(stream_value: int) -> list:
    """Remove and return the cached workspaces of one stream (the GEMM split-K slabs, the weight-gradient slabs, the window-attention
    scratch).  A HIP graph must OWN the workspaces its kernels were recorded with: the caches are keyed by stream and every capture runs
    on the same capture stream, so a later capture that needed a larger buffer used to replace -- and thereby free -- the one an earlier
    graph's kernels still point at; when that buffer lived in the pool of a graph that no longer existed, the next empty_cache() unmapped
    it and the earlier graph's replay ended in a GPU memory access fault (round 5, full test suite).  coin_amd.graphs and the teacher's
    capture call this before a capture (nothing inherited) and after it (the entries move into the graph's own record)."""
    out = []
    for cache in (_GEMM_WS, _WGRAD_WS, _WATTN_WS):
        for k in [k for k in cache if int(k[1] or 0) == int(stream_value)]:
            out.append(cache.pop(k))
    return out


_WGRAD_WS: dict = {}


def conv_wgrad_ok(cout: int, cin: int, m: Optional[int] = None) -> bool:
    """Channel counts coin_conv_wgrad_bf16 serves: multiples of 256, or -- on the persistent kernel, whose 32-bit buffer offsets bound
    the pixel count `m` -- odd multiples of 128 (half-valid edge tiles)."""
    if cout % 128 or cin % 128:
        return False
    if cout % 256 == 0 and cin % 256 == 0:
        return True
    return m is None or (m + 320) * max(cout, cin) * 2 < 0x7F000000


def conv_wgrad(gy: torch.Tensor, x: torch.Tensor, spatial: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
    """dW [Cout, Ktot] fp32 = gy[M, Cout]^T . Acol[M, Ktot] (coin_conv_wgrad_bf16); `spatial=(H, W, Cin)` selects the implicit 3x3 /
    pad 1 form (Ktot = 9*Cin, (ky, kx, ci) order), else the 1x1 form (x is [M, Cin])."""
    _dev(gy, x)
    if gy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not gy.is_contiguous() or not x.is_contiguous() or gy.shape[0] != x.shape[0]:
        raise CoinHipError("conv_wgrad needs contiguous bf16 [M, Cout] / [M, Cin] operands")
    m, cout, cin = gy.shape[0], gy.shape[1], x.shape[1]
    mode, h, w = (0, 0, 0) if spatial is None else (1, spatial[0], spatial[1])
    ktot = cin if spatial is None else 9 * cin
    nbytes = _lib.lib().coin_conv_wgrad_workspace_bytes(m, cout, ktot)
    if nbytes == 0:
        raise CoinHipError("conv_wgrad: Cout and Cin must be multiples of 128")
    key = (gy.device, _stream().value, nbytes)
    ws = _WGRAD_WS.get(key)  # one slab buffer per (stream, size), reused (calls on one stream are ordered)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=gy.device)
    dw = torch.empty((cout, ktot), dtype=torch.float32, device=gy.device)
    with _timed("coin_conv_wgrad_bf16", 2 * m * cout * ktot, (m, cout, ktot, 3 if mode else 1)):
        check(_lib.lib().coin_conv_wgrad_bf16(_p(gy), _p(x), mode, h, w, cin, m, cout, ktot, _p(dw), _p(ws), _stream()), "coin_conv_wgrad_bf16")
    return dw


def window_attn_fwd(qkv: torch.Tensor, bias64: torch.Tensor, mask64: Optional[torch.Tensor], heads: int, scale: float) -> torch.Tensor:
    """Window attention (coin_window_attn_fwd).  qkv bf16 [B, T, 3*heads*32] (layout [3][heads][32] along the last axis); bias64 fp32
    [heads, 64, 64] and mask64 fp32 [nW, 64, 64] padded as the header says -> bf16 [B, T, heads*32]."""
    _dev(qkv, bias64, mask64)
    b, t, c3 = qkv.shape
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous() or c3 != 3 * heads * 32:
        raise CoinHipError("window_attn_fwd needs a contiguous bf16 [B, T, 3*heads*32] tensor")
    _f32c(bias64, "bias64")
    if bias64.shape != (heads, 64, 64) or (mask64 is not None and (mask64.dim() != 3 or mask64.shape[1:] != (64, 64) or b % mask64.shape[0])):
        raise CoinHipError("window_attn_fwd: bias must be [heads, 64, 64], mask [nW, 64, 64] with B % nW == 0")
    if mask64 is not None:
        _f32c(mask64, "mask64")
    out = torch.empty((b, t, heads * 32), dtype=torch.bfloat16, device=qkv.device)
    check(_lib.lib().coin_window_attn_fwd(_p(qkv), _p(bias64), _p(mask64), _p(out), b, mask64.shape[0] if mask64 is not None else 1, heads, t, 32,
                                          float(scale), _stream()), "coin_window_attn_fwd")
    return out


_WATTN_WS: dict = {}


def window_attn_bwd(qkv: torch.Tensor, bias64: torch.Tensor, mask64: Optional[torch.Tensor], dout: torch.Tensor, heads: int, scale: float):
    """Backward of `window_attn_fwd` (coin_window_attn_bwd) -> (dqkv bf16 like qkv, dbias fp32 [heads, T, T])."""
    _dev(qkv, bias64, mask64, dout)
    b, t, c3 = qkv.shape
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous() or c3 != 3 * heads * 32:
        raise CoinHipError("window_attn_bwd needs a contiguous bf16 [B, T, 3*heads*32] tensor")
    if dout.dtype != torch.bfloat16 or not dout.is_contiguous() or tuple(dout.shape) != (b, t, heads * 32):
        raise CoinHipError("window_attn_bwd: dout must be a contiguous bf16 [B, T, heads*32] tensor")
    _f32c(bias64, "bias64")
    if bias64.shape != (heads, 64, 64) or (mask64 is not None and (mask64.dim() != 3 or mask64.shape[1:] != (64, 64) or b % mask64.shape[0])):
        raise CoinHipError("window_attn_bwd: bias must be [heads, 64, 64], mask [nW, 64, 64] with B % nW == 0")
    if mask64 is not None:
        _f32c(mask64, "mask64")
    nbytes = _lib.lib().coin_window_attn_bwd_workspace_bytes(b, heads)
    key = (str(qkv.device), torch.cuda.current_stream(qkv.device).cuda_stream if qkv.is_cuda else 0)
    ws = _WATTN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WATTN_WS[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=qkv.device)
    dqkv = torch.empty_like(qkv)
    dbias = torch.empty((heads, t, t), dtype=torch.float32, device=qkv.device)
    check(_lib.lib().coin_window_attn_bwd(_p(qkv), _p(bias64), _p(mask64), _p(dout), _p(dqkv), _p(dbias), _p(ws), b,
                                          mask64.shape[0] if mask64 is not None else 1, heads, t, 32, float(scale), _stream()), "coin_window_attn_bwd")
    return dqkv, dbias


def roi_align_fwd_levels(feats: Sequence[torch.Tensor], scales: Sequence[float], rois: torch.Tensor, levels: torch.Tensor, output_size: Tuple[int, int],
                         sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    """Multi-level RoIAlign in ONE launch (coin_roi_align_fwd_levels): feats = contiguous NHWC maps [N, H_l, W_l, C] of one dtype, RoI r
    is pooled from feats[levels[r]] (int32, device) -> [R, ph, pw, C]."""
    _dev(*feats, rois, levels)
    if not 1 <= len(feats) <= 4 or len(scales) != len(feats):
        raise CoinHipError("roi_align_fwd_levels: 1..4 pyramid levels with one scale each")
    n, _, _, c = feats[0].shape
    for f in feats:
        if not f.is_contiguous() or f.dim() != 4 or f.shape[0] != n or f.shape[3] != c or f.dtype != feats[0].dtype:
            raise CoinHipError("roi_align_fwd_levels: every level must be a contiguous [N, H, W, C] map of the same N, C and dtype")
    rois = _f32c(rois, "rois")
    if levels.dtype != torch.int32 or not levels.is_contiguous() or levels.numel() != rois.shape[0]:
        raise CoinHipError("roi_align_fwd_levels: levels must be a contiguous int32 [R] tensor")
    ph, pw = output_size
    r = rois.shape[0]
    out = torch.empty((r, ph, pw, c), dtype=feats[0].dtype, device=feats[0].device)
    table = (_lib.RoiLevel * len(feats))()
    for i, (f, s) in enumerate(zip(feats, scales)):
        table[i].feat, table[i].H, table[i].W, table[i].spatial_scale = f.data_ptr(), f.shape[1], f.shape[2], float(s)
    check(_lib.lib().coin_roi_align_fwd_levels(ctypes.cast(table, ctypes.c_void_p), len(feats), n, c, _p(rois), _p(levels), r, ph, pw,
                                               int(sampling_ratio), int(aligned), _p(out), _dt(feats[0]), _stream()), "coin_roi_align_fwd_levels")
    return out


def roi_align_bwd_level(grad_out: torch.Tensor, rois: torch.Tensor, levels: torch.Tensor, level: int, feat_shape: Sequence[int], spatial_scale: float,
                        sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    """float32 gradient map [N, H, W, C] of pyramid level `level`: the RoIs with levels[r] == level only (coin_roi_align_bwd_level)."""
    _dev(grad_out, rois, levels)
    if not grad_out.is_contiguous():
        raise CoinHipError("grad_out must be contiguous")
    rois = _f32c(rois, "rois")
    n, h, w, c = feat_shape
    r, ph, pw, _ = grad_out.shape
    g = torch.empty((n, h, w, c), dtype=torch.float32, device=grad_out.device)
    check(_lib.lib().coin_roi_align_bwd_level(_p(grad_out), n, c, h, w, _p(rois), _p(levels), int(level), r, ph, pw, float(spatial_scale),
                                              int(sampling_ratio), int(aligned), _p(g), _dt(grad_out), _stream()), "coin_roi_align_bwd_level")
    return g


def _nbt(t: Optional[torch.Tensor]):
    if t is not None and (t.dtype != torch.int64 or t.numel() != 1):
        raise CoinHipError("num_batches_tracked must be a one-element int64 tensor")
    return t


def conv_stats_finalize(part: torch.Tensor, m: int, n: int, rows: int, eps: float, momentum: float,
                        running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
                        num_batches_tracked: Optional[torch.Tensor] = None):
    """Per-channel batch mean / rstd (+ in-place running statistics, + the module's batch counter) from `conv_gemm`'s partials
    ([row tiles, 3, n] floats; the tile height -- 128 or 256 rows -- follows from the tensor's length, see `conv_gemm`)."""
    _dev(part, running_mean, running_var, _nbt(num_batches_tracked))
    tiles = part.numel() // (3 * n)
    if tiles * 3 * n != part.numel() or tiles not in ((m + 127) // 128, (m + 255) // 256):
        raise CoinHipError(f"conv_stats_finalize: {part.numel()} floats are not the partials of a [{m}, {n}] output")
    tile_rows = 256 if tiles == (m + 255) // 256 else 128   # (m <= 128: one tile either way)
    mean = torch.empty(n, dtype=torch.float32, device=part.device)
    rstd = torch.empty(n, dtype=torch.float32, device=part.device)
    check(_lib.lib().coin_conv_gemm_stats_finalize(_p(part), m, n, int(rows), tile_rows, float(eps), float(momentum), _p(mean), _p(rstd), _p(running_mean),
                                                   _p(running_var), _p(num_batches_tracked), _stream()), "coin_conv_gemm_stats_finalize")
    return mean, rstd


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise CoinHipError("transpose2d needs a contiguous 2-D tensor")
    m, n = x.shape
    out = torch.empty((n, m), dtype=x.dtype, device=x.device)
    check(_lib.lib().coin_transpose2d(_p(x), _p(out), m, n, _dt(x), _stream()), "coin_transpose2d")
    return out


def bias_act_bwd(dc: torch.Tensor, c: Optional[torch.Tensor], act: int, act_alpha: float = 0.01,
                 dbias: Optional[torch.Tensor] = None, want_dz: bool = True):
    """Returns (dZ, dbias) for C = act(Z + bias)."""
    _dev(dc, c, dbias)
    if dc.dim() != 2 or dc.stride(1) != 1:
        raise CoinHipError("bias_act_bwd needs a row-major 2-D tensor")
    m, n = dc.shape
    if c is not None and (c.shape != dc.shape or c.stride() != dc.stride() or c.dtype != dc.dtype):
        raise CoinHipError("bias_act_bwd: C must match dC")
    dz = torch.empty_strided(dc.shape, dc.stride(), dtype=dc.dtype, device=dc.device) if want_dz else None
    if dbias is None:
        dbias = torch.zeros(n, dtype=torch.float32, device=dc.device)
    ws = torch.empty(((m + 255) // 256) * n, dtype=torch.float32, device=dc.device)   # per-256-row-block column sums (joined in block order)
    check(_lib.lib().coin_bias_act_bwd(_p(dc), _p(c), _p(dz), dc.stride(0), m, n, _p(dbias), act, float(act_alpha),
                                       _dt(dc), _p(ws), _stream()), "coin_bias_act_bwd")
    return dz, dbias


def cosine_logits_fwd(feats: torch.Tensor, text: torch.Tensor, inv_scale: float):
    _dev(feats, text)
    text = _f32c(text, "text")
    if feats.dim() != 2 or feats.stride(1) != 1 or feats.shape[1] != text.shape[1]:
        raise CoinHipError("cosine_logits: feats [R,D] row-major, text [Kc,D]")
    r, d = feats.shape
    kc = text.shape[0]
    scores = torch.empty((r, kc), dtype=torch.float32, device=feats.device)
    inv_norm = torch.empty((r,), dtype=torch.float32, device=feats.device)
    check(_lib.lib().coin_cosine_logits_fwd(_p(feats), feats.stride(0), _p(text), r, d, kc, float(inv_scale), _p(scores),
                                            _p(inv_norm), _dt(feats), _stream()), "coin_cosine_logits_fwd")
    return scores, inv_norm


def cosine_logits_bwd(d_scores: torch.Tensor, feats: torch.Tensor, text: torch.Tensor, scores: torch.Tensor,
                      inv_norm: torch.Tensor, inv_scale: float, need_text_grad: bool = True):
    _dev(d_scores, feats, text, scores, inv_norm)
    d_scores = _f32c(d_scores.contiguous(), "d_scores")
    r, d = feats.shape
    kc = text.shape[0]
    d_feats = torch.empty_strided(feats.shape, feats.stride(), dtype=feats.dtype, device=feats.device)
    d_text = torch.zeros_like(text) if need_text_grad else None
    ws = torch.empty(((r + 15) // 16) * kc * d, dtype=torch.float32, device=feats.device) if need_text_grad else None   # per-row-block partials
    check(_lib.lib().coin_cosine_logits_bwd(_p(d_scores), _p(feats), feats.stride(0), _p(text), _p(scores), _p(inv_norm),
                                            r, d, kc, float(inv_scale), _p(d_feats), _p(d_text), _dt(feats), _p(ws), _stream()),
          "coin_cosine_logits_bwd")
    return d_feats, d_text


# --------------------------------------------------------------------------- losses
def _scalar(device) -> torch.Tensor:
    return torch.empty((), dtype=torch.float32, device=device)


def mil_ce(x: torch.Tensor, target: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
           weights: Optional[torch.Tensor] = None, avg_positives: bool = False, reduction: str = "mean",
           want_grad: bool = True):
    _dev(x, target, labels, weights)
    x = _f32c(x, "x")
    r, c = x.shape
    if target is not None:
        target = _f32c(target, "target")
    if labels is not None and (labels.dtype != torch.int64 or not labels.is_contiguous()):
        raise CoinHipError("labels must be contiguous int64")
    if weights is not None:
        weights = _f32c(weights, "weights")
    loss = _scalar(x.device)
    grad = torch.empty_like(x) if want_grad else None
    check(_lib.lib().coin_mil_ce_fwd_bwd(_p(x), x.stride(0) if r else c, _p(target), _p(labels), _p(weights), r, c,
                                         int(avg_positives), int(reduction == "mean"), _p(loss), _p(grad), _stream()),
          "coin_mil_ce_fwd_bwd")
    return loss, grad


def mil_focal(x: torch.Tensor, alpha: torch.Tensor, target: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
              gamma: float = 1.5, avg_positives: bool = True, weights: Optional[torch.Tensor] = None, reduction: str = "mean",
              want_grad: bool = True):
    """MILFocalLoss (coin/utils/losses.py:36-73): alpha [C] = the class weights; the reference takes the mean over rows without row
    weights (`weights` / reduction 'sum' serve the fixed-shape sampler's validity mask)."""
    _dev(x, alpha, target, labels, weights)
    if weights is not None:
        weights = _f32c(weights, "weights")
    x, alpha = _f32c(x, "x"), _f32c(alpha, "alpha")
    r, c = x.shape
    if alpha.numel() != c:
        raise CoinHipError("mil_focal: alpha must hold one weight per class")
    if target is not None:
        target = _f32c(target, "target")
    if labels is not None and (labels.dtype != torch.int64 or not labels.is_contiguous()):
        raise CoinHipError("labels must be contiguous int64")
    loss = _scalar(x.device)
    grad = torch.empty_like(x) if want_grad else None
    check(_lib.lib().coin_mil_focal_fwd_bwd(_p(x), x.stride(0) if r else c, _p(target), _p(labels), _p(weights), _p(alpha), float(gamma),
                                            r, c, int(avg_positives), int(reduction == "mean"), _p(loss), _p(grad), _stream()),
          "coin_mil_focal_fwd_bwd")
    return loss, grad


def kl_div(x: torch.Tensor, q: torch.Tensor, mode: int, row_mask: Optional[torch.Tensor] = None, eps: float = 1e-7,
           want_grad: bool = True):
    """mode 0: x logits [R,C]; 1: x probabilities [R,C]; 2: x binary logits [R], q [R]."""
    _dev(x, q, row_mask)
    x = _f32c(x, "x")
    q = _f32c(q, "q")
    if mode == 2:
        r, c, ldx, ldq = x.shape[0], 2, 1, 1
    else:
        r, c = x.shape
        ldx, ldq = c, c
    if row_mask is not None:
        if row_mask.dtype == torch.bool:
            row_mask = row_mask.to(torch.uint8)
        if row_mask.dtype != torch.uint8 or not row_mask.is_contiguous():
            raise CoinHipError("row_mask must be contiguous bool/uint8")
    loss = _scalar(x.device)
    grad = torch.empty_like(x) if want_grad else None
    check(_lib.lib().coin_kl_div_fwd_bwd(_p(x), ldx, _p(q), ldq, _p(row_mask), r, c, mode, float(eps), _p(loss), _p(grad),
                                         _stream()), "coin_kl_div_fwd_bwd")
    return loss, grad


def box_reg_l1(proposals: torch.Tensor, gt_boxes: torch.Tensor, pred_deltas: torch.Tensor, gt_classes: torch.Tensor,
               num_fg_classes: int, weights: Sequence[float], normalizer: float, want_grad: bool = True):
    _dev(proposals, gt_boxes, pred_deltas, gt_classes)
    proposals, gt_boxes, pred_deltas = _f32c(proposals, "proposals"), _f32c(gt_boxes, "gt_boxes"), _f32c(pred_deltas, "pred_deltas")
    if gt_classes.dtype != torch.int64 or not gt_classes.is_contiguous():
        raise CoinHipError("gt_classes must be contiguous int64")
    r = proposals.shape[0]
    loss = _scalar(pred_deltas.device)
    grad = torch.empty_like(pred_deltas) if want_grad else None
    wx, wy, ww, wh = [float(v) for v in weights]
    check(_lib.lib().coin_box_reg_l1_fwd_bwd(_p(proposals), _p(gt_boxes), _p(pred_deltas), _p(gt_classes), r,
                                             int(num_fg_classes), wx, wy, ww, wh, float(normalizer), _p(loss), _p(grad),
                                             _stream()), "coin_box_reg_l1_fwd_bwd")
    return loss, grad


def l1_mean(a: torch.Tensor, b: torch.Tensor, want_grad: bool = True):
    _dev(a, b)
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if a.shape != b.shape:
        raise CoinHipError("l1_mean: shape mismatch")
    loss = _scalar(a.device)
    grad = torch.empty_like(a) if want_grad else None
    check(_lib.lib().coin_l1_mean_fwd_bwd(_p(a), _p(b), a.numel(), _p(loss), _p(grad), _stream()), "coin_l1_mean_fwd_bwd")
    return loss, grad


def rpn_losses(logits: torch.Tensor, labels: torch.Tensor, deltas: torch.Tensor, anchors: torch.Tensor,
               matched_gt: torch.Tensor, min_label: int = 0, want_grad: bool = True):
    """logits [N,A] f32, labels [N,A] int8, deltas [N,A,4], anchors [A,4], matched_gt [N,A,4] -> sums."""
    _dev(logits, labels, deltas, anchors, matched_gt)
    logits, deltas, anchors, matched_gt = (_f32c(logits, "logits"), _f32c(deltas, "deltas"), _f32c(anchors, "anchors"),
                                           _f32c(matched_gt, "matched_gt"))
    if labels.dtype != torch.int8 or not labels.is_contiguous():
        raise CoinHipError("labels must be contiguous int8")
    a_total, a_img = logits.numel(), anchors.shape[0]
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    g_logits = torch.empty_like(logits) if want_grad else None
    g_deltas = torch.empty_like(deltas) if want_grad else None
    ws = torch.empty(2 * 1024, dtype=torch.float32, device=logits.device)   # COIN_RPN_LOSS_MAX_BLOCKS partial pairs
    check(_lib.lib().coin_rpn_losses_fwd_bwd(_p(logits), _p(labels), _p(deltas), _p(anchors), _p(matched_gt), a_total,
                                             a_img, int(min_label), ctypes.c_void_p(out.data_ptr()),
                                             ctypes.c_void_p(out.data_ptr() + 4), _p(g_logits), _p(g_deltas), _p(ws), _stream()),
          "coin_rpn_losses_fwd_bwd")
    return out[0], out[1], g_logits, g_deltas


# --------------------------------------------------------------------------- fused BatchNorm / pooling (NHWC)
BN_MAX_PARTS = 768  # COIN_BN_MAX_PARTS in include/coin_hip.h


def _nhwc(t: torch.Tensor, name: str):
    if t.dim() != 4 or not t.is_contiguous():
        raise CoinHipError(f"{name} must be a contiguous [N,H,W,C] tensor")
    return t.shape


def bn_stats(x: torch.Tensor, eps: float, momentum: float, running_mean: Optional[torch.Tensor] = None,
             running_var: Optional[torch.Tensor] = None, num_batches_tracked: Optional[torch.Tensor] = None):
    """Batch mean / rstd of x [N,H,W,C] (+ in-place running-statistics update, + the module's batch counter)."""
    _dev(x, running_mean, running_var, _nbt(num_batches_tracked))
    n, h, w, c = _nhwc(x, "x")
    ws = torch.empty(BN_MAX_PARTS * 2 * c, dtype=torch.float32, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    with _timed("coin_bn_stats", _nbytes(x)):
        check(_lib.lib().coin_bn_stats(_p(x), n, h, w, c, float(eps), float(momentum), _p(ws), _p(mean), _p(rstd), _p(running_mean),
                                       _p(running_var), _p(num_batches_tracked), _dt(x), _stream()), "coin_bn_stats")
    return mean, rstd


def bn_apply_fwd(x: torch.Tensor, mean, rstd, gamma, beta, residual: Optional[torch.Tensor], relu: bool, pool: int, want_mask: bool = False):
    """want_mask (pool 1 with a residual, pool 0): also return the ReLU bit mask [N*H*W, C/8] (uint8; C/4 groups for float32) that
    `bn_bwd` takes instead of the saved output / the residual input -> (y, mask)."""
    _dev(x, mean, rstd, gamma, beta, residual)
    n, h, w, c = _nhwc(x, "x")
    mask = None
    if want_mask:
        if pool == 2 or (pool == 1 and residual is None):
            raise CoinHipError("the ReLU bit mask exists for pool 1 with a residual and for pool 0")
        mask = torch.empty((n * h * w, c // (8 if x.dtype == torch.bfloat16 else 4)), dtype=torch.uint8, device=x.device)
    # pool: 1 = none, 2 = 2x2 average pool, 0 = global spatial mean (y is [N,1,1,C]; the full activation is never written)
    y = torch.empty((n, 1, 1, c) if pool == 0 else (n, h // pool, w // pool, c), dtype=x.dtype, device=x.device)
    if residual is not None and (residual.shape != x.shape or pool == 2 or residual.dtype != x.dtype or not residual.is_contiguous()):
        raise CoinHipError("residual must match the pre-pool activation (contiguous NHWC, same dtype)")
    with _timed("coin_bn_apply_fwd", _nbytes(x) + _nbytes(residual) + _nbytes(y) + _nbytes(mask)):
        check(_lib.lib().coin_bn_apply_fwd(_p(x), _p(_f32c(mean, "mean")), _p(_f32c(rstd, "rstd")), _p(_f32c(gamma, "gamma")),
                                           _p(_f32c(beta, "beta")), _p(residual), _p(y), _p(mask), n, h, w, c, int(relu), int(pool), _dt(x), _stream()),
              "coin_bn_apply_fwd")
    return (y, mask) if want_mask else y


def bn_bwd(x: torch.Tensor, dy: torch.Tensor, y: Optional[torch.Tensor], mean, rstd, gamma, beta, relu: bool, pool: int,
           want_dres: bool, mask: Optional[torch.Tensor] = None):
    """-> dx [N,H,W,C], dgamma [C], dbeta [C], d_residual (or None).
    `y`: the saved forward output (pool 1, only needed when the forward added a residual) or, for pool 0 (global mean), the
    forward's residual INPUT (the activation was never stored; the ReLU mask is recomputed).  `mask`: the forward's ReLU bit mask
    (`bn_apply_fwd(..., want_mask=True)`) -- replaces `y` in either role."""
    _dev(x, dy, y, mask)
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_contiguous()):
        raise CoinHipError("mask must be the contiguous uint8 tensor bn_apply_fwd returned")
    n, h, w, c = _nhwc(x, "x")
    if not dy.is_contiguous() or dy.dtype != x.dtype:
        raise CoinHipError("dy must be contiguous NHWC of x's dtype")
    dsums = torch.empty((BN_MAX_PARTS + 1) * 2 * c, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    # algorithmic bytes of the two-pass backward: the channel sums must be complete before any dx can be formed and the
    # tensors are far larger than the caches, so every input is streamed twice; outputs once
    alg = 2 * (_nbytes(x) + _nbytes(dy) + (_nbytes(mask) if mask is not None else _nbytes(y))) + _nbytes(dx) + _nbytes(dres)
    with _timed("coin_bn_bwd", alg):
        check(_lib.lib().coin_bn_bwd(_p(x), _p(dy), _p(y if mask is None else None), _p(mask), _p(mean), _p(rstd), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), n, h, w, c,
                                     int(relu), int(pool), _p(dsums), _p(dx), _p(dres), _dt(x), _stream()), "coin_bn_bwd")
    return dx, dsums[c:2 * c], dsums[:c], dres


def avgpool2_fwd(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    n, h, w, c = _nhwc(x, "x")
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    check(_lib.lib().coin_avgpool2_fwd(_p(x), _p(y), n, h, w, c, _dt(x), _stream()), "coin_avgpool2_fwd")
    return y


def avgpool2_bwd(dy: torch.Tensor, in_shape) -> torch.Tensor:
    _dev(dy)
    n, h, w, c = in_shape
    if not dy.is_contiguous():
        raise CoinHipError("dy must be contiguous NHWC")
    dx = torch.empty(tuple(in_shape), dtype=dy.dtype, device=dy.device)
    check(_lib.lib().coin_avgpool2_bwd(_p(dy), _p(dx), n, h, w, c, _dt(dy), _stream()), "coin_avgpool2_bwd")
    return dx


# --------------------------------------------------------------------------- NMS
def nms_batched(boxes: torch.Tensor, counts: torch.Tensor, iou_threshold: float, max_keep: int):
    """boxes [B,n_max,4] f32 (rows sorted by descending score), counts [B] int32 -> (keep [B,n_max] int32, num_keep [B])."""
    _dev(boxes, counts)
    boxes = _f32c(boxes, "boxes")
    if counts.dtype != torch.int32 or not counts.is_contiguous():
        raise CoinHipError("counts must be contiguous int32")
    b, n_max = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((b, n_max), dtype=torch.int32, device=boxes.device)
    num = torch.zeros((b,), dtype=torch.int32, device=boxes.device)
    if b == 0 or n_max == 0:
        return keep, num
    ws = torch.empty(_lib.lib().coin_nms_workspace_bytes(b, n_max), dtype=torch.uint8, device=boxes.device)
    check(_lib.lib().coin_nms_batched(_p(boxes), _p(counts), b, n_max, float(iou_threshold), int(max_keep), _p(ws), _p(keep),
                                      _p(num), _stream()), "coin_nms_batched")
    return keep, num


# --------------------------------------------------------------------------- anchor / proposal labelling
def anchor_match(gt_boxes: Sequence[torch.Tensor], anchors: torch.Tensor, lo: float, hi: float, labels: Tuple[int, int, int],
                 empty_label: int, allow_low_quality: bool, want_boxes: bool = True):
    """detectron2 Matcher(thresholds [lo, hi], labels) over pairwise_iou(gt_i, anchors) for a batch of images in one launch
    sequence (coin_anchor_match).  gt_boxes: per image a [G_i, 4] float32 tensor (G_i may be 0); anchors [A, 4] shared by the images,
    or [N, A, 4]: a candidate set per image.
    -> matched gt index [N, A] int64, label [N, A] int8, matched gt box [N, A, 4] (or None)."""
    anchors = _f32c(anchors, "anchors")
    _dev(anchors, *gt_boxes)
    per_image = anchors.dim() == 3       # [N, A, 4]: every image has its own candidate set
    n = len(gt_boxes)
    if per_image and anchors.shape[0] != n:
        raise CoinHipError("anchor_match: a [N, A, 4] box set needs one row block per image")
    a = anchors.shape[1] if per_image else anchors.shape[0]
    offs = [0]
    for g in gt_boxes:
        offs.append(offs[-1] + int(g.shape[0]))
    total = offs[-1]
    if total:
        nz = [g.reshape(-1, 4).float() for g in gt_boxes if g.shape[0]]
        cat = (nz[0] if len(nz) == 1 else torch.cat(nz)).contiguous()
    else:
        cat = None
    matched = torch.empty((n, a), dtype=torch.int64, device=anchors.device)
    lab = torch.empty((n, a), dtype=torch.int8, device=anchors.device)
    mb = torch.empty((n, a, 4), dtype=torch.float32, device=anchors.device) if want_boxes else None
    ws = torch.empty(max(total, 1), dtype=torch.int32, device=anchors.device) if allow_low_quality else None
    c_offs = (ctypes.c_int * (n + 1))(*offs)
    check(_lib.lib().coin_anchor_match(_p(cat), c_offs, n, _p(anchors), a, int(per_image), float(lo), float(hi), int(labels[0]), int(labels[1]),
                                       int(labels[2]), int(empty_label), int(bool(allow_low_quality)), _p(matched), _p(lab), _p(mb), _p(ws), _stream()),
          "coin_anchor_match")
    return matched, lab, mb


def sample_labels(cls: torch.Tensor, keys: torch.Tensor, bg_label: int, num_samples: int, pos_cap: int) -> torch.Tensor:
    """cls [N, M] int8 / int64 (-1 ignore, bg_label negative, else positive), keys [N, M] float32 in [0, 1) -> int8 [N, M]:
    1 for the min(#pos, pos_cap) positives and 0 for the min(#neg, num_samples - #chosen pos) negatives with the smallest keys
    (ties: lowest index), -1 elsewhere (coin_sample_labels)."""
    _dev(cls, keys)
    if cls.dim() != 2 or keys.shape != cls.shape or cls.dtype not in (torch.int8, torch.int64) or not cls.is_contiguous():
        raise CoinHipError("sample_labels: cls must be a contiguous [N, M] int8 / int64 tensor and keys float32 of the same shape")
    keys = _f32c(keys, "keys")
    out = torch.empty(cls.shape, dtype=torch.int8, device=cls.device)
    check(_lib.lib().coin_sample_labels(_p(cls), int(cls.dtype == torch.int64), _p(keys), cls.shape[0], cls.shape[1], int(bg_label), int(num_samples),
                                        int(pos_cap), _p(out), _stream()), "coin_sample_labels")
    return out


# --------------------------------------------------------------------------- two-view input augmentation (uint8 [H, W, 3] images)
AUG_COPY, AUG_BRIGHTNESS, AUG_CONTRAST, AUG_SATURATION, AUG_HUE, AUG_GRAYSCALE, AUG_SOLARIZE = range(7)


def _u8img(t: torch.Tensor, name: str) -> Tuple[int, int]:
    _dev(t)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3 or not t.is_contiguous():
        raise CoinHipError(f"{name} must be a contiguous uint8 [H, W, 3] image")
    return int(t.shape[0]), int(t.shape[1])


def aug_resize_bilinear(img: torch.Tensor, out_h: int, out_w: int, flip_h: bool = False) -> torch.Tensor:
    """PIL `Image.resize((out_w, out_h), BILINEAR)` (+ `np.flip(axis=1)` when flip_h) of a uint8 [H, W, 3] device image, bit-exact."""
    h, w = _u8img(img, "img")
    out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=img.device)
    tmp = torch.empty((h, out_w, 3), dtype=torch.uint8, device=img.device) if (out_h != h and out_w != w) else None
    check(_lib.lib().coin_aug_resize_bilinear_u8(_p(img), h, w, _p(out), int(out_h), int(out_w), int(bool(flip_h)), _p(tmp), _stream()),
          "coin_aug_resize_bilinear_u8")
    return out


def aug_point_op(img: torch.Tensor, op: int, fparam: float = 0.0, iparam: int = 0, out_chw: bool = False) -> torch.Tensor:
    """One COIN_AUG_* point operation (Pillow ImageEnhance / HSV hue shift / convert("L") / solarize arithmetic, bit-exact).
    out_chw: the result is [3, H, W] (what coin_normalize_pad reads) instead of [H, W, 3]."""
    h, w = _u8img(img, "img")
    out = torch.empty((3, h, w) if out_chw else (h, w, 3), dtype=torch.uint8, device=img.device)
    ws = torch.empty(2, dtype=torch.int64, device=img.device) if op == AUG_CONTRAST else None
    check(_lib.lib().coin_aug_point_op_u8(_p(img), _p(out), h, w, int(op), float(fparam), int(iparam), _p(ws), int(bool(out_chw)), _stream()),
          "coin_aug_point_op_u8")
    return out


def aug_gaussian_blur(img: torch.Tensor, radius: float) -> torch.Tensor:
    """PIL `ImageFilter.GaussianBlur(radius)` (3 + 3 extended box blurs in 8.24 fixed point), bit-exact."""
    h, w = _u8img(img, "img")
    out, tmp = torch.empty_like(img), torch.empty_like(img)
    check(_lib.lib().coin_aug_gaussian_blur_u8(_p(img), _p(out), h, w, float(radius), _p(tmp), _stream()), "coin_aug_gaussian_blur_u8")
    return out


# --------------------------------------------------------------------------- streams of bytes
def normalize_pad(images: Sequence[torch.Tensor], mean: Sequence[float], std: Sequence[float],
                  size_divisibility: int = 0, layout: int = COIN_NCHW, dtype: torch.dtype = torch.float32):
    """uint8 [3,h,w] device images -> one zero-padded normalised batch (+ list of true sizes)."""
    _dev(*images)
    sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
    hp, wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
    if size_divisibility > 1:
        d = size_divisibility
        hp, wp = (hp + d - 1) // d * d, (wp + d - 1) // d * d
    nb = len(images)
    shape = (nb, 3, hp, wp) if layout == COIN_NCHW else (nb, hp, wp, 3)
    out = torch.empty(shape, dtype=dtype, device=images[0].device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    for i, im in enumerate(images):
        if im.dtype != torch.uint8 or not im.is_contiguous() or im.shape[0] != 3:
            raise CoinHipError("images must be contiguous uint8 [3,h,w]")
        check(_lib.lib().coin_normalize_pad(_p(im), sizes[i][0], sizes[i][1], m, s, _p(out), i, hp, wp, layout, _dt(out),
                                            _stream()), "coin_normalize_pad")
    return out, sizes


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class SgdTable:
    """Device table of (param, grad, momentum, lr, wd) descriptors for the one-launch SGD step."""

    def __init__(self, params: Sequence[torch.Tensor], lrs: Sequence[float], wds: Sequence[float],
                 shadows: Optional[Sequence[Optional[torch.Tensor]]] = None):
        _dev(*params)
        self.params = list(params)
        self.bufs = [torch.zeros_like(p) for p in self.params]  # same (dense) strides as the parameter: the kernel is elementwise on storage
        self.shadows = list(shadows) if shadows is not None else [None] * len(self.params)
        self.lrs, self.wds = list(lrs), list(wds)
        self.first = True
        self.max_numel = max((p.numel() for p in self.params), default=0)
        self._host = (_lib.SgdTensor * max(len(self.params), 1))()
        nbytes = ctypes.sizeof(self._host)
        self._dev = torch.empty(nbytes, dtype=torch.uint8, device=self.params[0].device) if self.params else None
        # pinned staging ring: the table is re-uploaded asynchronously whenever a gradient pointer changed (gradients are
        # released every step and re-allocated by autograd); a slot is reused only after its copy has executed
        self._stage = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.STAGES)] if self.params else []
        self._stage_ev = [None] * self.STAGES
        self._stage_i = 0
        self._grads_key = None

    STAGES = 4

    def _upload(self, grads):
        for i, (p, g, b, s) in enumerate(zip(self.params, grads, self.bufs, self.shadows)):
            e = self._host[i]
            if g is None:  # torch.optim.SGD skips parameters without a gradient (no weight decay, no momentum update)
                e.numel = 0
                continue
            if not (_dense(p) and g.stride() == p.stride() and b.stride() == p.stride() and p.dtype == torch.float32 and g.dtype == torch.float32):
                raise CoinHipError("SGD needs dense float32 params with identically laid out grads (contiguous or channels_last)")
            if s is not None and (s.stride() != p.stride() or s.dtype != torch.bfloat16):
                raise CoinHipError("a bf16 shadow must share its parameter's layout")
            e.param, e.grad, e.momentum_buf = p.data_ptr(), g.data_ptr(), b.data_ptr()
            e.bf16_shadow = s.data_ptr() if s is not None else None
            e.numel, e.lr, e.weight_decay = p.numel(), float(self.lrs[i]), float(self.wds[i])
        k = self._stage_i
        self._stage_i = (k + 1) % self.STAGES
        if self._stage_ev[k] is not None:
            self._stage_ev[k].synchronize()
        ctypes.memmove(self._stage[k].data_ptr(), ctypes.addressof(self._host), ctypes.sizeof(self._host))
        self._dev.copy_(self._stage[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stage_ev[k] = ev

    def step(self, grads: Sequence[torch.Tensor], momentum: float, inv_loss_scale: float = 1.0,
             lrs: Optional[Sequence[float]] = None, lr_scale: float = 1.0,
             shadows: Optional[Sequence[Optional[torch.Tensor]]] = None, gate: Optional[torch.Tensor] = None):
        """`lrs` (re)defines the per-tensor base learning rates held in the device table (re-uploaded only when they or the
        gradient pointers change); `lr_scale` is the per-step schedule factor passed as a kernel argument; `gate` (one fp32 on the
        device): the launch is a no-op when it holds 0."""
        if gate is not None:
            _dev(gate)
            if gate.dtype != torch.float32 or gate.numel() != 1:
                raise CoinHipError("sgd gate must be a one-element float32 device tensor")
        if shadows is not None:
            self.shadows = list(shadows)
        key = (tuple(g.data_ptr() if g is not None else 0 for g in grads), tuple(lrs) if lrs is not None else tuple(self.lrs),
               tuple(s.data_ptr() if s is not None else 0 for s in self.shadows))
        if lrs is not None:
            self.lrs = list(lrs)
        if key != self._grads_key:
            self._upload(grads)
            self._grads_key = key
        check(_lib.lib().coin_sgd_step(_p(self._dev), len(self.params), self.max_numel, float(momentum),
                                       float(inv_loss_scale), float(lr_scale), int(self.first), _p(gate), _stream()), "coin_sgd_step")
        self.first = False


class WdTable:
    """Device table for `coin_weight_dgrad_layout`: (bf16 source weight, bf16 destination in the data-gradient layout, cout, cin, ks)."""

    def __init__(self, entries):
        self.n = len(entries)
        host = (_lib.WdTensor * max(self.n, 1))()
        self.max_tiles = 0
        for i, (src, dst, cout, cin, ks) in enumerate(entries):
            _dev(src, dst)
            if src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or cout % 8 or cin % 8 or src.numel() != cout * cin * ks * ks or dst.numel() != src.numel():
                raise CoinHipError("weight_dgrad_layout: bf16 weights with cout % 8 == 0 and cin % 8 == 0")
            host[i].src, host[i].dst, host[i].cout, host[i].cin, host[i].ks = src.data_ptr(), dst.data_ptr(), cout, cin, ks
            self.max_tiles = max(self.max_tiles, ks * ks * ((cout + 63) // 64) * ((cin + 63) // 64))
        self._keep = entries
        self._dev = torch.frombuffer(memoryview(host).cast("B"), dtype=torch.uint8).to(entries[0][0].device) if self.n else None

    def run(self):
        if self.n:
            check(_lib.lib().coin_weight_dgrad_layout(_p(self._dev), self.n, self.max_tiles, _stream()), "coin_weight_dgrad_layout")


class EmaTable:
    def __init__(self, teacher: Sequence[torch.Tensor], student: Sequence[torch.Tensor]):
        _dev(*teacher, *student)
        self.n = len(teacher)
        self.max_numel = max((t.numel() for t in teacher), default=0)
        host = (_lib.EmaTensor * max(self.n, 1))()
        for i, (t, s) in enumerate(zip(teacher, student)):
            if not (_dense(t) and s.stride() == t.stride() and t.dtype == torch.float32 and s.dtype == torch.float32):
                raise CoinHipError("EMA needs dense float32 tensors with identical layout (contiguous or channels_last)")
            host[i].teacher, host[i].student, host[i].numel = t.data_ptr(), s.data_ptr(), t.numel()
        self._keep = (list(teacher), list(student))
        self._teacher_ptrs = frozenset(t.data_ptr() for t in teacher)
        self._dev = torch.frombuffer(memoryview(host).cast("B"), dtype=torch.uint8).to(teacher[0].device) if self.n else None

    def update(self, keep: float):
        check(_lib.lib().coin_ema_update(_p(self._dev), self.n, self.max_numel, float(keep), _stream()), "coin_ema_update")
        from . import layers  # late import (layers imports this module): the teacher's bf16 weight shadows are now stale
        layers.invalidate_storage(self._teacher_ptrs)   # by storage: the table's tensors are state_dict() aliases, not the Parameters
