"""Autograd bindings of the HIP kernels (``coin_amd.kernels``): the differentiable ops the modules use.

Every op here runs ONLY on the GPU through ``libcoin_hip.so``; there is no torch-composite fallback.
Reference call sites are cited per class (paths under /root/reference).
"""
from __future__ import annotations

import weakref
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import kernels as K
from ._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, COIN_NHWC, CoinHipError


def _pad_rows(x: torch.Tensor, mult: int) -> torch.Tensor:
    m = x.shape[0]
    if m % mult == 0:
        return x
    pad = x.new_zeros((mult - m % mult,) + tuple(x.shape[1:]))
    return torch.cat([x, pad], dim=0)


# --------------------------------------------------------------------------- compute-dtype weight shadows
# bf16 throughput mode keeps fp32 master parameters (state dict, optimizer) and a persistent bf16 copy ("shadow") of each.
# The one-launch SGD kernel refreshes the shadow together with the master (coin_sgd_tensor.bf16_shadow), so a training step
# contains no per-parameter fp32->bf16 cast kernels (torch.autocast issues one per weight per step: ~230 launches here).
class _Shadow:
    __slots__ = ("ref", "tensor", "version", "ptr")


_SHADOWS: dict = {}


def _shadow_entry(param: torch.Tensor, dtype: torch.dtype) -> "_Shadow":
    e = _SHADOWS.get(id(param))
    if e is not None and e.ref() is not param:
        e = None
    if e is None or e.tensor.dtype != dtype or e.tensor.shape != param.shape or e.tensor.device != param.device:
        e = _Shadow()
        e.ref = weakref.ref(param, lambda _r, k=id(param): _SHADOWS.pop(k, None))
        e.tensor = torch.empty_like(param, dtype=dtype)  # preserve_format: same dense strides as the master (SGD kernel is elementwise)
        e.version, e.ptr = -1, 0
        _SHADOWS[id(param)] = e
    if e.version != param._version or e.ptr != param.data_ptr():
        # the master was written by something other than the fused SGD / EMA kernels (init, load_state_dict, .data = ...)
        with torch.no_grad():
            e.tensor.copy_(param.detach())
        e.version, e.ptr = param._version, param.data_ptr()
    return e


def shadow_of(param: torch.Tensor) -> Optional[torch.Tensor]:
    """The live compute-dtype copy of `param` (None if it never had one); used by the optimizer's tensor table."""
    e = _SHADOWS.get(id(param))
    return e.tensor if e is not None and e.ref() is param and e.tensor.stride() == param.stride() else None


def invalidate_shadows(params: Sequence[torch.Tensor]) -> None:
    """Call after writing masters through raw pointers WITHOUT refreshing their shadows (the EMA kernel)."""
    for p in params:
        e = _SHADOWS.get(id(p))
        if e is not None:
            e.version = -1


def invalidate_storage(ptrs) -> None:
    """Everything derived from the storages at `ptrs` (a set of data_ptr()s) is stale: bf16 weight shadows and the frozen-norm
    constants.  For writers that only know raw pointers -- `EmaTable` is built from `state_dict()` tensors, which are detached
    aliases of the Parameters (another id(), the same storage), and its kernel bumps no `_version` (round-3 ADVICE, high)."""
    for e in _SHADOWS.values():
        if e.ptr in ptrs:
            e.version = -1
    for k in [k for k, e in _DGRAD.items() if e.ptr in ptrs]:
        _DGRAD[k].version = -1
    for cache in _DERIVED_CACHES:   # entries are (key, ...) with key = ((data_ptr, _version), ..., [dtype])
        for k in [k for k, hit in cache.items() if any(isinstance(it, tuple) and it[0] in ptrs for it in hit[0])]:
            cache.pop(k, None)


_DERIVED_CACHES: list = []   # per-module constants derived from parameters / buffers, keyed by (data_ptr, _version) tuples


# --------------------------------------------------------------------------- data-gradient layouts of the weights
# dX of a convolution / linear layer is the same contraction over (flipped tap, Cout) with the weight re-laid [Cin][ky'][kx'][Cout]
# (a linear weight [N, K]: its transpose).  The re-layout used to be issued per backward call (flip + permute + copy: 46 + 11 launches
# per step).  Now every weight that has been differentiated once keeps a persistent bf16 copy in that layout, and ONE launch
# (coin_weight_dgrad_layout over a device table) rewrites all of them from the compute shadows, lazily at the first backward use after
# an optimizer step (`weights_updated`).
class _Dgrad:
    __slots__ = ("ref", "tensor", "src", "version", "ptr", "dims")


_DGRAD: dict = {}
_DGRAD_STATE = {"dirty": False, "table": None, "event": None, "stream": None}


def weights_updated() -> None:
    """Called by the optimizer after its (raw-pointer) update of masters and shadows: the dgrad layouts are stale."""
    if _DGRAD:
        _DGRAD_STATE["dirty"] = True


def _run_dgrad_table(device) -> None:
    st = _DGRAD_STATE
    if st["table"] is None:
        live = [x for x in _DGRAD.values() if x.ref() is not None]
        st["table"] = K.WdTable([(x.src, x.tensor, *x.dims) for x in live])
        st["live"] = live
    st["table"].run()
    for x in st["live"]:
        p = x.ref()
        if p is None:
            continue
        # stamp only layouts built from a CURRENT shadow: a master rewritten with a version bump (load_state_dict, init_, copy_) whose
        # shadow has not been refreshed by a forward yet keeps version -1 and takes the reference copy at its next use (round-4 ADVICE)
        sh = _SHADOWS.get(id(p))
        fresh = sh is not None and sh.ref() is p and sh.version == p._version and sh.ptr == p.data_ptr()
        x.version, x.ptr = (p._version, p.data_ptr()) if fresh else (-1, 0)
    st["dirty"] = False
    st["stream"] = torch.cuda.current_stream(device)
    st["event"] = torch.cuda.Event()
    st["event"].record(st["stream"])


def refresh_dgrad_layouts() -> None:
    """Run the pending one-launch refresh of the data-gradient layouts NOW, eagerly, on the current stream (coin_amd.graphs calls it
    before a backward graph is captured or replayed: inside a capture the launch would be recorded, not executed, while the host-side
    stamps said it had run)."""
    if _DGRAD_STATE["dirty"] and _DGRAD and not torch.cuda.is_current_stream_capturing():
        live = [x for x in _DGRAD.values() if x.ref() is not None]
        if live:
            _run_dgrad_table(live[0].tensor.device)


def _dgrad_reference(wq: torch.Tensor, ks: int) -> torch.Tensor:
    if wq.dim() == 2:
        return wq.t().contiguous()
    co, ci = wq.shape[0], wq.shape[1]
    return wq.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(ci, ks * ks * co)


def dgrad_weight(param: Optional[torch.Tensor], wq: torch.Tensor, ks: int) -> torch.Tensor:
    """`wq` (bf16 compute copy of `param`: conv weight [Co, Ci, k, k] in channels-last memory, or linear weight [N, K]) in the
    data-gradient layout [Ci, k*k*Co] (linear: [K, N])."""
    co, ci = wq.shape[0], wq.shape[1]
    dense = wq.is_contiguous() if wq.dim() == 2 else wq.is_contiguous(memory_format=torch.channels_last)
    if param is None or not wq.is_cuda or wq.dtype != torch.bfloat16 or co % 8 or ci % 8 or not dense:
        return _dgrad_reference(wq, ks)
    e = _DGRAD.get(id(param))
    if e is not None and (e.ref() is not param or e.src.data_ptr() != wq.data_ptr() or e.dims != (co, ci, ks)):
        e = None
    if e is None:
        e = _Dgrad()
        e.ref = weakref.ref(param, lambda _r, k=id(param): (_DGRAD.pop(k, None), _DGRAD_STATE.__setitem__("table", None)))
        e.tensor = torch.empty((ci, ks * ks * co), dtype=torch.bfloat16, device=wq.device)
        e.src, e.dims, e.version, e.ptr = wq, (co, ci, ks), -1, 0
        _DGRAD[id(param)] = e
        _DGRAD_STATE["table"] = None
    st = _DGRAD_STATE
    capturing = torch.cuda.is_current_stream_capturing()
    if st["dirty"] and not capturing:
        # one launch for every registered weight, on the stream of the first backward node that needs one; nodes on other streams wait
        _run_dgrad_table(wq.device)
    elif st["event"] is not None and not capturing and torch.cuda.current_stream(wq.device) != st["stream"]:
        torch.cuda.current_stream(wq.device).wait_event(st["event"])
    # (under a graph capture nothing is launched or waited for here: coin_amd.graphs refreshed the layouts eagerly before the capture,
    # and a new entry registered during the capture takes the reference copy below, recorded into the graph)
    if e.version != param._version or e.ptr != param.data_ptr():
        # first use, or the master was written by something other than the fused SGD kernel (init, load_state_dict)
        with torch.no_grad():
            e.tensor.copy_(_dgrad_reference(wq, ks))
        e.version, e.ptr = param._version, param.data_ptr()
    return e.tensor


class _ShadowCast(Function):
    @staticmethod
    def forward(ctx, param, shadow):
        return shadow.detach()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g.float(), None


def compute_weight(param: Optional[torch.Tensor], dtype: torch.dtype) -> Optional[torch.Tensor]:
    """`param` in the compute dtype: the parameter itself (fp32 mode), else its persistent shadow, wired into autograd so
    that the fp32 master receives the gradient."""
    if param is None or param.dtype == dtype or not param.is_cuda:
        return param
    e = _shadow_entry(param, dtype)
    if param.requires_grad and torch.is_grad_enabled():
        return _ShadowCast.apply(param, e.tensor)
    return e.tensor


def compute_dtype_of(x: torch.Tensor) -> torch.dtype:
    """Dtype the weights should take for input x: the autocast dtype inside an autocast region, else x's own."""
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype


def _no_library_conv_under_capture(what: str) -> None:
    """The library's convolutions must not be recorded into a HIP graph on this stack: replayed, single MIOpen launches (forward 1x1 on
    the res4 map, the backward-weights kernels of the 128-channel convolutions) produced garbage as soon as the process state differed
    from the capture's (another allocation pattern, an empty_cache(), a bare replay) -- measured with tools/bb_bisect.py (round 5; in the git history) and
    tools/miopen_graph_probe.py, DESIGN.md section 3.17.  A stretch that is to be captured keeps every convolution on the hand-written
    kernels (`conv_gemm_everywhere`).  Every library convolution is counted here: coin_amd.graphs compares the count across the dry run
    that precedes a capture and refuses the capture when it moved (the stretch then stays eager, with a warning); reaching one UNDER a
    capture raises as a last resort."""
    LIBRARY_CONV_CALLS[0] += 1
    if STRICT_CAPTURE[0] and torch.cuda.is_current_stream_capturing():
        raise CoinHipError(f"{what}: a library convolution inside a captured stretch")


STRICT_CAPTURE = [False]   # set by coin_amd.graphs around its captures
LIBRARY_CONV_CALLS = [0]   # library convolutions issued so far (coin_amd.graphs compares it across the dry run that precedes a capture)


def conv2d(x: torch.Tensor, conv: torch.nn.Conv2d) -> torch.Tensor:
    """conv(x) with the weight (and bias) taken from their compute-dtype shadows."""
    if x.is_cuda:
        _no_library_conv_under_capture("conv2d")
    dt = compute_dtype_of(x)
    if x.dtype != dt:
        x = x.to(dt)
    return torch.nn.functional.conv2d(x, compute_weight(conv.weight, dt), compute_weight(conv.bias, dt), conv.stride, conv.padding,
                                      conv.dilation, conv.groups)


# --------------------------------------------------------------------------- convolutions on the hand-written GEMM
# res5 runs on the RoI tiles ([R*196, 1024] ... [R*49, 2048] rows): its 1x1 convolutions are NHWC GEMMs and its 3x3 convolutions
# implicit GEMMs; forward and data-gradient go through coin_conv_gemm_bf16 (with the BatchNorm statistics of the output taken in
# the epilogue), the weight gradient through coin_conv_wgrad_bf16 (channel counts that are multiples of 256; the library's
# contraction otherwise).  bf16 compute mode only: the fp32 parity mode keeps the library's fp32 convolutions.
CONV_GEMM = {"enabled": False, "min_rows": 32768, "wgrad": True}


class conv_gemm_everywhere:
    """Inside: every convolution the hand-written GEMM can serve takes it, whatever its row count (a stretch that is replayed as a HIP
    graph must not contain library convolutions, see `_no_library_conv_under_capture`)."""

    def __enter__(self):
        self.prev = CONV_GEMM["min_rows"]
        CONV_GEMM["min_rows"] = 0

    def __exit__(self, *a):
        CONV_GEMM["min_rows"] = self.prev
        return False


def library_free(convs) -> bool:
    """True when forward, data gradient and weight gradient of every convolution in `convs` run on the hand-written kernels (bf16 mode,
    under `conv_gemm_everywhere`)."""
    for c in convs:
        ks = c.kernel_size
        if ks not in ((1, 1), (3, 3)) or c.stride != (1, 1) or c.padding != (ks[0] // 2, ks[1] // 2) or c.dilation != (1, 1) or c.groups != 1:
            return False
        if c.bias is not None or c.in_channels % 64 or c.out_channels % 64 or not K.conv_wgrad_ok(c.out_channels, c.in_channels):
            return False
    return bool(CONV_GEMM["enabled"] and CONV_GEMM["wgrad"])


def _conv_gemm_ok(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    if not (CONV_GEMM["enabled"] and x.is_cuda and x.dim() == 4 and compute_dtype_of(x) == torch.bfloat16):
        return False
    ks = conv.kernel_size
    if ks not in ((1, 1), (3, 3)) or conv.stride != (1, 1) or conv.padding != (ks[0] // 2, ks[1] // 2) or conv.dilation != (1, 1):
        return False
    if conv.groups != 1 or conv.bias is not None or conv.in_channels % 64 or conv.out_channels % 64:
        return False
    rows = x.shape[0] * x.shape[2] * x.shape[3]
    if rows >= CONV_GEMM["min_rows"]:
        return True
    # long-K 3x3 convolutions with few output tiles (layer3's conv2 on the res4 map: 65 tiles for 256 CUs): the split-K plan cuts every
    # tile along K, measured 47 / 47 / 55 us (forward / dgrad / wgrad) against the library's 50 / 132 us (tools/gemm_lab, l3.x.conv2)
    return ks == (3, 3) and rows >= CONV_GEMM["min_rows"] // 2 and conv.in_channels % 256 == 0 and conv.out_channels % 256 == 0


class _ConvGemm(Function):
    """nn.Conv2d (1x1, or 3x3 / pad 1; stride 1, no bias) of the CLIP Bottleneck (coin/modeling/utils.py:40-58,77-90) on
    channels-last bf16 activations."""

    @staticmethod
    def forward(ctx, x, weight, wq, stats_rows, fork=False):
        xn = _as_nhwc(x)
        n, h, w, c = xn.shape
        co, ks = wq.shape[0], wq.shape[2]
        wk = wq.permute(0, 2, 3, 1)
        wk = (wk if wk.is_contiguous() else wk.contiguous()).reshape(co, ks * ks * c)
        out, part = K.conv_gemm(xn.reshape(n * h * w, c), wk, spatial=(h, w, c) if ks == 3 else None, stats_rows=stats_rows)
        ctx.save_for_backward(x, wq)
        ctx.ks = ks
        ctx.param = weakref.ref(weight)
        ctx.pool_tap = fork == "pool"
        # the statistics partials are never differentiated: without this the engine materialises a zero tensor of their size for every
        # backward call (35 fill launches per pre-train step, tools/opsites.py)
        ctx.set_materialize_grads(False)
        if part is None:
            part = out.new_zeros(0, dtype=torch.float32)
        ctx.mark_non_differentiable(part)
        y = out.view(n, h, w, co).permute(0, 3, 1, 2)
        # fork: also hand the input back as the tap of the block's OTHER consumer (identity / downsample branch), so that this
        # node receives both gradients of x and forms their sum in the dgrad epilogue instead of autograd's separate add pass.
        # fork == "pool": the other consumer starts with AvgPool2d(2) (the stride-2 Bottleneck's downsample branch, utils.py:60-75):
        # the tap is the POOLED input, and its gradient is folded into the same epilogue at quarter weight -- the avg-pool backward
        # (a full-size tensor written, then read as the residual) is never materialised.
        if fork == "pool":
            return y, part, K.avgpool2_fwd(xn).permute(0, 3, 1, 2)
        return (y, part, x.view_as(x)) if fork else (y, part)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gpart, g_tap=None):
        x, wq = ctx.saved_tensors
        if gy is None:   # the convolution's output reached no loss: only the other consumer's gradient of x (if any) flows on
            return (g_tap if ctx.needs_input_grad[0] else None), None, None, None, None
        gyn = _as_nhwc(gy)
        if gyn.dtype != torch.bfloat16:
            gyn = gyn.to(torch.bfloat16)
        dx, dw = _conv_gemm_grads(x, wq, gyn, ctx.ks, ctx.needs_input_grad[0], ctx.needs_input_grad[1], g_tap, param=ctx.param(),
                                  tap_pooled=ctx.pool_tap)
        return dx, dw, None, None, None


def _conv_gemm_grads(x, wq, gyn, ks, need_dx, need_dw, g_tap=None, param=None, tap_pooled=False):
    """(dx, dw) of the GEMM convolutions; gyn: bf16 NHWC gradient of the output, x: the saved (logical NCHW) input, wq: bf16 weight,
    param: the fp32 master of wq (key of the persistent data-gradient layout)."""
    pad = ks // 2
    n, h, w, co = gyn.shape
    ci = wq.shape[1]
    dx = dw = None
    if need_dx:
        # dgrad = the same contraction over (flipped tap, Cout): weight re-laid [Cin][ky'][kx'][Cout] (persistent copy, see dgrad_weight)
        wd = dgrad_weight(param, wq, ks)
        res = None
        gx = None
        if g_tap is not None:
            res = _as_nhwc(g_tap)
            res = res if res.dtype == torch.bfloat16 else res.to(torch.bfloat16)
            if tap_pooled:   # gradient of avg_pool2(x): [n, h // 2, w // 2, ci]
                done = K.conv_gemm(gyn.reshape(n * h * w, co), wd, spatial=(h, w, co) if ks == 3 else None,
                                   residual=res.reshape(n * (h // 2) * (w // 2), ci), residual_pool=(h, w))
                if done is not None:
                    gx = done[0]
                else:        # a shape the persistent kernel does not serve: materialise the pool's backward
                    res = K.avgpool2_bwd(res.contiguous(), (n, h, w, ci))
            if gx is None:
                res = res.reshape(n * h * w, ci)
        if gx is None:
            gx, _ = K.conv_gemm(gyn.reshape(n * h * w, co), wd, spatial=(h, w, co) if ks == 3 else None, residual=res)
        dx = gx.view(n, h, w, ci).permute(0, 3, 1, 2)
    if need_dw:
        if CONV_GEMM["wgrad"] and K.conv_wgrad_ok(co, ci, n * h * w):
            xn = _as_nhwc(x)
            dw2 = K.conv_wgrad(gyn.reshape(n * h * w, co), xn.reshape(n * h * w, ci), spatial=(h, w, ci) if ks == 3 else None)
            dw = dw2.view(co, ks, ks, ci).permute(0, 3, 1, 2)  # fp32, already in the parameter's channels-last layout
        else:
            _no_library_conv_under_capture("weight gradient")
            dw = torch.ops.aten.convolution_backward(gyn.permute(0, 3, 1, 2), x, wq, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1].float()
    return dx, dw


class _ConvGemmBiasRelu(Function):
    """relu(conv(x) + bias) for a 3x3 / pad 1 (or 1x1) convolution WITH bias on the hand-written GEMM: the RPN head's shared
    convolution (detectron2 0.5 `StandardRPNHead.forward`, called at coin/modeling/proposal_generator/rpn.py:65 -- 1024 -> 1024 channels on res4, 313 GFLOP
    per pass at the timed shape, the largest convolution left on the library).  The GEMM stores bf16(conv), one streaming pass adds
    the bias and clamps (the frozen-BatchNorm apply kernel with mean 0 / scale 1 / shift = bias)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wq):
        xn = _as_nhwc(x)
        n, h, w, c = xn.shape
        co, ks = wq.shape[0], wq.shape[2]
        wk = wq.permute(0, 2, 3, 1)
        wk = (wk if wk.is_contiguous() else wk.contiguous()).reshape(co, ks * ks * c)
        z, _ = K.conv_gemm(xn.reshape(n * h * w, c), wk, spatial=(h, w, c) if ks == 3 else None)
        one, zero = _unit_consts(co, x.device)
        y = K.bn_apply_fwd(z.view(n, h, w, co), zero, one, one, bias.detach().float().contiguous(), None, True, 1)
        ctx.save_for_backward(x, wq, y)
        ctx.ks = ks
        ctx.param = weakref.ref(weight)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, wq, y = ctx.saved_tensors
        gyn = _as_nhwc(gy)
        if gyn.dtype != torch.bfloat16:
            gyn = gyn.to(torch.bfloat16)
        n, h, w, co = gyn.shape
        dz, dbias = K.bias_act_bwd(gyn.reshape(n * h * w, co), y.view(n * h * w, co), K.ACT_RELU)
        dx, dw = _conv_gemm_grads(x, wq, dz.view(n, h, w, co), ctx.ks, ctx.needs_input_grad[0], ctx.needs_input_grad[1], param=ctx.param())
        return dx, dw, (dbias if ctx.needs_input_grad[2] else None), None


_UNIT: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}


def _unit_consts(c: int, device):
    key = (c, str(device))
    if key not in _UNIT:
        _UNIT[key] = (torch.ones(c, dtype=torch.float32, device=device), torch.zeros(c, dtype=torch.float32, device=device))
    return _UNIT[key]


def conv_bias_relu(x: torch.Tensor, conv: torch.nn.Conv2d, min_rows: int = 8192) -> torch.Tensor:
    """relu(conv(x)) for a convolution with bias: on the hand-written GEMM in the bf16 mode when the shape fits (stride 1, 'same'
    padding, Cin % 64 == 0, Cout % 256 == 0, at least `min_rows` output pixels), else the library convolution + relu."""
    ks = conv.kernel_size
    if (CONV_GEMM["enabled"] and x.is_cuda and x.dim() == 4 and compute_dtype_of(x) == torch.bfloat16 and conv.bias is not None
            and ks in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.padding == (ks[0] // 2, ks[1] // 2) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.in_channels % 64 == 0 and conv.out_channels % 256 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] >= min_rows):
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        wq = _shadow_entry(conv.weight, torch.bfloat16).tensor
        return _ConvGemmBiasRelu.apply(x, conv.weight, conv.bias, wq)
    return torch.nn.functional.relu(conv2d(x, conv))


def conv2d_gemm(x: torch.Tensor, conv: torch.nn.Conv2d, stats_rows: Optional[int] = None, fork: bool = False):
    """-> (conv(x), statistics partials or None [, tap of x for the block's identity branch]); caller checked `_conv_gemm_ok`."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    wq = _shadow_entry(conv.weight, torch.bfloat16).tensor
    if fork:
        y, part, tap = _ConvGemm.apply(x, conv.weight, wq, stats_rows, fork)
        return y, (part if stats_rows is not None else None), tap
    y, part = _ConvGemm.apply(x, conv.weight, wq, stats_rows)
    return y, (part if stats_rows is not None else None)


def conv_bn_act(x: torch.Tensor, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, relu: bool, residual: Optional[torch.Tensor] = None,
                pool: int = 1, fork: bool = False):
    """conv -> train-mode BatchNorm (+ identity) (+ ReLU) (+ pool): with the hand-written GEMM the BatchNorm statistics come out of
    the convolution's epilogue (no statistics pass over the activation).
    fork=True -> (result, tap): `tap` is x for the block's other consumer (identity / downsample branch); when the convolution runs
    on the hand-written GEMM and x needs a gradient, the two gradients of x are summed in the dgrad epilogue (coin_conv_gemm_bf16's
    R operand) instead of by a separate elementwise pass.  fork="pool" -> `tap` is avg_pool2(x) (the stride-2 block's downsample
    branch starts with AvgPool2d(2)); on the GEMM path the pool's backward is folded into that epilogue too."""
    if _conv_gemm_ok(x, conv):
        do_fork = fork if (fork and torch.is_grad_enabled() and x.requires_grad) else False
        if bn.training and bn.momentum is not None:
            nv = _VALID_ROWS[0]
            rows = (x.shape[0] if nv is None or nv >= x.shape[0] else int(nv)) * x.shape[2] * x.shape[3]
            y, part, *tap = conv2d_gemm(x, conv, stats_rows=rows, fork=do_fork)
            out = bn_act(y, bn, relu, residual, pool, stats_part=(part, rows))
        else:
            y, _, *tap = conv2d_gemm(x, conv, fork=do_fork)
            out = bn_act(y, bn, relu, residual, pool)
        if not fork:
            return out
        return out, (tap[0] if tap else (avg_pool2(x) if fork == "pool" else x))
    out = bn_act(conv2d(x, conv), bn, relu, residual, pool)
    if not fork:
        return out
    return out, (avg_pool2(x) if fork == "pool" else x)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    dt = compute_dtype_of(x)
    if x.dtype != dt:
        x = x.to(dt)
    return torch.nn.functional.linear(x, compute_weight(weight, dt), compute_weight(bias, dt))


# --------------------------------------------------------------------------- RoIAlign
class _ROIAlignNHWC(Function):
    """clip_roi_heads.py:172-176 -> ROIPooler -> torchvision roi_align(aligned=True, sampling_ratio=0)."""

    @staticmethod
    def forward(ctx, feat_nhwc, rois, output_size, spatial_scale, sampling_ratio, aligned):
        out = K.roi_align_fwd(feat_nhwc, rois, output_size, spatial_scale, sampling_ratio, aligned, COIN_NHWC)
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat_nhwc.shape), feat_nhwc.dtype, spatial_scale, sampling_ratio, aligned)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        shape, dtype, scale, sr, aligned = ctx.meta
        g = K.roi_align_bwd(grad_out.contiguous(), rois, shape, scale, sr, aligned, COIN_NHWC)
        return g.to(dtype), None, None, None, None, None


def roi_align(feat: torch.Tensor, rois: torch.Tensor, output_size: Tuple[int, int], spatial_scale: float,
              sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    """feat: logical [N,C,H,W] in channels_last memory format (NHWC bytes); returns logical
    [R,C,ph,pw], also channels_last, so that the res5 convolutions consume it without a copy."""
    if feat.dim() != 4:
        raise CoinHipError("roi_align expects a 4-D feature map")
    nhwc = feat.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()  # one-off layout change if the producer was not channels_last
    out = _ROIAlignNHWC.apply(nhwc, rois.float().contiguous(), tuple(output_size), float(spatial_scale), int(sampling_ratio), bool(aligned))
    return out.permute(0, 3, 1, 2)


class _ROIAlignLevels(Function):
    """Multi-level pooler of the FPN extension (no reference counterpart): RoI r from pyramid level levels[r]; one forward launch for all
    levels (coin_roi_align_fwd_levels), one atomic-free gather launch per level in the backward (coin_roi_align_bwd_level)."""

    @staticmethod
    def forward(ctx, rois, levels, output_size, scales, sampling_ratio, aligned, *feats_nhwc):
        out = K.roi_align_fwd_levels(feats_nhwc, scales, rois, levels, output_size, sampling_ratio, aligned)
        ctx.save_for_backward(rois, levels)
        ctx.meta = ([tuple(f.shape) for f in feats_nhwc], [f.dtype for f in feats_nhwc], tuple(scales), sampling_ratio, aligned)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        rois, levels = ctx.saved_tensors
        shapes, dtypes, scales, sr, aligned = ctx.meta
        go = grad_out.contiguous()
        grads = []
        for i, (shape, dt, sc) in enumerate(zip(shapes, dtypes, scales)):
            grads.append(K.roi_align_bwd_level(go, rois, levels, i, shape, sc, sr, aligned).to(dt) if ctx.needs_input_grad[6 + i] else None)
        return (None, None, None, None, None, None) + tuple(grads)


def roi_align_levels(feats: Sequence[torch.Tensor], scales: Sequence[float], rois: torch.Tensor, levels: torch.Tensor, output_size: Tuple[int, int],
                     sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    """feats: logical [N,C,H_l,W_l] maps in channels_last memory format, levels: pyramid level index per RoI -> logical [R,C,ph,pw]
    (channels_last).  Level selection happens inside the launch: each RoI is pooled once, on its own level."""
    nhwc = []
    for f in feats:
        v = f.permute(0, 2, 3, 1)
        nhwc.append(v if v.is_contiguous() else v.contiguous())
    out = _ROIAlignLevels.apply(rois.float().contiguous(), levels.to(torch.int32).contiguous(), tuple(output_size), tuple(float(s) for s in scales),
                                int(sampling_ratio), bool(aligned), *nhwc)
    return out.permute(0, 3, 1, 2)


# --------------------------------------------------------------------------- fused BatchNorm(train) [+res] [+ReLU] [+pool]
def _as_nhwc(t: torch.Tensor) -> torch.Tensor:
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


# Leading rows of the batch that are real (the rest is shape padding, see OpenVocabularyRes5ROIHeads._pooled): train-mode
# BatchNorm takes its statistics and its gradient sums over these rows only and gives the filler rows zero gradient.
_VALID_ROWS = [None]


class valid_rows:
    def __init__(self, n: Optional[int]):
        self.n = n

    def __enter__(self):
        self.prev = _VALID_ROWS[0]
        _VALID_ROWS[0] = self.n

    def __exit__(self, *a):
        _VALID_ROWS[0] = self.prev
        return False


class _BNAct(Function):
    """nn.BatchNorm2d(train) -> (+ identity) -> ReLU -> nn.AvgPool2d(2) of the CLIP Bottleneck (coin/modeling/utils.py:77-90)
    as two HIP streams forward and two backward (coin_bn_stats / coin_bn_apply_fwd / coin_bn_bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, pool, stats_part=None, nbt=None):
        xn = _as_nhwc(x)
        rn = _as_nhwc(residual) if residual is not None else None
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        nv = _VALID_ROWS[0]
        nv = None if (nv is None or nv >= xn.shape[0]) else int(nv)
        if stats_part is not None:  # taken in the producing convolution's epilogue (coin_conv_gemm_bf16) over the same valid rows
            part, rows = stats_part
            assert rows == (xn.shape[0] if nv is None else nv) * xn.shape[1] * xn.shape[2]
            mean, rstd = K.conv_stats_finalize(part, xn.shape[0] * xn.shape[1] * xn.shape[2], xn.shape[3], rows, eps, momentum, running_mean, running_var, nbt)
        else:
            mean, rstd = K.bn_stats(xn if nv is None else xn[:nv], eps, momentum, running_mean, running_var, nbt)
        # The backward needs the ReLU decisions of a block that added a residual (pool 1: they were only in the saved output; pool 0:
        # in x + the residual input).  They are kept as ONE BIT per element, written by the apply pass: both backward passes then
        # read 1/16 of the bytes (the [2048, 7, 7, 2048] outputs of res5: 26 MB instead of 411 MB, twice per block).
        want_mask = bool(relu) and residual is not None and pool in (0, 1) and xn.is_cuda
        if want_mask:
            y, mask = K.bn_apply_fwd(xn, mean, rstd, g, b, rn, relu, pool, want_mask=True)
        else:
            y, mask = K.bn_apply_fwd(xn, mean, rstd, g, b, rn, relu, pool), None
        ctx.relu, ctx.pool, ctx.has_res, ctx.nv = relu, pool, residual is not None, nv
        # without a residual coin_bn_bwd recomputes the mask from x; pool 0 without ReLU still needs nothing but x
        keep = None if mask is not None else (rn if pool == 0 else (y if (relu and pool == 1 and residual is not None) else None))
        ctx.save_for_backward(xn, keep, mean, rstd, g, b, mask)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xn, y, mean, rstd, g, b, mask = ctx.saved_tensors
        dyn = _as_nhwc(dy)
        if dyn.dtype != xn.dtype:
            dyn = dyn.to(xn.dtype)
        want_dres = ctx.has_res and ctx.needs_input_grad[3]
        nv = ctx.nv
        if nv is None:
            dx, dgamma, dbeta, dres = K.bn_bwd(xn, dyn, y, mean, rstd, g, b, ctx.relu, ctx.pool, want_dres, mask=mask)
        else:  # sums and dx over the real rows (leading, contiguous); the filler rows get zero gradient
            hw = xn.shape[1] * xn.shape[2]
            dxv, dgamma, dbeta, dresv = K.bn_bwd(xn[:nv], dyn[:nv], y[:nv] if y is not None else None, mean, rstd, g, b, ctx.relu, ctx.pool,
                                                 want_dres, mask=mask[:nv * hw] if mask is not None else None)
            dx = torch.zeros_like(xn)
            dx[:nv] = dxv
            dres = None
            if dresv is not None:
                dres = torch.zeros_like(xn)
                dres[:nv] = dresv
        return (dx.permute(0, 3, 1, 2), dgamma, dbeta, dres.permute(0, 3, 1, 2) if dres is not None else None,
                None, None, None, None, None, None, None, None)


def bn_act(x: torch.Tensor, bn: torch.nn.BatchNorm2d, relu: bool, residual: Optional[torch.Tensor] = None, pool: int = 1,
           stats_part=None) -> torch.Tensor:
    """Train-mode BatchNorm with batch statistics (per GPU, as the reference) fused with the elementwise tail.
    x / residual / result: logical [N,C,H,W] in channels-last memory format.
    pool: 1 = none, 2 = nn.AvgPool2d(2) fused in, 0 = global spatial mean fused in (result [N,C,1,1])."""
    if bn.training:
        if bn.momentum is None:
            raise CoinHipError("cumulative-average BatchNorm (momentum=None) is not used by the reference")
        rm = bn.running_mean if bn.track_running_stats else None
        rv = bn.running_var if bn.track_running_stats else None
        nbt = bn.num_batches_tracked if bn.track_running_stats else None
        if nbt is not None and not nbt.is_cuda:
            nbt.add_(1)
            nbt = None
        # (on the device the counter is incremented by the launch that updates the running statistics: 42 `add_` launches per step before)
        return _BNAct.apply(x, bn.weight, bn.bias, residual, rm, rv, float(bn.momentum), float(bn.eps), bool(relu), int(pool), stats_part, nbt)
    # eval mode (teacher inference): a per-channel affine map with the running statistics
    if x.is_cuda and not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad)):
        # no gradient wanted: the fused apply kernel with (running_mean, 1/sqrt(running_var + eps)) as the statistics
        # (the constants -- running mean, 1 / sqrt(running_var + eps), scale, shift in fp32 -- are cached per module and follow every
        # write of their sources: `_bn_constants`.  Formed per call they were 2-4 small launches per norm: 110 per teacher pass.)
        mean, rstd, g, b = _bn_constants(bn)
        y = K.bn_apply_fwd(_as_nhwc(x), mean, rstd, g, b, _as_nhwc(residual) if residual is not None else None, relu, pool)
        return y.permute(0, 3, 1, 2)
    scale = bn.weight * (bn.running_var + bn.eps).rsqrt()
    shift = bn.bias - bn.running_mean * scale
    y = x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)
    if residual is not None:
        y = y + residual
    if relu:
        y = torch.relu(y)
    if pool == 0:
        return y.mean(dim=[2, 3], keepdim=True)
    return avg_pool2(y) if pool == 2 else y


_FROZEN_CONSTS: dict = {}
_DERIVED_CACHES.append(_FROZEN_CONSTS)


def _bn_constants(bn: torch.nn.Module):
    """(running_mean, 1 / sqrt(running_var + eps), weight, bias) of a norm in fp32, cached per module; rebuilt when one of the four sources
    was written (version bump or new storage) -- writers that bump no version (the EMA kernel) call `invalidate_storage`."""
    src = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in src)
    hit = _FROZEN_CONSTS.get(id(bn))
    if hit is None or hit[0] != key or hit[1]() is not bn:
        with torch.no_grad():
            consts = (bn.running_mean.float().contiguous(), (bn.running_var.float() + bn.eps).rsqrt().contiguous(),
                      bn.weight.float().contiguous(), bn.bias.float().contiguous())
        hit = (key, weakref.ref(bn, lambda _r, k=id(bn): _FROZEN_CONSTS.pop(k, None)), consts)
        _FROZEN_CONSTS[id(bn)] = hit
    return hit[2]


def frozen_bn_fusable(x: torch.Tensor, channels: int) -> bool:
    """The frozen stages in the throughput mode: no gradient flows through them and the channel count fits the streaming kernels'
    thread mapping (coin_bn_apply_fwd: 8 bf16 channels per lane, a row of channel groups divides or is a multiple of a workgroup)."""
    if not x.is_cuda or x.dtype != torch.bfloat16 or (torch.is_grad_enabled() and x.requires_grad):
        return False
    ncg = channels // 8
    return channels % 8 == 0 and (256 % ncg == 0 if ncg <= 256 else ncg % 256 == 0)


def frozen_bn_act(x: torch.Tensor, bn: torch.nn.Module, relu: bool, residual: Optional[torch.Tensor] = None, pool: int = 1) -> torch.Tensor:
    """detectron2 FrozenBatchNorm2d (+ identity) (+ ReLU) (+ nn.AvgPool2d(2)) of a frozen CLIP stage (coin/modeling/utils.py:77-90 with
    the norms converted, :243-284) as ONE pass over the convolution's output: the statistics slots of coin_bn_apply_fwd carry the
    frozen running statistics.  Replaces the library's bias add, ReLU, residual add and ReLU launches (4 passes) on the largest
    activations of the network.  Caller checked `frozen_bn_fusable`."""
    mean, rstd, g, b = _bn_constants(bn)
    y = K.bn_apply_fwd(_as_nhwc(x), mean, rstd, g, b, _as_nhwc(residual) if residual is not None else None, relu, pool)
    return y.permute(0, 3, 1, 2)


class _AvgPool2(Function):
    @staticmethod
    def forward(ctx, x):
        xn = _as_nhwc(x)
        ctx.shape = tuple(xn.shape)
        return K.avgpool2_fwd(xn).permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return K.avgpool2_bwd(_as_nhwc(dy), ctx.shape).permute(0, 3, 1, 2)


def avg_pool2(x: torch.Tensor) -> torch.Tensor:
    """nn.AvgPool2d(2) on a channels-last tensor."""
    return _AvgPool2.apply(x)


# --------------------------------------------------------------------------- box-head linear
class _LinearAct(Function):
    """nn.Linear (+ LeakyReLU) of FastRCNNOutputLayers (fast_rcnn.py:237-251,331-337) on the MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, alpha, out_dtype, wq):
        if wq is None:  # wq: the weight's compute-dtype shadow (the gradient still goes to the fp32 master `weight`)
            wq = weight.to(x.dtype) if weight.dtype != x.dtype else weight
        y = K.gemm_nt(x, wq.contiguous(), bias.float() if bias is not None else None, act, alpha, out_dtype=out_dtype)
        ctx.act, ctx.alpha, ctx.has_bias = act, alpha, bias is not None
        ctx.param = weakref.ref(weight)
        ctx.save_for_backward(x, wq, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wq, y = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if y is not None and y.dtype != dy.dtype:
            y = y.to(dy.dtype)
        n = dy.shape[1]
        dbias = torch.zeros(n, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
        if ctx.act != ACT_NONE or ctx.has_bias:
            dz, dbias = K.bias_act_bwd(dy, y, ctx.act, ctx.alpha, dbias=dbias if ctx.has_bias else torch.zeros(n, dtype=torch.float32, device=dy.device),
                                       want_dz=ctx.act != ACT_NONE)
            if dz is None:
                dz = dy
        else:
            dz = dy
        dx = dw = None
        kmult = 64 if x.dtype == torch.bfloat16 else 16
        if ctx.needs_input_grad[0]:
            # dX[M,K] = dZ[M,N] . W[N,K]  ==  gemm_nt(dZ, W^T[K,N]); contraction N padded to the MFMA K-step
            # [K, N]: the persistent data-gradient layout (ks = 1) where the shape allows it, else a transposed copy per call
            wt = dgrad_weight(ctx.param(), wq, 1) if (n % kmult == 0 and wq.is_contiguous()) else K.transpose2d(wq)
            if n % kmult:
                padn = kmult - n % kmult
                dzp = torch.cat([dz, dz.new_zeros(dz.shape[0], padn)], dim=1)
                wt = torch.cat([wt, wt.new_zeros(wt.shape[0], padn)], dim=1)
            else:
                dzp = dz
            dx = K.gemm_nt(dzp, wt)
        if ctx.needs_input_grad[1]:
            if x.dtype == torch.bfloat16 and K.conv_wgrad_ok(n, x.shape[1]) and x.shape[0] >= 128:
                # dW[N,K] = dZ^T . X: both operands row-major over the contraction index M -- the transposed-read kernel of the res5
                # weight gradients takes them as they are (no transposes, no zero padding of M)
                dw = K.conv_wgrad(dz, x)
            else:
                # dW[N,K] = dZ^T[N,M] . X[M,K]  ==  gemm_nt(dZ^T, X^T); contraction M zero-padded
                dzt = K.transpose2d(_pad_rows(dz, kmult))   # [N, Mp]
                xt = K.transpose2d(_pad_rows(x, kmult))     # [K, Mp]
                dw = K.gemm_nt(dzt, xt, out_dtype=torch.float32)
        return dx, dw, (dbias if ctx.has_bias else None), None, None, None, None


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE, alpha: float = 0.01,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    wq = _shadow_entry(weight, x.dtype).tensor if (weight.dtype != x.dtype and weight.is_cuda) else None
    return _LinearAct.apply(x.contiguous(), weight, bias, act, alpha, out_dtype or x.dtype, wq)


# --------------------------------------------------------------------------- cosine classifier
class _CosineLogits(Function):
    """FastRCNNOutputLayers.do_classify (fast_rcnn.py:343-346): L2-normalise both sides, dot, / logit_scale."""

    @staticmethod
    def forward(ctx, feats, text, inv_scale):
        scores, inv_norm = K.cosine_logits_fwd(feats, text, inv_scale)
        ctx.save_for_backward(feats, text, scores, inv_norm)
        ctx.inv_scale = inv_scale
        return scores

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        feats, text, scores, inv_norm = ctx.saved_tensors
        df, dt = K.cosine_logits_bwd(ds.float().contiguous(), feats, text, scores, inv_norm, ctx.inv_scale,
                                     need_text_grad=ctx.needs_input_grad[1])
        return df, dt, None


def cosine_logits(feats: torch.Tensor, text: torch.Tensor, inv_scale: float) -> torch.Tensor:
    return _CosineLogits.apply(feats.contiguous(), text.float().contiguous(), float(inv_scale))


# --------------------------------------------------------------------------- fused losses
class _ScalarLoss(Function):
    """Common shape: the kernel returns (loss, dloss/dinput for unit upstream); backward rescales."""

    @staticmethod
    def forward(ctx, inp, fn):
        loss, grad = fn(inp.detach())
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def mil_cross_entropy(x, target=None, labels=None, weights=None, avg_positives=False, reduction="mean"):
    """coin/utils/losses.py:13-34."""
    x = x.float().contiguous()
    return _ScalarLoss.apply(x, lambda t: K.mil_ce(t, target=target, labels=labels, weights=weights,
                                                   avg_positives=avg_positives, reduction=reduction))


def mil_focal_loss(x, alpha, target=None, labels=None, gamma=1.5, avg_positives=True, weights=None, reduction="mean"):
    """coin/utils/losses.py:36-73 (MILFocalLoss(class_num, alpha=classes_weight), gamma 1.5)."""
    x = x.float().contiguous()
    a = alpha.to(device=x.device, dtype=torch.float32).contiguous()
    return _ScalarLoss.apply(x, lambda t: K.mil_focal(t, a, target=target, labels=labels, gamma=gamma, avg_positives=avg_positives,
                                                      weights=weights, reduction=reduction))


def kl_div_from_logits(scores, q, row_mask=None):
    """KLDivLoss('mean')(log(softmax(scores)+1e-7), q)  (fast_rcnn.py:538,544)."""
    return _ScalarLoss.apply(scores.float().contiguous(), lambda t: K.kl_div(t, q.float().contiguous(), 0, row_mask))


def kl_div_from_probs(p, q, row_mask=None):
    """KLDivLoss('mean')(log(p+1e-7), q)  (fast_rcnn.py:526)."""
    return _ScalarLoss.apply(p.float().contiguous(), lambda t: K.kl_div(t, q.float().contiguous(), 1, row_mask))


def kl_div_binary(logits, q, row_mask):
    """RPN objectness distillation (rpn.py:331-335)."""
    return _ScalarLoss.apply(logits.float().contiguous(), lambda t: K.kl_div(t, q.float().contiguous(), 2, row_mask))


def box_reg_l1(proposals, gt_boxes, pred_deltas, gt_classes, num_fg_classes, weights, normalizer):
    """fast_rcnn.py:601-646."""
    return _ScalarLoss.apply(pred_deltas.float().contiguous(),
                             lambda t: K.box_reg_l1(proposals.float().contiguous(), gt_boxes.float().contiguous(), t,
                                                    gt_classes.contiguous(), num_fg_classes, weights, normalizer))


def l1_mean(a, b):
    """nn.L1Loss('mean') (fast_rcnn.py:351)."""
    return _ScalarLoss.apply(a.float().contiguous(), lambda t: K.l1_mean(t, b.detach().float().contiguous()))


class _RpnLosses(Function):
    """DualTeacherRPN.losses BCE + L1 (rpn.py:300-324); returns the two SUMS."""

    @staticmethod
    def forward(ctx, logits, deltas, labels, anchors, matched_gt, min_label):
        cls, loc, gl, gd = K.rpn_losses(logits.detach(), labels, deltas.detach(), anchors, matched_gt, min_label)
        ctx.save_for_backward(gl, gd)
        return cls, loc

    @staticmethod
    @once_differentiable
    def backward(ctx, gcls, gloc):
        gl, gd = ctx.saved_tensors
        return gl * gcls, gd * gloc, None, None, None, None


def rpn_losses(logits, deltas, labels, anchors, matched_gt, min_label=0):
    return _RpnLosses.apply(logits.float().contiguous(), deltas.float().contiguous(), labels.to(torch.int8).contiguous(),
                            anchors.float().contiguous(), matched_gt.float().contiguous(), int(min_label))
