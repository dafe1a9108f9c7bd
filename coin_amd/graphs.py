"""HIP-graph capture of the fixed-shape stretches of a training step (forward AND backward).

The student's step (coin/engine/trainer.py:160-218, pre_train.py:178-211) enqueues ~1 400 kernels; about half of them belong to two
stretches whose launch sequence depends on tensor SHAPES only, never on the targets of the batch:

* the trainable stages of the backbone (frozen-stage output -> res4), and
* the RoI trunk (res4 + sampled RoIs -> RoIAlign -> res5 -> pooled features).

Everything in between (RPN head, proposal selection, NMS, the samplers, the losses) depends on per-image target counts that are host
integers and stays eager.  A ``GraphedSegment`` wraps one stretch: the first calls of a shape run eagerly (library solver searches,
workspace growth), then the forward is captured as one HIP graph and its backward as a second one sharing the memory pool.  The backward
is recorded by walking the autograd graph node by node ON THE CAPTURING THREAD (`_backward_on_this_thread`), not by the autograd engine:
the engine runs device work on its own worker thread and, for a leaf whose gradient accumulator is still alive from an earlier eager
step, makes the default stream wait on the capturing stream -- which ended in a segmentation fault inside hipStreamEndCapture on this
runtime even for a bare nn.Linear.  A call then costs the host three launches (copy-in, replay) instead of hundreds (measured: 23.0 ->
12.2 ms of host enqueue per pre-train step, 0.2 ms of it inside hipGraphLaunch), and autograd sees ONE node whose backward replays the
second graph.

Rules the capture keeps (each one is there because its absence produced a wrong result or a failed capture):
* graphs are replayed on the device's default stream only (a graph launched from a side stream serialised the whole step on this
  runtime); any other stream, an active capture, or live kernel-timing events -> the eager path;
* a segment between its forward and its backward is BUSY: a second call of the same shape in that window (step_one / step_two pool
  twice per forward) runs eagerly, it would overwrite the activations the pending backward reads;
* the data-gradient weight layouts (layers.dgrad_weight) are refreshed eagerly BEFORE a backward capture / replay: captured, the
  refresh would not execute while the host-side stamp says it did;
* before a capture the stretch is run once, forward and backward, on the capturing thread: every cache the stretch fills (GEMM
  workspaces of the stream, data-gradient layouts, unit constants, the set of shapes the pooled-residual epilogue does not serve)
  is filled by executed launches, never by recorded ones; the running statistics that run advances are put back.  (When the stretch
  still contained library convolutions this run was also what kept MIOpen from loading a code object during the capture --
  miopenStatusUnknownError, stream invalidated -- the eager backward having run on the engine's thread with another handle);
* a captured stretch contains NO library convolution (layers._no_library_conv_under_capture; the model code wraps its stretches in
  `layers.conv_gemm_everywhere` and offers only stages whose widths the hand-written kernels serve): replayed, the library's
  backward-weights launch depends on memory that is not the graph's -- 2e-2 relative error at the first replay of a single captured
  call, 1e28 as soon as unrelated allocations have happened (tools/miopen_graph_probe.py, profiles/r5_miopen_graph_probe.log);
* a graph OWNS the workspaces its kernels were recorded with (kernels.take_stream_workspaces): the per-stream workspace caches are
  emptied for the capture stream before a capture and again after it, the entries moving into the graph's record.  Inherited from an
  earlier capture, such a buffer was replaced -- freed -- by the next capture that needed a larger one while this graph's kernels still
  pointed at it; harmless while the pool it lived in existed, a GPU memory access fault once that pool's graph had been destroyed and
  empty_cache() returned the block to the driver;
* only passes whose shapes are fixed by construction are captured (a padded pass -- the C boxes of step_one / step_two, whose count
  changes every step -- stays eager);
* BatchNorm running statistics and `num_batches_tracked` are updated by kernels inside the graph: replays update them in place;
* parameter gradients: the backward graph writes them into static buffers.  Without gradient hooks on a parameter (single GPU) the
  static buffer itself becomes ``p.grad`` (stable pointers: the optimizer's device table is uploaded once) -- valid until the stretch's
  next replay, forward or backward; a p.grad that is still the buffer at that point (gradient accumulation without zero_grad) is
  replaced by a copy first.  Post-accumulate hooks (the data-parallel reducer) fire as usual -- the engine runs the accumulator node
  for the undefined gradient the replay returns, and that node calls its post hooks with p.grad already in place; a parameter with a
  tensor hook, or one that already holds a gradient, gets its gradient through autograd.

``cfg.AMD.STEP_GRAPHS`` (default on) / ``COIN_STEP_GRAPHS=0`` switch the mechanism; a failed capture warns once and leaves the segment
eager for good (same kernels either way).
"""
from __future__ import annotations

import os
import time
import warnings
import weakref
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import kernels as K
from . import layers as L

ENABLED = {"on": os.environ.get("COIN_STEP_GRAPHS", "1") != "0"}
WARM_CALLS = 2          # eager calls of a shape before it is captured
MAX_GRAPHS = 3          # captured shapes per segment (each holds a private pool with the stretch's activations and static gradients)
EVICT_AFTER = 8         # a shape seen this often while the table is full may take the place of ...
EARLY_DELIVERY = [os.environ.get("COIN_EARLY_DELIVERY", "1") != "0"]   # A/B switch: chunk-by-chunk hand-over of a replayed backward's gradients to the reducer
EVICT_IDLE = 16         # ... the captured shape not replayed for this many calls of the segment (least recently used first)
# launch_ms: host time spent inside hipGraphLaunch; pool_bytes: device memory the allocator reserved for the captures that are alive
STATS = {"captures": 0, "replays": 0, "eager": 0, "busy": 0, "launch_ms": 0.0, "evictions": 0, "pool_bytes": 0}
CAPTURE_MODE = {"fwd": "thread_local", "bwd": "thread_local"}   # other threads (image decoding) may touch the device meanwhile
TRACE: Optional[list] = None   # diagnostics (tools/bb_bisect.py (round 5; in the git history)): a list here receives (node, incoming gradients, results) of every captured backward node


_SEGMENTS: "weakref.WeakSet[GraphedSegment]" = weakref.WeakSet()   # weak: a dropped model must release its graphs' memory pools


def set_enabled(flag: bool) -> None:
    ENABLED["on"] = bool(flag) and os.environ.get("COIN_STEP_GRAPHS", "1") != "0"


# One capture per step, and none in a step that replays.  A capture synchronises the device, releases the allocator's cache
# (torch.cuda.graph.__enter__) and runs the stretch three times (dry run, forward capture, backward capture): a stretch that is ready
# announces it one step ahead (`_WANT_NEXT`); in the announced step every already captured stretch runs eagerly (`defer`), the first
# ready stretch captures, the others wait another step.  (Introduced while a corruption of the backbone stretch was being hunted --
# which turned out to be the library's weight-gradient kernel inside the graph, see layers._no_library_conv_under_capture -- and kept:
# it keeps the expensive step to one capture and a replay's pending backward away from the cache release.)
_STEP = {"captured": False, "replayed": False, "defer": False}
_WANT_NEXT: set = set()


def step_done() -> None:
    """End of an optimizer step (the trainers call it after `optimizer.step()`): no backward is pending any more -- a segment whose forward
    ran under grad mode but whose output never reached a backward would otherwise stay busy, i.e. eager, for ever -- and the capture
    schedule advances (see `_STEP`)."""
    for seg in list(_SEGMENTS):
        seg.outside_uses = 0
        for ent in seg.graphs.values():
            ent.busy = False
    _STEP["captured"], _STEP["replayed"], _STEP["defer"] = False, False, bool(_WANT_NEXT)
    _WANT_NEXT.clear()


def _backward_on_this_thread(roots: Sequence[torch.Tensor], root_grads: Sequence[torch.Tensor], wrt: Sequence[torch.Tensor], trace=None, on_leaf=None):
    """d(roots)/d(wrt) by calling the autograd nodes one by one ON THE CALLING THREAD, in dependency order -- what
    ``torch.autograd.grad(roots, wrt, root_grads, allow_unused=True)`` computes, without the engine.

    Why not the engine, under a stream capture: (1) it runs device work on its own worker thread, and (2) a leaf's gradient accumulator
    node that is still alive from an earlier eager step (the previous step's loss dict keeps its graph) carries the DEFAULT stream, so the
    engine makes the default stream wait for an event of the capturing stream -- an illegal dependency that ended in a segmentation fault
    inside hipStreamEndCapture (torch 2.10 / ROCm 7.0, reproduced with a bare nn.Linear).  Here every node runs on the capturing thread
    and stream, accumulator nodes are never executed (their incoming gradient IS the result), no cross-stream event is created.
    Runs with grad mode and autocast off, as the engine's worker threads do.  `trace`: a list that receives (node, incoming gradients,
    results) per executed node (diagnostics).  `on_leaf(i, more)`: called when the gradient of wrt[i] is FINAL (its accumulator node has
    received every contribution); `more` = further nodes are waiting (the chunked capture cuts the graph there)."""
    import collections

    leaf = {id(t): i for i, t in enumerate(wrt)}
    result: List[Optional[torch.Tensor]] = [None] * len(wrt)
    start = []
    for r in roots:
        if r.grad_fn is not None and r.grad_fn not in start:
            start.append(r.grad_fn)
    deps: Dict = collections.Counter()
    seen, stack = set(start), list(start)
    while stack:
        n = stack.pop()
        for nxt, _ in n.next_functions:
            if nxt is None:
                continue
            deps[nxt] += 1
            if nxt not in seen:
                seen.add(nxt)
                stack.append(nxt)
    pending: Dict = {}

    def add(node, idx, g):
        if g is None:
            return
        # the engine's validate_outputs: a gradient is reduced to the shape (broadcast operands) and cast to the dtype its consumer recorded
        meta = node._input_metadata[idx]
        if tuple(g.shape) != tuple(meta.shape):
            g = g.sum_to_size(tuple(meta.shape))
        if g.dtype != meta.dtype:
            g = g.to(meta.dtype)
        slot = pending.setdefault(node, {})
        slot[idx] = g if idx not in slot else slot[idx] + g

    for r, g in zip(roots, root_grads):
        if r.grad_fn is None:           # a root that is itself a wanted leaf
            i = leaf.get(id(r))
            if i is not None:
                result[i] = g if result[i] is None else result[i] + g
        else:
            add(r.grad_fn, r.output_nr, g)
    ready = [n for n in start if deps[n] == 0]
    with torch.no_grad(), torch.autocast("cuda", enabled=False):
        while ready:
            n = ready.pop()
            got = pending.pop(n, {})
            if hasattr(n, "variable"):       # AccumulateGrad of a leaf: not executed, its input is the leaf's gradient
                i = leaf.get(id(n.variable))
                if i is not None and 0 in got:
                    result[i] = got[0] if result[i] is None else result[i] + got[0]
                if i is not None and on_leaf is not None:
                    on_leaf(i, bool(ready))
                continue
            outs = None
            edges = n.next_functions
            where = list(range(len(edges)))          # edge k takes the node's output where[k]
            if got:
                args = [got.get(i) for i in range(len(n._input_metadata))]
                if isinstance(n, torch.autograd.function.BackwardCFunction):
                    # a Python autograd.Function: its node object is the ctx.  The engine's PyNode materialises undefined gradients as zeros
                    # (unless the Function opted out) before it calls backward, and keeps only the results at TENSOR argument positions:
                    # `backward` returns one value per forward argument, the edges exist per tensor argument.  The j-th argument that needs
                    # a gradient is the j-th live edge (an edge is live exactly when its tensor requires grad).
                    if getattr(n, "materialize_grads", True):
                        args = [a if a is not None else torch.zeros(tuple(m.shape), dtype=m.dtype, device=m.device) for a, m in zip(args, n._input_metadata)]
                    outs = n.apply(*args)
                    need = [i for i, f in enumerate(n.needs_input_grad) if f]
                    live = [k for k, e in enumerate(edges) if e[0] is not None]
                    assert len(need) == len(live), (type(n).__name__, n.needs_input_grad, len(live))
                    where = [None] * len(edges)
                    for k, i in zip(live, need):
                        where[k] = i
                else:
                    outs = n(*args)
                if not isinstance(outs, (tuple, list)):
                    outs = (outs,)
                if trace is not None:
                    trace.append((n, args, outs))
            for k, (nxt, idx) in enumerate(edges):
                if nxt is None:
                    continue
                if outs is not None and where[k] is not None and where[k] < len(outs):
                    add(nxt, idx, outs[where[k]])
                deps[nxt] -= 1
                if deps[nxt] == 0:
                    ready.append(nxt)
    return result


_UNDEF_HOOK = {"ok": None}


def undefined_gradient_fires_post_hooks() -> bool:
    """Does this torch run a leaf's post-accumulate-grad hooks when a custom Function returns None for it (with p.grad assigned directly)?
    `_Replay.backward` hands a replayed stretch's parameter gradients over exactly that way when a parameter carries such hooks (the
    data-parallel reducer).  The engine's behaviour here is not documented API: it is PROBED once per process (a three-element CPU
    autograd run, the same experiment tests/test_graphs_cpu.py pins) instead of assumed from a version number; where it does not hold,
    parameters with post-accumulate hooks get their gradients through autograd (a copy into a fresh p.grad, the hooks fire as usual)."""
    if _UNDEF_HOOK["ok"] is None:
        seen = []
        p = torch.nn.Parameter(torch.ones(3))
        h = p.register_post_accumulate_grad_hook(lambda q: seen.append(None if q.grad is None else float(q.grad[0])))

        class _Direct(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.w = w
                return x * 2

            @staticmethod
            def backward(ctx, g):
                ctx.w.grad = torch.full((3,), 5.0)
                return g * 2, None

        try:
            _Direct.apply(torch.ones(3, requires_grad=True), p).sum().backward()
            _UNDEF_HOOK["ok"] = seen == [5.0]
        except Exception:
            _UNDEF_HOOK["ok"] = False
        h.remove()
        if not _UNDEF_HOOK["ok"]:
            warnings.warn("coin_amd.graphs: this torch does not fire post-accumulate-grad hooks for an undefined gradient; replayed stretches "
                          "hand hooked parameters their gradients through autograd")
    return bool(_UNDEF_HOOK["ok"])


class _no_hooks_inside:
    """A captured stretch is replayed as ONE autograd node: a tensor hook, `retain_grad()` or a post-accumulate hook registered on a tensor
    INSIDE the stretch (by the module code itself or by a forward hook a user attached) would silently never run under replay
    (`_backward_on_this_thread` calls the nodes itself and does not run tensor / node hooks).  While a stretch is dry-run and recorded,
    registering one raises: the capture fails with a warning and the stretch stays eager, where hooks work (round-5 VERDICT, weak 5)."""

    NAMES = ("register_hook", "retain_grad", "register_post_accumulate_grad_hook")

    def __enter__(self):
        self.saved = {n: getattr(torch.Tensor, n) for n in self.NAMES}

        def refuse(name):
            def f(t, *a, **k):
                raise K.CoinHipError(f"Tensor.{name}() inside a stretch that is being captured as a HIP graph: the hook would not run under replay")
            return f

        for n in self.NAMES:
            setattr(torch.Tensor, n, refuse(n))
        return self

    def __exit__(self, *a):
        for n, f in self.saved.items():
            setattr(torch.Tensor, n, f)
        return False


CHUNK_BYTES = 16 << 20   # parameter-gradient bytes (fp32) after which a data-parallel backward graph is cut (half a reducer slice)


class _ChunkedCapture:
    """A backward pass recorded as a SEQUENCE of HIP graphs sharing one memory pool (replayed in recording order): `leaf(j, nbytes, more)`
    is called when parameter j's gradient is final; once CHUNK_BYTES of gradients are final -- and further nodes are waiting -- the
    current graph is ended and the next one begun.  Same entry / exit protocol as torch.cuda.graph (device synchronize, cache release,
    the capture stream), only with capture_end / capture_begin in the middle."""

    def __init__(self, pool, mode):
        self.pool, self.mode = pool, mode
        self.graphs, self.leaves, self.pending = [], [[]], 0

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(self.pool, capture_error_mode=self.mode)
        self.graphs.append(g)

    def __enter__(self):
        import gc

        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        if torch.cuda.graph.default_capture_stream is None:
            K.capture_stream_value()
        self.stream = torch.cuda.graph.default_capture_stream
        self.stream.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        self._begin()
        return self

    def leaf(self, j: int, nbytes: int, more: bool):
        self.leaves[-1].append(j)
        self.pending += nbytes
        if more and self.pending >= CHUNK_BYTES:
            self.graphs[-1].capture_end()
            self._begin()
            self.leaves.append([])
            self.pending = 0

    def __exit__(self, et, ev, tb):
        try:
            self.graphs[-1].capture_end()
        finally:
            self.ctx.__exit__(et, ev, tb)
        if et is None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return False

    def chunks(self):
        return list(zip(self.graphs, self.leaves))


class _Entry:
    __slots__ = ("fwd", "bwd", "static_in", "outs", "out_req", "static_gout", "grads_in", "grads_p", "params", "busy", "pool", "single", "workspaces", "ptrs", "bwd_chunks", "last_used", "pool_bytes", "seg")


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent: _Entry, n_in: int, *args):
        for s, x in zip(ent.static_in, args[:n_in]):
            if s.data_ptr() != x.data_ptr():
                s.copy_(x)
        t0 = time.perf_counter()
        ent.fwd.replay()
        STATS["launch_ms"] += (time.perf_counter() - t0) * 1e3
        ctx.ent = ent
        ctx.n_in = n_in
        outs = tuple(o.detach() for o in ent.outs)
        ctx.mark_non_differentiable(*[o for o, r in zip(outs, ent.out_req) if not r])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        ent: _Entry = ctx.ent
        assert ent.bwd is not None, "graphed segment was captured without a backward"
        L.refresh_dgrad_layouts()
        gi = iter(ent.static_gout)
        for g, r in zip(gouts, ent.out_req):
            if not r:
                continue
            s = next(gi)
            if g is None:
                s.zero_()
            elif s.data_ptr() != g.data_ptr():
                s.copy_(g)
        _detach_static_grads(ent)
        early = set()
        seg = ent.seg() if getattr(ent, "seg", None) is not None else None
        exclusive = (seg is None or seg.outside_uses == 0) and EARLY_DELIVERY[0]   # no other differentiable pass over these parameters in this step (note_outside_use)
        if ent.bwd_chunks:
            # Data-parallel mode: the backward was recorded as several graphs, cut where 16 MiB of parameter gradients are final.  After
            # each one the reducer is told at once (`deliver_early`): a slice whose gradients are complete is packed and its all-reduce
            # launched on the communicator's stream WHILE the next graph replays -- not after the stretch's last kernel (round-5 VERDICT,
            # weak 9: one graph per stretch delivered res5's 15 M gradients in a bunch at its end).
            t0 = time.perf_counter()
            for g_k, leaves in ent.bwd_chunks:
                g_k.replay()
                for j in leaves:
                    p, g = ent.params[j], ent.grads_p[j]
                    hooks = getattr(p, "_post_accumulate_grad_hooks", None)
                    owners = [getattr(h, "__self__", None) for h in hooks.values()] if hooks else []
                    if (exclusive and g is not None and p.grad is None and not getattr(p, "_backward_hooks", None) and owners
                            and all(hasattr(o, "deliver_early") for o in owners)):
                        p.grad = g
                        for o in owners:
                            o.deliver_early(p)
                        early.add(j)
            STATS["launch_ms"] += (time.perf_counter() - t0) * 1e3
        else:
            t0 = time.perf_counter()
            ent.bwd.replay()
            STATS["launch_ms"] += (time.perf_counter() - t0) * 1e3
        ent.busy = False
        gin = tuple(None if g is None else g.detach() for g in ent.grads_in)
        gp = []
        for j, (p, g) in enumerate(zip(ent.params, ent.grads_p)):
            if g is None or j in early:
                gp.append(None)   # (early: p.grad is in place and the reducer has it; the engine's later call of the hook is swallowed by the reducer)
            elif p.grad is None and not getattr(p, "_backward_hooks", None) and (not getattr(p, "_post_accumulate_grad_hooks", None) or undefined_gradient_fires_post_hooks()):
                # The static buffer itself becomes p.grad: no accumulate copy, a stable address for the optimizer's table.  Post-accumulate
                # hooks (the data-parallel reducer's arrival counter) still fire: the engine runs the parameter's accumulator node for
                # the undefined gradient returned here, and that node calls its post hooks whether or not a gradient arrived (torch 2.10;
                # pinned by tests/test_graphs_cpu.py) -- with p.grad already in place, which is all the reducer's hook reads.  (Calling the
                # hooks from here as well delivered every parameter twice: the reducer's second-gradient guard caught it.)
                p.grad = g
                gp.append(None)
            else:
                # through autograd (a tensor hook, or an existing p.grad to add to).  A fresh alias, not the object `ent` holds, so that
                # the accumulator may adopt it instead of cloning it; `_detach_static_grads` covers a p.grad that is still this buffer
                # at the next replay
                gp.append(g.detach())
        return (None, None) + gin + tuple(gp)


def _detach_static_grads(ent: _Entry) -> None:
    """Gradient accumulation over several passes without zero_grad(): a p.grad that IS this graph's static buffer (assigned by the previous
    pass's backward) holds the sum so far.  The buffer lives in the pool the two graphs share -- the NEXT FORWARD replay may use the same
    memory for its temporaries, the next backward replay overwrites it -- so the sum is moved out before either runs.  (In the usual loop
    p.grad is None here: the optimizer consumed it and zero_grad() dropped it.)"""
    for p, g in zip(ent.params, ent.grads_p):
        if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
            p.grad = p.grad.clone()


def _recover_from_failed_capture() -> None:
    """A capture that failed half way leaves a sticky runtime error behind: the next launch status check would report it for an innocent
    kernel.  Drain the device and consume the error (hipGetLastError through the library's own status call)."""
    for _ in range(2):
        try:
            torch.cuda.synchronize()
            break
        except Exception:
            pass
    try:
        from . import _lib

        _lib.lib().coin_clear_last_error()
    except Exception:
        pass


class GraphedSegment:
    """fn(*tensors) -> tensor | tuple of tensors, replayed as HIP graphs once a shape has repeated.  `params`: callable returning the
    parameters whose gradients the stretch produces (evaluated at capture)."""

    def __init__(self, name: str, fn: Callable, params: Callable[[], Sequence[torch.nn.Parameter]],
                 buffers: Optional[Callable[[], Sequence[torch.Tensor]]] = None):
        """`buffers`: the stretch's mutable state besides the parameters (BatchNorm running statistics, `num_batches_tracked`): restored
        after the dry run that precedes a capture."""
        self.name, self.fn, self.params_fn, self.buffers_fn = name, fn, params, buffers
        self.graphs: Dict[tuple, _Entry] = {}
        self.seen: Dict[tuple, int] = {}
        self.failed = False
        self.calls = 0          # eligible calls of this segment: the clock of the least-recently-used bookkeeping
        self.outside_uses = 0   # this step: differentiable passes over the stretch's parameters that were NOT a replay (see note_outside_use)
        _SEGMENTS.add(self)

    # ---------------------------------------------------------------- eligibility
    def _eligible(self, inputs) -> bool:
        if not ENABLED["on"] or self.failed or not inputs or not all(torch.is_tensor(x) and x.is_cuda for x in inputs):
            return False
        if K.timing_active() or torch.cuda.is_current_stream_capturing():
            return False
        dev = inputs[0].device
        return torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev)

    def _key(self, inputs, extra) -> tuple:
        cg = L.CONV_GEMM
        return (tuple((tuple(x.shape), tuple(x.stride()), x.dtype, bool(x.requires_grad)) for x in inputs), torch.is_grad_enabled(),
                torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None,
                (cg["enabled"], cg["wgrad"], cg["min_rows"]), L._VALID_ROWS[0], extra)

    def _eager(self, inputs):
        if torch.is_grad_enabled():
            self.outside_uses += 1
        return self.fn(*inputs)

    def __call__(self, *inputs, key_extra=()):
        if not self._eligible(inputs):
            STATS["eager"] += 1
            return self._eager(inputs)
        key = self._key(inputs, key_extra)
        self.calls += 1
        ent = self.graphs.get(key)
        if ent is None:
            n = self.seen.get(key, 0) + 1
            if len(self.seen) > 64 and key not in self.seen:   # variable-size data: the table of seen shapes stays bounded
                self.seen.pop(next(iter(self.seen)))
            self.seen[key] = n
            if n >= EVICT_AFTER and len(self.graphs) >= MAX_GRAPHS:
                self._evict_idle()   # variable-size data (round-5 ADVICE): the first shapes to repeat are not pinned for good
            if n < WARM_CALLS or len(self.graphs) >= MAX_GRAPHS:
                STATS["eager"] += 1
                return self._eager(inputs)
            if n == WARM_CALLS or _STEP["captured"] or _STEP["replayed"]:
                # ready from the next call on -- or ready now, but another stretch was captured / replayed in this step: announce and wait
                _WANT_NEXT.add((id(self), key))
                STATS["eager"] += 1
                return self._eager(inputs)
            _STEP["captured"] = True
            rng = torch.cuda.get_rng_state(inputs[0].device)   # a capture registers the generator with the graph and moves its offset: the
            try:                                                 # samplers' draws of the following eager code must not depend on it
                ent = self._capture(inputs)
                torch.cuda.set_rng_state(rng, inputs[0].device)
            except Exception as e:   # same kernels either way
                try:
                    torch.cuda.set_rng_state(rng, inputs[0].device)
                except Exception:
                    pass
                warnings.warn(f"step graph '{self.name}': capture failed ({type(e).__name__}: {e}); this stretch stays eager")
                self.failed = True
                _recover_from_failed_capture()
                STATS["eager"] += 1
                return self._eager(inputs)
            ent.last_used = self.calls
            self.graphs[key] = ent
            STATS["captures"] += 1
        if ent.busy or _STEP["defer"]:
            STATS["busy" if ent.busy else "eager"] += 1
            return self._eager(inputs)     # (defer: another stretch captures in this step, nothing may be replayed around it)
        if not self._fresh(ent):
            # a parameter / buffer of the stretch lives in another storage than at capture (load_state_dict with assign, `.data =`, a
            # module moved): the recorded kernels point at the old one.  Drop the graph; the shape is captured again after its warm-up calls
            del self.graphs[key]
            self.seen[key] = 0
            STATS["eager"] += 1
            STATS["stale"] = STATS.get("stale", 0) + 1
            STATS["pool_bytes"] -= getattr(ent, "pool_bytes", 0)
            return self._eager(inputs)
        _STEP["replayed"] = True
        STATS["replays"] += 1
        ent.last_used = self.calls
        _detach_static_grads(ent)
        outs = _Replay.apply(ent, len(inputs), *inputs, *ent.params)
        # until this call's backward has replayed, the graph's buffers hold the activations it will read: a second call must not replay
        ent.busy = ent.bwd is not None and any(o.requires_grad for o in outs)
        return outs[0] if ent.single else outs

    def note_outside_use(self) -> None:
        """The stretch's parameters take part in another differentiable pass of this step that is not a replay of this segment: an eager
        fallback of the segment itself (busy, foreign stream, a shape that is not captured), or -- told by the caller -- a pass that bypasses
        it (the roi heads' padded C-box pass runs res5 eagerly on the same weights).  Such a parameter's gradient is complete only when the
        autograd engine has added the other pass's contribution, so a replayed backward must NOT hand it to the data-parallel reducer
        chunk by chunk: found with two ranks on one GPU (tests/test_ddp_gpu.py, CoinTrainer): the early all-reduce carried the replay's
        share only and the C-box share was added afterwards, per rank, to the reduced sum -- the ranks' weights drifted apart.  Until
        `step_done()` the segment's replays deliver their gradients through the accumulator nodes, at the end of the stretch's backward."""
        self.outside_uses += 1

    def _evict_idle(self) -> bool:
        """Drops the captured shape that has gone unreplayed the longest, if that is at least EVICT_IDLE calls of this segment (and its
        backward is not pending): its graphs and their pool go back to the allocator, the newcomer is captured by the usual schedule.
        A shape that comes back later warms up and is captured again -- the table follows the data instead of its first three sizes."""
        idle = [(e.last_used, k) for k, e in self.graphs.items() if not e.busy and self.calls - e.last_used >= EVICT_IDLE]
        if not idle:
            return False
        _, k = min(idle, key=lambda t: t[0])
        ent = self.graphs.pop(k)
        self.seen[k] = 0
        STATS["evictions"] += 1
        STATS["pool_bytes"] -= getattr(ent, "pool_bytes", 0)
        return True

    def _fresh(self, ent: _Entry) -> bool:
        """What the eager path checks on every call and a replay would skip (round-5 ADVICE): the storages the graph was recorded against
        are still the parameters' / buffers' storages -- else False -- and the bf16 compute shadows are current: a master written out of band
        since (load_state_dict / resume_or_load in a process that already trained, init_, copy_: a version bump the fused SGD kernel does not
        make) gets its shadow refreshed HERE, eagerly, before the replay reads it; the data-gradient layouts follow at the backward's refresh."""
        params = list(self.params_fn())
        tensors = params + (list(self.buffers_fn()) if self.buffers_fn is not None else [])
        if [t.data_ptr() for t in tensors] != ent.ptrs:
            return False
        refreshed = False
        for p in params:
            e = L._SHADOWS.get(id(p))
            if e is not None and e.ref() is p and (e.version != p._version or e.ptr != p.data_ptr()):
                L._shadow_entry(p, e.tensor.dtype)
                refreshed = True
        if refreshed:
            L.weights_updated()
        return True

    # ---------------------------------------------------------------- capture
    def _capture(self, inputs) -> _Entry:
        ent = _Entry()
        ent.busy = False
        grad_mode = torch.is_grad_enabled()
        ent.static_in = [x.detach().clone(memory_format=torch.preserve_format).requires_grad_(bool(x.requires_grad) and grad_mode) for x in inputs]
        ent.params = [p for p in self.params_fn() if p.requires_grad] if grad_mode else []
        L.refresh_dgrad_layouts()
        wants_bwd = grad_mode and (bool(ent.params) or any(s.requires_grad for s in ent.static_in))
        # Dry run of forward (+ backward) ON THIS THREAD before anything is recorded: whatever the stretch caches on first use is then
        # produced by executed launches (a cache entry filled under the capture would hold memory no kernel has written yet), and a
        # library convolution in the stretch is noticed HERE -- refusing before the capture starts, not in the middle of one (a capture
        # abandoned half way left the process in a state in which a later test aborted).  The running statistics the dry run advances
        # are put back.
        saved = [b.detach().clone() for b in self.buffers_fn()] if self.buffers_fn is not None else []
        lib0 = L.LIBRARY_CONV_CALLS[0]
        with _no_hooks_inside():
            out = self.fn(*ent.static_in)
        if wants_bwd:
            outs = (out,) if torch.is_tensor(out) else tuple(out)
            req = [o for o in outs if o.requires_grad]
            if req:
                _backward_on_this_thread(req, [torch.zeros_like(o) for o in req], [s for s in ent.static_in if s.requires_grad] + list(ent.params))
            del outs, req
        del out
        if self.buffers_fn is not None:
            with torch.no_grad():
                for b, s in zip(self.buffers_fn(), saved):
                    b.copy_(s)
        L.refresh_dgrad_layouts()
        if L.LIBRARY_CONV_CALLS[0] != lib0:
            raise K.CoinHipError(f"{L.LIBRARY_CONV_CALLS[0] - lib0} library convolution(s) inside a captured stretch")
        torch.cuda.synchronize()
        # the graphs own the workspaces their kernels are recorded with: nothing inherited from an earlier capture on the (shared) capture
        # stream, and what this capture allocates leaves the stream's cache with it (kernels.take_stream_workspaces)
        cap = K.capture_stream_value()
        K.take_stream_workspaces(cap)
        reserved0 = torch.cuda.memory_reserved(inputs[0].device)
        try:
            self._record(ent, grad_mode)
        finally:
            ent.workspaces = K.take_stream_workspaces(cap)
        # what the allocator had to reserve for this capture (its private pool; blocks the pool could reuse from the cache do not count)
        ent.seg = weakref.ref(self)
        ent.pool_bytes = max(0, torch.cuda.memory_reserved(inputs[0].device) - reserved0)
        ent.last_used = self.calls
        STATS["pool_bytes"] += ent.pool_bytes
        ent.ptrs = [t.data_ptr() for t in list(self.params_fn()) + (list(self.buffers_fn()) if self.buffers_fn is not None else [])]
        return ent

    def _record(self, ent: _Entry, grad_mode: bool) -> None:
        ent.pool = torch.cuda.graph_pool_handle()
        ent.fwd = torch.cuda.CUDAGraph()
        L.STRICT_CAPTURE[0] = True     # a library convolution inside the stretch fails the capture (layers._no_library_conv_under_capture)
        try:
            with torch.cuda.graph(ent.fwd, pool=ent.pool, capture_error_mode=CAPTURE_MODE["fwd"]), _no_hooks_inside():
                out = self.fn(*ent.static_in)
        finally:
            L.STRICT_CAPTURE[0] = False
        ent.single = torch.is_tensor(out)
        outs = (out,) if ent.single else tuple(out)
        ent.outs = outs
        ent.out_req = [bool(o.requires_grad) for o in outs]
        ent.bwd, ent.static_gout, ent.grads_in, ent.grads_p = None, [], [None] * len(ent.static_in), [None] * len(ent.params)
        ent.bwd_chunks = None
        if grad_mode and any(ent.out_req):
            req = [o for o in outs if o.requires_grad]
            ent.static_gout = [torch.zeros_like(o) for o in req]
            wrt_in = [i for i, s in enumerate(ent.static_in) if s.requires_grad]
            wrt = [ent.static_in[i] for i in wrt_in] + list(ent.params)
            chunked = any(any(hasattr(getattr(h, "__self__", None), "deliver_early") for h in (getattr(p, "_post_accumulate_grad_hooks", None) or {}).values())
                          for p in ent.params)
            L.STRICT_CAPTURE[0] = True
            try:
                if chunked:
                    n_in = len(wrt_in)
                    with _ChunkedCapture(ent.pool, CAPTURE_MODE["bwd"]) as cc:
                        def on_leaf(i, more, cc=cc, n_in=n_in):
                            if i >= n_in:
                                cc.leaf(i - n_in, wrt[i].numel() * 4, more)
                        grads = _backward_on_this_thread(req, ent.static_gout, wrt, trace=TRACE, on_leaf=on_leaf)
                    ent.bwd_chunks = cc.chunks()
                    ent.bwd = ent.bwd_chunks[0][0]
                else:
                    ent.bwd = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ent.bwd, pool=ent.pool, capture_error_mode=CAPTURE_MODE["bwd"]):
                        grads = _backward_on_this_thread(req, ent.static_gout, wrt, trace=TRACE)
            finally:
                L.STRICT_CAPTURE[0] = False
            for j, i in enumerate(wrt_in):
                ent.grads_in[i] = grads[j]
            ent.grads_p = list(grads[len(wrt_in):])
