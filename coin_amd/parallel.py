"""Data-parallel gradient averaging for the trainers (plain data parallelism over images, one process per GPU,
the only collective is the gradient all-reduce; the reference wraps the model in DistributedDataParallel, pre_train.py:59-62,
trainer.py:66-72).

Why not torch's DistributedDataParallel here: the step keeps its gradients "stolen" (``p.grad`` is whatever tensor autograd
produced; nothing is accumulated in place) and updates all ~170 tensors with ONE table-driven SGD launch.  DDP on top of that
copies every gradient into its buckets with one small kernel per parameter and rewrites the gradient pointers every step, which
measured 3-4 ms per step on a single rank (88 -> 81 views/s) before any byte crosses xGMI.  ``GradReducer`` does what the step
needs and nothing else:

  * the gradients live in a flat fp32 arena cut into slices of <= ``slice_mb`` (32 MiB: large enough for xGMI's per-link rings to
    run at bandwidth, small enough that the first slice leaves while backward is still running); slices follow the REVERSE
    parameter order, i.e. the order in which backward produces gradients (res5 / box head first);
  * a post-accumulate hook per parameter counts arrivals; when a slice is complete its gradients are packed with ONE
    multi-tensor copy and the slice is all-reduced asynchronously (RCCL: on the communicator's own stream, overlapping the rest
    of backward); slices are always launched in index order, so every rank issues the same sequence of collectives whatever
    its autograd graph looks like, and ``finalize()`` flushes slices whose parameters received no gradient with zeros;
  * after the pack ``p.grad`` is a view of the arena (the optimizer reads the averaged gradient there); the trainers drop the
    gradients every step (``set_to_none``), so autograd hands over fresh tensors and the pack copy is paid every step
    (0.8 ms of 40, measured with ``cfg.AMD.GRAD_ARENA``);
  * ONE backward per step: a parameter must not receive a second gradient between two ``finalize()`` calls (gradient
    accumulation over several backward calls would launch its slice with a partial sum) -- a second arrival raises;
  * the division by the world size is folded into the SGD launch (``inv_scale``), not a kernel of its own.
BatchNorm statistics, prototypes and the sampler RNG stay per rank, exactly as with ``broadcast_buffers=False`` in the reference.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Slice:
    __slots__ = ("params", "views", "flat", "arrived", "launched", "work", "events", "seen", "stamp")


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, slice_mb: float = 32.0):
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.slices: List[_Slice] = []
        self._slice_of = {}
        import os

        slice_mb = float(os.environ.get("COIN_REDUCER_SLICE_MB", slice_mb))   # measurements
        self._cap = max(int(slice_mb * (1 << 20) // 4), 1)
        # First pass: slices in REVERSE parameter order (roughly the order in which backward produces gradients).  Roughly is not good
        # enough for overlap: slices are launched strictly in index order, and the first slice then held the box predictor's text prompts,
        # whose gradient arrives LAST (the text encoder ran first in the forward, so the engine runs its backward last) -- every slice
        # waited for it and all collectives were issued after the backward's last kernel (round 6, tests/test_ddp_gpu.py).  The first
        # backward pass therefore records the order in which the gradients actually arrive, rank 0's order is agreed on by all ranks, and
        # the slices are rebuilt along it (`_reslice`, once, at the end of the first `finalize`).
        self._arrival: Optional[List[int]] = []
        self._build(list(reversed(self.params)))
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _build(self, ordered: List[torch.nn.Parameter]):
        self.slices = []
        self._slice_of = {}
        cur: List[torch.nn.Parameter] = []
        n = 0
        for p in ordered:
            if cur and n + p.numel() > self._cap:
                self._close(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self._close(cur)
        # one extra fp32 behind the LAST slice's gradients: a per-step flag that is summed over the ranks by that slice's own all-reduce
        # (`set_flag` / `flag`): "how many ranks had this kind of gradient" without a collective or a host round trip of its own
        self.flag = None        # (a parameter list without a trainable parameter has no slices and no flag: nothing to reduce)
        if self.slices:
            last = self.slices[-1]
            flat = torch.zeros(last.flat.numel() + 1, dtype=torch.float32, device=last.flat.device)
            off = 0
            views = []
            for p, v in zip(last.params, last.views):
                views.append(flat[off:off + p.numel()].as_strided(v.shape, v.stride()))
                off += p.numel()
            last.flat, last.views = flat, views
            self.flag = flat[-1:]
        self._next = 0          # next slice index to launch (collectives are issued in index order on every rank)

    def _reslice(self):
        """Once, after the first backward pass: rebuild the slices in the order the gradients arrived (parameters that received none keep
        their place at the end).  Every rank must cut the same slices: rank 0's order is broadcast.  The old arena stays alive through the
        p.grad views the optimizer is about to read; the next backward fills the new one."""
        pos = {pid: i for i, pid in enumerate(self._arrival)}
        self._arrival = None
        order = sorted(range(len(self.params)), key=lambda i: (pos.get(id(self.params[i]), 1 << 30), -i))
        if self.world_size > 1 and self.params:
            t = torch.tensor(order, dtype=torch.int64, device=self.params[0].device)
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            order = [int(i) for i in t.tolist()]
        old_flag = self.flag
        self._build([self.params[i] for i in order])
        if old_flag is not None and self.flag is not None:
            self.flag.copy_(old_flag)   # the caller reads this step's flag after finalize() returns

    def _close(self, params: List[torch.nn.Parameter]):
        s = _Slice()
        s.params = params
        total = sum(p.numel() for p in params)
        s.flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        s.views, off = [], 0
        for p in params:
            # same dense strides as the parameter (channels-last conv weights stay channels-last): the pack is a plain copy and the
            # optimizer sees the layout it expects
            s.views.append(s.flat[off:off + p.numel()].as_strided(p.shape, p.stride()) if _dense(p) else s.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        s.arrived, s.launched, s.work, s.events, s.seen, s.stamp = 0, False, None, [], set(), None
        idx = len(self.slices)
        self.slices.append(s)
        for p in params:
            self._slice_of[id(p)] = idx

    # ---- backward hooks
    # Autograd runs a backward node on the stream its forward ran on, and this package puts part of the forward on side streams (text
    # encoder, anchor labelling: OpenVocabularyRCNN._overlap_side_work).  A hook may therefore fire on ANY of those streams, and the
    # slice it completes holds gradients produced on others.  Every arrival leaves an event on its stream; the stream that packs a
    # slice waits for the events of that slice first.  (Found when GPU_MAX_HW_QUEUES was raised: with 4 hardware queues the streams
    # happened to serialise and the missing dependency did not show; with 8 the packed gradients were read too early -> NaN.)
    def deliver_early(self, p: torch.nn.Parameter):
        """p.grad is in place NOW (coin_amd.graphs: a chunk of a replayed backward has produced it): count the arrival at once, and swallow
        the engine's own call of the hook for this parameter later in the same backward pass (its accumulator node still runs, for the
        undefined gradient the replay returns)."""
        if self._early is None:
            self._early = set()
        self._early.add(id(p))
        self._arrive(p)

    _early = None

    def _on_grad(self, p: torch.nn.Parameter):
        if self._early and id(p) in self._early:
            self._early.discard(id(p))
            # The engine's accumulator node has run for a parameter that was handed over early.  It must have had nothing to add: if the
            # slice's collective is already launched and the arena was written since (the accumulator adds in place into p.grad = the arena
            # view), another pass's contribution reached this rank's copy AFTER the all-reduce -- the ranks would drift apart silently
            # (found with two ranks on one GPU; coin_amd.graphs.GraphedSegment.note_outside_use is what prevents it).  Fail loudly.
            s = self.slices[self._slice_of[id(p)]]
            if s.launched and s.stamp is not None and s.flat._version != s.stamp:
                raise RuntimeError("GradReducer: a parameter whose gradient was delivered early (a replayed step graph) received another "
                                   "contribution after its slice was all-reduced; the other pass must be announced with "
                                   "GraphedSegment.note_outside_use() before the backward")
            return
        self._arrive(p)

    def _arrive(self, p: torch.nn.Parameter):
        if self._arrival is not None:
            self._arrival.append(id(p))
        s = self.slices[self._slice_of[id(p)]]
        if id(p) in s.seen or s.launched:
            raise RuntimeError("GradReducer: a parameter received a second gradient before finalize() (one backward per step; "
                               "restrict extra backward calls to other parameters with backward(inputs=...))")
        s.seen.add(id(p))
        s.arrived += 1
        if p.grad is not None and p.grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(p.grad.device))
            s.events.append(ev)
        self._launch_ready()

    def _launch_ready(self):
        while self._next < len(self.slices) and self.slices[self._next].arrived >= len(self.slices[self._next].params):
            self._launch(self.slices[self._next])
            self._next += 1

    _pack_events: list = None

    @torch.no_grad()
    def _launch(self, s: _Slice):
        if s.events:
            cur = torch.cuda.current_stream(s.flat.device)
            for ev in s.events:
                cur.wait_event(ev)
            s.events = []
        have = [(v, p.grad) for v, p in zip(s.views, s.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for v, p in zip(s.views, s.params) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g.to(torch.float32) if g.dtype != torch.float32 else g for _, g in have])
        for v, p in zip(s.views, s.params):
            p.grad = v
        s.launched = True
        if self.world_size > 1 or _FORCE[0]:
            s.work = dist.all_reduce(s.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        s.stamp = s.flat._version   # (views share their base's version counter: any later in-place write of a p.grad of this slice moves it)
        if s.flat.is_cuda:   # whoever consumes the arena (finalize, on the caller's stream) waits for this pack
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(s.flat.device))
            if self._pack_events is None:
                self._pack_events = []
            self._pack_events.append(done)

    def set_flag(self, value: float) -> None:
        """This rank's contribution to `flag` (read it after `finalize()`: the sum over the ranks).  Call BEFORE the backward pass of the
        step: the last slice may be launched by a hook, on whichever stream that backward node runs."""
        if self.flag is None:   # no trainable parameter: nothing is reduced, nothing carries a flag
            return
        self.flag.fill_(float(value))
        if self.flag.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flag.device))
            self.slices[-1].events.append(ev)

    # ---- called by the trainer after backward
    def finalize(self) -> float:
        """Launch what is left (parameters without a gradient this step contribute zeros), wait for the collectives on the current
        stream and re-arm.  -> the factor the optimizer must apply to the summed gradients (1 / world size)."""
        while self._next < len(self.slices):
            self._launch(self.slices[self._next])
            self._next += 1
        if self._pack_events:
            cur = torch.cuda.current_stream()
            for ev in self._pack_events:
                cur.wait_event(ev)
            self._pack_events = []
        for s in self.slices:
            if s.work is not None:
                s.work.wait()
                s.work = None
            s.arrived, s.launched, s.events, s.stamp = 0, False, [], None
            s.seen.clear()
        self._next = 0
        if self._early:
            self._early.clear()
        if self._arrival is not None:   # the FIRST finalize of every rank, whatever arrived on it: the agreement is a collective
            import os

            if os.environ.get("COIN_REDUCER_RESLICE", "1") != "0":   # (A/B measurements)
                self._reslice()
            else:
                self._arrival = None
        return 1.0 / self.world_size

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


_FORCE = [False]  # single-rank dry run of the collective path (COIN_FORCE_DDP=1)


def force_collectives(flag: bool) -> None:
    _FORCE[0] = bool(flag)


def _dense(p: torch.Tensor) -> bool:
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Initial synchronisation of the replicas (what DistributedDataParallel's constructor does; pre_train.py:275-277,
    trainer.py:257-260 `_sync_params_and_buffers`): parameters AND buffers from rank `src`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        # one collective per dtype instead of one per tensor (~600 tiny broadcasts for the detector: each is a full ring hop over xGMI):
        # pack -> broadcast -> unpack.  Duplicated (tied) tensors are sent once.
        seen, by_dtype = set(), {}
        for t in list(module.parameters()) + list(module.buffers()):
            if id(t) in seen:
                continue
            seen.add(id(t))
            by_dtype.setdefault(t.dtype, []).append(t.data)
        for dtype, tensors in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in tensors]) if len(tensors) > 1 else tensors[0].reshape(-1).clone()
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))   # logical order on both sides: any memory format
                off += n
