"""The package's side streams, one per ROLE and device, created together, in a fixed order, the first time any of them is asked for.

Why: the HIP runtime binds a stream to one of a few hardware queues when the stream is created / first used, and two streams that share
a queue serialise.  Which queue a stream gets depends on how many streams the PROCESS made before it -- so the same CoinTrainer step ran at
52 ms in a fresh process and at 63-64 ms in a process where a PRETrainer had lived first (round 5; round 6 sweep
`tools/td_stream_sweep.sh`: 50-54 ms with 0, 1, 2, 4, 5, 6 or 7 streams used before the trainer, 64 ms with exactly 3).  With the roles
below every process -- whichever trainer comes first, however many trainers it builds -- has the same four streams in the same creation
order, made before any torch-internal one (graph warm-up streams, RCCL's), and a second trainer re-uses the first one's streams instead
of drawing new ones.  Sharing a role between two objects is safe: a stream is an ordering domain, its users wait on events, not on
each other's identity.

  side     pre-train: text encoder + anchor labelling beside the backbone (OpenVocabularyRCNN._overlap_side_work)
  look     pre-train: the NEXT batch's frozen stem beside the proposal chain (_launch_lookahead)
  capture  torch.cuda.graph's capture stream (coin_amd.graphs, the teacher's and the text encoder's graphs)
  teacher  targetDET: the EMA teacher's pass + matching beside the student's step (CoinTrainer)
"""
from __future__ import annotations

from typing import Dict

import torch

# Creation order = the order in which the runtime hands out its hardware queues (4 by default; the DEFAULT stream holds the first one, and
# the fifth stream of a process wraps around onto it).  `teacher` first: it is the one stream that must never share a queue with the
# default stream (its whole point is to run beside the student's step); `capture` last: what wraps onto the default stream's queue is the
# stream that is only used while a graph is being recorded, when nothing else runs.
ROLES = ("teacher", "side", "look", "capture")
_STREAMS: Dict[int, Dict[str, "torch.cuda.Stream"]] = {}


def role_stream(device, role: str) -> "torch.cuda.Stream":
    assert role in ROLES, role
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    table = _STREAMS.get(idx)
    if table is None:
        table = {}
        with torch.cuda.device(idx):
            live = not torch.cuda.is_current_stream_capturing()
            if live:
                # the default stream takes its queue FIRST (a trainer asks for its role streams before it has launched anything: without
                # this the first role stream got the queue the default stream was bound to a moment later -- measured: the pre-train
                # step at 51 instead of 31 ms, its side streams serialised with the main one)
                torch.empty(64, device=f"cuda:{idx}").zero_()
                torch.cuda.synchronize(idx)
            for r in ROLES:
                table[r] = torch.cuda.Stream(device=idx)
            if live:
                for r in ROLES:   # first use in creation order: the runtime may bind a stream to its hardware queue lazily
                    with torch.cuda.stream(table[r]):
                        torch.empty(64, device=f"cuda:{idx}").zero_()
                torch.cuda.synchronize(idx)
        _STREAMS[idx] = table
        if torch.cuda.graph.default_capture_stream is None:
            torch.cuda.graph.default_capture_stream = table["capture"]
    return table[role]
