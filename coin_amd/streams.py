"""The package's side streams: ONE per role and device for the whole process, made the first time the role is used.

  side     pre-train: text encoder + anchor labelling beside the backbone (OpenVocabularyRCNN._overlap_side_work)
  look     pre-train: the NEXT batch's frozen stem beside the proposal chain (_launch_lookahead)
  capture  torch.cuda.graph's capture stream (coin_amd.graphs, the teacher's and the text encoder's graphs)
  teacher  targetDET: the EMA teacher's pass + matching beside the student's step (CoinTrainer)

A second model / trainer built in the same process re-uses the first one's streams instead of drawing new ones from torch's pool
(sharing a role is safe: a stream is an ordering domain, its users wait on events, not on each other's identity), so the number of
streams a process holds does not grow with the objects it builds.

What was tried in round 6 and measured worse (kept here because the failure modes are easy to walk into again): the HIP runtime binds
every stream to one of GPU_MAX_HW_QUEUES hardware queues, and streams that share a queue serialise -- round 5 saw the targetDET step at
64 instead of 52 ms in a process where a PRETrainer had lived first.  (1) Creating all four role streams up front, in a fixed order,
at trainer construction made that step the same 47.5 ms in both kinds of process -- and the PRE-TRAIN step 51 ms instead of 31 when the
streams were made before the default stream's first launch, and 56 instead of 32 ms under an RCCL process group (tools/ddp_ab.sh), with
or without (2) accepting only candidates whose spin kernels overlap the default stream's (the probe accepted the first three candidates
every time; whatever serialises there is not visible to 0.15 ms spin kernels).  Streams made lazily, at the point of first use, are
what every measured configuration of a job runs well with: pre-train 31.0 ms (no process group) / 32.3 ms (1-rank RCCL), targetDET
47-48 ms in a process of its own.  NOT resolved: the targetDET step in a process where a PRETrainer ran first still shows a 57-59 ms floor
(`tools/bench_targetdet.py --after-pretrain 8`; also with the teacher's stream created before the others) -- the reference runs the two
trainings as two jobs, and so does bench.py's `secondary` block.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

ROLES = ("side", "look", "capture", "teacher")
_STREAMS: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}


def role_stream(device, role: str) -> "torch.cuda.Stream":
    assert role in ROLES, role
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _STREAMS.get((idx, role))
    if s is None:
        s = _STREAMS[(idx, role)] = torch.cuda.Stream(device=idx)
        if role == "capture" and torch.cuda.graph.default_capture_stream is None:
            torch.cuda.graph.default_capture_stream = s
    return s
