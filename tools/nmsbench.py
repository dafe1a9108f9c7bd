#!/usr/bin/env python3
"""coin_nms_batched at the pre-train step's shape: 4 images x 12 000 score-sorted proposals (dense anchors decoded with small random
deltas: the heavy-overlap regime of a real RPN), IoU 0.7, 2 000 kept.  Prints us per call and a checksum of the kept indices."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

from coin_amd import _lib

if os.environ.get("NMS_LIB"):   # another build of the library (A/B of a kernel change)
    _lib.LIB_PATH = os.path.abspath(os.environ["NMS_LIB"])
from coin_amd import kernels as K


def main():
    g = torch.Generator().manual_seed(0)
    b, n = 4, 12000
    cx = torch.rand(b, n, generator=g) * 1333
    cy = torch.rand(b, n, generator=g) * 800
    w = torch.exp(torch.rand(b, n, generator=g) * 2.5 + 3.0)
    h = torch.exp(torch.rand(b, n, generator=g) * 2.5 + 3.0)
    boxes = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).clamp(min=0).cuda().contiguous()
    counts = torch.full((b,), n, dtype=torch.int32, device="cuda")
    for thr, keep in ((0.7, 2000), (0.7, 12000), (0.3, 2000)):
        k, num = K.nms_batched(boxes, counts, thr, keep)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            K.nms_batched(boxes, counts, thr, keep)
        e.record()
        torch.cuda.synchronize()
        chk = int(sum(int(k[i, : int(num[i])].long().sum()) for i in range(b)))
        print(f"iou {thr} max_keep {keep}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us per call; kept {num.tolist()} checksum {chk}")


if __name__ == "__main__":
    main()
