#!/usr/bin/env python3
"""Development tool: what does a ONE-rank RCCL all-reduce cost on the device?  (the single-rank dry run of the collective path pays it
per gradient slice; with N > 1 ranks the same call moves data over xGMI instead)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for mb in (1, 8, 32, 64):
    t = torch.ones(mb * (1 << 20) // 4, device="cuda")
    for _ in range(3):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(10):
        dist.all_reduce(t)
    e.record()
    host = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    print(f"{mb:3d} MiB: device {s.elapsed_time(e) / 10:.3f} ms per all_reduce, host {host:.3f} ms per call, {mb / 1024 / (s.elapsed_time(e) / 10 / 1e3):.0f} GiB/s")
dist.destroy_process_group()
