#!/usr/bin/env python3
"""What does one tiny device-to-device copy cost on a busy queue?  N back-to-back ops with the host far ahead (device time by events):
`b.copy_(a)` (hipMemcpyAsync -> the runtime's blit kernel + its barrier packets), `a.clone()`, an elementwise kernel (`torch.add(a, 0, out=b)`),
for 128-byte and 64 KiB tensors; and the same ops interleaved with a 100 us GEMM (does the copy's barrier drain the queue?)."""
import torch


def t(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for numel in (32, 16384):
    a = torch.randn(numel, device="cuda")
    b = torch.empty_like(a)
    print(f"{numel * 4:6d} B  copy_ {t(lambda: b.copy_(a)):6.2f} us   clone {t(lambda: a.clone()):6.2f} us   add(out=) {t(lambda: torch.add(a, 0, out=b)):6.2f} us   "
          f"mul {t(lambda: a * 1):6.2f} us")
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
a = torch.randn(32, device="cuda")
b = torch.empty_like(a)
base = t(lambda: x @ x, 200)
print(f"GEMM alone {base:7.1f} us; GEMM + copy_ {t(lambda: (x @ x, b.copy_(a)), 200):7.1f} us; GEMM + add(out=) {t(lambda: (x @ x, torch.add(a, 0, out=b)), 200):7.1f} us; "
      f"GEMM + 4 copy_ {t(lambda: (x @ x, b.copy_(a), b.copy_(a), b.copy_(a), b.copy_(a)), 200):7.1f} us; GEMM + 4 add {t(lambda: (x @ x, torch.add(a, 0, out=b), torch.add(a, 0, out=b), torch.add(a, 0, out=b), torch.add(a, 0, out=b)), 200):7.1f} us")
