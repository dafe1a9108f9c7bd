"""Seeded weights / inputs for the parity cases whose tensors are too large to commit (RN50-width res5: 15 M parameters, a
[64,1024,14,14] RoI batch).  The golden generator (tests/golden/gen_golden.py, runs the REFERENCE's modules) and the tests
(oracle, HIP product) both call these functions, so all three see identical values; the committed fixture then only holds
the small outputs.  Everything is drawn from CPU generators (torch's CPU Philox/mt19937 streams are a function of the seed and
the torch build, which is the same image here and on the GPU box; the fixture stores input checksums to catch a drift)."""
import math

import torch


def fill_module(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Deterministic non-trivial values for every parameter and float buffer, keyed by the state-dict name (so modules with the
    same key set get the same values whatever their class)."""
    import zlib

    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd):
            t = sd[name]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()))  # keyed by NAME: extra / missing keys elsewhere do not shift it
            if name.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) + 0.5
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.1
            elif t.dim() == 1 and name.endswith("weight"):      # norm scale
                v = torch.rand(t.shape, generator=g) + 0.5
            elif t.dim() == 1:                                   # biases
                v = torch.randn(t.shape, generator=g) * 0.1
            else:                                                # conv / linear / embedding: He-style
                fan_in = t[0].numel() if t.dim() > 1 else t.numel()
                v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / max(fan_in, 1))
            t.copy_(v.to(t.dtype))
    return module


def randn(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed)) * scale


def checksum(t: torch.Tensor) -> float:
    """Order-independent-enough fingerprint of an input tensor (fp64 sum of x * a fixed ramp)."""
    f = t.detach().double().reshape(-1)
    ramp = torch.arange(f.numel(), dtype=torch.float64) % 977 + 1.0
    return float((f * ramp).sum())
