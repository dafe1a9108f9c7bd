#!/usr/bin/env python3
"""Which source lines of the product call the small aten ops of a step?  (development tool)

One steady-state bench step under a TorchDispatchMode: every aten call is charged to the innermost coin_amd/ frame on the Python
stack (forward, optimizer, trainer code); ops issued by the autograd engine's thread are not seen by the mode and are listed by
the second pass (autograd function names with counts).

    python tools/dispatch_count.py [--out gpurun_out/dispatch_count.txt]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/dispatch_count.txt")
    args = ap.parse_args()
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode

    import bench
    from coin_amd.engine import PRETrainer

    cfg = bench.build_cfg(1, "cuda:0", "bf16")
    torch.manual_seed(cfg.SEED)
    tr = PRETrainer(cfg)
    for _ in range(4):
        tr.run_step()
    torch.cuda.synchronize()
    counts = collections.Counter()
    skip = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::permute", "aten::transpose", "aten::t", "aten::slice", "aten::select",
            "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::unbind", "aten::split",
            "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::_local_scalar_dense", "aten::size", "aten::stride",
            "aten::is_", "aten::sym_", "aten::lift_fresh", "aten::unflatten", "aten::narrow", "aten::chunk", "aten::contiguous", "aten::result_type")

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, a=(), kw=None):
            out = func(*a, **(kw or {}))
            name = func._schema.name
            if not name.startswith(skip):
                t = out[0] if isinstance(out, (tuple, list)) and out else out
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    frame = "?"
                    for f in reversed(traceback.extract_stack(limit=40)[:-1]):
                        if "/coin_amd/" in f.filename and not f.filename.endswith("kernels.py"):
                            frame = f"{f.filename[f.filename.index('/coin_amd/') + 1:]}:{f.lineno} {f.name}"
                            break
                    counts[(name, frame)] += 1
            return out

    with Mode():
        tr.run_step()
    torch.cuda.synchronize()
    lines = [f"aten calls with a device result in one step (python-thread ops only): {sum(counts.values())}"]
    by_frame = collections.Counter()
    for (name, frame), n in counts.items():
        by_frame[frame] += n
    lines.append("--- by source line")
    for frame, n in by_frame.most_common(60):
        ops = ", ".join(f"{k[0][6:]}x{v}" for k, v in sorted(counts.items(), key=lambda kv: -kv[1]) if k[1] == frame)[:150]
        lines.append(f"{n:5d}  {frame}   [{ops}]")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:70]))


if __name__ == "__main__":
    main()
