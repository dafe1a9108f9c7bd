#!/usr/bin/env python3
"""Why does firing the reducer's hooks straight from a replay trip its second-gradient guard?  (development probe)
Patches coin_amd.graphs._Replay.backward to the direct variant, wraps GradReducer._on_grad to report who delivers a parameter twice."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COIN_FORCE_DDP"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import bench
from coin_amd import graphs as G
from coin_amd import parallel as P
from coin_amd.engine import PRETrainer

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
P._FORCE[0] = True
cfg = bench.build_cfg(1, "cuda:0", "bf16")
torch.manual_seed(cfg.SEED)
tr = PRETrainer(cfg)
names = {id(p): n for n, p in tr.model.named_parameters()}
who = {}
orig = P.GradReducer._on_grad


def on_grad(self, p):
    s = self.slices[self._slice_of[id(p)]]
    here = "".join(traceback.format_stack(limit=6)[:-1])
    if id(p) in s.seen or s.launched:
        print("SECOND ARRIVAL", names.get(id(p)), "slice launched:", s.launched, "\n--- first:\n", who.get(id(p), "?")[-900:], "\n--- now:\n", here[-900:], flush=True)
        raise SystemExit(1)
    who[id(p)] = here
    return orig(self, p)


P.GradReducer._on_grad = on_grad
for h in tr.reducer._hooks:
    h.remove()
tr.reducer._hooks = [p.register_post_accumulate_grad_hook(tr.reducer._on_grad) for p in tr.reducer.params]
real_backward = G._Replay.backward


@staticmethod
@torch.autograd.function.once_differentiable
def backward(ctx, *gouts):
    ent = ctx.ent
    out = real_backward.__wrapped__(ctx, *gouts) if hasattr(real_backward, "__wrapped__") else real_backward(ctx, *gouts)
    return out


# direct variant: after the stock backward returned the gradients for autograd, deliver them ourselves instead
def direct(ctx, *gouts):
    res = list(G._Replay._stock(ctx, *gouts))
    ent = ctx.ent
    base = 2 + len(ent.grads_in)
    for i, p in enumerate(ent.params):
        g = res[base + i]
        if g is not None and p.grad is None:
            p.grad = g
            for h in list(getattr(p, "_post_accumulate_grad_hooks", {}).values()):
                h(p)
            res[base + i] = None
    return tuple(res)


G._Replay._stock = staticmethod(G._Replay.backward)
G._Replay.backward = staticmethod(direct)
for i in range(8):
    who.clear()
    rec = tr.run_step()
    print("STEP", i, float(sum(rec.values())), G.STATS["replays"], flush=True)
print("no second arrival in 8 steps")
