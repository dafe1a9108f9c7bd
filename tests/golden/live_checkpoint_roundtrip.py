#!/usr/bin/env python3
"""Build-container only (needs /root/reference): both directions of SURVEY section 8f-2 against the reference's OWN loaders.

    python tests/golden/live_checkpoint_roundtrip.py <scratch dir>

1. re-creates the committed artefacts of tests/golden/ckpt/ with `gen_golden.case_checkpoint_formats` (the reference's save code) and
   checks that they are reproduced tensor for tensor;
2. lets the PRODUCT write a pre-train checkpoint, a CoinTrainer checkpoint and a result cache (`PRETrainer.save`,
   `coin_amd.checkpoint.save_cointrainer_checkpoint`, `CloudResults.save`) and reads them back with the reference's
   `PRETrainer.resume_or_load` (pre_train.py:238-279), `CoinTrainer.resume_or_load` (trainer.py:220-262, both forms of MODEL.WEIGHTS) through
   its `DetectionTSCheckpointer` / `EnsembleTSModel` -- weights, optimizer momentum, scheduler state, iteration, AP histories and cached
   results must arrive unchanged.
Run as a subprocess by tests/test_reference_live.py (the import shim rewrites sys.modules).  Prints OK lines; any mismatch raises.
"""
import functools
import logging
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))                       # tests/: cpu_shim, e2e_util
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import gen_golden as G  # noqa: E402

shim = G.shim
scratch = sys.argv[1]
os.makedirs(scratch, exist_ok=True)
_load = functools.partial(torch.load, map_location="cpu", weights_only=False)


def same(a, b, where=""):
    if torch.is_tensor(a):
        assert torch.is_tensor(b) and a.shape == b.shape and torch.equal(a, b), where
    elif isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), (where, list(a.keys())[:5], list(b.keys())[:5])
        for k in a:
            same(a[k], b[k], f"{where}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), where
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{where}[{i}]")
    elif hasattr(a, "get_fields"):      # Instances
        assert type(a).__name__ == type(b).__name__ and tuple(a.image_size) == tuple(b.image_size), where
        same(dict(a.get_fields()), dict(b.get_fields()), where)
    elif hasattr(a, "tensor"):          # Boxes
        same(a.tensor, b.tensor, where)
    else:
        assert a == b, (where, a, b)


# ------------------------------------------------------------------ 1. the committed artefacts are what the reference writes
G.HERE = scratch
G.case_checkpoint_formats()
for f in sorted(os.listdir(os.path.join(HERE, G.CKPT_DIR))):
    same(_load(os.path.join(scratch, G.CKPT_DIR, f)), _load(os.path.join(HERE, G.CKPT_DIR, f)), f)
    print("OK reproduced", f)

# ------------------------------------------------------------------ 2. the product writes, the reference reads
from cpu_shim import cpu_kernels  # noqa: E402
from e2e_util import tiny_product_detector  # noqa: E402

from coin_amd.checkpoint import CloudResults, load_file, save_cointrainer_checkpoint  # noqa: E402
from coin_amd.config import get_cfg  # noqa: E402
from coin_amd.engine import PRETrainer  # noqa: E402
from coin_amd.modeling.text_encoder import CKGNet  # noqa: E402
from coin_amd.solver import build_lr_scheduler, build_optimizer  # noqa: E402

tr = G.import_ref_trainer()
shim.install_checkpoint_stack()
pt = shim.ref("coin.engine.pre_train")
dc = shim.ref("coin.checkpoint.detection_checkpoint")
ts = shim.ref("coin.modeling.meta_arch.ts_ensemble")
sb = shim.ref("coin.solver.build")
sched = shim.ref("coin.solver.lr_scheduler")
torch.load = functools.partial(torch.load, weights_only=False)   # trainer.py:224,230 call torch.load bare (torch 1.9 semantics)


def product_cfg():
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "SOLVER.BASE_LR", 0.01, "SOLVER.STEPS", (5, 9), "SOLVER.FACTOR_LIST", (1, 0.1, 0.01), "SOLVER.WARMUP_ITERS", 3,
                         "SOLVER.WARMUP_FACTOR", 0.001])
    return cfg


def some_steps(model, opt, sc, n, seed):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        opt.zero_grad()
        sum((p * torch.randn(p.shape, generator=g)).sum() for p in model.parameters() if p.requires_grad).backward()
        opt.step()
        sc.step()


def ref_groups(m):
    ov = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0}]
    return sb.get_default_optimizer_params(m, base_lr=0.01, weight_decay_norm=0.0, bias_lr_factor=1.0, weight_decay_bias=1e-4, overrides=ov, only_text_encoder=None)


def ref_sched(opt):
    return sched.WarmupTwoStageMultiStepLR(opt, [5, 9], factor_list=[1, 0.1, 0.01], warmup_factor=0.001, warmup_iters=3, warmup_method="linear")


def momentum_of(ref_opt):
    return [ref_opt.state[g["params"][0]].get("momentum_buffer") for g in ref_opt.param_groups]


with cpu_kernels():
    cfg = product_cfg()
    # the cached cloud results the product holds (any reference-layout store will do: the committed one, read by the product)
    cloud = CloudResults.load(os.path.join(HERE, G.CKPT_DIR, "GDINO_collect.pth"))
    # ---- product pre-train checkpoint
    torch.manual_seed(31)
    p = object.__new__(PRETrainer)
    p.cfg, p.device, p.rank, p.model = cfg, torch.device("cpu"), 0, tiny_product_detector()
    p.optimizer = build_optimizer(cfg, p.model, name="all")
    p.scheduler = build_lr_scheduler(cfg, p.optimizer)
    p.collect_model, p.iter, p.ap_50 = cloud, 5, {4: 12.5}
    some_steps(p.model, p.optimizer, p.scheduler, 2, 32)
    pre_path = os.path.join(scratch, "pre_train_CLIP_0000004.pth")
    p.save(pre_path, iteration=4)
    collect_path = os.path.join(scratch, "product_GDINO_collect.pth")
    cloud.save(collect_path)
    # ---- product CoinTrainer checkpoint
    torch.manual_seed(33)
    c = types.SimpleNamespace(cfg=cfg, offline_teacher=tiny_product_detector(), model=tiny_product_detector(), merge=CKGNet(32, 32, G.K + 1, head_num=4),
                              iter=7, model_CLOUD=cloud, ap_50_student={3: 41.5}, ap_50_offline_teacher={3: 40.0})
    c.optimizer, c.optimizer_merge = build_optimizer(cfg, c.model, name="all"), build_optimizer(cfg, c.merge, name="all")
    c.scheduler, c.scheduler_merge = build_lr_scheduler(cfg, c.optimizer), build_lr_scheduler(cfg, c.optimizer_merge)
    some_steps(c.model, c.optimizer, c.scheduler, 3, 34)
    some_steps(c.merge, c.optimizer_merge, c.scheduler_merge, 2, 35)
    coin_path = os.path.join(scratch, "model_0000006.pth")
    save_cointrainer_checkpoint(c, coin_path)
    prod_mom = {"s": [b.clone() for b in c.optimizer.state_dict()["momentum_buffers"]], "m": [b.clone() for b in c.optimizer_merge.state_dict()["momentum_buffers"]],
                "p": [b.clone() for b in p.optimizer.state_dict()["momentum_buffers"]]}

cfgns = lambda w: types.SimpleNamespace(MODEL=types.SimpleNamespace(WEIGHTS=w))
log = logging.getLogger("live")


# ---- reference PRETrainer.resume_or_load(resume=True) on the product's pre-train checkpoint
def ref_pretrainer(weights):
    model = G.build_detector(seed=41)
    opt = torch.optim.SGD(ref_groups(model), lr=0.01, momentum=0.9, weight_decay=1e-4)
    f = types.SimpleNamespace(model=model, optimizer=opt, scheduler=ref_sched(opt), cfg=cfgns(weights), start_iter=0, ap_50={},
                              collect_model=G._ckpt_cloud_results(42, ["x/y.png"]))
    f.checkpointer = dc.DetectionTSCheckpointer(model, scratch, optimizer=opt, scheduler=f.scheduler)
    return f


f = ref_pretrainer(pre_path)
pt.PRETrainer.resume_or_load(f, resume=True)
same({k: v for k, v in f.model.state_dict().items()}, {k: v.detach() for k, v in p.model.state_dict().items()}, "pre/model")
assert f.start_iter == 5 and f.ap_50 == {4: 12.5}
same(momentum_of(f.optimizer), prod_mom["p"], "pre/momentum")
assert f.scheduler.last_epoch == p.scheduler.last_epoch and [g["lr"] for g in f.optimizer.param_groups] == [g["lr"] for g in p.optimizer.param_groups]
same(f.collect_model.get_results(), _load(os.path.join(HERE, G.CKPT_DIR, "GDINO_collect.pth"))["results"], "pre/results")
print("OK reference PRETrainer.resume_or_load read the product's pre-train checkpoint")


# ---- reference CoinTrainer.resume_or_load on the product's files
def ref_cointrainer(weights):
    student, teacher, merge = G.build_detector(seed=51), G.build_detector(seed=52), G.build_ckg(53)
    online = G._ckpt_cloud_results(54, ["x/y.png"])
    online.delete_model()
    os_, om = torch.optim.SGD(ref_groups(student), lr=0.01, momentum=0.9, weight_decay=1e-4), torch.optim.SGD(ref_groups(merge), lr=0.01, momentum=0.9, weight_decay=1e-4)
    f = types.SimpleNamespace(model=student, offline_teacher=teacher, merge=merge, model_CLOUD=online, optimizer=os_, optimizer_merge=om, scheduler=ref_sched(os_),
                              scheduler_merge=ref_sched(om), cfg=cfgns(weights), logger=log, start_iter=0, ap_50_student={}, ap_50_offline_teacher={})
    f.ensem_ts_model = ts.EnsembleTSModel(teacher, online, student, merge, scratch)
    f.checkpointer = dc.DetectionTSCheckpointer(f.ensem_ts_model, scratch, optimizer=os_, optimizer_merge=om, scheduler=f.scheduler, scheduler_merge=f.scheduler_merge)
    f.load_aps = lambda ck: tr.CoinTrainer.load_aps(f, ck)
    return f


f = ref_cointrainer(coin_path)
tr.CoinTrainer.resume_or_load(f, resume=True)
for ref_m, prod_m, tag in ((f.model, c.model, "student"), (f.offline_teacher, c.offline_teacher, "teacher"), (f.merge, c.merge, "merge")):
    same(dict(ref_m.state_dict()), {k: v.detach() for k, v in prod_m.state_dict().items()}, "coin/" + tag)
same(momentum_of(f.optimizer), prod_mom["s"], "coin/momentum")
same(momentum_of(f.optimizer_merge), prod_mom["m"], "coin/momentum_merge")
assert f.start_iter == 7 and f.scheduler.last_epoch == c.scheduler.last_epoch and f.scheduler_merge.last_epoch == c.scheduler_merge.last_epoch
assert f.ap_50_student == {3: 41.5} and f.ap_50_offline_teacher == {3: 40.0}
same(f.model_CLOUD.get_results(), _load(os.path.join(HERE, G.CKPT_DIR, "GDINO_collect.pth"))["results"], "coin/online_results")
print("OK reference CoinTrainer.resume_or_load(resume=True) read the product's checkpoint")

f = ref_cointrainer(coin_path)
tr.CoinTrainer.resume_or_load(f, resume=False)
assert f.start_iter == 7 and f.scheduler.last_epoch == 6 and all(b is None for b in momentum_of(f.optimizer))
print("OK reference CoinTrainer.resume_or_load(resume=False)")

f = ref_cointrainer(pre_path + "+" + collect_path)
student0 = {k: v.clone() for k, v in f.model.state_dict().items()}
tr.CoinTrainer.resume_or_load(f, resume=False)
same(dict(f.offline_teacher.state_dict()), {k: v.detach() for k, v in p.model.state_dict().items()}, "two-path/teacher")
same(dict(f.model.state_dict()), student0, "two-path/student untouched")
same(f.model_CLOUD.get_results(), _load(os.path.join(HERE, G.CKPT_DIR, "GDINO_collect.pth"))["results"], "two-path/results")
rec = f.model_CLOUD("foggy/JPEGImages/c_3.png")     # GDINO_COLLECTOR.forward: a deep copy of the cached record
assert type(rec["RPN"]["instances"]).__name__ == "MyInstances" and type(rec["RCNN"]["instances"]).__module__ == "detectron2.structures.instances"
print("OK reference CoinTrainer.resume_or_load read 'pre_train.pth+GDINO_collect.pth' written by the product")
