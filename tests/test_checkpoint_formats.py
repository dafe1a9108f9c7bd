"""SURVEY section 8f-2, pinned: the product reads artefacts WRITTEN BY THE REFERENCE'S OWN CODE.

`tests/golden/ckpt/*.pth` were produced by `tests/golden/gen_golden.py::case_checkpoint_formats`, i.e. by the reference's
`PRETrainer.save` / `collect_results` (coin/engine/pre_train.py:138-161), `CoinTrainer.save` (coin/engine/trainer.py:122-137) over its
`EnsembleTSModel` (coin/modeling/meta_arch/ts_ensemble.py:24-37) and `DetectionTSCheckpointer` (coin/checkpoint/detection_checkpoint.py),
with `MyInstances` (coin/utils/util.py:188-267) and detectron2-path `Instances` / `Boxes` pickled inside.  Here (no reference, no
detectron2: this file also runs on the GPU box) `coin_amd/checkpoint.py` and the trainers' `resume_or_load` must take every tensor,
optimizer slot, scheduler field and cached result out of them.  The opposite direction -- files the product writes, read by the
reference's loaders -- is `tests/test_reference_live.py::test_reference_loads_the_files_the_product_writes`.
"""
import os
import types

import pytest
import torch

from cpu_shim import cpu_kernels
from e2e_util import tiny_product_detector
from golden_util import K


@pytest.fixture(autouse=True)
def _host_kernels():
    """No GPU is needed for file formats: the optimizer's device table is the test-only CPU stand-in (tests/cpu_shim.py)."""
    with cpu_kernels():
        yield

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt")


def _cfg(**kv):
    from coin_amd.config import get_cfg

    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "SOLVER.BASE_LR", 0.01, "SOLVER.STEPS", (5, 9), "SOLVER.FACTOR_LIST", (1, 0.1, 0.01), "SOLVER.WARMUP_ITERS", 3,
                         "SOLVER.WARMUP_FACTOR", 0.001] + [x for k, v in kv.items() for x in (k, v)])
    return cfg


def _raw(name):
    from coin_amd.checkpoint import load_file

    return load_file(os.path.join(CKPT, name))


def _fresh_pretrainer(weights, seed=0):
    """A PRETrainer around the tiny detector of the goldens (same module tree / state-dict keys as the reference's build), without the
    data side: what `resume_or_load` touches."""
    from coin_amd.engine import PRETrainer
    from coin_amd.solver import build_lr_scheduler, build_optimizer

    torch.manual_seed(seed)
    t = object.__new__(PRETrainer)
    t.cfg = _cfg(**{"MODEL.WEIGHTS": weights})
    t.device = torch.device("cpu")
    t.model = tiny_product_detector()
    t.optimizer = build_optimizer(t.cfg, t.model, name="all")
    t.scheduler = build_lr_scheduler(t.cfg, t.optimizer)
    t.collect_model, t.iter, t.start_iter, t._next_batch = None, 0, 0, None
    return t


def _same_results(mine, theirs):
    assert list(mine) == list(theirs)
    for ds in theirs:
        assert list(mine[ds]) == list(theirs[ds])
        for fn, rec in theirs[ds].items():
            got = mine[ds][fn]
            assert {k: v for k, v in got.items() if not isinstance(v, dict)} == {k: v for k, v in rec.items() if not isinstance(v, dict)}
            for tag in ("RCNN", "RPN"):
                a, b = got[tag]["instances"], rec[tag]["instances"]
                assert a.image_size == b.image_size and list(a.get_fields()) == list(b.get_fields()) == ["pred_boxes", "scores", "pred_classes", "probs"]
                assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.probs, b.probs) and torch.equal(a.pred_classes, b.pred_classes)


def test_the_artefacts_carry_the_references_class_paths_and_layout():
    raw = open(os.path.join(CKPT, "GDINO_collect.pth"), "rb").read()
    assert b"detectron2.structures.instances" in raw and b"detectron2.structures.boxes" in raw and b"coin.utils.util" in raw and b"MyInstances" in raw
    assert b"oracle" not in raw and b"coin_amd" not in raw
    blob = _raw("GDINO_collect.pth")
    assert list(blob) == ["results"] and list(blob["results"]) == ["foggytrain_0.02"] and len(blob["results"]["foggytrain_0.02"]) == 3
    ck = _raw("model_0000006.pth")
    assert set(ck) == {"model", "optimizer", "optimizer_merge", "scheduler", "scheduler_merge", "iteration", "ap_50_student", "ap_50_offline_teacher", "online_results"}
    assert all(k.startswith(("offline_teacher.", "model_student.", "merge_model.")) for k in ck["model"])     # online_teacher: its model was deleted
    assert set(_raw("CLIP_-000001.pth")) == {"model", "optimizer", "scheduler", "iteration", "results", "load_models"}
    assert set(_raw("pre_train_CLIP_0000004.pth")) == {"model", "optimizer", "scheduler", "iteration", "results"}


@pytest.mark.parametrize("resume", [False, True])
def test_pretrainer_resume_or_load_reads_the_references_pretrain_checkpoint(resume):
    path = os.path.join(CKPT, "pre_train_CLIP_0000004.pth")
    blob = _raw("pre_train_CLIP_0000004.pth")
    t = _fresh_pretrainer(path)
    sd0 = {k: v.clone() for k, v in t.model.state_dict().items()}
    t.resume_or_load(resume=resume)
    sd = t.model.state_dict()
    assert set(sd) == set(blob["model"]), set(sd) ^ set(blob["model"])           # the state-dict layout, key for key
    assert all(torch.equal(sd[k], blob["model"][k]) for k in sd) and any(not torch.equal(sd[k], sd0[k]) for k in sd)
    assert t.start_iter == t.iter == 5
    _same_results(t.collect_model.get_results(), blob["results"])
    res = t.collect_model("foggy/JPEGImages/b_000002.png")                          # a deep copy per call, as the collector hands out
    assert res["height"] == 104 and res["RCNN"]["instances"].probs.shape[1] == K + 1
    bufs = t.optimizer.state_dict().get("momentum_buffers")
    if resume:
        # one parameter per group in the reference's order: group i of the file is parameter i of the product's optimizer
        assert len(blob["optimizer"]["param_groups"]) == len(t.optimizer.param_groups) == 72
        for i, g in enumerate(t.optimizer.param_groups):
            ref = blob["optimizer"]["param_groups"][i]
            assert g["lr"] == ref["lr"] and g["weight_decay"] == ref["weight_decay"] and g["base_lr"] == ref["initial_lr"]
            assert torch.equal(bufs[i].cpu(), blob["optimizer"]["state"][i]["momentum_buffer"]) and bufs[i].shape == t.optimizer.params[i].shape
        assert t.scheduler.last_epoch == blob["scheduler"]["last_epoch"] == 2 and list(t.scheduler.base_lrs) == blob["scheduler"]["base_lrs"]
        assert [g["lr"] for g in t.optimizer.param_groups] == pytest.approx(blob["scheduler"]["_last_lr"], rel=1e-6)
    else:
        assert bufs is None or t.optimizer.state_dict().get("first", True)


def test_the_collection_runs_file_keeps_the_fresh_weights_but_hands_over_the_results():
    """CLIP_-000001.pth carries `load_models: False` (pre_train.py:142,265-268): the initialisation of THIS run stays, the cached teacher
    results and the iteration counter are taken."""
    t = _fresh_pretrainer(os.path.join(CKPT, "CLIP_-000001.pth"), seed=3)
    sd0 = {k: v.clone() for k, v in t.model.state_dict().items()}
    t.resume_or_load(resume=False)
    assert all(torch.equal(v, sd0[k]) for k, v in t.model.state_dict().items())
    assert t.start_iter == 0
    _same_results(t.collect_model.get_results(), _raw("CLIP_-000001.pth")["results"])


def _fresh_cointrainer(seed=0):
    from coin_amd.modeling.text_encoder import CKGNet
    from coin_amd.solver import build_lr_scheduler, build_optimizer

    torch.manual_seed(seed)
    cfg = _cfg()
    t = types.SimpleNamespace(cfg=cfg, device=torch.device("cpu"), offline_teacher=tiny_product_detector(), model=tiny_product_detector(),
                              merge=CKGNet(32, 32, K + 1, head_num=4), iter=0, start_iter=0, model_CLOUD=None, ap_50_student=None, ap_50_offline_teacher=None)
    t.optimizer, t.optimizer_merge = build_optimizer(cfg, t.model, name="all"), build_optimizer(cfg, t.merge, name="all")
    t.scheduler, t.scheduler_merge = build_lr_scheduler(cfg, t.optimizer), build_lr_scheduler(cfg, t.optimizer_merge)
    return t


@pytest.mark.parametrize("resume", [False, True])
def test_cointrainer_loads_the_references_checkpoint(resume):
    from coin_amd.checkpoint import load_cointrainer_weights, split_ensemble_state_dict

    blob = _raw("model_0000006.pth")
    parts = split_ensemble_state_dict(blob["model"])
    t = _fresh_cointrainer()
    load_cointrainer_weights(t, os.path.join(CKPT, "model_0000006.pth"), resume=resume)
    for module, part in ((t.offline_teacher, "offline_teacher"), (t.model, "student"), (t.merge, "merge")):
        sd = module.state_dict()
        assert set(sd) == set(parts[part]), (part, set(sd) ^ set(parts[part]))
        assert all(torch.equal(sd[k], parts[part][k]) for k in sd), part
    assert t.start_iter == t.iter == 7
    _same_results(t.model_CLOUD.get_results(), blob["online_results"])
    if resume:
        for opt, name in ((t.optimizer, "optimizer"), (t.optimizer_merge, "optimizer_merge")):
            bufs = opt.state_dict()["momentum_buffers"]
            assert len(bufs) == len(blob[name]["param_groups"])
            assert all(torch.equal(b.cpu(), blob[name]["state"][i]["momentum_buffer"]) for i, b in enumerate(bufs))
        assert t.scheduler.last_epoch == t.scheduler_merge.last_epoch == 3
        assert t.ap_50_student == {3: 41.5, 6: 43.25} and t.ap_50_offline_teacher == {3: 40.0, 6: 40.5}
    else:
        assert t.scheduler.last_epoch == t.scheduler_merge.last_epoch == 6       # trainer.py:246-247


def test_cointrainer_starts_from_pretrain_checkpoint_plus_collected_results():
    """MODEL.WEIGHTS = "pre_train_CLIP_xxx.pth+GDINO_collect.pth" (trainer.py:222-234), both written by the reference."""
    from coin_amd.checkpoint import load_cointrainer_weights

    t = _fresh_cointrainer(seed=5)
    student0 = {k: v.clone() for k, v in t.model.state_dict().items()}
    load_cointrainer_weights(t, os.path.join(CKPT, "pre_train_CLIP_0000004.pth") + "+" + os.path.join(CKPT, "GDINO_collect.pth"))
    pre = _raw("pre_train_CLIP_0000004.pth")["model"]
    sd = t.offline_teacher.state_dict()
    assert set(sd) == set(pre) and all(torch.equal(sd[k], pre[k]) for k in sd)
    assert all(torch.equal(v, student0[k]) for k, v in t.model.state_dict().items())     # the student keeps its own initialisation
    _same_results(t.model_CLOUD.get_results(), _raw("GDINO_collect.pth")["results"])
    rec = t.model_CLOUD("foggy/JPEGImages/c_3.png")
    assert rec["image_id"] == "c_3" and rec["RPN"]["instances"].pred_boxes.tensor.shape[1] == 4
