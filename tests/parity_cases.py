"""Device-generic parity cases: the same comparison of the PRODUCT with the reference's golden vectors runs
(a) on the GPU-less build container with the kernels shimmed (``tests/cpu_shim.py``; host logic only) and
(b) on the MI355X through ``libcoin_hip.so`` (``-m gpu``), so that every integer / index decision the device
path takes is checked bit for bit where the reference's arithmetic is integer, and to 1e-4 where it is fp32.

The reference's samplers draw ``torch.randperm`` from the CPU generator (the goldens were captured on the CPU).
A device permutation comes from a different stream, so on the GPU the product's ``randperm`` calls are served the
SAME CPU permutation, moved to the device (``reference_randperm``): everything around the draw -- IoU matrix,
Matcher, nonzero / index bookkeeping, label scatter -- runs on the device and must reproduce the reference exactly.
"""
import contextlib
import copy
import random

import numpy as np
import torch

from e2e_util import _inst, tiny_product_detector
from golden_util import K, T, close, load, load_weights


@contextlib.contextmanager
def reference_randperm():
    real = torch.randperm

    def randperm(n, *a, device=None, **kw):
        out = real(n, *a, **kw)  # the CPU generator's stream (= the reference's)
        return out.to(device) if device is not None else out

    torch.randperm = randperm
    try:
        yield
    finally:
        torch.randperm = real


@contextlib.contextmanager
def kernels_for(device):
    """CPU: oracle-backed stand-ins for the HIP kernels (host-logic check).  GPU: the real library, nothing patched but the
    source of the random permutations."""
    if str(device) == "cpu":
        from cpu_shim import cpu_kernels

        with cpu_kernels():
            yield
    else:
        with reference_randperm():
            yield


def _eq(a, b, what):
    a = a.detach().cpu()
    b = torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.equal(a.to(b.dtype), b), f"{what}: {int((a.to(b.dtype) != b).sum())} of {b.numel()} entries differ"


# ------------------------------------------------------------------------------------------ A3: anchor labelling (rpn.py:120-254)
def rpn_labelling_losses_and_proposals(device):
    from coin_amd.structures import Boxes, ImageList, Instances

    z = load("rpn")
    with kernels_for(device):
        pg = tiny_product_detector().proposal_generator
        load_weights(pg, z)
        pg.to(device).train()
        feats = {"res4": T(z["feat"]).to(device).contiguous(memory_format=torch.channels_last)}
        sizes = [tuple(int(v) for v in s) for s in z["image_sizes"]]
        images = ImageList(torch.zeros(2, 3, 96, 128, device=device), sizes)
        gts = []
        for i, s in enumerate(sizes):
            t = Instances(s)
            t.gt_boxes = Boxes(T(z[f"gt{i}.boxes"]).to(device))
            gts.append(t)
        anchors = pg.anchor_generator([feats["res4"]])
        _eq(anchors[0].tensor, z["anchors"], "anchors")
        torch.manual_seed(103)
        labels, matched = pg.label_and_sample_anchors(anchors, gts, "pre_train")
        _eq(torch.stack(labels), z["labels"], "sampled anchor labels (pre_train)")
        _eq(torch.stack(matched), z["matched_boxes"], "matched boxes (pre_train)")
        torch.manual_seed(103)
        props, losses = pg(images, feats, gts, branch="pre_train")
        for k in ("loss_rpn_cls", "loss_rpn_loc"):
            assert abs(float(losses[k]) - float(z["loss::" + k])) < 1e-5 * max(1.0, abs(float(z["loss::" + k]))), (k, float(losses[k]))
        for i, p in enumerate(props):
            close(p.proposal_boxes.tensor, z[f"prop{i}.boxes"], 1e-5, f"proposals {i}")
            close(p.objectness_logits, z[f"prop{i}.logits"], 1e-5, f"proposal logits {i}")
        # step_two with (A, None, C) targets (rpn.py:199-254): image 1 has no A box
        dual = []
        for i, s in enumerate(sizes):
            a, c = Instances(s), Instances(s)
            a.gt_boxes = Boxes(T(z[f"s.a{i}.boxes"]).to(device))
            c.gt_boxes = Boxes(T(z[f"s.c{i}.boxes"]).to(device))
            c.gt_probs = T(z[f"s.c{i}.probs"]).to(device)
            dual.append((a, None, c))
        torch.manual_seed(104)
        lab, mb, idx, dl = pg.label_and_sample_anchors(anchors, [[d[0] for d in dual], [d[2] for d in dual]], "step_two")
        _eq(torch.stack(lab), z["s_labels"], "sampled anchor labels (step_two)")
        _eq(torch.stack(idx), z["s_matched_idxs"], "matched C index (step_two)")
        _eq(torch.stack(dl), z["s_dist_labels"], "distillation labels (step_two)")
        _eq(torch.stack(mb), z["s_matched_boxes"], "matched A boxes (step_two)")
        torch.manual_seed(104)
        _, losses2 = pg(images, feats, dual, branch="step_two")
        ref2 = {k[7:]: float(z[k]) for k in z.files if k.startswith("sloss::")}
        assert set(losses2) == set(ref2)
        for k, v in ref2.items():
            assert abs(float(losses2[k]) - v) < 1e-5 * max(1.0, abs(v)), (k, float(losses2[k]), v)


# ------------------------------------------------------------------------------------------ A4: RoI labelling + sampling (clip_roi_heads.py:283-399)
def _cmp_inst_exact(got, z, prefix):
    from coin_amd.structures import Boxes

    names = [k[len(prefix) + 1:] for k in z.files if k.startswith(prefix + ".")]
    assert names and set(names) == set(got.get_fields()), (prefix, names, sorted(got.get_fields()))
    for name in names:
        v = got.get(name)
        _eq(v.tensor if isinstance(v, Boxes) else v, z[f"{prefix}.{name}"], f"{prefix}.{name}")  # copies of the inputs: bit-exact


def roi_label_and_sample(device):
    from coin_amd.structures import Boxes, Instances

    z = load("roi_sampling")
    size = (96, 128)
    with kernels_for(device):
        rh = tiny_product_detector().roi_heads
        props, targets = [], []
        for i in range(2):
            p = Instances(size)
            p.proposal_boxes = Boxes(T(z[f"in{i}.boxes"]).to(device))
            p.objectness_logits = T(z[f"in{i}.logits"]).to(device)
            props.append(p)
            targets.append(_inst(z, f"t{i}", size).to(device))
        torch.manual_seed(114)
        out = rh.label_and_sample_proposals(props, targets, "pre_train")
        for i, (fg, bg) in enumerate(out):
            _cmp_inst_exact(fg, z, f"o{i}.fg")
            _cmp_inst_exact(bg, z, f"o{i}.bg")
        props, A, B, C = [], [], [], []
        for i in range(2):
            p = Instances(size)
            p.proposal_boxes = Boxes(T(z[f"s.in{i}.boxes"]).to(device))
            p.objectness_logits = T(z[f"s.in{i}.logits"]).to(device)
            props.append(p)
            A.append(_inst(z, f"s.a{i}", size).to(device))
            B.append(_inst(z, f"s.b{i}", size).to(device))
            C.append(_inst(z, f"s.c{i}", size).to(device))
        torch.manual_seed(115)
        out = rh.label_and_sample_proposals(props, [A, B, C], "step_two")
        for i, (a, b, bg) in enumerate(out):
            _cmp_inst_exact(a, z, f"s.o{i}.a")
            _cmp_inst_exact(b, z, f"s.o{i}.b")
            _cmp_inst_exact(bg, z, f"s.o{i}.bg")


# ------------------------------------------------------------------------------------------ A7 / A9 / A11: predictor + pre_train losses (fast_rcnn.py:318-438)
def box_predictor_pretrain(device, tag, tol=1e-4):
    """FastRCNNOutputLayers.forward + losses('pre_train') of the PRODUCT in the reference's (fg, bg) layout vs box_predictor_pretrain_*.npz
    (incl. an image without RoIs, no foreground at all, the clipart soft-target branch and CLOUD.LOSS_TYPE MILFocalLoss)."""
    from coin_amd.box_ops import Box2BoxTransform
    from coin_amd.modeling.fast_rcnn import FastRCNNOutputLayers
    from coin_amd.structures import ShapeSpec
    from golden_util import LOSS_W

    z = load(f"box_predictor_pretrain_{tag}")
    with kernels_for(device):
        te = tiny_product_detector().roi_heads.box_predictor.text_encoder
        bp = FastRCNNOutputLayers(ShapeSpec(channels=64, height=1, width=1), text_encoder=te, pooling_type="meanpool",
                                  box2box_transform=Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=32, classes_weight=[1.0] * K + [0.9],
                                  loss_type="MILFocalLoss" if tag == "focal" else "MILCrossEntropy", cls_agnostic_bbox_reg=True, loss_weight=LOSS_W,
                                  batch_size_per_image=32, cls_b_thresh=0.3, dataset=(str(z["dataset"]),), prototype_update_rate=0.9996)
        load_weights(bp, z)
        bp.to(device).train()
        props = [(_inst(z, f"p{i}.fg", (96, 128)).to(device), _inst(z, f"p{i}.bg", (96, 128)).to(device)) for i in range(int(z["n_img"]))]
        x = T(z["x"]).to(device).requires_grad_(True)
        preds = bp(x, "pre_train")
        (scores, lta), deltas, feats = preds
        close(scores.cpu(), z["scores"], tol, "scores")
        close(deltas.cpu(), z["deltas"], tol, "deltas")
        close(feats.cpu(), z["feats"], tol, "feats")
        losses = bp.losses(preds, props, None, "pre_train", update_prototype=bool(z["update_prototype"]))
        ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
        assert set(losses) == set(ref)
        for k, v in ref.items():
            assert abs(float(losses[k]) - v) < tol * max(1.0, abs(v)), (tag, k, float(losses[k]), v)
        sum(losses.values()).backward()
        close(x.grad.cpu(), z["gx"], tol, "gx")
        params = dict(bp.named_parameters())
        for k in z.files:
            if k.startswith("g::"):
                g = params[k[3:]].grad
                close((g if g is not None else torch.zeros_like(params[k[3:]])).cpu(), z[k], tol, k)
        close(bp.text_encoder.per_class_feat.cpu(), z["prototype_after"], 1e-6, "prototype")


# ------------------------------------------------------------------------------------------ whole forward + backward WITH the samplers in the loop
def e2e_pretrain_with_samplers(device, tol_loss=1e-4, tol_grad=1e-4):
    """OpenVocabularyRCNN.forward('pre_train') with NOTHING fed in: RPN, NMS, anchor labelling, RoI matching + sampling all run and
    must draw the reference's samples (e2e_pretrain.npz was captured with torch.manual_seed(123))."""
    z = load("e2e_pretrain")
    with kernels_for(device):
        model = tiny_product_detector()
        load_weights(model, z)
        model.to(device)
        model.train()
        batch = []
        for i in range(2):
            img = T(z[f"img{i}"])
            size = (img.shape[1], img.shape[2])
            batch.append({"image": img.to(device), "height": size[0], "width": size[1], "RCNN": _inst(z, f"rcnn{i}", size), "RPN": _inst(z, f"rpn{i}", size)})
        torch.manual_seed(123)
        losses = model(batch, branch="pre_train", update_prototype=True)
        ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
        assert set(losses) == set(ref)
        for k, v in ref.items():
            assert abs(float(losses[k]) - v) < tol_loss * max(1.0, abs(v)), (k, float(losses[k]), v)
        sum(losses.values()).backward()
        params = dict(model.named_parameters())
        return {k[3:]: params[k[3:]].grad.detach().float().cpu() for k in z.files if k.startswith("g::")}, {k[3:]: T(z[k]) for k in z.files if k.startswith("g::")}, model, z


# ------------------------------------------------------------------------------------------ A15: CoinTrainer.run_step (trainer.py:160-218)
def cointrainer_scripted_iteration(device, tol=1e-5):
    """`CoinTrainer.run_step` (product) against the iteration scripted with the reference's own pieces
    (tests/golden/gen_golden.py:case_e2e_coin_step): A/B/C targets, every loss incl. loss_merge_grad, the CKG parameters after the
    merge optimizer step and student parameters after the student optimizer step.  Boundary P: the reference's sampled anchors /
    RoIs are fed in; the matcher gets the stored teacher detections (the product's own teacher inference is compared as a set)."""
    from coin_amd.engine import CoinTrainer
    from coin_amd.modeling.text_encoder import CKGNet
    from coin_amd.solver import FusedSGD, get_default_optimizer_params
    from coin_amd.structures import Boxes

    z = load("e2e_coin_step")
    dev = torch.device(device)
    overrides = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0}]
    with kernels_for(device):
        student, teacher = tiny_product_detector(), tiny_product_detector()
        load_weights(student, z, "s::")
        load_weights(teacher, z, "t::")
        merge = CKGNet(32, 32, K + 1, head_num=4)
        load_weights(merge, z, "m::")
        student.to(dev), teacher.to(dev), merge.to(dev)
        teacher.roi_heads.box_predictor.test_score_thresh = 0.05
        for p in teacher.parameters():
            p.requires_grad = False
        batch, cloud, sizes = [], {}, []
        for i in range(2):
            img = T(z[f"img{i}"]).to(dev)
            s = (img.shape[1], img.shape[2])
            sizes.append(s)
            name = f"img{i}.png"
            batch.append({"image": img, "height": s[0], "width": s[1], "file_name": name, "image_id": f"id{i}", "random_flip": "no"})
            cloud[name] = {"file_name": name, "image_id": f"id{i}", "height": s[0], "width": s[1], "RCNN": {"instances": _inst(z, f"cloud{i}", s)},
                           "RPN": {"instances": _inst(z, f"cloud{i}", s)}}
        stored = [{"instances": _inst(z, f"det{i}", s).to(dev)} for i, s in enumerate(sizes)]
        # the product's own teacher inference finds the same detections
        teacher.eval()
        with torch.no_grad():
            own = teacher([{k: b[k] for k in ("image", "height", "width")} for b in batch], branch="test")
        teacher.train()
        for o, st in zip(own, stored):
            assert len(o["instances"]) == len(st["instances"])
            close(torch.sort(o["instances"].scores, descending=True).values.cpu(), torch.sort(st["instances"].scores, descending=True).values.cpu(),
                  max(tol, 1e-5), "teacher scores")
        teacher_forward = teacher.forward
        teacher.forward = lambda bi, branch=None, **kw: (teacher_forward(bi, branch=branch, **kw), copy.deepcopy(stored))[1]

        tr = object.__new__(CoinTrainer)
        ns = lambda **kw: type("NS", (), kw)()
        tr.cfg = ns(CLOUD=ns(BURN_UP_STEP=0, OFFLINE_TEACHER_UPDATE_ITER=1, EMA_KEEP_RATE_OFFLINE=1.0, PROTOTYPE_UPDATE_START=0,
                             MATCHER=ns(IOU_THRESHOLDS=0.5)))
        tr.device, tr.world_size, tr.rank = dev, 1, 0
        tr.model, tr.offline_teacher, tr.merge = student, teacher, merge
        tr.ddp_model, tr.ddp_merge = student, merge
        groups = lambda m: get_default_optimizer_params(m, base_lr=0.01, weight_decay_norm=0.0, bias_lr_factor=1.0, weight_decay_bias=1e-4,
                                                        overrides=overrides, only_text_encoder=None)
        tr.optimizer = FusedSGD(groups(student), lr=0.01, momentum=0.9, weight_decay=1e-4)
        tr.optimizer_merge = FusedSGD(groups(merge), lr=0.01, momentum=0.9, weight_decay=1e-4)
        tr.scheduler = tr.scheduler_merge = ns(step=lambda self=None: None)
        tr._data_loader_iter = iter([(copy.deepcopy(batch), copy.deepcopy(batch))])
        tr.model_CLOUD = lambda fn: copy.deepcopy(cloud[fn])
        tr.iter, tr.max_iter, tr.WEIGHT_FOR_BOX_A, tr._ema, tr._pending, tr.last_losses = 0, 1, 0.5, None, None, None
        student.train()
        # boundary P: the samplers return what the reference's samplers drew
        sampled = [(_inst(z, f"s{i}.a", s).to(dev), _inst(z, f"s{i}.b", s).to(dev), _inst(z, f"s{i}.bg", s).to(dev)) for i, s in enumerate(sizes)]
        student.roi_heads.label_and_sample_proposals = lambda proposals, targets, branch: sampled
        lab, mb = T(z["anchor_labels"]).to(dev), T(z["anchor_matched_boxes"]).to(dev)
        idx, dl = T(z["anchor_matched_idxs"]).to(dev), T(z["anchor_dist_labels"]).to(dev)
        student.proposal_generator.label_and_sample_anchors = lambda anchors, gt, branch: (list(lab), list(mb), list(idx), list(dl))
        seen = {}
        match = tr.match_boxes
        tr.match_boxes = lambda b, o, **kw: seen.setdefault("targets", match(b, o, **kw))
        random.seed(77)
        record = tr.run_step()
    rcnn, rpn = seen["targets"]
    assert [[len(t[0]), len(t[1]), len(t[2])] for t in rcnn] == z["n_abc"].tolist()
    for i in range(2):
        for name, inst in (("a", rcnn[i][0]), ("b", rcnn[i][1]), ("c", rcnn[i][2]), ("rpn_a", rpn[i][0]), ("rpn_c", rpn[i][2])):
            for k, v in inst.get_fields().items():
                close((v.tensor if isinstance(v, Boxes) else v).cpu(), z[f"{name}{i}.{k}"], 1e-5, f"{name}{i}.{k}")
    ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
    assert set(record) == set(ref)
    for k, v in ref.items():
        assert abs(float(record[k]) - v) < 1e-4 * max(1.0, abs(v)), (k, float(record[k]), v)
    for k, v in merge.state_dict().items():
        close(v.cpu(), z["m_after::" + k], tol, "merge " + k)
    sd = student.state_dict()
    for k in z.files:
        if k.startswith("s_after::"):
            close(sd[k[9:]].cpu(), z[k], tol, "student " + k[9:])
    assert tr.iter == 1 and tr.WEIGHT_FOR_BOX_A == 0.5


def cointrainer_two_iterations_through_constructor(device, tol=1e-5, teacher_stream=True):
    """The REAL `CoinTrainer(cfg, data_loader, cloud_results)` constructor (models, optimizers and LR schedulers from the cfg, teacher
    stream on the GPU), two consecutive `run_step` + `prepare_next` iterations against the two iterations scripted with the
    reference's own pieces in the reference's order (tests/golden/gen_golden.py:case_e2e_coin_two_steps; trainer.py:149-218,
    ts_ensemble.py:39-69).  An EMA is due at both iterations, so the fixture is only reproduced when
      * the EMA of iteration 1 reads the weights the optimizer of iteration 0 wrote (teacher parameters after each EMA),
      * the teacher pass of iteration 1 runs on the EMA'd teacher (detection count and scores change from 52/43 to 45/40 boxes),
      * iteration 1's matching already fuses the A boxes (WEIGHT_FOR_BOX_A 1.0 -> 0.5 in after_step),
      * the second optimizer step does not overtake the EMA (student parameters after both steps).
    Boundary P as everywhere: the samplers return what the reference's samplers drew, the matcher gets the stored detections after the
    product's own teacher pass has been compared with them as a set."""
    import os

    from coin_amd.config import get_cfg
    from coin_amd.engine import CoinTrainer
    from coin_amd.structures import Boxes
    from golden_util import tiny_tokens

    z = load("e2e_coin_two_steps")
    dev = torch.device(device)
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "GDINO", "foggy_synthetic.yaml")
    cfg = get_cfg()
    cfg.merge_from_file(root)
    cfg.merge_from_list([
        "MODEL.DEVICE", str(device), "AMD.COMPUTE_DTYPE", "fp32", "AMD.TEACHER_STREAM", bool(teacher_stream), "AMD.TEACHER_GRAPH", not teacher_stream,
        "AMD.SYNC_FREE", False, "AMD.SYNC_FREE_STEP", False,
        "AMD.CLASS_NAMES", ["car", "person", "bus"], "AMD.TEXT_TEMPLATES", 2, "DATASETS.TRAIN_UNLABEL", ("foggytrain_0.02",),
        "AMD.ARCH.LAYERS", [1, 1, 2, 2], "AMD.ARCH.WIDTH", 8, "AMD.ARCH.TEXT_WIDTH", 32, "AMD.ARCH.TEXT_LAYERS", 2, "AMD.ARCH.TEXT_HEADS", 2,
        "AMD.ARCH.TEXT_DIM", 32, "AMD.ARCH.CONTEXT_LENGTH", 16, "AMD.ARCH.VOCAB_SIZE", 64, "MODEL.MERGE_DIM", 32, "MODEL.BACKBONE.FREEZE_AT", 2,
        "MODEL.ANCHOR_GENERATOR.SIZES", [[32, 64, 128]], "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64, "MODEL.RPN.POSITIVE_FRACTION", 0.5,
        "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 200, "MODEL.RPN.PRE_NMS_TOPK_TEST", 120, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 60, "MODEL.RPN.POST_NMS_TOPK_TEST", 40,
        "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.ROI_HEADS.POSITIVE_FRACTION", 0.25, "MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.05,
        "MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG", True, "CLOUD.CLASSES_WEIGHT", [1.0, 1.0, 1.0, 0.9], "CLOUD.CLS_B_THRESH", 0.3,
        "CLOUD.BURN_UP_STEP", 0, "CLOUD.OFFLINE_TEACHER_UPDATE_ITER", 1, "CLOUD.EMA_KEEP_RATE_OFFLINE", float(z["keep_rate"]), "CLOUD.PROTOTYPE_UPDATE_START", 0,
        "SOLVER.BASE_LR", 0.01, "SOLVER.MOMENTUM", 0.9, "SOLVER.WEIGHT_DECAY", 1e-4, "SOLVER.WEIGHT_DECAY_NORM", 0.0, "SOLVER.WEIGHT_DECAY_BIAS", 1e-4,
        "SOLVER.BIAS_LR_FACTOR", 1.0, "SOLVER.LR_SCHEDULER_NAME", "WarmupMultiStepLR", "SOLVER.STEPS", (1000,), "SOLVER.WARMUP_ITERS", 0,
        "SOLVER.MAX_ITER", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2,
        "SOLVER.PER_MODULE_PARAM_WEIGHT", [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0,
                                            "add_in_embedding": 1.0, "logit_scale": 0.0}]])
    batches, cloud, sizes, stored = [], {}, [], []
    for it in range(2):
        batch, sz = [], []
        for i in range(2):
            img = T(z[f"it{it}::img{i}"]).to(dev)
            s = (img.shape[1], img.shape[2])
            sz.append(s)
            name = f"it{it}_img{i}.png"
            batch.append({"image": img, "height": s[0], "width": s[1], "file_name": name, "image_id": f"id{it}_{i}", "random_flip": "no"})
            cloud[name] = {"file_name": name, "image_id": f"id{it}_{i}", "height": s[0], "width": s[1],
                           "RCNN": {"instances": _inst(z, f"it{it}::cloud{i}", s)}, "RPN": {"instances": _inst(z, f"it{it}::cloud{i}", s)}}
        batches.append(batch)
        sizes.append(sz)
        stored.append([{"instances": _inst(z, f"it{it}::det{i}", s).to(dev)} for i, s in enumerate(sz)])
    loader = [(copy.deepcopy(b), copy.deepcopy(b)) for b in batches]   # (strong, weak): the scripted iterations feed the same views to both
    with kernels_for(device):
        torch.manual_seed(0)
        tr = CoinTrainer(cfg, data_loader=loader, cloud_results=lambda fn: copy.deepcopy(cloud[fn]))
        student, teacher, merge = tr.model, tr.offline_teacher, tr.merge
        for m in (student, teacher):
            m.roi_heads.box_predictor.text_encoder.encoder.tokenized_prompts.copy_(tiny_tokens())
        load_weights(student, z, "s::")
        load_weights(teacher, z, "t::")
        load_weights(merge, z, "m::")
        assert tr.WEIGHT_FOR_BOX_A == 1.0 and tr.iter == 0                      # trainer.py:111
        names = [k[len("it0::t_ema::"):] for k in z.files if k.startswith("it0::t_ema::")]
        state = {"teacher_calls": 0, "match_calls": 0, "student_calls": 0}
        teacher_forward = teacher.forward

        def teacher_pass(bi, branch=None, **kw):
            it = state["teacher_calls"]
            state["teacher_calls"] += 1
            tsd = teacher.state_dict()
            for k in names:   # the EMA that precedes this pass has read the right student weights
                close(tsd[k].float().cpu(), T(z[f"it{it}::t_ema::{k}"]).float(), tol, f"teacher after EMA {it}: {k}")
            own = teacher_forward(bi, branch=branch, **kw)
            for o, st in zip(own, stored[it]):
                assert len(o["instances"]) == len(st["instances"]), (it, len(o["instances"]), len(st["instances"]))
                close(torch.sort(o["instances"].scores, descending=True).values.cpu(), torch.sort(st["instances"].scores, descending=True).values.cpu(),
                      max(tol, 1e-5), f"teacher scores, iteration {it}")
            return copy.deepcopy(stored[it])

        teacher.forward = teacher_pass
        seen, match = [], tr.match_boxes

        def match_boxes(b, o, **kw):
            random.seed(77 + state["match_calls"])
            state["match_calls"] += 1
            seen.append(match(b, o, **kw))
            return seen[-1]

        tr.match_boxes = match_boxes
        # boundary P: per iteration, the samplers return what the reference's samplers drew
        def sampled(proposals, targets, branch):
            it = state["student_calls"]
            return [(_inst(z, f"it{it}::s{i}.a", s).to(dev), _inst(z, f"it{it}::s{i}.b", s).to(dev), _inst(z, f"it{it}::s{i}.bg", s).to(dev))
                    for i, s in enumerate(sizes[it])]

        def anchors(anc, gt, branch):
            it = state["student_calls"]
            g = lambda k: list(T(z[f"it{it}::{k}"]).to(dev))
            return g("anchor_labels"), g("anchor_matched_boxes"), g("anchor_matched_idxs"), g("anchor_dist_labels")

        student.roi_heads.label_and_sample_proposals = sampled
        student.proposal_generator.label_and_sample_anchors = anchors
        records = []
        for it in range(2):
            rec = tr.run_step()
            state["student_calls"] += 1
            records.append({k: float(v.detach()) for k, v in rec.items()})
            if os.environ.get("COIN_TEST_DEBUG"):
                for k in sorted(records[-1]):
                    print(it, k, records[-1][k], float(z[f"it{it}::loss::{k}"]))
            assert tr.iter == it + 1 and tr.WEIGHT_FOR_BOX_A == 0.5
            for k, v in merge.state_dict().items():
                close(v.cpu(), z[f"it{it}::m_after::" + k], tol, f"merge {k} after iteration {it}")
            sd = student.state_dict()
            for k in names:
                close(sd[k].float().cpu(), T(z[f"it{it}::s_after::" + k]).float(), tol, f"student {k} after iteration {it}")
            tr.prepare_next()          # iteration it+1's EMA / teacher pass / matching (teacher stream on the GPU); no-op after the last one
            if it == 0:
                assert state["teacher_calls"] == 2 and state["match_calls"] == 2, state   # iteration 1 was prepared BEFORE its run_step
        if dev.type == "cuda":
            torch.cuda.synchronize()
            assert (tr._teacher_stream is not None) == bool(teacher_stream)
    assert state["teacher_calls"] == 2
    for it in range(2):
        rcnn, rpn = seen[it]
        assert [[len(t[0]), 0 if t[1] is None else len(t[1]), len(t[2])] for t in rcnn] == z[f"it{it}::n_abc"].tolist(), it
        for i in range(2):
            for name, inst in (("a", rcnn[i][0]), ("b", rcnn[i][1]), ("c", rcnn[i][2]), ("rpn_a", rpn[i][0]), ("rpn_c", rpn[i][2])):
                for k, v in inst.get_fields().items():
                    close((v.tensor if isinstance(v, Boxes) else v).cpu(), z[f"it{it}::{name}{i}.{k}"], 1e-5, f"it{it} {name}{i}.{k}")
        ref = {k[len(f"it{it}::loss::"):]: float(z[k]) for k in z.files if k.startswith(f"it{it}::loss::")}
        assert set(records[it]) == set(ref), (sorted(records[it]), sorted(ref))
        bad = {k: (records[it][k], v) for k, v in ref.items() if not abs(records[it][k] - v) < 1e-4 * max(1.0, abs(v))}
        assert not bad, (it, bad)
    for k, v in merge.state_dict().items():
        close(v.cpu(), z["it1::m_after::" + k], tol, "merge " + k)
    sd = student.state_dict()
    for k in names:
        close(sd[k].float().cpu(), T(z["it1::s_after::" + k]).float(), tol, "student " + k)


# ------------------------------------------------------------------------------------------ sync-free (packed) losses
def losses_packed_pretrain(device):
    """`losses_packed` (fixed-shape rows with per-row labels / validity) == FastRCNNOutputLayers.losses('pre_train') of the reference
    on the same rows, with invalid filler rows appended (they must not change anything)."""
    from coin_amd.box_ops import Box2BoxTransform
    from coin_amd.modeling.fast_rcnn import FastRCNNOutputLayers
    from coin_amd.modeling.roi_heads import PackedSamples
    from coin_amd.structures import ShapeSpec
    from golden_util import LOSS_W

    dev = torch.device(device)
    for tag in ("a", "empty_image", "no_fg", "clipart", "focal"):
        z = load(f"box_predictor_pretrain_{tag}")
        with kernels_for(device):
            det = tiny_product_detector()
            bp = FastRCNNOutputLayers(ShapeSpec(channels=64, height=1, width=1), text_encoder=det.roi_heads.box_predictor.text_encoder,
                                      pooling_type="meanpool", box2box_transform=Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=32,
                                      classes_weight=[1.0] * K + [0.9], loss_type="MILFocalLoss" if tag == "focal" else "MILCrossEntropy",
                                      cls_agnostic_bbox_reg=True, loss_weight=LOSS_W, batch_size_per_image=32, cls_b_thresh=0.3, dataset=(str(z["dataset"]),),
                                      prototype_update_rate=0.9996)
            load_weights(bp, z)
            bp.to(dev).train()
            n_img = int(z["n_img"])
            boxes, cls, gtb, prs = [], [], [], []
            for i in range(n_img):
                fg, bg = _inst(z, f"p{i}.fg", (96, 128)).to(dev), _inst(z, f"p{i}.bg", (96, 128)).to(dev)
                boxes += [fg.proposal_boxes.tensor, bg.proposal_boxes.tensor]
                cls += [fg.gt_classes_offline, bg.gt_classes]
                gtb += [fg.gt_boxes.tensor, bg.proposal_boxes.tensor]
                prs += [fg.gt_probs_offline, torch.zeros(len(bg), K + 1, device=dev)]
            x = T(z["x"]).to(dev)
            # 3 invalid filler rows (fewer candidates than the batch size): they must not change anything
            filler = torch.tensor([[1.0, 1, 20, 20]] * 3, device=dev)
            x_ext = torch.cat([x, torch.randn(3, x.shape[1]).to(dev)]).requires_grad_(True)
            ps = PackedSamples(torch.cat(boxes + [filler]), torch.cat(cls + [torch.full((3,), -1, device=dev)]),
                               torch.cat(gtb + [filler]), torch.cat(prs + [torch.zeros(3, K + 1, device=dev)]), 0)
            preds = bp(x_ext, "pre_train")
            losses = bp.losses_packed(preds, ps, update_prototype=bool(z["update_prototype"]))
            ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
            assert set(losses) == set(ref)
            for k, v in ref.items():
                assert abs(float(losses[k]) - v) < 1e-4 * max(1.0, abs(v)), (tag, k, float(losses[k]), v)
            sum(losses.values()).backward()
            close(x_ext.grad[: x.shape[0]], z["gx"], 1e-5 if str(device) == "cpu" else 1e-4, "gx")
            assert float(x_ext.grad[x.shape[0]:].abs().max()) == 0.0
            close(bp.text_encoder.per_class_feat, z["prototype_after"], 1e-6, "prototype")


def losses_packed_step(device, tag):
    """`losses_packed_step` (row roles + masks, no host-side row counts) == FastRCNNOutputLayers.losses(step_one / step_two) of
    the reference on the same A / B / background rows, with filler rows interleaved; the masked `merge_grad_loss` reproduces
    gradient_discrepancy_loss and the CKG gradients."""
    from coin_amd.modeling.roi_heads import PackedStepSamples
    from coin_amd.modeling.text_encoder import CKGNet
    from e2e_util import _product_box_predictor

    dev = torch.device(device)
    z = load(f"box_predictor_{tag}")
    branch = str(z["branch"])
    with kernels_for(device):
        bp = _product_box_predictor()
        load_weights(bp, z)
        merge = CKGNet(32, 32, K + 1, head_num=4)
        load_weights(merge, z, "m::")
        bp.to(dev).train()
        merge.to(dev)
        n_img = int(z["n_img"])
        size = (96, 128)
        x_ref = T(z["x"]).to(dev)
        rows, role, cls, con, coff, gtb, pon, poff, boxes = [], [], [], [], [], [], [], [], []
        cursor = 0
        zero_p = lambda n: torch.zeros(n, K + 1, device=dev)
        zl = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
        for i in range(n_img):
            a, b, g = (_inst(z, f"p{i}.{t}", size).to(dev) for t in ("a", "b", "bg"))
            for inst, r in ((a, 0), (b, 1), (g, 2)):
                n = len(inst)
                rows.append(x_ref[cursor:cursor + n])
                cursor += n
                role.append(torch.full((n,), r, device=dev))
                boxes.append(inst.proposal_boxes.tensor)
                if r == 0:
                    cls.append(inst.gt_classes), con.append(zl(n)), coff.append(zl(n))
                    gtb.append(inst.gt_boxes.tensor), pon.append(inst.gt_probs_online), poff.append(inst.gt_probs_offline)
                elif r == 1:
                    cls.append(zl(n)), con.append(inst.gt_classes_online), coff.append(inst.gt_classes_offline)
                    gtb.append(inst.gt_boxes.tensor), pon.append(inst.gt_probs_online), poff.append(inst.gt_probs_offline)
                else:
                    cls.append(inst.gt_classes), con.append(zl(n)), coff.append(zl(n))
                    gtb.append(inst.proposal_boxes.tensor), pon.append(zero_p(n)), poff.append(zero_p(n))
            # two filler rows after every image
            fb = torch.tensor([[1.0, 1, 20, 20]] * 2, device=dev)
            rows.append(torch.randn(2, x_ref.shape[1]).to(dev)), role.append(torch.full((2,), -1, device=dev)), boxes.append(fb)
            cls.append(zl(2)), con.append(zl(2)), coff.append(zl(2))
            gtb.append(fb), pon.append(zero_p(2)), poff.append(zero_p(2))
        assert cursor == x_ref.shape[0]
        x = torch.cat(rows).requires_grad_(True)
        role_t = torch.cat(role)
        has_b = bool((role_t == 1).any())
        ps = PackedStepSamples(torch.cat(boxes), role_t, torch.cat(cls), torch.cat(con), torch.cat(coff), torch.cat(gtb), torch.cat(pon),
                               torch.cat(poff), 0, n_img, has_b)
        cs = [_inst(z, f"p{i}.c", size).to(dev) for i in range(n_img)]
        xc = T(z["xc"]).to(dev)
        preds = bp(x, branch)
        cpred = bp(xc, branch, return_feats=False) if xc.shape[0] else None
        losses = bp.losses_packed_step(preds, ps, cpred, cs if xc.shape[0] else None, merge, branch, update_prototype=bool(z["update_prototype"]))
        ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
        got = {k: float(v) for k, v in losses.items()}
        if "loss_merge_a" in losses:
            lg = bp.merge_grad_loss()
            got["loss_merge_grad"] = float(lg)
            (lg + losses["loss_merge_base"]).backward(inputs=list(merge.parameters()), retain_graph=True)
            for n, p in merge.named_parameters():
                close(p.grad, z["mg::" + n], 2e-4, n)
        if "loss_cls_b" in got and "loss_cls_b" not in ref:  # no B row passed the threshold: the reference omits the (zero) term
            assert got.pop("loss_cls_b") == 0.0
        assert set(got) == set(ref), (sorted(got), sorted(ref))
        for k, v in ref.items():
            assert abs(got[k] - v) < 1e-4 * max(1.0, abs(v)), (tag, k, got[k], v)
        skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"] + ([] if branch == "step_two" else ["loss_cls_b"])
        sum(v for k, v in losses.items() if k not in skip).backward()
        real = role_t >= 0
        close(x.grad[real], z["gx"], 1e-5 if str(device) == "cpu" else 1e-4, "gx")
        assert float(x.grad[~real].abs().max()) == 0.0
        te = bp.text_encoder
        for name, buf in (("prototype_after", te.per_class_feat), ("prototype_b_online_after", te.prototype_b_online), ("prototype_b_offline_after", te.prototype_b_offline)):
            close(buf, z[name], 1e-6, name)
        params = dict(bp.named_parameters())
        for k in z.files:
            if k.startswith("g::"):
                close(params[k[3:]].grad, z[k], 1e-4, k)


# ------------------------------------------------------------------------------------------ (f)-4: CLIP-teacher relabelling (clip_rcnn.py:87-132)
def clip_relabel(device):
    """coin_amd CLIP meta-arch + CLIPRes5ROIHeads + AttentionPool2d against the reference's CLIP teacher (tests/golden/clip_relabel.npz):
    attention pooling, relabelled boxes with their full probabilities, background filter."""
    from coin_amd.modeling.backbone import CLIP_IMAGE
    from coin_amd.modeling.meta_arch import CLIP
    from coin_amd.modeling.roi_heads import CLIPRes5ROIHeads, ROIPooler
    from coin_amd.modeling.text_encoder import CLIP_TEXT
    from coin_amd.structures import Boxes, Instances

    dev = torch.device(device)
    z = load("clip_relabel")
    with kernels_for(device):
        bb = CLIP_IMAGE("RN50", freeze_at=2, layers=(1, 1, 2, 2), width=8, attnpool_dim=32, attnpool_heads=4)
        load_weights(bb, z, "bb::")
        toks = torch.zeros(K + 1, 16, dtype=torch.int)
        for i in range(K + 1):
            seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
            toks[i, : len(seq)] = torch.tensor(seq)
        te = CLIP_TEXT("RN50", ["car", "person", "bus", "backgroud"], embed_dim=32, context_length=16, vocab_size=64, width=32, heads=2, layers=2,
                       tokenized_prompts=toks, n_templates=2)
        load_weights(te, z, "te::")
        heads = CLIPRes5ROIHeads(in_features=["res4"], pooler=ROIPooler(14, (1.0 / 16,), 0, "ROIAlignV2"), text_encoder=te)
        model = CLIP(backbone=bb, roi_heads=heads, pixel_mean=[0.48145466, 0.4578275, 0.40821073], pixel_std=[0.26862954, 0.26130258, 0.27577711],
                     device=str(device))
        model.to(dev).eval()
        with torch.no_grad():
            close(bb.attnpool(T(z["attn_x"]).to(dev)), z["attn_y"], 1e-5 if str(device) == "cpu" else 1e-4, "attnpool")
        h, w = (int(v) for v in z["hw"])
        probs = T(z["probs"]).to(dev)

        def inst(n=None):
            r = Instances((h, w))
            r.pred_boxes = Boxes(T(z["boxes"]).to(dev))
            r.scores, r.pred_classes, r.probs = probs[:, :-1].max(1).values, probs[:, :-1].argmax(1), probs
            return r if n is None else r[:n]

        pre = {"file_name": "x.png", "image_id": "x", "height": h, "width": w, "RCNN": {"instances": inst()}, "RPN": {"instances": inst(4)}}
        binp = [{"image": T(z["img"]).to(dev), "height": h, "width": w, "file_name": "x.png", "image_id": "x"}]
        out = model(binp, pre)
        for tag in ("RCNN", "RPN"):
            got = out[tag]["instances"]
            assert len(got) == int(z["n_" + tag])
            assert torch.equal(got.pred_classes.cpu(), T(z[f"out_{tag}.pred_classes"]).long())
            close(got.probs, z[f"out_{tag}.probs"], 1e-4, tag + " probs")
            close(got.scores, z[f"out_{tag}.scores"], 1e-4, tag + " scores")
            close(got.pred_boxes.tensor, z[f"out_{tag}.pred_boxes"], 0, tag + " boxes")
        assert out["height"] == h and out["file_name"] == "x.png"
        with torch.no_grad():
            te.per_class_feat[K] = T(z["bg_embedding_2"]).to(dev)
        assert len(model(binp, pre)["RCNN"]["instances"]) == int(z["n2_RCNN"])
