#!/usr/bin/env python3
"""Capture golden vectors from the reference's OWN modules (run in the build container only).

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz / *.json

Every case seeds its RNG, builds the reference module with explicit constructor
kwargs (``from_config`` needs a detectron2 CfgNode and CLIP downloads), runs it on
CPU fp32 and stores inputs, weights and outputs.  The files are DATA (arrays and
scalars); no reference source text is stored.  Third-party arithmetic underneath
(detectron2/torchvision/fvcore) is ``oracle/d2.py`` - see ``_ref_shim.py``.

SURVEY.md §8c lists the cases: G1 layer4, G2 res4, G3/G4 box predictor + pre_train
losses, G5/G6 step_one/two + CKG + gradient-discrepancy, G7 MIL losses, G8 text
encoder, G9 CKG, G10 LR schedule, G11 box fusion / flip-scale, plus RPN labelling
and losses, RoI sampling, an end-to-end tiny detector step (pre_train and
step_two), the inference path and the optimizer parameter groups.
"""
from __future__ import annotations

import copy
import importlib
import json
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))   # where the fixtures are written (tests/test_reference_live.py points it at a scratch directory)
_SRC = HERE                                          # where this script lives
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import _ref_shim as shim  # noqa: E402

shim.install()
from oracle import d2  # noqa: E402

LOG = logging.getLogger("gen_golden")
torch.set_num_threads(4)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB, {len(out)} arrays)")


def sd_arrays(module, prefix="w::"):
    return {prefix + k: v for k, v in module.state_dict().items()}


# --------------------------------------------------------------------------- #
# tiny reference model factory
# --------------------------------------------------------------------------- #
K = 3  # foreground classes of the tiny cases
CLASSES = ["car", "person", "bus", "backgroud"]
WIDTH = 8  # ModifiedResNet width -> res4 = 128 ch, res5 = 256 ch
TEXT_DIM = 32
CTX = 16


def tiny_tokens():
    """[K+1, CTX] int tokens 'SOS a photo of a X X X X {cls} . EOT' over a 64-word toy vocabulary."""
    sos, eot, dot, x = 62, 63, 5, 6
    toks = torch.zeros(len(CLASSES), CTX, dtype=torch.int)
    for i in range(len(CLASSES)):
        seq = [sos, 1, 2, 3, 1, x, x, x, x, 10 + i, dot, eot]
        toks[i, : len(seq)] = torch.tensor(seq)
    return toks


def build_text_encoder():
    ct = shim.ref("coin.modeling.text_encoder.clip_text")
    enc = ct.TEXT_ENCODER(TEXT_DIM, CTX, 64, 32, 2, 2, (tiny_tokens(), 4, 4))
    enc.eval()
    enc.load_embedding(32)
    enc.float()
    enc.freeze_encoder()
    te = object.__new__(ct.CLIP_TEXT)
    nn.Module.__init__(te)
    te.type, te.target_device, te.classes = "tiny", "cpu", list(CLASSES)
    te.encoder = enc
    feat = F.normalize(torch.randn(len(CLASSES), TEXT_DIM), dim=1)
    te.register_buffer("per_class_feat", feat)
    te.register_buffer("prototype_b_online", feat.clone())
    te.register_buffer("prototype_b_offline", feat.clone())
    return te


def build_box_predictor(text_encoder, in_ch=256, dataset=("foggytrain_0.02",), loss_type="MILCrossEntropy"):
    fr = shim.ref("coin.modeling.roi_heads.fast_rcnn")
    return fr.FastRCNNOutputLayers(
        d2.ShapeSpec(channels=in_ch, height=1, width=1),
        text_encoder=text_encoder,
        pooling_type="meanpool",
        box2box_transform=d2.Box2BoxTransform((10.0, 10.0, 5.0, 5.0)),
        text_dim=TEXT_DIM,
        classes_weight=[1.0] * K + [0.9],
        loss_type=loss_type,
        test_score_thresh=0.05,
        test_nms_thresh=0.5,
        test_topk_per_image=100,
        cls_agnostic_bbox_reg=True,
        smooth_l1_beta=0.0,
        box_reg_loss_type="smooth_l1",
        loss_weight={
            "loss_box_reg": 1.0, "loss_box_reg_offline": 1.0, "loss_box_reg_online": 1.0, "loss_cls": 1.0,
            "loss_text_align": 10.0, "loss_distillation": 0.1, "loss_cls_b": 0.1,
        },
        batch_size_per_image=32,
        cls_b_thresh=0.3,
        dataset=dataset,
        prototype_update_rate=0.9996,
    )


class TinyBackbone(nn.Module):
    """Stand-in for CLIP_IMAGE (needs a CLIP download): holds the reference ModifiedResNet under the same
    attribute path (`encoder.visual`) and exposes the members the detector touches."""

    size_divisibility = 0

    def __init__(self, freeze_at=2):
        super().__init__()
        mu = shim.ref("coin.modeling.utils")
        self.encoder = nn.Module()
        self.encoder.visual = mu.ModifiedResNet(
            layers=(1, 1, 2, 2), output_dim=TEXT_DIM, heads=4, width=WIDTH, out_features=["res4"], freeze_at=freeze_at, depth=50
        )
        self.encoder.attnpool = None
        # non-trivial BN statistics / affine so that frozen and train-mode BN are both exercised
        g = torch.Generator().manual_seed(7)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, d2.FrozenBatchNorm2d)):
                m.weight.data = torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
                m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
                m.running_var.data = torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75

    @property
    def layer4(self):
        return self.encoder.visual.layer4

    @property
    def attnpool(self):
        return self.encoder.attnpool

    def output_shape(self):
        return self.encoder.visual.output_shape()

    def forward(self, x):
        return self.encoder.visual(x)


def build_detector(seed=0, bg_train=True):
    torch.manual_seed(seed)
    rcnn = shim.ref("coin.modeling.meta_arch.clip_rcnn")
    rh = shim.ref("coin.modeling.roi_heads.clip_roi_heads")
    rpn = shim.ref("coin.modeling.proposal_generator.rpn")
    backbone = TinyBackbone()
    te = build_text_encoder()
    bp = build_box_predictor(te)
    roi_heads = rh.OpenVocabularyRes5ROIHeads(
        in_features=["res4"],
        pooler=d2.ROIPooler(output_size=14, scales=(1.0 / 16,), sampling_ratio=0, pooler_type="ROIAlignV2"),
        box_predictor=bp,
        pooling_type="meanpool",
        mask_head=None,
        logger=logging.getLogger("ref"),
        BG_TRAIN=bg_train,
        num_classes=K,
        batch_size_per_image=32,
        positive_fraction=0.25,
        proposal_matcher=d2.Matcher([0.5], [0, 1], allow_low_quality_matches=False),
        proposal_append_gt=True,
    )
    anchor_gen = d2.DefaultAnchorGenerator(sizes=[[32, 64, 128]], aspect_ratios=[[0.5, 1.0, 2.0]], strides=[16])
    pg = rpn.DualTeacherRPN(
        BG_TRAIN=bg_train,
        in_features=["res4"],
        head=d2.StandardRPNHead(128, 9),
        anchor_generator=anchor_gen,
        anchor_matcher=d2.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
        box2box_transform=d2.Box2BoxTransform((1.0, 1.0, 1.0, 1.0)),
        batch_size_per_image=64,
        positive_fraction=0.5,
        pre_nms_topk=(200, 120),
        post_nms_topk=(60, 40),
        nms_thresh=0.7,
        min_box_size=0.0,
        anchor_boundary_thresh=-1.0,
        loss_weight={"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0, "loss_rpn_distillation": 0.1},
        box_reg_loss_type="smooth_l1",
        smooth_l1_beta=0.0,
    )
    model = rcnn.OpenVocabularyRCNN(
        backbone=backbone, proposal_generator=pg, roi_heads=roi_heads,
        pixel_mean=[0.48145466, 0.4578275, 0.40821073], pixel_std=[0.26862954, 0.26130258, 0.27577711],
        device="cpu", vis_period=0, input_format="RGB", logger=logging.getLogger("ref"),
    )
    # the zero-initialised bn3.weight of CLIP (clip_backbone.py:56-61) would hide the residual branch: keep non-zero.
    # larger head weights so that logits/deltas are not ~0
    with torch.no_grad():
        bp.cls_score.weight.normal_(std=0.05)
        bp.bbox_pred.weight.normal_(std=0.02)
        for l in (pg.rpn_head.conv, pg.rpn_head.objectness_logits, pg.rpn_head.anchor_deltas):
            l.weight.normal_(std=0.03)
    model.train()
    return model


def rand_boxes(n, h, w, g, min_size=16.0, max_size=None):
    max_size = max_size or min(h, w) * 0.7
    bw = torch.rand(n, generator=g) * (max_size - min_size) + min_size
    bh = torch.rand(n, generator=g) * (max_size - min_size) + min_size
    x0 = torch.rand(n, generator=g) * (w - bw)
    y0 = torch.rand(n, generator=g) * (h - bh)
    return torch.stack([x0, y0, x0 + bw, y0 + bh], dim=1)


def rand_probs(n, g, sharp=3.0):
    p = torch.softmax(sharp * torch.randn(n, K + 1, generator=g), dim=1)
    # background column forced smallest (SURVEY §8d synthetic cache recipe)
    p[:, -1] = p.min(dim=1).values * 0.5
    return p / p.sum(dim=1, keepdim=True)


def make_pretrain_batch(seed, sizes=((96, 128), (80, 112))):
    """batched_inputs for branch='pre_train': uint8 images + RCNN/RPN target Instances (Appendix A.2)."""
    g = torch.Generator().manual_seed(seed)
    MyInstances = shim.ref("coin.utils.util").MyInstances
    batch = []
    for i, (h, w) in enumerate(sizes):
        img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
        n = 5 + i
        boxes = rand_boxes(n, h, w, g)
        probs = rand_probs(n, g)
        rc = MyInstances((h, w))
        rc.gt_boxes = d2.Boxes(boxes.clone())
        rc.gt_classes_offline = probs[:, :-1].argmax(1)
        rc.gt_probs_offline = probs
        rc.gt_scores_offline = probs[:, :-1].max(1).values
        rp = MyInstances((h, w))
        rp.gt_boxes = d2.Boxes(boxes.clone())
        rp.gt_classes = probs[:, :-1].argmax(1)
        batch.append({"image": img, "RCNN": rc, "RPN": rp, "height": h, "width": w, "file_name": f"img{i}.png"})
    return batch


def instances_arrays(prefix, inst):
    out = {}
    for k, v in inst.get_fields().items():
        out[f"{prefix}.{k}"] = v.tensor if isinstance(v, d2.Boxes) else v
    return out


def grads_of(model, names):
    sd = dict(model.named_parameters())
    return {"g::" + n: (sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in names}


# --------------------------------------------------------------------------- #
# cases
# --------------------------------------------------------------------------- #
def case_mil_losses():
    L = shim.ref("coin.utils.losses")
    torch.manual_seed(11)
    x = torch.randn(40, 9) * 2.0
    hard = F.one_hot(torch.randint(0, 9, (40,)), 9).float()
    soft = torch.rand(40, 9) * (torch.rand(40, 9) > 0.5)
    soft[0] = 0.0
    soft[0, 3] = 0.7
    wts = torch.where(hard[:, -1] > 0, torch.tensor(0.9), torch.tensor(1.0))
    mil = L.MILCrossEntropy()
    out = {"x": x, "hard": hard, "soft": soft, "weights": wts}
    out["ce_hard_avg_w_mean"] = mil(x, hard, weights=wts, avg_positives=True)
    out["ce_hard_noavg_mean"] = mil(x, hard, avg_positives=False)
    out["ce_soft_avg_sum"] = mil(x, soft + 1e-3, weights=wts, avg_positives=True, reduction="sum")
    out["ce_soft_noavg_w_mean"] = mil(x, soft + 1e-3, weights=wts, avg_positives=False)
    out["ce_empty"] = mil(x[:0], hard[:0], weights=wts[:0], avg_positives=True)
    xg = x.clone().requires_grad_(True)
    mil(xg, hard, weights=wts, avg_positives=True).backward()
    out["ce_hard_avg_w_mean_grad"] = xg.grad
    alpha = torch.tensor([1.0] * 8 + [0.9])
    foc = L.MILFocalLoss(9, alpha)
    out["focal_alpha"] = alpha
    out["focal_hard_avg"] = foc(x, hard, avg_positives=True)
    out["focal_soft_noavg"] = foc(x, soft + 1e-3, avg_positives=False)
    npz("mil_losses", **out)


def case_bottleneck():
    mu = shim.ref("coin.modeling.utils")
    torch.manual_seed(21)
    b1 = mu.Bottleneck(64, 32, stride=2)
    b2 = mu.Bottleneck(128, 32, stride=1)
    net = nn.Sequential(b1, b2)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(std=0.1)
    net.train()
    x = torch.randn(6, 64, 14, 14, requires_grad=True)
    before = sd_arrays(net)
    before = {k: v.clone() for k, v in before.items()}
    y = net(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    after = {"after::" + k: v for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
    npz("bottleneck_layer4", x=x, y=y, gy=gy, gx=x.grad, g_conv1=b1.conv1.weight.grad, g_conv2=b1.conv2.weight.grad,
        g_down=b1.downsample[1].weight.grad, g_bn2_w=b1.bn2.weight.grad, g_b2_conv3=b2.conv3.weight.grad, **before, **after)


def case_resnet():
    torch.manual_seed(22)
    bb = TinyBackbone(freeze_at=2)
    with torch.no_grad():
        for n, p in bb.named_parameters():
            if n.endswith("bn3.weight"):
                p.uniform_(0.3, 0.8)
    bb.train()
    before = {k: v.clone() for k, v in sd_arrays(bb).items()}
    x = torch.randn(2, 3, 64, 96)
    y = bb(x)["res4"]
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    v = bb.encoder.visual
    req = {n: p.requires_grad for n, p in bb.named_parameters()}
    npz("resnet_res4", x=x, res4=y, gy=gy, g_l2_conv1=v.layer2[0].conv1.weight.grad, g_l3_1_conv2=v.layer3[1].conv2.weight.grad,
        g_l3_bn1_b=v.layer3[0].bn1.bias.grad,
        frozen_names=np.array([n for n, r in req.items() if not r]), **before)


def _pretrain_proposals(g, nfg=(5, 0, 3), nbg=(9, 6, 0), size=(96, 128)):
    """Per-image (fg, bg) Instances as label_and_sample_proposals(pre_train) emits (Appendix A.4)."""
    props = []
    h, w = size
    for f, b in zip(nfg, nbg):
        fg = d2.Instances(size)
        fg.proposal_boxes = d2.Boxes(rand_boxes(f, h, w, g))
        fg.objectness_logits = torch.randn(f, generator=g)
        fg.gt_boxes = d2.Boxes(fg.proposal_boxes.tensor + torch.randn(f, 4, generator=g) * 2.0)
        p = rand_probs(f, g)
        fg.gt_classes_offline = p[:, :-1].argmax(1)
        fg.gt_probs_offline = p
        fg.gt_scores_offline = p[:, :-1].max(1).values
        bg = d2.Instances(size)
        bg.proposal_boxes = d2.Boxes(rand_boxes(b, h, w, g))
        bg.objectness_logits = torch.randn(b, generator=g)
        bg.gt_classes = torch.full((b,), K, dtype=torch.int64)
        props.append((fg, bg))
    return props


def _pack_pretrain_props(props):
    out = {}
    for i, (fg, bg) in enumerate(props):
        out.update(instances_arrays(f"p{i}.fg", fg))
        out.update(instances_arrays(f"p{i}.bg", bg))
    return out


def case_box_predictor_pretrain():
    for tag, nfg, nbg, dataset, upd in [
        ("focal", (5, 0, 3), (9, 6, 4), ("foggytrain_0.02",), True),        # CLOUD.LOSS_TYPE MILFocalLoss (fast_rcnn.py:581-582)
        ("a", (5, 0, 3), (9, 6, 4), ("foggytrain_0.02",), True),
        # NB an image with fg but 0 bg trips the reference's own assert (fast_rcnn.py:383-385: `[-0:]` selects
        # every row), so that shape is not a valid input; an image with neither fg nor bg is.
        ("empty_image", (4, 0, 2), (7, 0, 3), ("foggytrain_0.02",), True),
        ("no_fg", (0, 0), (6, 5), ("foggytrain_0.02",), True),
        ("clipart", (5, 2), (6, 3), ("cliparttrain",), False),             # class_cross_loss1 branch
    ]:
        torch.manual_seed(31)
        te = build_text_encoder()
        bp = build_box_predictor(te, in_ch=64, dataset=dataset, loss_type="MILFocalLoss" if tag == "focal" else "MILCrossEntropy")
        with torch.no_grad():
            bp.cls_score.weight.normal_(std=0.05)
            bp.bbox_pred.weight.normal_(std=0.02)
        bp.train()
        g = torch.Generator().manual_seed(32)
        props = _pretrain_proposals(g, nfg, nbg)
        R = sum(nfg) + sum(nbg)
        x = torch.randn(R, 64, generator=g).requires_grad_(True)
        before = {k: v.clone() for k, v in sd_arrays(bp).items()}
        preds = bp(x, "pre_train")
        (scores, lta), deltas, feats = preds
        losses = bp.losses(preds, props, None, "pre_train", update_prototype=upd)
        total = sum(losses.values())
        total.backward()
        out = dict(x=x, scores=scores, loss_text_align_raw=lta, deltas=deltas, feats=feats, gx=x.grad,
                   prototype_after=te.per_class_feat, update_prototype=np.array(upd), dataset=np.array(dataset[0]),
                   **{"loss::" + k: v for k, v in losses.items()}, **_pack_pretrain_props(props), **before,
                   **grads_of(bp, ["trans.0.weight", "trans.2.bias", "trans.4.weight", "cls_score.weight", "cls_score.bias",
                                   "bbox_pred.weight", "text_encoder.encoder.embedding_tmp", "text_encoder.encoder.add_in_embedding"]))
        out["n_img"] = np.array(len(props))
        npz(f"box_predictor_pretrain_{tag}", **out)


def _step_proposals(g, na, nb, nbg, nc, size=(96, 128)):
    props, cs = [], []
    h, w = size
    for a, b, bgn, c in zip(na, nb, nbg, nc):
        A = d2.Instances(size)
        A.proposal_boxes = d2.Boxes(rand_boxes(a, h, w, g))
        A.objectness_logits = torch.randn(a, generator=g)
        A.gt_boxes = d2.Boxes(A.proposal_boxes.tensor + torch.randn(a, 4, generator=g) * 2.0)
        pon, poff = rand_probs(a, g), rand_probs(a, g)
        A.gt_classes = pon[:, :-1].argmax(1)
        A.gt_scores_online, A.gt_scores_offline = pon.max(1).values, poff.max(1).values
        A.gt_probs_online, A.gt_probs_offline = pon, poff
        B = d2.Instances(size)
        B.proposal_boxes = d2.Boxes(rand_boxes(b, h, w, g))
        B.objectness_logits = torch.randn(b, generator=g)
        B.gt_boxes = d2.Boxes(B.proposal_boxes.tensor + torch.randn(b, 4, generator=g) * 2.0)
        pon, poff = rand_probs(b, g), rand_probs(b, g)
        B.gt_classes_online = pon[:, :-1].argmax(1)
        B.gt_classes_offline = (B.gt_classes_online + 1) % K
        B.gt_scores_online, B.gt_scores_offline = pon.max(1).values, poff.max(1).values
        B.gt_probs_online, B.gt_probs_offline = pon, poff
        BG = d2.Instances(size)
        BG.proposal_boxes = d2.Boxes(rand_boxes(bgn, h, w, g))
        BG.objectness_logits = torch.randn(bgn, generator=g)
        BG.gt_classes = torch.full((bgn,), K, dtype=torch.int64)
        C = d2.Instances(size)
        C.gt_boxes = d2.Boxes(rand_boxes(c, h, w, g))
        pc = rand_probs(c, g)
        C.gt_classes = pc[:, :-1].argmax(1)
        C.gt_scores = pc.max(1).values
        C.gt_probs = pc
        props.append((A, B, BG))
        cs.append(C)
    return props, cs


def build_ckg(seed=41, head_num=4):
    torch.manual_seed(seed)
    ckg = shim.ref("coin.modeling.merge.ckg")
    return ckg.CKGNet(hidden_size=TEXT_DIM, all_head_size=TEXT_DIM, num_classes=K + 1, logger=None, head_num=head_num)


def case_box_predictor_step():
    L = shim.ref("coin.utils.losses")
    for tag, branch, na, nb, nbg, nc, upd in [
        ("one", "step_one", (4, 3), (2, 1), (8, 6), (3, 2), True),
        ("two", "step_two", (4, 3), (2, 2), (8, 6), (3, 0), True),
        ("two_nobg_noC", "step_two", (5, 2), (1, 2), (0, 0), (0, 0), True),
        ("two_noB", "step_two", (4, 3), (0, 0), (5, 6), (2, 2), True),
        ("one_noproto", "step_one", (4, 3), (2, 1), (8, 6), (3, 2), False),
    ]:
        torch.manual_seed(51)
        te = build_text_encoder()
        te.prototype_b_online.copy_(F.normalize(torch.randn(K + 1, TEXT_DIM), dim=1))
        bp = build_box_predictor(te, in_ch=64)
        with torch.no_grad():
            bp.cls_score.weight.normal_(std=0.05)
            bp.bbox_pred.weight.normal_(std=0.02)
        bp.train()
        merge = build_ckg()
        g = torch.Generator().manual_seed(52)
        props, cs = _step_proposals(g, na, nb, nbg, nc)
        R = sum(na) + sum(nb) + sum(nbg)
        x = torch.randn(R, 64, generator=g).requires_grad_(True)
        xc = torch.randn(sum(nc), 64, generator=g)
        before = {k: v.clone() for k, v in sd_arrays(bp).items()}
        before.update({k: v.clone() for k, v in sd_arrays(merge, "m::").items()})
        preds = bp(x, branch)
        if sum(nc):
            cpreds = bp(xc, branch, return_feats=False)
            losses = bp.losses((preds, cpreds), (props, cs), merge, branch, update_prototype=upd)
        else:
            losses = bp.losses((preds, ((None, None), None)), (props, None), merge, branch, update_prototype=upd)
        out = dict(x=x, xc=xc, branch=np.array(branch), update_prototype=np.array(upd), scores=preds[0][0], deltas=preds[1],
                   feats=preds[2], prototype_after=te.per_class_feat, prototype_b_online_after=te.prototype_b_online,
                   prototype_b_offline_after=te.prototype_b_offline, **before)
        for i, ((A, B, BG), C) in enumerate(zip(props, cs)):
            out.update(instances_arrays(f"p{i}.a", A))
            out.update(instances_arrays(f"p{i}.b", B))
            out.update(instances_arrays(f"p{i}.bg", BG))
            out.update(instances_arrays(f"p{i}.c", C))
        out["n_img"] = np.array(len(props))
        out.update({"loss::" + k: v for k, v in losses.items()})
        if "loss_merge_a" in losses:
            # trainer.py:192-197: CKG update via gradient alignment, then student loss
            holder = type("M", (), {})()
            holder.roi_heads = type("R", (), {})()
            holder.roi_heads.box_predictor = bp
            lg = L.gradient_discrepancy_loss(holder, 1e4 * losses["loss_merge_a"], 1e4 * losses["loss_merge_b"])
            out["loss::loss_merge_grad"] = lg
            (lg + losses["loss_merge_base"]).backward(retain_graph=True)
            out.update({"mg::" + n: p.grad.clone() for n, p in merge.named_parameters()})
            bp.zero_grad()
            merge.zero_grad()
            x.grad = None
        skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"] + ([] if branch == "step_two" else ["loss_cls_b"])
        total = sum(v for k, v in losses.items() if k not in skip)
        total.backward()
        out["gx"] = x.grad
        out.update(grads_of(bp, ["trans.0.weight", "trans.4.bias", "cls_score.weight", "bbox_pred.weight", "text_encoder.encoder.embedding_tmp"]))
        npz(f"box_predictor_{tag}", **out)


def case_text_encoder():
    torch.manual_seed(61)
    te = build_text_encoder()
    enc = te.encoder
    y = enc(None, add=True)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    toks = torch.randint(1, 60, (5, CTX), dtype=torch.int)
    toks[:, 9] = 63
    toks[:, 10:] = 0
    with torch.no_grad():
        y_fixed = enc(toks, add=False)
    npz("text_encoder", y_added=y, gy=gy, g_embedding_tmp=enc.embedding_tmp.grad, g_add_in=enc.add_in_embedding.grad,
        tokens_fixed=toks, y_fixed=y_fixed, **sd_arrays(te))
    # real CLIP tokenisation of the learnable prompt (data: token ids only)
    ct = shim.ref("coin.modeling.text_encoder.clip_text")
    inst = object.__new__(ct.CLIP_TEXT)
    names = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle", "backgroud"]
    toks, tmp_len, add_n = ct.CLIP_TEXT.get_token(inst, "a photo of a {}.", names, 4)
    tmpl = shim.ref("coin.modeling.utils").MODIFIED_REGION_CLIP_TEMPLATES
    fixed = ct.CLIP_TEXT.tokenize(inst, [t.format("foggy cityscapes style", "car") for t in tmpl[:6]])
    npz("clip_tokens", prompt_tokens=toks, prompt_tmp_len=np.array(tmp_len), add_prompt_num=np.array(add_n),
        classes=np.array(names), fixed_tokens_car=fixed, n_templates=np.array(len(tmpl)))


def case_ckg():
    merge = build_ckg(71)
    g = torch.Generator().manual_seed(72)
    x = torch.randn(7, TEXT_DIM, generator=g)
    poff = F.normalize(torch.randn(K + 1, TEXT_DIM, generator=g), dim=1)
    pon = F.normalize(torch.randn(K + 1, TEXT_DIM, generator=g), dim=1)
    a, b = rand_probs(7, g), rand_probs(7, g)
    y = merge(x, poff, pon, a, b)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    npz("ckg", x=x, proto_off=poff, proto_on=pon, probs_off=a, probs_on=b, y=y, gy=gy,
        **sd_arrays(merge, "m::"), **{"mg::" + n: p.grad for n, p in merge.named_parameters()})


def case_lr_and_fusion():
    sched = shim.ref("coin.solver.lr_scheduler")
    p = [nn.Parameter(torch.zeros(1)), nn.Parameter(torch.zeros(1))]
    table = {}
    for name, steps, factors, warm in [("pretrain", (40,), (1, 0.1), 8), ("final", (40, 45, 60), (1, 0.1, 0.5, 0.1), 8)]:
        opt = torch.optim.SGD([{"params": [p[0]], "lr": 0.001}, {"params": [p[1]], "lr": 0.0001}], lr=0.001, momentum=0.9)
        s = sched.WarmupTwoStageMultiStepLR(opt, list(steps), factor_list=list(factors), gamma=0.1, warmup_factor=0.001,
                                            warmup_iters=warm, warmup_method="linear")
        lrs = []
        for _ in range(70):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt.step()
            s.step()
        table[name] = lrs
    nms = shim.ref("coin.layers.nms")
    g = torch.Generator().manual_seed(81)
    ba, bb = rand_boxes(6, 200, 300, g), rand_boxes(6, 200, 300, g)
    sa, sb = torch.rand(6, generator=g), torch.rand(6, generator=g)
    fused = nms.weighted_box_fusion_split(ba, bb, sa, sb)
    # BASE_Trainer.process (base.py:80-126) is a method of a DefaultTrainer subclass; call it unbound.
    base_src_mod = shim.ref("coin.utils.util")
    MyInstances = base_src_mod.MyInstances
    import importlib.util as iu
    import types
    # engine/base.py imports data/evaluation packages; import it with those names stubbed
    for n in ("coin.data", "coin.data.build", "coin.data.dataset_mapper", "coin.evaluation"):
        shim._mod(n)
    base = shim.ref("coin.engine.base")
    res = {}
    for flip in ("no", "horizontal", "vertical"):
        inst = d2.Instances((200, 300))
        inst.pred_boxes = d2.Boxes(ba.clone())
        inst.scores = sa.clone()
        inst.pred_classes = torch.arange(6) % 3
        inst.probs = rand_probs(6, torch.Generator().manual_seed(82))
        out = base.BASE_Trainer.process(None, inst, (200, 300), (160, 270), flip)
        res["proc_" + flip] = out.gt_boxes.tensor
        out_t = base.BASE_Trainer.process(None, inst, (200, 300), (160, 270), flip, thresh=0.5)
        res["proc_thresh_" + flip] = out_t.gt_boxes.tensor
    npz("lr_fusion_process", lr_pretrain=np.array(table["pretrain"]), lr_final=np.array(table["final"]),
        box_a=ba, box_b=bb, score_a=sa, score_b=sb, fused=fused, proc_scores=sa, **res)


def case_optimizer_groups():
    sb = shim.ref("coin.solver.build")
    model = build_detector(seed=91)
    overrides = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "backbone.encoder.attnpool": 0.1,
                  "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0, "anchor_generator": 1.0}]
    params = sb.get_default_optimizer_params(model, base_lr=0.001, weight_decay_norm=0.0, bias_lr_factor=1.0,
                                             weight_decay_bias=1e-4, overrides=overrides, only_text_encoder=None)
    id2name = {id(p): n for n, p in model.named_parameters()}
    rows = [{"name": id2name[id(g["params"][0])], "lr": g["lr"], "weight_decay": g.get("weight_decay")} for g in params]
    with open(os.path.join(HERE, "optimizer_groups.json"), "w") as f:
        json.dump(rows, f, indent=0)
    print(f"  wrote optimizer_groups.json ({len(rows)} groups)")


def case_rpn():
    """DualTeacherRPN label_and_sample_anchors + losses, pre_train and step_two, RNG seeded (labels stored)."""
    model = build_detector(seed=101)
    pg = model.proposal_generator
    g = torch.Generator().manual_seed(102)
    feats = {"res4": torch.randn(2, 128, 6, 8, generator=g)}
    images = d2.ImageList(torch.zeros(2, 3, 96, 128), [(96, 128), (90, 120)])
    MyInstances = shim.ref("coin.utils.util").MyInstances

    def tgt(n, size):
        t = MyInstances(size)
        t.gt_boxes = d2.Boxes(rand_boxes(n, size[0], size[1], g, min_size=24.0))
        t.gt_classes = torch.randint(0, K, (n,), generator=g)
        return t

    gts = [tgt(3, (96, 128)), tgt(2, (90, 120))]
    torch.manual_seed(103)
    props, losses = pg(images, feats, gts, branch="pre_train")
    torch.manual_seed(103)
    anchors = pg.anchor_generator([feats["res4"]])
    labels, matched = pg.label_and_sample_anchors(anchors, gts, "pre_train")
    out = dict(feat=feats["res4"], image_sizes=np.array(images.image_sizes), anchors=anchors[0].tensor,
               **{f"gt{i}.boxes": t.gt_boxes.tensor for i, t in enumerate(gts)},
               labels=torch.stack(labels), matched_boxes=torch.stack(matched),
               **{"loss::" + k: v for k, v in losses.items()}, **sd_arrays(pg, "w::"))
    for i, p in enumerate(props):
        out[f"prop{i}.boxes"] = p.proposal_boxes.tensor
        out[f"prop{i}.logits"] = p.objectness_logits
    # step_two: (A, None, C) per image
    def ainst(n, size):
        t = tgt(n, size)
        return t

    def cinst(n, size):
        t = MyInstances(size)
        t.gt_boxes = d2.Boxes(rand_boxes(n, size[0], size[1], g, min_size=24.0))
        t.gt_probs = rand_probs(n, g)
        t.gt_classes = t.gt_probs[:, :-1].argmax(1)
        return t

    dual = [(ainst(2, (96, 128)), None, cinst(2, (96, 128))), (ainst(0, (90, 120)), None, cinst(3, (90, 120)))]
    torch.manual_seed(104)
    props2, losses2 = pg(images, feats, dual, branch="step_two")
    torch.manual_seed(104)
    lab2, mb2, idx2, dl2 = pg.label_and_sample_anchors(anchors, [[d[0] for d in dual], [d[2] for d in dual]], "step_two")
    for i, d in enumerate(dual):
        out[f"s.a{i}.boxes"] = d[0].gt_boxes.tensor
        out[f"s.c{i}.boxes"] = d[2].gt_boxes.tensor
        out[f"s.c{i}.probs"] = d[2].gt_probs
    out.update(s_labels=torch.stack(lab2), s_matched_boxes=torch.stack(mb2), s_matched_idxs=torch.stack(idx2),
               s_dist_labels=torch.stack(dl2), **{"sloss::" + k: v for k, v in losses2.items()})
    npz("rpn", **out)


def case_roi_sampling():
    model = build_detector(seed=111)
    rh = model.roi_heads
    g = torch.Generator().manual_seed(112)
    size = (96, 128)
    batch = make_pretrain_batch(113, sizes=(size, size))
    props = []
    for _ in range(2):
        p = d2.Instances(size)
        p.proposal_boxes = d2.Boxes(rand_boxes(50, 96, 128, g))
        p.objectness_logits = torch.randn(50, generator=g)
        props.append(p)
    # make a handful of proposals overlap the targets strongly
    for p, b in zip(props, batch):
        n = len(b["RCNN"])
        p.proposal_boxes.tensor[:n] = b["RCNN"].gt_boxes.tensor + torch.randn(n, 4, generator=g)
    torch.manual_seed(114)
    sampled = rh.label_and_sample_proposals([copy.deepcopy(p) for p in props], [b["RCNN"] for b in batch], branch="pre_train")
    out = {}
    for i, (p, b, (fg, bg)) in enumerate(zip(props, batch, sampled)):
        out[f"in{i}.boxes"], out[f"in{i}.logits"] = p.proposal_boxes.tensor.clone(), p.objectness_logits.clone()
        out.update(instances_arrays(f"t{i}", b["RCNN"]))
        out.update(instances_arrays(f"o{i}.fg", fg))
        out.update(instances_arrays(f"o{i}.bg", bg))
    # step branch
    na, nb, nc = (3, 2), (2, 1), (2, 2)
    sp, cs = _step_proposals(g, na, nb, (0, 0), nc)
    A = [x[0] for x in sp]
    B = [x[1] for x in sp]
    for lst in (A, B):
        for t in lst:
            t.remove("proposal_boxes")
            t.remove("objectness_logits")
    for p, a, b, c in zip(props, A, B, cs):
        k = 0
        for t in (a, b, c):
            n = len(t)
            p.proposal_boxes.tensor[10 + k: 10 + k + n] = t.gt_boxes.tensor + torch.randn(n, 4, generator=g)
            k += n
    torch.manual_seed(115)
    sampled2 = rh.label_and_sample_proposals([copy.deepcopy(p) for p in props], [A, B, cs], branch="step_two")
    for i, (p, a, b, c, (oa, ob, obg)) in enumerate(zip(props, A, B, cs, sampled2)):
        out[f"s.in{i}.boxes"], out[f"s.in{i}.logits"] = p.proposal_boxes.tensor, p.objectness_logits
        out.update(instances_arrays(f"s.a{i}", a))
        out.update(instances_arrays(f"s.b{i}", b))
        out.update(instances_arrays(f"s.c{i}", c))
        out.update(instances_arrays(f"s.o{i}.a", oa))
        out.update(instances_arrays(f"s.o{i}.b", ob))
        out.update(instances_arrays(f"s.o{i}.bg", obg))
    npz("roi_sampling", **out)


E2E_GRADS = [
    "backbone.encoder.visual.layer2.0.conv1.weight", "backbone.encoder.visual.layer3.1.bn2.weight",
    "backbone.encoder.visual.layer4.0.conv1.weight", "backbone.encoder.visual.layer4.1.conv2.weight",
    "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.anchor_deltas.bias",
    "roi_heads.box_predictor.trans.0.weight", "roi_heads.box_predictor.cls_score.weight",
    "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_predictor.text_encoder.encoder.embedding_tmp",
]


def _capture_sampling(model):
    """Wrap the two samplers so that the golden also records what they emitted (parity boundary P)."""
    rec = {}
    rh, pg = model.roi_heads, model.proposal_generator
    orig_rh, orig_pg = rh.label_and_sample_proposals, pg.label_and_sample_anchors

    def rh_wrap(proposals, targets, branch):
        rec["proposals_in"] = [(p.proposal_boxes.tensor.clone(), p.objectness_logits.clone()) for p in proposals]
        out = orig_rh(proposals, targets, branch=branch)
        rec["sampled"] = out
        return out

    def pg_wrap(anchors, gt_instances, branch):
        out = orig_pg(anchors, gt_instances, branch)
        rec["anchor_labels"] = out
        return out

    rh.label_and_sample_proposals = rh_wrap
    pg.label_and_sample_anchors = pg_wrap
    return rec


def case_e2e_pretrain():
    model = build_detector(seed=121)
    batch = make_pretrain_batch(122)
    before = {k: v.clone() for k, v in sd_arrays(model).items()}
    rec = _capture_sampling(model)
    torch.manual_seed(123)
    losses = model(copy.deepcopy(batch), branch="pre_train", update_prototype=True)
    sum(losses.values()).backward()
    out = dict(**before, **{"loss::" + k: v for k, v in losses.items()}, **grads_of(model, E2E_GRADS))
    for i, b in enumerate(batch):
        out[f"img{i}"] = b["image"]
        out.update(instances_arrays(f"rcnn{i}", b["RCNN"]))
        out.update(instances_arrays(f"rpn{i}", b["RPN"]))
    for i, (fg, bg) in enumerate(rec["sampled"]):
        out.update(instances_arrays(f"s{i}.fg", fg))
        out.update(instances_arrays(f"s{i}.bg", bg))
    for i, (b, l) in enumerate(rec["proposals_in"]):
        out[f"rpn_out{i}.boxes"], out[f"rpn_out{i}.logits"] = b, l
    out["anchor_labels"] = torch.stack(rec["anchor_labels"][0])
    out["anchor_matched_boxes"] = torch.stack(rec["anchor_labels"][1])
    out["prototype_after"] = model.roi_heads.box_predictor.text_encoder.per_class_feat
    out["after::layer4.0.bn1.running_mean"] = model.backbone.layer4[0].bn1.running_mean
    out["after::layer3.0.bn1.running_var"] = model.backbone.encoder.visual.layer3[0].bn1.running_var
    npz("e2e_pretrain", **out)


def case_e2e_step_and_inference():
    model = build_detector(seed=131)
    merge = build_ckg(132)
    batch = make_pretrain_batch(133)
    g = torch.Generator().manual_seed(134)
    MyInstances = shim.ref("coin.utils.util").MyInstances
    rc, rp = [], []
    for b in batch:
        h, w = b["height"], b["width"]
        props, cs = _step_proposals(g, (3,), (2,), (0,), (2,), size=(h, w))
        A, B, _ = props[0]
        C = cs[0]
        for t in (A, B):
            t.remove("proposal_boxes")
            t.remove("objectness_logits")
        RA = d2.Instances((h, w))
        RA.gt_boxes = d2.Boxes(A.gt_boxes.tensor.clone())
        RA.gt_classes = A.gt_classes.clone()
        RC = d2.Instances((h, w))
        RC.gt_boxes = d2.Boxes(C.gt_boxes.tensor.clone())
        RC.gt_probs = C.gt_probs.clone()
        RC.gt_classes = C.gt_classes.clone()
        rc.append((A, B, C))
        rp.append((RA, None, RC))
    before = {k: v.clone() for k, v in sd_arrays(model).items()}
    before.update({k: v.clone() for k, v in sd_arrays(merge, "m::").items()})
    rec = _capture_sampling(model)
    torch.manual_seed(135)
    losses = model(copy.deepcopy(batch), merge, (rc, rp), branch="step_two", update_prototype=True)
    skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"]
    sum(v for k, v in losses.items() if k not in skip).backward()
    out = dict(**before, **{"loss::" + k: v for k, v in losses.items()}, **grads_of(model, E2E_GRADS))
    for i, b in enumerate(batch):
        out[f"img{i}"] = b["image"]
        out.update(instances_arrays(f"a{i}", rc[i][0]))
        out.update(instances_arrays(f"b{i}", rc[i][1]))
        out.update(instances_arrays(f"c{i}", rc[i][2]))
        out.update(instances_arrays(f"rpn_a{i}", rp[i][0]))
        out.update(instances_arrays(f"rpn_c{i}", rp[i][2]))
    for i, (a, b_, bg) in enumerate(rec["sampled"]):
        out.update(instances_arrays(f"s{i}.a", a))
        out.update(instances_arrays(f"s{i}.b", b_))
        out.update(instances_arrays(f"s{i}.bg", bg))
    for i, (b, l) in enumerate(rec["proposals_in"]):
        out[f"rpn_out{i}.boxes"], out[f"rpn_out{i}.logits"] = b, l
    lab, mb, idx, dl = rec["anchor_labels"]
    out.update(anchor_labels=torch.stack(lab), anchor_matched_boxes=torch.stack(mb), anchor_matched_idxs=torch.stack(idx),
               anchor_dist_labels=torch.stack(dl))
    npz("e2e_step_two", **out)

    # inference path (clip_rcnn.py:381-426, fast_rcnn.py:116-175,648-671) on a fresh model in eval mode
    model = build_detector(seed=141)
    with torch.no_grad():
        model.roi_heads.box_predictor.cls_score.weight.normal_(std=0.3)
    model.eval()
    model.roi_heads.box_predictor.test_score_thresh = 0.05
    batch = make_pretrain_batch(142)
    for b in batch:
        b["height"], b["width"] = b["height"] * 2, b["width"] * 2  # exercise _postprocess rescale
    with torch.no_grad():
        res = model([{k: v for k, v in b.items() if k in ("image", "height", "width")} for b in batch], branch="test")
    out = dict(**sd_arrays(model))
    for i, (b, r) in enumerate(zip(batch, res)):
        out[f"img{i}"] = b["image"]
        out[f"hw{i}"] = np.array([b["height"], b["width"]])
        out.update(instances_arrays(f"det{i}", r["instances"]))
    npz("inference", **out)


def case_ema():
    ts = shim.ref("coin.modeling.meta_arch.ts_ensemble")
    torch.manual_seed(151)
    s = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    t = copy.deepcopy(s)
    with torch.no_grad():
        for p in s.parameters():
            p.add_(torch.randn_like(p))
        s[1].running_mean.normal_()
        s[1].num_batches_tracked.fill_(5)
    ens = object.__new__(ts.EnsembleTSModel)
    nn.Module.__init__(ens)
    ens.offline_teacher, ens.model_student = t, s
    before = {"t::" + k: v.clone() for k, v in t.state_dict().items()}
    ens.update_params(0.9996, "offline")
    npz("ema", **before, **{"s::" + k: v for k, v in s.state_dict().items()}, **{"after::" + k: v for k, v in t.state_dict().items()})

# --------------------------------------------------------------------------- #
# CoinTrainer: dual-teacher box matching (trainer.py:338-485, util.py:434-507)
# --------------------------------------------------------------------------- #
def import_ref_trainer():
    """coin/engine/trainer.py imports the whole training stack; everything that is not the matching logic is stubbed."""
    for n in ("coin.data", "coin.data.build", "coin.data.dataset_mapper", "coin.evaluation", "coin.engine.hooks", "coin.checkpoint",
              "coin.checkpoint.detection_checkpoint", "fvcore.common", "fvcore.common.checkpoint", "detectron2.engine.hooks"):
        if n not in sys.modules:
            shim._mod(n)
    shim.install()
    ma = sys.modules["coin.modeling.meta_arch"]
    if not hasattr(ma, "build_model"):
        ma.build_model = None
        ma.EnsembleTSModel = shim.ref("coin.modeling.meta_arch.ts_ensemble").EnsembleTSModel
    if not hasattr(sys.modules["detectron2.engine"], "hooks"):
        sys.modules["detectron2.engine"].hooks = sys.modules["detectron2.engine.hooks"]
    return shim.ref("coin.engine.trainer")


def _teacher_inst(boxes, classes, probs, size):
    """What BASE_Trainer.process leaves on an online / offline result: gt_boxes, gt_classes, scores, probs."""
    util = shim.ref("coin.utils.util")
    inst = util.MyInstances(size)
    inst.gt_boxes = d2.Boxes(boxes.clone())
    inst.gt_classes = classes.clone()
    inst.probs = probs.clone()
    inst.scores = probs[:, :-1].max(dim=1).values.clone()
    return inst


def _match_cases():
    """name -> (online boxes/classes, offline boxes/classes) designed to hit every branch of match_dual_teacher."""
    B = lambda rows: torch.tensor(rows, dtype=torch.float32)
    C = lambda xs: torch.tensor(xs, dtype=torch.long)
    base_on = B([[10, 10, 60, 70], [80, 20, 140, 90], [30, 100, 90, 150], [150, 60, 200, 120], [5, 5, 25, 25], [100, 100, 160, 150]])
    cases = {}
    # normal: A (same class), B (different class), offline-only C, online-only C, one online box matched by two offline boxes
    cases["normal"] = (base_on, C([0, 1, 2, 0, 1, 2]),
                       B([[12, 12, 62, 68], [82, 22, 138, 88], [28, 98, 92, 152], [210, 10, 250, 60], [101, 99, 158, 149], [98, 102, 161, 152], [60, 160, 100, 190]]),
                       C([0, 2, 2, 1, 2, 0, 1]))
    cases["online_empty"] = (B([]).reshape(0, 4), C([]), B([[12, 12, 62, 68], [82, 22, 138, 88], [28, 98, 92, 152], [210, 10, 250, 60]]), C([0, 2, 2, 1]))
    cases["offline_empty"] = (base_on[:4], C([0, 1, 2, 0]), B([]).reshape(0, 4), C([]))
    cases["both_empty"] = (B([]).reshape(0, 4), C([]), B([]).reshape(0, 4), C([]))
    # the teacher's class-wise NMS emits the SAME box under several labels: duplicates matched (same label present / absent) and unmatched
    cases["offline_duplicates"] = (base_on, C([0, 1, 2, 0, 1, 2]),
                                   B([[12, 12, 62, 68], [12, 12, 62, 68], [82, 22, 138, 88], [82, 22, 138, 88], [28, 98, 92, 152], [220, 130, 260, 180], [220, 130, 260, 180],
                                      [220, 130, 260, 180], [150, 62, 199, 118]]),
                                   C([0, 1, 2, 0, 2, 0, 1, 2, 0]))
    # two online boxes with IoU >= 0.95 and different classes, both matched: online_boxes_merging (consistent / inconsistent offline votes)
    cases["online_self_overlap"] = (B([[10, 10, 60, 70], [10, 10, 60, 70.5], [80, 20, 140, 90], [80.2, 20, 140, 90], [30, 100, 90, 150]]), C([0, 1, 1, 2, 2]),
                                    B([[11, 11, 61, 69], [81, 21, 139, 89], [79, 19, 141, 91], [29, 99, 91, 151]]), C([0, 1, 2, 2]))
    return cases


def case_match_dual_teacher():
    import random
    tr = import_ref_trainer()
    out_all = {}
    g = torch.Generator().manual_seed(171)
    size = (200, 300)
    for name, (bon, con, boff, coff) in _match_cases().items():
        pon, poff = rand_probs(len(bon), g), rand_probs(len(boff), g)
        # make argmax of the teacher probabilities agree with the labels (as the collectors guarantee)
        for p, c in ((pon, con), (poff, coff)):
            if len(c):
                top = p[:, :-1].max(dim=1).values
                p[torch.arange(len(c)), c] = top + 0.05
                p /= p.sum(dim=1, keepdim=True)
        if name == "online_empty":  # scores around the 0.8 split of trainer.py:351
            poff[0] = torch.tensor([0.9, 0.04, 0.03, 0.03])
            poff[1] = torch.tensor([0.05, 0.05, 0.85, 0.05])
        out_all.update({f"{name}::on_boxes": bon, f"{name}::on_classes": con, f"{name}::on_probs": pon,
                        f"{name}::off_boxes": boff, f"{name}::off_classes": coff, f"{name}::off_probs": poff})
        for wname, weight in (("w1", 1.0), ("w05", 0.5)):
            for tag in ("RCNN", "RPN"):
                stub = type("Stub", (), {})()
                stub.cfg = type("Cfg", (), {})()
                stub.cfg.CLOUD = type("Cloud", (), {})()
                stub.cfg.CLOUD.MATCHER = type("Matcher", (), {"IOU_THRESHOLDS": 0.5})()
                stub.WEIGHT_FOR_BOX_A = weight
                stub.merge_boxes = lambda *a, _s=stub: tr.CoinTrainer.merge_boxes(_s, *a)
                online = {"RCNN": _teacher_inst(bon, con, pon, size), "RPN": _teacher_inst(bon, con, pon, size)}
                offline = _teacher_inst(boff, coff, poff, size)
                random.seed(1234)
                a, b, c = tr.CoinTrainer.match_dual_teacher(stub, online, offline, tag, torch.device("cpu"))
                key = f"{name}::{wname}::{tag}"
                out_all.update(instances_arrays(key + "::a", a))
                if b is not None:
                    out_all.update(instances_arrays(key + "::b", b))
                out_all.update(instances_arrays(key + "::c", c))
                out_all[key + "::n"] = np.array([len(a), -1 if b is None else len(b), len(c)])
    npz("match_dual_teacher", **out_all)

def case_e2e_coin_step():
    """One whole target-detector iteration (coin/engine/trainer.py:160-218) scripted line by line with the reference's own pieces:
    teacher inference -> CoinTrainer.match_boxes -> student step_two forward with the CKG module -> gradient_discrepancy_loss +
    CKG optimizer step -> student loss + optimizer step.  (The trainer class itself needs the whole detectron2 training stack.)"""
    import random
    tr = import_ref_trainer()
    base = shim.ref("coin.engine.base")
    Lref = shim.ref("coin.utils.losses")
    sb = shim.ref("coin.solver.build")
    student, teacher, merge = build_detector(seed=151), build_detector(seed=152), build_ckg(153)
    with torch.no_grad():
        teacher.roi_heads.box_predictor.cls_score.weight.normal_(std=0.3)
    teacher.roi_heads.box_predictor.test_score_thresh = 0.05
    batch = make_pretrain_batch(154)
    for i, b in enumerate(batch):
        b["image_id"], b["random_flip"] = f"id{i}", "no"
        del b["RCNN"], b["RPN"]
    before = {k: v.clone() for k, v in sd_arrays(student, "s::").items()}
    before.update({k: v.clone() for k, v in sd_arrays(teacher, "t::").items()})
    before.update({k: v.clone() for k, v in sd_arrays(merge, "m::").items()})
    # 1. teacher inference on the weak views (trainer.py:174-178)
    teacher.eval()
    with torch.no_grad():
        offline = teacher([{k: v for k, v in b.items() if k in ("image", "height", "width")} for b in batch], branch="test")
    teacher.train()
    # cached cloud-detector results derived from the teacher's detections so that A, B and both kinds of C are populated
    g = torch.Generator().manual_seed(156)
    cloud = {}
    for b, o in zip(batch, offline):
        det = o["instances"]
        n = min(6, len(det))
        boxes = det.pred_boxes.tensor[:n] + 1.5 * torch.randn(n, 4, generator=g)
        probs = det.probs[:n].clone()
        for j in range(n):
            if j % 3 == 1:  # a different label than the teacher's -> B
                probs[j, :K] = probs[j, :K].roll(1)
        extra = rand_boxes(2, b["height"], b["width"], g)
        pe = rand_probs(2, g)
        boxes, probs = torch.cat([boxes, extra]), torch.cat([probs, pe])

        def inst():
            r = d2.Instances((b["height"], b["width"]))
            r.pred_boxes = d2.Boxes(boxes.clone())
            r.scores = probs[:, :-1].max(1).values
            r.pred_classes = probs[:, :-1].argmax(1)
            r.probs = probs.clone()
            return r

        cloud[b["file_name"]] = {"file_name": b["file_name"], "image_id": b["image_id"], "height": b["height"], "width": b["width"],
                                 "RCNN": {"instances": inst()}, "RPN": {"instances": inst()}}
    stub = type("Stub", (), {})()
    stub.cfg = type("Cfg", (), {})()
    stub.cfg.CLOUD = type("Cloud", (), {})()
    stub.cfg.CLOUD.MATCHER = type("Matcher", (), {"IOU_THRESHOLDS": 0.5})()
    stub.WEIGHT_FOR_BOX_A = 0.5
    stub.model_CLOUD = lambda fn: copy.deepcopy(cloud[fn])
    stub.process = lambda *a, **k: base.BASE_Trainer.process(stub, *a, **k)
    stub.preprocess_results = lambda *a, **k: base.BASE_Trainer.preprocess_results(stub, *a, **k)
    stub.merge_boxes = lambda *a: tr.CoinTrainer.merge_boxes(stub, *a)
    stub.match_dual_teacher = lambda *a: tr.CoinTrainer.match_dual_teacher(stub, *a)
    random.seed(77)
    rcnn, rpn = tr.CoinTrainer.match_boxes(stub, batch, copy.deepcopy(offline))
    # 2. student step (trainer.py:184-205) with plain SGD over the reference's parameter groups
    overrides = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0}]
    groups = lambda m, ov: sb.get_default_optimizer_params(m, base_lr=0.01, weight_decay_norm=0.0, bias_lr_factor=1.0, weight_decay_bias=1e-4,
                                                            overrides=ov, only_text_encoder=None)
    opt_s = torch.optim.SGD(groups(student, overrides), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt_m = torch.optim.SGD(groups(merge, overrides), lr=0.01, momentum=0.9, weight_decay=1e-4)
    rec = _capture_sampling(student)
    torch.manual_seed(155)
    record = student(copy.deepcopy(batch), merge, (rcnn, rpn), branch="step_two", update_prototype=True)
    opt_s.zero_grad()
    opt_m.zero_grad()
    assert "loss_merge_a" in record, sorted(record)
    record["loss_merge_grad"] = Lref.gradient_discrepancy_loss(student, 1e4 * record["loss_merge_a"], 1e4 * record["loss_merge_b"])
    (record["loss_merge_grad"] + record["loss_merge_base"]).backward(retain_graph=True)
    opt_m.step()
    opt_s.zero_grad()
    opt_m.zero_grad()
    skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"]
    sum(v for k, v in record.items() if k not in skip).backward()
    opt_s.step()
    out = dict(**before, **{"loss::" + k: v for k, v in record.items()})
    for i, (b, o) in enumerate(zip(batch, offline)):
        out[f"img{i}"] = b["image"]
        out.update(instances_arrays(f"det{i}", o["instances"]))
        out.update(instances_arrays(f"cloud{i}", cloud[b["file_name"]]["RCNN"]["instances"]))
        for name, t in (("a", rcnn[i][0]), ("b", rcnn[i][1]), ("c", rcnn[i][2]), ("rpn_a", rpn[i][0]), ("rpn_c", rpn[i][2])):
            out.update(instances_arrays(f"{name}{i}", t))
    for i, (a, b_, bg) in enumerate(rec["sampled"]):
        out.update(instances_arrays(f"s{i}.a", a))
        out.update(instances_arrays(f"s{i}.b", b_))
        out.update(instances_arrays(f"s{i}.bg", bg))
    lab, mb, idx, dl = rec["anchor_labels"]
    out.update(anchor_labels=torch.stack(lab), anchor_matched_boxes=torch.stack(mb), anchor_matched_idxs=torch.stack(idx),
               anchor_dist_labels=torch.stack(dl))
    out.update({"m_after::" + k: v for k, v in merge.state_dict().items()})
    names = ["roi_heads.box_predictor.trans.0.weight", "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.bias",
             "backbone.encoder.visual.layer3.0.conv1.weight", "backbone.encoder.visual.layer4.1.bn2.weight",
             "proposal_generator.rpn_head.conv.weight", "roi_heads.box_predictor.text_encoder.encoder.embedding_tmp"]
    sd = student.state_dict()
    out.update({"s_after::" + k: sd[k] for k in names})
    out["n_abc"] = np.array([[len(t[0]), len(t[1]), len(t[2])] for t in rcnn])
    npz("e2e_coin_step", **out)


def case_e2e_coin_two_steps():
    """TWO consecutive target-detector iterations (coin/engine/trainer.py:149-218) scripted with the reference's own pieces, in the
    reference's order: [EMA of the offline teacher, ts_ensemble.py:39-69] -> teacher inference on the weak views -> match_boxes ->
    student step_two + CKG step + student step -> after_step (WEIGHT_FOR_BOX_A 1.0 -> 0.5, trainer.py:150-157).  BURN_UP_STEP = 0 and
    OFFLINE_TEACHER_UPDATE_ITER = 1: an EMA is due at BOTH iterations, the second one reads the weights the first optimizer step
    wrote, the second teacher pass reads the EMA'd teacher, and the second matching runs with the fused A boxes.  Pins the
    product's `prepare_next` pipelining (teacher stream) to the reference's ordering."""
    import random
    tr = import_ref_trainer()
    base = shim.ref("coin.engine.base")
    Lref = shim.ref("coin.utils.losses")
    sb = shim.ref("coin.solver.build")
    ts = shim.ref("coin.modeling.meta_arch.ts_ensemble")
    keep = 0.9
    student, teacher, merge = build_detector(seed=161), build_detector(seed=172), build_ckg(163, head_num=8)   # 8 heads: what CKGNet.from_config builds (ckg.py:95-107)
    with torch.no_grad():
        teacher.roi_heads.box_predictor.cls_score.weight.normal_(std=0.3)
        student.roi_heads.box_predictor.cls_score.weight.normal_(std=0.3)
    teacher.roi_heads.box_predictor.test_score_thresh = 0.05
    batches = [make_pretrain_batch(164), make_pretrain_batch(165)]
    for it, batch in enumerate(batches):
        for i, b in enumerate(batch):
            b["file_name"] = f"it{it}_img{i}.png"
            b["image_id"], b["random_flip"] = f"id{it}_{i}", "no"
            del b["RCNN"], b["RPN"]
    out = {k: v.clone() for k, v in sd_arrays(student, "s::").items()}
    out.update({k: v.clone() for k, v in sd_arrays(teacher, "t::").items()})
    out.update({k: v.clone() for k, v in sd_arrays(merge, "m::").items()})
    ens = object.__new__(ts.EnsembleTSModel)
    nn.Module.__init__(ens)
    ens.offline_teacher, ens.model_student = teacher, student
    overrides = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0}]
    groups = lambda m, ov: sb.get_default_optimizer_params(m, base_lr=0.01, weight_decay_norm=0.0, bias_lr_factor=1.0, weight_decay_bias=1e-4,
                                                            overrides=ov, only_text_encoder=None)
    opt_s = torch.optim.SGD(groups(student, overrides), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt_m = torch.optim.SGD(groups(merge, overrides), lr=0.01, momentum=0.9, weight_decay=1e-4)
    stub = type("Stub", (), {})()
    stub.cfg = type("Cfg", (), {})()
    stub.cfg.CLOUD = type("Cloud", (), {})()
    stub.cfg.CLOUD.MATCHER = type("Matcher", (), {"IOU_THRESHOLDS": 0.5})()
    stub.WEIGHT_FOR_BOX_A = 1.0                                          # trainer.py:111
    cloud = {}
    stub.model_CLOUD = lambda fn: copy.deepcopy(cloud[fn])
    stub.process = lambda *a, **k: base.BASE_Trainer.process(stub, *a, **k)
    stub.preprocess_results = lambda *a, **k: base.BASE_Trainer.preprocess_results(stub, *a, **k)
    stub.merge_boxes = lambda *a: tr.CoinTrainer.merge_boxes(stub, *a)
    stub.match_dual_teacher = lambda *a: tr.CoinTrainer.match_dual_teacher(stub, *a)
    rec = _capture_sampling(student)
    g = torch.Generator().manual_seed(166)
    names = ["roi_heads.box_predictor.trans.0.weight", "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.bias",
             "backbone.encoder.visual.layer3.0.conv1.weight", "backbone.encoder.visual.layer4.1.bn2.weight",
             "backbone.encoder.visual.layer4.1.bn2.running_mean", "backbone.encoder.visual.layer3.0.bn1.num_batches_tracked",
             "proposal_generator.rpn_head.conv.weight", "roi_heads.box_predictor.text_encoder.encoder.embedding_tmp",
             "roi_heads.box_predictor.text_encoder.per_class_feat", "roi_heads.box_predictor.text_encoder.prototype_b_online",
             "roi_heads.box_predictor.text_encoder.prototype_b_offline"]
    for it, batch in enumerate(batches):
        T_ = f"it{it}::"
        # trainer.py:170-172: the EMA comes first (BURN_UP_STEP = 0, OFFLINE_TEACHER_UPDATE_ITER = 1)
        ens.update_params(keep_rate=keep, name="offline")
        tsd = teacher.state_dict()
        out.update({T_ + "t_ema::" + k: tsd[k].clone() for k in names})
        teacher.eval()
        with torch.no_grad():
            offline = teacher([{k: v for k, v in b.items() if k in ("image", "height", "width")} for b in batch], branch="test")
        teacher.train()
        for b, o in zip(batch, offline):
            det = o["instances"]
            n = min(6, len(det))
            boxes = det.pred_boxes.tensor[:n] + 1.5 * torch.randn(n, 4, generator=g)
            probs = det.probs[:n].clone()
            for j in range(n):
                if j % 3 == 1:  # a different label than the teacher's -> B
                    probs[j, :K] = probs[j, :K].roll(1)
            extra = rand_boxes(2, b["height"], b["width"], g)
            pe = rand_probs(2, g)
            boxes, probs = torch.cat([boxes, extra]), torch.cat([probs, pe])

            def inst(boxes=boxes, probs=probs, b=b):
                r = d2.Instances((b["height"], b["width"]))
                r.pred_boxes = d2.Boxes(boxes.clone())
                r.scores = probs[:, :-1].max(1).values
                r.pred_classes = probs[:, :-1].argmax(1)
                r.probs = probs.clone()
                return r

            cloud[b["file_name"]] = {"file_name": b["file_name"], "image_id": b["image_id"], "height": b["height"], "width": b["width"],
                                     "RCNN": {"instances": inst()}, "RPN": {"instances": inst()}}
        random.seed(77 + it)
        rcnn, rpn = tr.CoinTrainer.match_boxes(stub, batch, copy.deepcopy(offline))
        torch.manual_seed(155 + it)
        record = student(copy.deepcopy(batch), merge, (rcnn, rpn), branch="step_two", update_prototype=True)
        opt_s.zero_grad()
        opt_m.zero_grad()
        LOG.warning("it %d: detections %s  A/B/C %s", it, [len(o["instances"]) for o in offline], [[len(t[0]), len(t[1]), len(t[2])] for t in rcnn])
        assert "loss_merge_a" in record, sorted(record)
        record["loss_merge_grad"] = Lref.gradient_discrepancy_loss(student, 1e4 * record["loss_merge_a"], 1e4 * record["loss_merge_b"])
        (record["loss_merge_grad"] + record["loss_merge_base"]).backward(retain_graph=True)
        opt_m.step()
        opt_s.zero_grad()
        opt_m.zero_grad()
        skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"]
        sum(v for k, v in record.items() if k not in skip).backward()
        opt_s.step()
        stub.WEIGHT_FOR_BOX_A = 0.5                                      # after_step, trainer.py:150-157 (iter >= BURN_UP_STEP)
        out.update({T_ + "loss::" + k: v.detach().clone() for k, v in record.items()})
        for i, (b, o) in enumerate(zip(batch, offline)):
            out[T_ + f"img{i}"] = b["image"]
            out.update(instances_arrays(T_ + f"det{i}", o["instances"]))
            out.update(instances_arrays(T_ + f"cloud{i}", cloud[b["file_name"]]["RCNN"]["instances"]))
            for name, t in (("a", rcnn[i][0]), ("b", rcnn[i][1]), ("c", rcnn[i][2]), ("rpn_a", rpn[i][0]), ("rpn_c", rpn[i][2])):
                out.update(instances_arrays(T_ + f"{name}{i}", t))
        for i, (a, b_, bg) in enumerate(rec["sampled"]):
            out.update(instances_arrays(T_ + f"s{i}.a", a))
            out.update(instances_arrays(T_ + f"s{i}.b", b_))
            out.update(instances_arrays(T_ + f"s{i}.bg", bg))
        lab, mb, idx, dl = rec["anchor_labels"]
        out.update({T_ + "anchor_labels": torch.stack(lab), T_ + "anchor_matched_boxes": torch.stack(mb), T_ + "anchor_matched_idxs": torch.stack(idx),
                    T_ + "anchor_dist_labels": torch.stack(dl)})
        out[T_ + "n_abc"] = np.array([[len(t[0]), len(t[1]), len(t[2])] for t in rcnn])
        out.update({T_ + "m_after::" + k: v.clone() for k, v in merge.state_dict().items()})
        sd = student.state_dict()
        out.update({T_ + "s_after::" + k: sd[k].clone() for k in names})
    out["keep_rate"] = np.array(keep)
    npz("e2e_coin_two_steps", **out)


# --------------------------------------------------------------------------- #
# Evaluation: Pascal-VOC AP as the reference computes it (coin/evaluation/cloud_pascal_voc_evaluation.py)
# --------------------------------------------------------------------------- #
def _write_voc(root, gts, split="val"):
    os.makedirs(os.path.join(root, "Annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "ImageSets", "Main"), exist_ok=True)
    with open(os.path.join(root, "ImageSets", "Main", split + ".txt"), "w") as f:
        f.write("\n".join(gts.keys()) + "\n")
    for image_id, objs in gts.items():
        parts = ["<annotation>", f"<filename>{image_id}.png</filename>", "<size><width>300</width><height>200</height><depth>3</depth></size>"]
        for name, box, difficult, with_flags in objs:
            flags = f"<pose>Unspecified</pose><truncated>0</truncated><difficult>{difficult}</difficult>" if with_flags else ""
            parts.append(f"<object><name>{name}</name>{flags}<bndbox><xmin>{box[0]}</xmin><ymin>{box[1]}</ymin><xmax>{box[2]}</xmax><ymax>{box[3]}</ymax></bndbox></object>")
        parts.append("</annotation>")
        with open(os.path.join(root, "Annotations", image_id + ".xml"), "w") as f:
            f.write("".join(parts))


def case_voc_eval():
    import tempfile
    classes = ["car", "person", "bus"]
    rng = np.random.default_rng(181)
    # detectron2 pieces the evaluator touches: metadata, PathManager, comm.gather, the base class
    root = tempfile.mkdtemp(prefix="voc_golden_")
    meta = type("Meta", (), {"dirname": root, "split": "val", "thing_classes": classes, "year": 2007})()
    sys.modules["detectron2.data"].MetadataCatalog = type("MC", (), {"get": staticmethod(lambda name: meta)})
    sys.modules["detectron2.utils.comm"].gather = lambda data, dst=0: [data]
    sys.modules["detectron2.utils.file_io"].PathManager = type("PM", (), {"open": staticmethod(lambda p, mode="r", **k: open(p, mode)),
                                                                           "get_local_path": staticmethod(lambda p: p)})
    sys.modules["detectron2.evaluation"].DatasetEvaluator = object
    ev = shim.ref("coin.evaluation.cloud_pascal_voc_evaluation")
    gts, dets = {}, {}
    for i in range(12):
        image_id = f"img_{i:03d}"
        objs = []
        for _ in range(int(rng.integers(0, 5))):
            x0, y0 = int(rng.integers(0, 200)), int(rng.integers(0, 120))
            w, h = int(rng.integers(15, 90)), int(rng.integers(15, 70))
            objs.append((classes[int(rng.integers(0, 3))], (x0, y0, x0 + w, y0 + h), int(rng.random() < 0.2), bool(rng.random() < 0.8)))
        gts[image_id] = objs
        rows = []
        for name, box, _, _ in objs:  # detections near the ground truth (some with the wrong label), duplicates, and clutter
            if rng.random() < 0.85:
                j = rng.normal(0, 4, 4)
                cls = classes.index(name) if rng.random() < 0.8 else int(rng.integers(0, 3))
                rows.append((box[0] + j[0], box[1] + j[1], box[2] + j[2], box[3] + j[3], float(rng.random() * 0.6 + 0.4), cls))
                if rng.random() < 0.3:
                    rows.append((box[0] + j[1], box[1] + j[0], box[2] + j[3], box[3] + j[2], float(rng.random() * 0.5 + 0.2), cls))
        for _ in range(int(rng.integers(0, 4))):
            x0, y0 = rng.random() * 220, rng.random() * 140
            rows.append((x0, y0, x0 + 20 + rng.random() * 60, y0 + 20 + rng.random() * 40, float(rng.random() * 0.5), int(rng.integers(0, 3))))
        dets[image_id] = np.array(rows, dtype=np.float64).reshape(-1, 6)
    _write_voc(root, gts)
    out = {"classes": np.array(classes), "image_ids": np.array(list(gts.keys()))}
    for image_id, objs in gts.items():
        out[f"gt::{image_id}::names"] = np.array([o[0] for o in objs], dtype="<U16")
        out[f"gt::{image_id}::boxes"] = np.array([o[1] for o in objs], dtype=np.int64).reshape(-1, 4)
        out[f"gt::{image_id}::difficult"] = np.array([o[2] for o in objs], dtype=np.int64)
        out[f"gt::{image_id}::with_flags"] = np.array([o[3] for o in objs], dtype=bool)
        out[f"det::{image_id}"] = dets[image_id]
    for year in (2007, 2012):
        meta.year = year
        cfg = type("Cfg", (), {"OUTPUT_DIR": root, "TEST": type("T", (), {"SAVE_DETECTION_PKLS": False})()})()
        e = ev.Cloud_PascalVOCDetectionEvaluator(cfg, "synthetic_voc_val")
        e.reset()
        for image_id, d in dets.items():
            inst = d2.Instances((200, 300))
            inst.pred_boxes = d2.Boxes(torch.from_numpy(d[:, :4]).float())
            inst.scores = torch.from_numpy(d[:, 4]).float()
            inst.pred_classes = torch.from_numpy(d[:, 5]).long()
            e.process([{"image_id": image_id}], [{"instances": inst}])
        res = e.evaluate()["bbox"]
        out[f"res{year}::keys"] = np.array(list(res.keys()))
        out[f"res{year}::values"] = np.array([float(v) for v in res.values()])
    # voc_ap on hand-made curves, both metrics
    rec = np.array([0.1, 0.2, 0.2, 0.4, 0.4, 0.7, 1.0])
    prec = np.array([1.0, 1.0, 0.67, 0.75, 0.6, 0.55, 0.3])
    out["ap_curve_rec"], out["ap_curve_prec"] = rec, prec
    out["ap_curve"] = np.array([ev.voc_ap(rec, prec, True), ev.voc_ap(rec, prec, False)])
    npz("voc_eval", **out)


def case_voc_dataset():
    """coin/data/datasets/pascal_voc.py:25-83 on a synthetic tree (an unknown class, a difficult object, an image without objects)."""
    import tempfile
    root = tempfile.mkdtemp(prefix="voc_ds_")
    gts = {"a_001": [("car", (1, 1, 300, 200), 0, True), ("tram", (10, 20, 50, 60), 0, True), ("person", (33, 44, 77, 99), 1, True)],
           "a_002": [], "b_7": [("bus", (5, 6, 7, 8), 0, False)]}
    _write_voc(root, gts, split="train")
    sys.modules["detectron2.utils.file_io"].PathManager = type("PM", (), {"open": staticmethod(lambda p, mode="r", **k: open(p, mode)),
                                                                           "get_local_path": staticmethod(lambda p: p)})
    sys.modules["detectron2.structures"].BoxMode = type("BoxMode", (), {"XYXY_ABS": 0})
    sys.modules["detectron2.data"].DatasetCatalog = None
    if "coin.data" in sys.modules and not hasattr(sys.modules["coin.data"], "__path__"):
        del sys.modules["coin.data"]
    for n, rel in (("coin.data", "coin/data"), ("coin.data.datasets", "coin/data/datasets")):
        m = types.ModuleType(n)
        m.__path__ = [os.path.join(shim.REFERENCE_ROOT, rel)]
        sys.modules[n] = m
    pv = importlib.import_module("coin.data.datasets.pascal_voc")
    dicts = pv.load_voc_instances(root, "train", ["car", "person", "bus"], "png")
    for d in dicts:
        d["file_name"] = os.path.relpath(d["file_name"], root)
    with open(os.path.join(HERE, "voc_dataset.json"), "w") as f:
        json.dump({"gts": {k: [[n, list(b), dif, fl] for n, b, dif, fl in v] for k, v in gts.items()}, "dicts": dicts}, f, indent=0)
    print("  wrote voc_dataset.json")


def case_clip_relabel():
    """CLIP-teacher relabelling of the cloud detector's boxes (pre-training data collection): AttentionPool2d (utils.py:93-125),
    CLIPRes5ROIHeads (clip_roi_heads.py:19-87) and CLIP.get_clip_result / preprocess_boxes (clip_rcnn.py:87-151)."""
    mu = shim.ref("coin.modeling.utils")
    rh = shim.ref("coin.modeling.roi_heads.clip_roi_heads")
    rcnn = shim.ref("coin.modeling.meta_arch.clip_rcnn")
    torch.manual_seed(191)
    backbone = TinyBackbone()
    backbone.encoder.attnpool = mu.AttentionPool2d(7, WIDTH * 32, 4, TEXT_DIM)
    with torch.no_grad():
        for lin in (backbone.encoder.attnpool.q_proj, backbone.encoder.attnpool.k_proj, backbone.encoder.attnpool.v_proj, backbone.encoder.attnpool.c_proj):
            lin.weight.normal_(std=(WIDTH * 32) ** -0.5)
            lin.bias.normal_(std=0.02)
    te = build_text_encoder()
    with torch.no_grad():  # spread the class embeddings so that the boxes get different labels, background included
        te.per_class_feat.copy_(F.normalize(torch.randn(K + 1, TEXT_DIM), dim=1))
        te.encoder.logit_scale.fill_(float(np.log(100.0)))
    heads = rh.CLIPRes5ROIHeads(in_features=["res4"], pooler=d2.ROIPooler(output_size=14, scales=(1.0 / 16,), sampling_ratio=0, pooler_type="ROIAlignV2"),
                                text_encoder=te)
    os.makedirs("/tmp/clip_labels_golden", exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp/clip_labels_golden")  # the constructor creates ./clip_labels
    try:
        model = rcnn.CLIP(backbone=backbone, roi_heads=heads, pixel_mean=[0.48145466, 0.4578275, 0.40821073],
                          pixel_std=[0.26862954, 0.26130258, 0.27577711], device="cpu")
    finally:
        os.chdir(cwd)
    model.eval()
    g = torch.Generator().manual_seed(192)
    # attention pooling alone
    xa = torch.randn(5, WIDTH * 32, 7, 7, generator=g)
    with torch.no_grad():
        ya = backbone.encoder.attnpool(xa)
    # one image stored at 2x the network input size, cloud boxes in stored-image pixels
    img = torch.randint(0, 64, (3, 96, 128), generator=g, dtype=torch.uint8)
    img[0, :48, :64] += 190   # four differently coloured quadrants so that boxes in different places get different features
    img[1, :48, 64:] += 190
    img[2, 48:, :64] += 190
    img[:, 48:, 64:] += 120
    h, w = 192, 256
    boxes = torch.tensor([[8.0, 8, 100, 80], [140, 10, 250, 90], [10, 110, 120, 185], [150, 105, 250, 188], [60, 40, 200, 150],
                          [5, 5, 60, 60], [180, 120, 255, 190]])
    probs = rand_probs(7, g)

    def inst():
        r = d2.Instances((h, w))
        r.pred_boxes = d2.Boxes(boxes.clone())
        r.scores = probs[:, :-1].max(1).values
        r.pred_classes = probs[:, :-1].argmax(1)
        r.probs = probs.clone()
        return r

    pre = {"file_name": "x.png", "image_id": "x", "height": h, "width": w, "RCNN": {"instances": inst()}, "RPN": {"instances": inst()[:4]}}
    out = model([{"image": img, "file_name": "x.png", "image_id": "x", "height": h, "width": w}], pre)
    res = dict(**{k: v.clone() for k, v in sd_arrays(backbone, "bb::").items()}, **{k: v.clone() for k, v in sd_arrays(te, "te::").items()},
               attn_x=xa, attn_y=ya, img=img, hw=np.array([h, w]), boxes=boxes, probs=probs)
    for tag in ("RCNN", "RPN"):
        res.update(instances_arrays("out_" + tag, out[tag]["instances"]))
        res["n_" + tag] = np.array(len(out[tag]["instances"]))
    # second run: the background embedding points at box 4's feature -> that box (and whatever else prefers it) is filtered out
    with torch.no_grad():
        feats = model.backbone(model.preprocess_image([{"image": img}]).tensor)
        tmp = inst()
        tmp.proposal_boxes = model.preprocess_boxes(tmp.pred_boxes, [{"image": img, "height": h, "width": w}])
        region = backbone.encoder.attnpool(heads._shared_roi_transform([feats["res4"]], [tmp.proposal_boxes], backbone.layer4))
        te.per_class_feat[K] = F.normalize(region[4], dim=0)
    out2 = model([{"image": img, "file_name": "x.png", "image_id": "x", "height": h, "width": w}], pre)
    res["bg_embedding_2"] = te.per_class_feat[K].clone()
    res.update(instances_arrays("out2_RCNN", out2["RCNN"]["instances"]))
    res["n2_RCNN"] = np.array(len(out2["RCNN"]["instances"]))
    npz("clip_relabel", **res)


# --------------------------------------------------------------------------- #
# real-width cases (RN50 res5 on RoI tiles, predictor at D = 1024 / 9 classes): weights and inputs are SEEDED (tests/seeded.py),
# the fixture holds the reference's outputs (sub-sampled where large)
def _sub(t, *steps):
    idx = tuple(slice(None, None, s) for s in steps)
    return t[idx].clone()


def case_real_width():
    sys.path.insert(0, os.path.abspath(os.path.join(_SRC, "..")))
    import seeded

    mu = shim.ref("coin.modeling.utils")
    # ---- res5 = ModifiedResNet.layer4 at RN50 width (utils.py:184-186): Bottleneck(1024, 512, stride 2) + 2 x Bottleneck(2048, 512)
    torch.manual_seed(0)
    net = nn.Sequential(mu.Bottleneck(1024, 512, stride=2), mu.Bottleneck(2048, 512), mu.Bottleneck(2048, 512))
    seeded.fill_module(net, 501)
    net.train()
    x = seeded.randn((64, 1024, 14, 14), 502).requires_grad_(True)
    gy = seeded.randn((64, 2048), 503)
    y = net(x).mean(dim=[2, 3])
    (y * gy).sum().backward()
    p = dict(net.named_parameters())
    out = {"x_checksum": np.array(seeded.checksum(x)), "w_checksum": np.array(seeded.checksum(p["0.conv2.weight"])), "y": y,
           "gx_sub": _sub(x.grad, 8, 32, 1, 1), "g::0.conv1.weight_sub": _sub(p["0.conv1.weight"].grad, 4, 8, 1, 1),
           "g::0.conv2.weight_sub": _sub(p["0.conv2.weight"].grad, 8, 8, 1, 1), "g::0.downsample.0.weight_sub": _sub(p["0.downsample.0.weight"].grad, 16, 8, 1, 1),
           "g::1.conv3.weight_sub": _sub(p["1.conv3.weight"].grad, 16, 4, 1, 1), "g::2.conv1.weight_sub": _sub(p["2.conv1.weight"].grad, 4, 16, 1, 1)}
    for n, v in p.items():
        if v.dim() == 1:
            out["g::" + n] = v.grad
    out.update({"after::" + k: v for k, v in net.state_dict().items() if "running" in k})
    npz("real_width_res5", **out)

    # ---- FastRCNNOutputLayers at the benchmark's head sizes: 2048 -> trans -> cls_score 1024-d cosine logits vs 9 classes, bbox_pred
    ct = shim.ref("coin.modeling.text_encoder.clip_text")
    fr = shim.ref("coin.modeling.roi_heads.fast_rcnn")
    k, dim = 8, 1024
    classes = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle", "backgroud"]
    toks = torch.zeros(k + 1, CTX, dtype=torch.int)
    for i in range(k + 1):
        seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
        toks[i, : len(seq)] = torch.tensor(seq)
    torch.manual_seed(0)
    enc = ct.TEXT_ENCODER(dim, CTX, 64, 32, 2, 2, (toks, 4, 4))
    enc.eval()
    enc.load_embedding(32)
    enc.float()
    enc.freeze_encoder()
    te = object.__new__(ct.CLIP_TEXT)
    nn.Module.__init__(te)
    te.type, te.target_device, te.classes = "tiny", "cpu", classes
    te.encoder = enc
    feat = F.normalize(seeded.randn((k + 1, dim), 511), dim=1)
    te.register_buffer("per_class_feat", feat)
    te.register_buffer("prototype_b_online", feat.clone())
    te.register_buffer("prototype_b_offline", feat.clone())
    bp = fr.FastRCNNOutputLayers(
        d2.ShapeSpec(channels=2048, height=1, width=1), text_encoder=te, pooling_type="meanpool",
        box2box_transform=d2.Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=dim, classes_weight=[1.0] * k + [0.9],
        loss_type="MILCrossEntropy", test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100, cls_agnostic_bbox_reg=True,
        smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1",
        loss_weight={"loss_box_reg": 1.0, "loss_box_reg_offline": 1.0, "loss_box_reg_online": 1.0, "loss_cls": 1.0, "loss_text_align": 10.0,
                     "loss_distillation": 0.1, "loss_cls_b": 0.1},
        batch_size_per_image=256, cls_b_thresh=0.3, dataset=("foggytrain_0.02",), prototype_update_rate=0.9996)
    # the large matrices are seeded; the (small) text encoder keeps its own initialisation and is stored in the fixture
    seeded.fill_module(bp.trans, 512), seeded.fill_module(bp.cls_score, 515), seeded.fill_module(bp.bbox_pred, 516)
    with torch.no_grad():  # the seeded fill is He-scaled; the heads keep the reference's small initial scale (fast_rcnn.py:255-258)
        bp.cls_score.weight.mul_(0.2)
        bp.bbox_pred.weight.mul_(0.05)
    te_before = {"w::text_encoder." + n: v.clone() for n, v in te.state_dict().items()}
    bp.train()
    g = torch.Generator().manual_seed(513)
    size = (800, 1333)
    props = []
    for nf, nb in ((64, 192), (48, 208)):
        fg = d2.Instances(size)
        fg.proposal_boxes = d2.Boxes(rand_boxes(nf, size[0], size[1], g, 32.0, 400.0))
        fg.objectness_logits = torch.randn(nf, generator=g)
        fg.gt_boxes = d2.Boxes(fg.proposal_boxes.tensor + torch.randn(nf, 4, generator=g) * 4.0)
        pr = torch.softmax(3.0 * torch.randn(nf, k + 1, generator=g), dim=1)
        pr[:, -1] = pr.min(dim=1).values * 0.5
        pr = pr / pr.sum(dim=1, keepdim=True)
        fg.gt_classes_offline = pr[:, :-1].argmax(1)
        fg.gt_probs_offline = pr
        fg.gt_scores_offline = pr[:, :-1].max(1).values
        bg = d2.Instances(size)
        bg.proposal_boxes = d2.Boxes(rand_boxes(nb, size[0], size[1], g, 32.0, 400.0))
        bg.objectness_logits = torch.randn(nb, generator=g)
        bg.gt_classes = torch.full((nb,), k, dtype=torch.int64)
        props.append((fg, bg))
    xh = seeded.randn((512, 2048), 514).abs().requires_grad_(True)   # mean-pooled post-ReLU features are non-negative
    preds = bp(xh, "pre_train")
    (scores, lta), deltas, feats = preds
    losses = bp.losses(preds, props, None, "pre_train", update_prototype=True)
    sum(losses.values()).backward()
    q = dict(bp.named_parameters())
    out = {"x_checksum": np.array(seeded.checksum(xh)), "w_checksum": np.array(seeded.checksum(q["trans.2.weight"])), "scores": scores, "deltas": deltas,
           "feats_sub": _sub(feats, 4, 8), "gx_sub": _sub(xh.grad, 4, 8), "prototype_after": te.per_class_feat,
           "g::trans.0.weight_sub": _sub(q["trans.0.weight"].grad, 8, 16), "g::trans.2.weight_sub": _sub(q["trans.2.weight"].grad, 8, 8),
           "g::trans.4.weight_sub": _sub(q["trans.4.weight"].grad, 16, 8), "g::cls_score.weight_sub": _sub(q["cls_score.weight"].grad, 8, 16),
           "g::bbox_pred.weight": q["bbox_pred.weight"].grad, "g::trans.0.bias": q["trans.0.bias"].grad, "g::cls_score.bias": q["cls_score.bias"].grad,
           "g::text_encoder.encoder.embedding_tmp": q["text_encoder.encoder.embedding_tmp"].grad,
           "g::text_encoder.encoder.add_in_embedding": q["text_encoder.encoder.add_in_embedding"].grad,
           **{"loss::" + n: v for n, v in losses.items()}, **_pack_pretrain_props(props), "n_img": np.array(2)}
    out.update(te_before)
    npz("real_width_box_predictor", **out)


def case_rn101():
    """BASELINE configs[3] (targetDET, RN101 backbone, BDD100K: 7 classes, D = 512, MERGE_DIM 512 -- configs/coin/GDINO/clipart.yaml:4-8,
    coin/modeling/utils.py:184-186 with layers (3, 4, 23, 3), coin/data/datasets/builtin.py:162): the reference's ModifiedResNet-101
    trunk (frozen stem + layer1, train-mode BN in layer2 / layer3) on a small image and its CKGNet at 512 dims / 8 classes.
    Weights and inputs are seeded (tests/seeded.py); res5 of RN101 is RN50's (real_width_res5)."""
    sys.path.insert(0, os.path.abspath(os.path.join(_SRC, "..")))
    import seeded

    mu = shim.ref("coin.modeling.utils")
    torch.manual_seed(0)
    net = mu.ModifiedResNet(layers=(3, 4, 23, 3), output_dim=512, heads=32, width=64, out_features=["res4"], freeze_at=0, depth=101)
    seeded.fill_module(net, 601)       # filled BEFORE freezing: the FrozenBatchNorm conversion copies these statistics
    net.freeze(2)
    net.train()
    x = seeded.randn((2, 3, 96, 128), 602)
    y = net(x)["res4"]
    gy = seeded.randn(tuple(y.shape), 603)
    (y * gy).sum().backward()
    p = dict(net.named_parameters())
    out = {"x_checksum": np.array(seeded.checksum(x)), "w_checksum": np.array(seeded.checksum(p["layer3.22.conv2.weight"])), "res4": y,
           "frozen_names": np.array([n for n, q in p.items() if not q.requires_grad]),
           "g::layer3.22.conv3.weight_sub": _sub(p["layer3.22.conv3.weight"].grad, 8, 4, 1, 1), "g::layer3.11.conv2.weight_sub": _sub(p["layer3.11.conv2.weight"].grad, 4, 4, 1, 1),
           "g::layer3.0.conv1.weight_sub": _sub(p["layer3.0.conv1.weight"].grad, 2, 8, 1, 1), "g::layer2.0.conv1.weight": p["layer2.0.conv1.weight"].grad,
           "g::layer3.22.bn3.weight": p["layer3.22.bn3.weight"].grad, "g::layer3.5.bn1.bias": p["layer3.5.bn1.bias"].grad, "g::layer2.3.bn2.weight": p["layer2.3.bn2.weight"].grad,
           "after::layer3.22.bn1.running_mean": net.state_dict()["layer3.22.bn1.running_mean"], "after::layer2.0.bn3.running_var": net.state_dict()["layer2.0.bn3.running_var"]}
    npz("rn101_res4", **out)
    ckg = shim.ref("coin.modeling.merge.ckg")
    torch.manual_seed(0)
    merge = ckg.CKGNet(hidden_size=512, all_head_size=512, num_classes=8, logger=None)   # 8 heads (ckg.py default), 7 classes + background
    seeded.fill_module(merge, 611)
    g = torch.Generator().manual_seed(612)
    xm = torch.randn(40, 512, generator=g)
    poff, pon = F.normalize(torch.randn(8, 512, generator=g), dim=1), F.normalize(torch.randn(8, 512, generator=g), dim=1)
    a, b = torch.softmax(3 * torch.randn(40, 8, generator=g), 1), torch.softmax(3 * torch.randn(40, 8, generator=g), 1)
    ym = merge(xm, poff, pon, a, b)
    gym = torch.randn(ym.shape, generator=g)
    (ym * gym).sum().backward()
    npz("rn101_ckg", x=xm, proto_off=poff, proto_on=pon, probs_off=a, probs_on=b, y=ym, gy=gym,
        w_checksum=np.array(seeded.checksum(dict(merge.named_parameters())[sorted(dict(merge.named_parameters()))[0]])),
        **{"mg::" + n + "_sub": _sub(q.grad, *([4, 4] if q.dim() == 2 else [1])) for n, q in merge.named_parameters()})


# --------------------------------------------------------------------------- #
# on-disk formats: artefacts WRITTEN BY THE REFERENCE'S OWN CODE
# --------------------------------------------------------------------------- #
CKPT_DIR = "ckpt"   # under HERE


def _ckpt_cloud_results(seed, names, classes_key="pred_classes"):
    """{dataset: {file: result}} assembled by the reference's GDINO_COLLECTOR.collect() (gdino_collector.py:51-75) from a stand-in cloud
    model that returns the GDINO-shaped dict (the detector itself is out of scope): collect() moves the instances to the CPU and files them."""
    gc = shim.ref("coin.modeling.meta_arch.gdino_collector")
    comm = sys.modules["detectron2.utils.comm"]
    comm.synchronize, comm.all_gather = (lambda: None), (lambda x: [x])
    g = torch.Generator().manual_seed(seed)

    class Cloud(nn.Module):
        device = torch.device("cpu")

        def forward(self, inputs):
            d = inputs[0]
            n = 3 + len(d["file_name"]) % 3
            boxes, probs = rand_boxes(n, d["height"], d["width"], g), rand_probs(n, g)

            def inst(cls):
                r = cls((d["height"], d["width"]))
                r.pred_boxes = d2.Boxes(boxes.clone())
                r.scores = probs[:, :-1].max(1).values
                r.pred_classes = probs[:, :-1].argmax(1)
                r.probs = probs.clone()
                return r

            MyInstances = shim.ref("coin.utils.util").MyInstances
            return {"file_name": d["file_name"], "image_id": d["image_id"], "height": d["height"], "width": d["width"],
                    "RCNN": {"instances": inst(d2.Instances)}, "RPN": {"instances": inst(MyInstances)}}

    col = gc.GDINO_COLLECTOR(model=Cloud())
    col.dataloader = {"foggytrain_0.02": [[{"file_name": n, "image_id": n.split("/")[-1][:-4], "height": 96 + 8 * i, "width": 128}] for i, n in enumerate(names)]}
    col.collect()
    return col


def case_checkpoint_formats():
    """The four artefacts that cross the hot-path boundary, written by the reference's own save code under the scaffolding-only checkpoint
    stack of _ref_shim.install_checkpoint_stack (fvcore Checkpointer / detectron2 DetectionCheckpointer restated from their published
    behaviour; the containers pickle under detectron2's class paths):
      CLIP_-000001.pth            PRETrainer.collect_results -> save(iteration=-1, load_models=False, 'CLIP')      pre_train.py:138-161
      pre_train_CLIP_0000004.pth  PRETrainer.save(4, True, 'pre_train_CLIP') after 2 optimizer steps (momentum present) pre_train.py:138-146,172-175
      model_0000006.pth           CoinTrainer.save(6) over EnsembleTSModel + 4 checkpointables + AP histories      trainer.py:122-137, ts_ensemble.py:24-37
      GDINO_collect.pth           torch.save({'results': model_CLOUD.get_results()})                               pre_train.py:152-153
    """
    tr = import_ref_trainer()
    shim.install_checkpoint_stack()
    pt = shim.ref("coin.engine.pre_train")
    dc = shim.ref("coin.checkpoint.detection_checkpoint")
    ts = shim.ref("coin.modeling.meta_arch.ts_ensemble")
    sb = shim.ref("coin.solver.build")
    sched = shim.ref("coin.solver.lr_scheduler")
    out = os.path.join(HERE, CKPT_DIR)
    os.makedirs(out, exist_ok=True)
    overrides = [{"backbone.encoder.visual": 0.1, "backbone.encoder.visual.layer4": 0.1, "embedding_tmp": 1.0, "add_in_embedding": 1.0, "logit_scale": 0.0}]
    groups = lambda m: sb.get_default_optimizer_params(m, base_lr=0.01, weight_decay_norm=0.0, bias_lr_factor=1.0, weight_decay_bias=1e-4,
                                                       overrides=overrides, only_text_encoder=None)
    mk_sched = lambda opt: sched.WarmupTwoStageMultiStepLR(opt, [5, 9], factor_list=[1, 0.1, 0.01], warmup_factor=0.001, warmup_iters=3, warmup_method="linear")

    def a_few_steps(model, opt, sc, n, seed):
        g = torch.Generator().manual_seed(seed)
        for _ in range(n):   # any gradient will do: the optimizer / scheduler STATE is what the file carries
            opt.zero_grad()
            sum((p * torch.randn(p.shape, generator=g)).sum() for p in model.parameters() if p.requires_grad).backward()
            opt.step()
            sc.step()

    # ---- pre-train side
    cloud = _ckpt_cloud_results(201, ["foggy/JPEGImages/a_000001.png", "foggy/JPEGImages/b_000002.png", "foggy/JPEGImages/c_3.png"])
    torch.save({"results": cloud.get_results()}, os.path.join(out, "GDINO_collect.pth"))        # pre_train.py:152-153
    model = build_detector(seed=202)
    opt = torch.optim.SGD(groups(model), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sc = mk_sched(opt)
    fake = type("FakePRETrainer", (), {})()
    fake.collect_model = cloud                                     # CLIP_COLLECTOR.get_results() hands out the same store layout
    fake.checkpointer = dc.DetectionTSCheckpointer(model, out, optimizer=opt, scheduler=sc)   # pre_train.py:94-99
    pt.PRETrainer.save(fake, iteration=-1, load_models=False, model_name="CLIP")                 # collect_results, pre_train.py:160
    a_few_steps(model, opt, sc, 2, 203)
    pt.PRETrainer.save(fake, iteration=4, load_models=True, model_name="pre_train_CLIP")
    # ---- target-detector side
    student, teacher, merge = build_detector(seed=204), build_detector(seed=205), build_ckg(206)
    online = _ckpt_cloud_results(207, ["foggy/JPEGImages/a_000001.png", "foggy/JPEGImages/d_4.png"])
    online.delete_model()                                           # trainer.py:56: the cloud model itself is dropped, the results stay
    opt_s, opt_m = torch.optim.SGD(groups(student), lr=0.01, momentum=0.9, weight_decay=1e-4), torch.optim.SGD(groups(merge), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sc_s, sc_m = mk_sched(opt_s), mk_sched(opt_m)
    a_few_steps(student, opt_s, sc_s, 3, 208)
    a_few_steps(merge, opt_m, sc_m, 3, 209)
    ens = ts.EnsembleTSModel(teacher, online, student, merge, out)                                # trainer.py:84
    fake = type("FakeCoinTrainer", (), {})()
    fake.checkpointer = dc.DetectionTSCheckpointer(ens, out, optimizer=opt_s, optimizer_merge=opt_m, scheduler=sc_s, scheduler_merge=sc_m)
    fake.ap_50_student, fake.ap_50_offline_teacher, fake.model_CLOUD = {3: 41.5, 6: 43.25}, {3: 40.0, 6: 40.5}, online
    tr.CoinTrainer.save(fake, 6)
    os.remove(os.path.join(out, "last_checkpoint"))
    for f in sorted(os.listdir(out)):
        print(f"  wrote {CKPT_DIR}/{f} ({os.path.getsize(os.path.join(out, f)) / 1024:.0f} KiB)")



CASES = [case_mil_losses, case_bottleneck, case_resnet, case_box_predictor_pretrain, case_box_predictor_step,
         case_text_encoder, case_ckg, case_lr_and_fusion, case_optimizer_groups, case_rpn, case_roi_sampling,
         case_e2e_pretrain, case_e2e_step_and_inference, case_ema, case_match_dual_teacher, case_e2e_coin_step, case_e2e_coin_two_steps, case_voc_eval, case_voc_dataset, case_clip_relabel, case_real_width, case_rn101, case_checkpoint_formats]

if __name__ == "__main__":
    only = set(sys.argv[1:])
    for c in CASES:
        if only and c.__name__ not in only:
            continue
        print(c.__name__)
        c()
