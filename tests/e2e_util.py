"""Shared helpers for the end-to-end parity tests (CPU host-logic tests, GPU parity tests, smoke()).

Parity boundary P: the reference's anchor labelling and RoI sampling draw from torch.randperm,
whose stream differs between CPU and GPU.  The end-to-end cases therefore FEED the sampled anchors labels and
sampled RoIs that the reference run produced (stored in tests/golden/e2e_*.npz) to both the HIP product and the
CPU oracle, and compare everything downstream: loss values and gradients.
"""
import os

import numpy as np
import torch

from golden_util import K, LOSS_W, T, instances, load, load_weights, tiny_detector


def _inst(z, prefix, size):
    from coin_amd.structures import Boxes, Instances

    inst = Instances(tuple(size))
    for k in z.files:
        if k.startswith(prefix + "."):
            name = k[len(prefix) + 1:]
            if "." in name:
                continue
            v = T(z[k])
            inst.set(name, Boxes(v) if name.endswith("boxes") else v, check_len=False)
    return inst


def tiny_product_detector():
    from coin_amd.box_ops import Box2BoxTransform, Matcher
    from coin_amd.modeling.backbone import CLIP_IMAGE
    from coin_amd.modeling.fast_rcnn import FastRCNNOutputLayers
    from coin_amd.modeling.meta_arch import OpenVocabularyRCNN
    from coin_amd.modeling.roi_heads import OpenVocabularyRes5ROIHeads, ROIPooler
    from coin_amd.modeling.rpn import DefaultAnchorGenerator, DualTeacherRPN, StandardRPNHead
    from coin_amd.modeling.text_encoder import CLIP_TEXT, prompt_tokens
    from coin_amd.structures import ShapeSpec

    toks = torch.zeros(K + 1, 16, dtype=torch.int)
    for i in range(K + 1):
        seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
        toks[i, : len(seq)] = torch.tensor(seq)
    te = CLIP_TEXT("RN50", ["car", "person", "bus", "backgroud"], embed_dim=32, context_length=16, vocab_size=64, width=32, heads=2,
                   layers=2, tokenized_prompts=toks, n_templates=2)
    bp = FastRCNNOutputLayers(ShapeSpec(channels=256, height=1, width=1), text_encoder=te, pooling_type="meanpool",
                              box2box_transform=Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=32, classes_weight=[1.0] * K + [0.9],
                              loss_type="MILCrossEntropy", test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100,
                              cls_agnostic_bbox_reg=True, loss_weight=LOSS_W, batch_size_per_image=32, cls_b_thresh=0.3,
                              dataset=("foggytrain_0.02",), prototype_update_rate=0.9996)
    rh = OpenVocabularyRes5ROIHeads(in_features=["res4"], pooler=ROIPooler(14, (1.0 / 16,), 0, "ROIAlignV2"), box_predictor=bp,
                                    pooling_type="meanpool", num_classes=K, batch_size_per_image=32, positive_fraction=0.25,
                                    proposal_matcher=Matcher([0.5], [0, 1], False))
    ag = DefaultAnchorGenerator([[32, 64, 128]], [[0.5, 1.0, 2.0]], [16])
    pg = DualTeacherRPN(in_features=["res4"], head=StandardRPNHead(128, 9), anchor_generator=ag,
                        anchor_matcher=Matcher([0.3, 0.7], [0, -1, 1], True), box2box_transform=Box2BoxTransform((1.0, 1.0, 1.0, 1.0)),
                        batch_size_per_image=64, positive_fraction=0.5, pre_nms_topk=(200, 120), post_nms_topk=(60, 40))
    bb = CLIP_IMAGE("RN50", freeze_at=2, layers=(1, 1, 2, 2), width=8)
    return OpenVocabularyRCNN(backbone=bb, proposal_generator=pg, roi_heads=rh, pixel_mean=[0.48145466, 0.4578275, 0.40821073],
                              pixel_std=[0.26862954, 0.26130258, 0.27577711], device="cpu", compute_dtype=torch.float32)




def _to_dev(inst, device):
    return inst.to(device)


def golden_pretrain_case():
    z = load("e2e_pretrain")
    case = {"z": z, "ref_losses": {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")},
            "ref_grads": {k[3:]: T(z[k]) for k in z.files if k.startswith("g::")}}
    case["images"] = [T(z[f"img{i}"]) for i in range(2)]
    case["sizes"] = [(im.shape[1], im.shape[2]) for im in case["images"]]
    case["anchor_labels"] = T(z["anchor_labels"])
    case["anchor_matched_boxes"] = T(z["anchor_matched_boxes"])
    return case


def _patch_samplers(model, sampled, labels, matched):
    model.roi_heads.label_and_sample_proposals = lambda proposals, targets, branch: sampled
    model.proposal_generator.label_and_sample_anchors = lambda anchors, gt, branch: ([l for l in labels], [m for m in matched])


def _grads(model, names, keep_dtype=False):
    p = dict(model.named_parameters())
    cv = (lambda t: t.detach().cpu()) if keep_dtype else (lambda t: t.detach().float().cpu())
    return {n: (cv(p[n].grad) if p[n].grad is not None else cv(torch.zeros_like(p[n]))) for n in names}


def run_product_pretrain(case, device="cuda:0", dtype=torch.float32, update_prototype=True):
    z = case["z"]
    model = tiny_product_detector()
    load_weights(model, z)
    model.to(device)
    model.set_compute_dtype(dtype)
    model.train()
    sampled = [(_inst(z, f"s{i}.fg", s).to(device), _inst(z, f"s{i}.bg", s).to(device)) for i, s in enumerate(case["sizes"])]
    _patch_samplers(model, sampled, case["anchor_labels"].to(device), case["anchor_matched_boxes"].to(device))
    batch = []
    for i, (img, s) in enumerate(zip(case["images"], case["sizes"])):
        batch.append({"image": img.to(device), "height": s[0], "width": s[1], "RCNN": _inst(z, f"rcnn{i}", s), "RPN": _inst(z, f"rpn{i}", s)})
    losses = model(batch, branch="pre_train", update_prototype=update_prototype)
    sum(losses.values()).backward()
    return {k: v.detach().float().cpu() for k, v in losses.items()}, _grads(model, case["ref_grads"].keys())


def _to_dtype(inst, dtype):
    """Float fields of an oracle Instances in `dtype` (fp64 runs of the oracle: tests/test_parity_gpu.py's precision reference)."""
    from oracle import d2

    if inst is None or dtype == torch.float32:
        return inst
    for k, v in list(inst.get_fields().items()):
        if isinstance(v, d2.Boxes):
            v.tensor = v.tensor.to(dtype)
        elif torch.is_tensor(v) and v.is_floating_point():
            inst.set(k, v.to(dtype))
    return inst


def run_oracle_pretrain(case, update_prototype=True, dtype=torch.float32):
    z = case["z"]
    model = tiny_detector()
    load_weights(model, z)
    model.to(dtype)
    model.train()
    sampled = [(_to_dtype(instances(z, f"s{i}.fg", s), dtype), _to_dtype(instances(z, f"s{i}.bg", s), dtype)) for i, s in enumerate(case["sizes"])]
    _patch_samplers(model, sampled, case["anchor_labels"], case["anchor_matched_boxes"].to(dtype))
    batch = []
    for i, (img, s) in enumerate(zip(case["images"], case["sizes"])):
        batch.append({"image": img, "height": s[0], "width": s[1], "RCNN": _to_dtype(instances(z, f"rcnn{i}", s), dtype),
                      "RPN": _to_dtype(instances(z, f"rpn{i}", s), dtype)})
    losses = model(batch, branch="pre_train", update_prototype=update_prototype)
    sum(losses.values()).backward()
    return {k: v.detach() for k, v in losses.items()}, _grads(model, case["ref_grads"].keys(), keep_dtype=True)


def run_oracle_step_two(case, dtype=torch.float32):
    """The oracle's step_two forward + student backward at boundary P (the reference's sampled anchors / RoIs fed in)."""
    from oracle import coin as OC

    z = case["z"]
    model = tiny_detector()
    load_weights(model, z)
    model.to(dtype)
    model.train()
    merge = OC.CKGNet(32, 32, K + 1, head_num=4)
    load_weights(merge, z, "m::")
    merge.to(dtype)
    sampled = [tuple(_to_dtype(instances(z, f"s{i}.{t}", s), dtype) for t in ("a", "b", "bg")) for i, s in enumerate(case["sizes"])]
    model.roi_heads.label_and_sample_proposals = lambda proposals, targets, branch: sampled
    lab, mb = T(z["anchor_labels"]), T(z["anchor_matched_boxes"]).to(dtype)
    idx, dl = T(z["anchor_matched_idxs"]), T(z["anchor_dist_labels"])
    model.proposal_generator.label_and_sample_anchors = lambda anchors, gt, branch: (list(lab), list(mb), list(idx), list(dl))
    batch, rc, rp = [], [], []
    for i, (img, s) in enumerate(zip(case["images"], case["sizes"])):
        batch.append({"image": img, "height": s[0], "width": s[1]})
        rc.append(tuple(_to_dtype(instances(z, f"{t}{i}", s), dtype) for t in ("a", "b", "c")))
        rp.append((_to_dtype(instances(z, f"rpn_a{i}", s), dtype), None, _to_dtype(instances(z, f"rpn_c{i}", s), dtype)))
    losses = model(batch, merge, (rc, rp), branch="step_two", update_prototype=True)
    skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"]
    sum(v for k, v in losses.items() if k not in skip).backward()
    return {k: v.detach() for k, v in losses.items()}, _grads(model, case["ref_grads"].keys(), keep_dtype=True)


def golden_step_case():
    z = load("e2e_step_two")
    case = {"z": z, "ref_losses": {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")},
            "ref_grads": {k[3:]: T(z[k]) for k in z.files if k.startswith("g::")}}
    case["images"] = [T(z[f"img{i}"]) for i in range(2)]
    case["sizes"] = [(im.shape[1], im.shape[2]) for im in case["images"]]
    return case


def run_product_step_two(case, device="cuda:0", dtype=torch.float32):
    from coin_amd.modeling.text_encoder import CKGNet

    z = case["z"]
    model = tiny_product_detector()
    load_weights(model, z)
    model.to(device)
    model.set_compute_dtype(dtype)
    model.train()
    merge = CKGNet(32, 32, K + 1, head_num=4)
    load_weights(merge, z, "m::")
    merge.to(device)
    sampled = [(_inst(z, f"s{i}.a", s).to(device), _inst(z, f"s{i}.b", s).to(device), _inst(z, f"s{i}.bg", s).to(device)) for i, s in enumerate(case["sizes"])]
    model.roi_heads.label_and_sample_proposals = lambda proposals, targets, branch: sampled
    lab, mb = T(z["anchor_labels"]).to(device), T(z["anchor_matched_boxes"]).to(device)
    idx, dl = T(z["anchor_matched_idxs"]).to(device), T(z["anchor_dist_labels"]).to(device)
    model.proposal_generator.label_and_sample_anchors = lambda anchors, gt, branch: (list(lab), list(mb), list(idx), list(dl))
    batch, rc, rp = [], [], []
    for i, (img, s) in enumerate(zip(case["images"], case["sizes"])):
        batch.append({"image": img.to(device), "height": s[0], "width": s[1]})
        rc.append(tuple(_inst(z, f"{t}{i}", s).to(device) for t in ("a", "b", "c")))
        rp.append((_inst(z, f"rpn_a{i}", s).to(device), None, _inst(z, f"rpn_c{i}", s).to(device)))
    losses = model(batch, merge, (rc, rp), branch="step_two", update_prototype=True)
    skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"]
    sum(v for k, v in losses.items() if k not in skip).backward()
    return {k: v.detach().float().cpu() for k, v in losses.items()}, _grads(model, case["ref_grads"].keys())


# ------------------------------------------------------------------------------------------ box predictor: step branches + CKG update
def _product_box_predictor(in_ch=64):
    from coin_amd.box_ops import Box2BoxTransform
    from coin_amd.modeling.fast_rcnn import FastRCNNOutputLayers
    from coin_amd.modeling.text_encoder import CLIP_TEXT
    from coin_amd.structures import ShapeSpec

    toks = torch.zeros(K + 1, 16, dtype=torch.int)
    for i in range(K + 1):
        seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
        toks[i, : len(seq)] = torch.tensor(seq)
    te = CLIP_TEXT("RN50", ["car", "person", "bus", "backgroud"], embed_dim=32, context_length=16, vocab_size=64, width=32, heads=2,
                   layers=2, tokenized_prompts=toks, n_templates=2)
    return FastRCNNOutputLayers(ShapeSpec(channels=in_ch, height=1, width=1), text_encoder=te, pooling_type="meanpool",
                                box2box_transform=Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=32, classes_weight=[1.0] * K + [0.9],
                                loss_type="MILCrossEntropy", test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100,
                                cls_agnostic_bbox_reg=True, loss_weight=LOSS_W, batch_size_per_image=32, cls_b_thresh=0.3,
                                dataset=("foggytrain_0.02",), prototype_update_rate=0.9996)


def run_product_box_predictor_step(tag, device="cpu"):
    """The reference's FastRCNNOutputLayers.losses(step_one / step_two) + trainer.py:192-197 (CKG update through
    gradient_discrepancy_loss) on the product's predictor; returns what the golden recorded."""
    from coin_amd.modeling.text_encoder import CKGNet

    z = load(f"box_predictor_{tag}")
    branch = str(z["branch"])
    bp = _product_box_predictor()
    load_weights(bp, z)
    merge = CKGNet(32, 32, K + 1, head_num=4)
    load_weights(merge, z, "m::")
    bp.to(device).train()
    merge.to(device)
    n_img = int(z["n_img"])
    size = (96, 128)
    props = [tuple(_inst(z, f"p{i}.{t}", size).to(device) for t in ("a", "b", "bg")) for i in range(n_img)]
    cs = [_inst(z, f"p{i}.c", size).to(device) for i in range(n_img)]
    x = T(z["x"]).to(device).requires_grad_(True)
    xc = T(z["xc"]).to(device)
    preds = bp(x, branch)
    upd = bool(z["update_prototype"])
    if xc.shape[0]:
        losses = bp.losses((preds, bp(xc, branch, return_feats=False)), (props, cs), merge, branch, update_prototype=upd)
    else:
        losses = bp.losses((preds, ((None, None), None)), (props, None), merge, branch, update_prototype=upd)
    out = {"losses": {k: float(v) for k, v in losses.items()}, "ref": {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}, "z": z}
    if "loss_merge_a" in losses:
        lg = bp.merge_grad_loss()
        out["losses"]["loss_merge_grad"] = float(lg)
        (lg + losses["loss_merge_base"]).backward(inputs=list(merge.parameters()), retain_graph=True)
        out["merge_grads"] = {n: p.grad.detach().cpu().clone() for n, p in merge.named_parameters()}
        assert all(p.grad is None for p in bp.parameters()), "the CKG update must not touch the detector's gradients"
    skip = ["loss_merge_grad", "loss_merge_a", "loss_merge_b", "loss_merge_base"] + ([] if branch == "step_two" else ["loss_cls_b"])
    sum(v for k, v in losses.items() if k not in skip).backward()
    out["gx"] = x.grad.detach().cpu()
    params = dict(bp.named_parameters())
    out["grads"] = {k[3:]: params[k[3:]].grad.detach().cpu() for k in z.files if k.startswith("g::")}
    return out


