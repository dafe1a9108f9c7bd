"""Parity cases at the benchmark's real layer widths (RN50 res5 on 14x14 RoI tiles, the box predictor at D = 1024 and 9 classes).

The fixtures (tests/golden/real_width_*.npz, written by gen_golden.py:case_real_width from the REFERENCE's modules) hold the
outputs; weights and inputs are re-created from seeds (tests/seeded.py) and fingerprinted against the fixture's checksums."""
import numpy as np
import torch

import seeded
from golden_util import T, close, load

CLASSES9 = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle", "backgroud"]
LOSS_W = {"loss_box_reg": 1.0, "loss_box_reg_offline": 1.0, "loss_box_reg_online": 1.0, "loss_cls": 1.0, "loss_text_align": 10.0,
          "loss_distillation": 0.1, "loss_cls_b": 0.1}
RES5_SUB = {"0.conv1.weight": (4, 8, 1, 1), "0.conv2.weight": (8, 8, 1, 1), "0.downsample.0.weight": (16, 8, 1, 1), "1.conv3.weight": (16, 4, 1, 1),
            "2.conv1.weight": (4, 16, 1, 1)}
HEAD_SUB = {"trans.0.weight": (8, 16), "trans.2.weight": (8, 8), "trans.4.weight": (16, 8), "cls_score.weight": (8, 16)}


def sub(t, steps):
    return t[tuple(slice(None, None, s) for s in steps)]


def _d(t):
    return torch.as_tensor(np.asarray(t)).double() if not torch.is_tensor(t) else t.detach().double().cpu()


def rel_err(a, b):
    """max |a - b| relative to max |b| (row P of SURVEY §8a: gradients are compared relative to the tensor's scale)."""
    a, b = _d(a), _d(b)
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)) if b.numel() else 0.0


def res5_inputs():
    z = load("real_width_res5")
    x = seeded.randn((64, 1024, 14, 14), 502)
    assert abs(seeded.checksum(x) - float(z["x_checksum"])) <= 1e-9 * abs(float(z["x_checksum"])), "seeded input drifted from the generator's"
    return z, x, seeded.randn((64, 2048), 503)


def run_res5(net, x, gy, device="cpu", dtype=torch.float32, mean_pool=None):
    """net: Sequential of 3 bottlenecks (oracle or product) already filled; -> (y [64,2048], gx, {param grads}, state dict)."""
    net.to(device=device)
    if dtype == torch.float64:
        net.double()
    net.train()
    xx = x.detach().clone().to(device=device, dtype=dtype if dtype == torch.float64 else torch.float32)
    if str(device) != "cpu":
        xx = xx.contiguous(memory_format=torch.channels_last)
    xx.requires_grad_(True)
    if mean_pool is not None:
        y = mean_pool(net, xx)
    else:
        y = net(xx).mean(dim=[2, 3])
    (y * gy.to(device=device, dtype=y.dtype)).sum().backward()
    return y.detach(), xx.grad.detach(), {n: p.grad.detach() for n, p in net.named_parameters()}, net.state_dict()


def l2_err(a, b):
    """||a - b|| / ||b|| (insensitive to the handful of elements that sit behind a ReLU whose pre-activation is within fp32 rounding
    of zero: one such flip moves ~1e4 input-gradient elements of this net by ~1e-2 of the tensor's maximum in ANY fp32 run)."""
    a, b = _d(a), _d(b)
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    return float((a - b).norm() / b.norm().clamp(min=1e-30)) if b.numel() else 0.0


def check_res5(z, y, gx, grads, sd, tol_y, tol_g, exact=None, what=""):
    """Against the reference's outputs.  Without `exact`: every tensor within `tol` of the reference (oracle on the CPU: same ops).
    With `exact` = (y, gx, grads) of an fp64 run: the forward stays at `tol_y` against the reference; a gradient is measured against
    fp64 (relative L2) and held to max(tol_g, 2 x the largest error the REFERENCE's own fp32 gradients show against fp64 on any of
    these tensors) -- what fp32 can hold on this net, measured, not assumed.  -> table rows for the log."""
    ey, egx, eg = exact if exact is not None else (None, None, None)
    items = [("gx", sub(gx, (8, 32, 1, 1)), z["gx_sub"], None if egx is None else sub(egx, (8, 32, 1, 1)))]
    for n, steps in RES5_SUB.items():
        items.append((n, sub(grads[n], steps), z[f"g::{n}_sub"], None if eg is None else sub(eg[n], steps)))
    for k in z.files:
        if k.startswith("g::") and not k.endswith("_sub"):
            items.append((k[3:], grads[k[3:]], z[k], None if eg is None else eg[k[3:]]))
    e = rel_err(y, z["y"])
    assert e <= tol_y, f"{what}y: rel err {e:.2e} > {tol_y:.0e} vs the reference"
    rows = [("y", e, None if ey is None else rel_err(y, ey), None if ey is None else rel_err(z["y"], ey), None, None)]
    if exact is None:
        for name, got, ref, _ in items:
            e = rel_err(got, ref)
            rows.append((name, e, None, None, None, None))
            assert e <= tol_g, f"{what}{name}: rel err {e:.2e} > {tol_g:.0e} vs the reference"
    else:
        for name, got, ref, ex in items:
            rows.append((name, rel_err(got, ref), l2_err(got, ex), l2_err(ref, ex), rel_err(got, ex), rel_err(ref, ex)))
        floor = max(r[3] for r in rows[1:])
        bound = max(tol_g, 2.0 * floor)
        bad = [(r[0], r[2]) for r in rows[1:] if r[2] > bound]
        assert not bad, f"{what}gradients further from fp64 than {bound:.2e} (reference fp32 floor {floor:.2e}): {bad}"
    for k in z.files:
        if k.startswith("after::"):
            close(sd[k[7:]].float().cpu(), z[k], 1e-5, what + k)
    return rows


# ------------------------------------------------------------------------------------------ predictor at D = 1024
def head_tokens(ctx=16):
    toks = torch.zeros(9, ctx, dtype=torch.int)
    for i in range(9):
        seq = [62, 1, 2, 3, 1, 6, 6, 6, 6, 10 + i, 5, 63]
        toks[i, : len(seq)] = torch.tensor(seq)
    return toks


def head_inputs():
    z = load("real_width_box_predictor")
    x = seeded.randn((512, 2048), 514).abs()
    assert abs(seeded.checksum(x) - float(z["x_checksum"])) <= 1e-9 * abs(float(z["x_checksum"]))
    return z, x


def fill_head(bp, z):
    """The generator's recipe (gen_golden.py:case_real_width): the three large blocks are seeded, the small text encoder (weights,
    prompt vectors, fixed class embeddings, prototypes) comes from the fixture."""
    seeded.fill_module(bp.trans, 512), seeded.fill_module(bp.cls_score, 515), seeded.fill_module(bp.bbox_pred, 516)
    with torch.no_grad():
        bp.cls_score.weight.mul_(0.2)
        bp.bbox_pred.weight.mul_(0.05)
    sd = {k[len("w::text_encoder."):]: T(z[k]) for k in z.files if k.startswith("w::text_encoder.")}
    own = bp.text_encoder.state_dict()
    assert set(sd) == set(own), (sorted(set(sd) ^ set(own)))
    bp.text_encoder.load_state_dict(sd)
    w = dict(bp.named_parameters())["trans.2.weight"]
    assert abs(seeded.checksum(w) - float(z["w_checksum"])) <= 1e-9 * abs(float(z["w_checksum"])), "seeded weights drifted from the generator's"
    return bp


class ForcedLeaky(torch.nn.Module):
    """LeakyReLU whose branch decisions are given (a boolean mask of the pre-activation's shape) instead of taken from the sign: the
    fp64 run of the oracle then follows the fp32 product through the same linear pieces.  Of the ~10^6 pre-activations of `trans` a
    few lie within fp32 rounding of zero; there ANY two fp32 evaluations may decide differently, and one flipped element moves the
    gradients of everything upstream by ~1e-4 of their maximum (measured: tools/debug_head.py (round 5; in the git history))."""

    def __init__(self, mask: torch.Tensor, slope: float = 0.01):
        super().__init__()
        self.mask, self.slope = mask, slope

    def forward(self, p):
        return torch.where(self.mask.to(p.device), p, self.slope * p)


def follow_leaky_decisions(oracle_bp, x, product_pre, atol=2e-5, max_flips=8):
    """Make the (fp64) oracle predictor take the product's LeakyReLU decisions in `trans`.  product_pre = the product's fp32
    pre-activations of trans.0 and trans.2.  A decision may differ from the fp64 one only where the fp64 pre-activation is within
    `atol` of zero, and only in a handful of places (asserted).  -> number of differing decisions."""
    t = oracle_bp.trans
    flips = 0
    with torch.no_grad():
        h = x.double()
        for li, pre in zip((0, 2), product_pre):
            p64 = torch.nn.functional.linear(h, t[li].weight.double(), t[li].bias.double())
            mask = _d(pre) > 0
            diff = mask != (p64 > 0)
            if diff.any():
                assert float(p64[diff].abs().max()) <= atol, f"trans.{li}: a LeakyReLU decision differs at |pre-activation| {float(p64[diff].abs().max()):.2e}"
            flips += int(diff.sum())
            t[li + 1] = ForcedLeaky(mask)
            h = torch.where(mask, p64, 0.01 * p64)
    assert flips <= max_flips, f"{flips} LeakyReLU decisions differ from fp64"
    return flips


def check_head(z, scores, deltas, losses, gx, grads, proto, tol, tol_g, what="", exact=None):
    """`exact` = (gx, grads) of an fp64 run of the oracle: gradients are then measured against fp64 and held to
    max(tol_g, 2 x the largest error of the reference's own fp32 gradients against fp64 over these tensors)."""
    close(scores, z["scores"], tol, what + "scores")
    close(deltas, z["deltas"], tol, what + "deltas")
    ref = {k[6:]: float(z[k]) for k in z.files if k.startswith("loss::")}
    assert set(losses) == set(ref)
    for k, v in ref.items():
        assert abs(float(losses[k]) - v) < tol * max(1.0, abs(v)), (what, k, float(losses[k]), v)
    items = [("gx", sub(gx, (4, 8)), z["gx_sub"], None if exact is None else sub(exact[0], (4, 8)))]
    for k in z.files:
        if k.startswith("g::"):
            name = k[3:]
            pick = (lambda t, n=name[:-4]: sub(t, HEAD_SUB[n])) if name.endswith("_sub") else (lambda t: t)
            base = name[:-4] if name.endswith("_sub") else name
            items.append((base, pick(grads[base]), z[k], None if exact is None else pick(exact[1][base])))
    if exact is None:
        for name, got, r, _ in items:
            e = rel_err(got, r)
            assert e <= tol_g, f"{what}{name}: rel err {e:.2e} > {tol_g:.0e}"
    else:
        rows = [(name, rel_err(got, ex), rel_err(r, ex), rel_err(got, r)) for name, got, r, ex in items]
        print("\n".join(f"{what}{n:44s} product vs fp64 {a:.2e}   reference vs fp64 {b:.2e}   product vs reference {c:.2e}" for n, a, b, c in rows))
        bound = max(tol_g, 2.0 * max(b for _, _, b, _ in rows))
        bad = [(n, a) for n, a, _, _ in rows if a > bound]
        assert not bad, f"{what}gradients further from fp64 than {bound:.2e}: {bad}"
    close(proto, z["prototype_after"], 1e-6, what + "prototype")


def oracle_head():
    from oracle import coin as OC

    enc = OC.TextEncoder(1024, 16, 64, 32, 2, 2, head_tokens(), 4, 4)
    te = OC.ClipText(enc, CLASSES9, torch.zeros(9, 1024))
    return OC.BoxPredictor(2048, te, 1024, [1.0] * 8 + [0.9], LOSS_W, 256, cls_b_thresh=0.3, dataset=("foggytrain_0.02",))


def product_head():
    from coin_amd.box_ops import Box2BoxTransform
    from coin_amd.modeling.fast_rcnn import FastRCNNOutputLayers
    from coin_amd.modeling.text_encoder import CLIP_TEXT
    from coin_amd.structures import ShapeSpec

    te = CLIP_TEXT("RN50", CLASSES9, embed_dim=1024, context_length=16, vocab_size=64, width=32, heads=2, layers=2,
                   tokenized_prompts=head_tokens(), n_templates=2)
    return FastRCNNOutputLayers(ShapeSpec(channels=2048, height=1, width=1), text_encoder=te, pooling_type="meanpool",
                                box2box_transform=Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), text_dim=1024, classes_weight=[1.0] * 8 + [0.9],
                                loss_type="MILCrossEntropy", test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100,
                                cls_agnostic_bbox_reg=True, loss_weight=LOSS_W, batch_size_per_image=256, cls_b_thresh=0.3,
                                dataset=("foggytrain_0.02",), prototype_update_rate=0.9996)


def run_head(bp, z, x, make_inst, device="cpu", dtype=torch.float32):
    """pre_train forward + losses + backward of a filled predictor -> what check_head needs."""
    bp.to(device).train()
    props = [(make_inst(z, f"p{i}.fg", (800, 1333)), make_inst(z, f"p{i}.bg", (800, 1333))) for i in range(int(z["n_img"]))]
    if str(device) != "cpu":
        props = [(a.to(device), b.to(device)) for a, b in props]
    if dtype == torch.float64:
        from e2e_util import _to_dtype

        bp.double()
        props = [(_to_dtype(a, dtype), _to_dtype(b, dtype)) for a, b in props]
    xx = x.detach().clone().to(device=device, dtype=dtype).requires_grad_(True)
    preds = bp(xx, "pre_train")
    (scores, lta), deltas, feats = preds
    losses = bp.losses(preds, props, None, "pre_train", update_prototype=True)
    sum(losses.values()).backward()
    grads = {n: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for n, p in bp.named_parameters()}
    return (scores.detach().cpu(), deltas.detach().cpu(), {k: float(v) for k, v in losses.items()}, xx.grad.detach().cpu(), grads,
            bp.text_encoder.per_class_feat.detach().cpu())


# ------------------------------------------------------------------------------------------ RN101 trunk + CKG at MERGE_DIM 512 (BASELINE configs[3])
RN101_SUB = {"layer3.22.conv3.weight": (8, 4, 1, 1), "layer3.11.conv2.weight": (4, 4, 1, 1), "layer3.0.conv1.weight": (2, 8, 1, 1)}


def rn101_inputs():
    z = load("rn101_res4")
    x = seeded.randn((2, 3, 96, 128), 602)
    assert abs(seeded.checksum(x) - float(z["x_checksum"])) <= 1e-9 * abs(float(z["x_checksum"]))
    return z, x, seeded.randn(tuple(z["res4"].shape), 603)


def run_rn101(net, x, gy, device="cpu", dtype=torch.float32):
    """net: a ModifiedResNet((3,4,23,3)) built with freeze_at=0 -> seeded fill -> freeze(2) here (the generator's order)."""
    seeded.fill_module(net, 601)
    net.freeze(2)
    net.to(device)
    if dtype == torch.float64:
        net.double()
    net.train()
    xx = x.detach().clone().to(device=device, dtype=dtype)
    if str(device) != "cpu":
        xx = xx.contiguous(memory_format=torch.channels_last)
    y = net(xx)["res4"]
    (y * gy.to(device=device, dtype=y.dtype)).sum().backward()
    p = dict(net.named_parameters())
    return y.detach(), {n: q.grad.detach() for n, q in p.items() if q.grad is not None}, net.state_dict(), [n for n, q in p.items() if not q.requires_grad]


def check_rn101(z, y, grads, sd, frozen, tol_y, tol_g, exact=None, what=""):
    """As check_res5: forward against the reference; gradients against the reference (oracle on the CPU) or, with `exact`
    (an fp64 run), within max(tol_g, 2 x the reference's own worst fp32 error) of fp64 (relative L2)."""
    assert sorted(frozen) == sorted(str(n) for n in z["frozen_names"]), "frozen parameter set differs from the reference's"
    e = rel_err(y, z["res4"])
    if exact is None:
        assert e <= tol_y, f"{what}res4: rel err {e:.2e} > {tol_y:.0e}"
    else:  # 34 bottlenecks of train-mode BatchNorm deep: the forward, too, is held to what the reference's fp32 run holds against fp64
        e_ref, e_got = rel_err(z["res4"], exact[0]), rel_err(y, exact[0])
        assert e_got <= max(tol_y, 2.0 * e_ref), f"{what}res4: {e_got:.2e} from fp64 (reference fp32: {e_ref:.2e})"
    items = []
    for k in z.files:
        if k.startswith("g::"):
            name = k[3:]
            if name.endswith("_sub"):
                items.append((name[:-4], lambda t, n=name[:-4]: sub(t, RN101_SUB[n]), z[k]))
            else:
                items.append((name, lambda t: t, z[k]))
    rows = [("res4", e, None if exact is None else rel_err(y, exact[0]), None if exact is None else rel_err(z["res4"], exact[0]))]
    if exact is None:
        for name, pick, ref in items:
            e = rel_err(pick(grads[name]), ref)
            rows.append((name, e, None, None))
            assert e <= tol_g, f"{what}{name}: rel err {e:.2e} > {tol_g:.0e} vs the reference"
    else:
        _, eg = exact
        for name, pick, ref in items:
            rows.append((name, rel_err(pick(grads[name]), ref), l2_err(pick(grads[name]), pick(eg[name])), l2_err(ref, pick(eg[name]))))
        floor = max(r[3] for r in rows[1:])
        bound = max(tol_g, 2.0 * floor)
        bad = [(r[0], r[2]) for r in rows[1:] if r[2] > bound]
        assert not bad, f"{what}gradients further from fp64 than {bound:.2e} (reference fp32 floor {floor:.2e}): {bad}"
    for k in z.files:
        if k.startswith("after::"):
            close(sd[k[7:]].float().cpu(), z[k], 1e-5, what + k)
    return rows


def check_rn101_ckg(merge, device="cpu", tol=1e-5):
    """CKGNet at hidden 512 / 8 classes / 8 heads against rn101_ckg.npz (forward + every parameter gradient, sub-sampled)."""
    z = load("rn101_ckg")
    seeded.fill_module(merge, 611)
    names = sorted(dict(merge.named_parameters()))
    assert abs(seeded.checksum(dict(merge.named_parameters())[names[0]]) - float(z["w_checksum"])) <= 1e-9 * abs(float(z["w_checksum"]))
    merge.to(device)
    t = lambda k: T(z[k]).to(device)
    y = merge(t("x"), t("proto_off"), t("proto_on"), t("probs_off"), t("probs_on"))
    close(y, z["y"], tol, "ckg y")
    (y * t("gy")).sum().backward()
    for n, q in merge.named_parameters():
        e = rel_err(sub(q.grad, (4, 4) if q.dim() == 2 else (1,)), z[f"mg::{n}_sub"])
        assert e <= max(tol, 1e-5) * 10, (n, e)
