"""The sync-free (random-key, fixed-shape) pre_train path computes the SAME quantities as the reference-shaped path:
`losses_packed` against the golden losses for the reference's sampled RoIs, `sample_masks` / `sample_packed` against the
counting rules of detectron2's subsample_labels, and the sync-free anchor labelling against the matcher."""
import numpy as np
import pytest
import torch

from cpu_shim import cpu_kernels
from e2e_util import _inst, tiny_product_detector
from golden_util import K, T, close, load, load_weights


def test_sample_masks_counts_and_membership():
    from coin_amd.box_ops import sample_masks

    g = torch.Generator().manual_seed(0)
    cls = torch.randint(-1, 5, (6, 300), generator=g)            # bg label = 4
    cls[0] = 4                                                    # no positives
    cls[1, :290] = -1                                             # almost everything ignored
    cls[2] = torch.where(torch.rand(300, generator=g) < 0.9, torch.tensor(1), torch.tensor(4))  # more positives than the cap
    torch.manual_seed(1)
    with cpu_kernels():  # coin_sample_labels by its definition (tests/cpu_shim.py); the kernel itself: tests/test_kernels_gpu.py
        _check_sample_masks(sample_masks, cls)


def _check_sample_masks(sample_masks, cls):
    pos, neg = sample_masks(cls, 64, 0.25, 4)
    is_pos, is_neg = (cls != -1) & (cls != 4), cls == 4
    assert not (pos & ~is_pos).any() and not (neg & ~is_neg).any()
    for i in range(6):
        n_pos = min(int(is_pos[i].sum()), 16)
        n_neg = min(int(is_neg[i].sum()), 64 - n_pos)
        assert int(pos[i].sum()) == n_pos and int(neg[i].sum()) == n_neg
    # uniformity: every candidate of a row is picked with the same frequency
    hits = torch.zeros(300)
    for s in range(400):
        torch.manual_seed(100 + s)
        p, _ = sample_masks(cls[2:3], 64, 0.25, 4)
        hits += p[0].float()
    cand = is_pos[2]
    freq = hits[cand] / 400
    assert abs(float(freq.mean()) - 16 / int(cand.sum())) < 1e-6 and float(freq.std()) < 0.03


def test_losses_packed_equal_reference_losses_on_the_same_samples():
    from parity_cases import losses_packed_pretrain

    losses_packed_pretrain("cpu")   # GPU twin: tests/test_parity_gpu.py


def test_sync_free_detector_step_matches_counting_rules():
    """Whole sync-free forward on CPU (shimmed kernels): 32 rows per image, <= 8 fg, labels consistent with IoU >= 0.5."""
    from coin_amd.structures import pairwise_iou, Boxes

    z = load("e2e_pretrain")
    with cpu_kernels():
        model = tiny_product_detector()
        load_weights(model, z)
        model.set_sync_free(True)
        model.train()
        batch = []
        for i in range(2):
            img = T(z[f"img{i}"])
            size = (img.shape[1], img.shape[2])
            batch.append({"image": img, "height": size[0], "width": size[1], "RCNN": _inst(z, f"rcnn{i}", size), "RPN": _inst(z, f"rpn{i}", size)})
        rec = {}
        orig = model.roi_heads.sample_packed
        model.roi_heads.sample_packed = lambda p, t: rec.setdefault("ps", orig(p, t))
        torch.manual_seed(5)
        losses = model(batch, branch="pre_train", update_prototype=True)
        assert set(losses) == {"loss_text_align", "loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
        assert all(torch.isfinite(v) for v in losses.values())
        sum(losses.values()).backward()
        ps = rec["ps"]
        assert ps.per_image == 32 and ps.gt_classes.shape[0] == 64
        for i in range(2):
            c = ps.gt_classes[i * 32:(i + 1) * 32]
            b = ps.boxes[i * 32:(i + 1) * 32]
            fg = (c >= 0) & (c < K)
            assert int(fg.sum()) <= 8 and int((c == K).sum()) == 32 - int(fg.sum()) - int((c < 0).sum())
            iou = pairwise_iou(batch[i]["RCNN"].gt_boxes, Boxes(b)).max(dim=0).values
            assert bool((iou[fg] >= 0.5).all()) and bool((iou[c == K] < 0.5).all())
            # fg rows carry the matched teacher box / class
            m = pairwise_iou(batch[i]["RCNN"].gt_boxes, Boxes(b)).argmax(dim=0)
            assert torch.equal(c[fg], batch[i]["RCNN"].gt_classes_offline[m[fg]])
            close(ps.gt_boxes[i * 32:(i + 1) * 32][fg], batch[i]["RCNN"].gt_boxes.tensor[m[fg]], 0)


@pytest.mark.parametrize("tag", ["one", "two", "two_nobg_noC", "two_noB", "one_noproto"])
def test_losses_packed_step_equal_reference_losses_on_the_same_samples(tag):
    from parity_cases import losses_packed_step

    losses_packed_step("cpu", tag)   # GPU twin: tests/test_parity_gpu.py


def _step_two_batch():
    z = load("e2e_step_two")
    sizes = [(T(z[f"img{i}"]).shape[1], T(z[f"img{i}"]).shape[2]) for i in range(2)]
    batch, rc, rp = [], [], []
    for i, s in enumerate(sizes):
        batch.append({"image": T(z[f"img{i}"]), "height": s[0], "width": s[1]})
        rc.append(tuple(_inst(z, f"{t}{i}", s) for t in ("a", "b", "c")))
        rp.append((_inst(z, f"rpn_a{i}", s), None, _inst(z, f"rpn_c{i}", s)))
    return z, batch, rc, rp


def test_sync_free_step_anchor_labelling_matches_reference_shaped_rules():
    """The deterministic outputs (matched A boxes, matched C index, distillation labels) are identical to the reference-shaped
    labelling; the sampled labels obey its counting rules and only ever pick anchors it could pick."""
    z, batch, rc, rp = _step_two_batch()
    with cpu_kernels():
        model = tiny_product_detector()
        load_weights(model, z)
        pg = model.proposal_generator
        anchors = pg.anchor_generator.for_hw((6, 8), "cpu")
        ia, ic = [g[0] for g in rp], [g[2] for g in rp]
        # add an image without any target and one with private boxes only
        from coin_amd.structures import Boxes, Instances

        empty_a = ia[0][0:0]
        ia2, ic2 = ia + [empty_a, empty_a], ic + [ic[0][0:0], ic[0]]
        torch.manual_seed(3)
        lab_s, mb_s, midx_s, dlab_s = pg.label_and_sample_anchors_step_sync_free(anchors, [ia2, ic2])
        torch.manual_seed(3)
        lab_r, mb_r, midx_r, dlab_r = pg.label_and_sample_anchors(anchors, [ia2, ic2], "step_two")
        for i in range(len(ia2)):
            close(mb_s[i], mb_r[i], 0, f"matched boxes {i}")
            assert torch.equal(midx_s[i], midx_r[i]) and torch.equal(dlab_s[i].long(), dlab_r[i].long())
            ls, lr = lab_s[i].long(), lab_r[i].long()
            # candidate sets before sampling: recompute them the reference way
            idx, lab = pg.anchor_matcher(__import__("coin_amd.structures", fromlist=["pairwise_iou"]).pairwise_iou(Boxes.cat([ia2[i].gt_boxes, ic2[i].gt_boxes]), anchors[0])) \
                if len(ia2[i]) + len(ic2[i]) else (None, torch.zeros(len(anchors[0]), dtype=torch.int8))
            if idx is None:
                assert bool((ls == -1).all()) and bool((lr == -1).all())
                continue
            in_c = (idx >= len(ia2[i])) & (idx < len(ia2[i]) + len(ic2[i]))
            cand_pos = (lab == 1) & ~in_c
            cand_neg = lab == 0
            if len(ia2[i]) == 0:
                cand_pos = cand_pos & False
                cand_neg = cand_neg & in_c
            assert not bool(((ls == 1) & ~cand_pos).any()) and not bool(((ls == 0) & ~cand_neg).any())
            n_pos = min(int(((lab == 1) & ~in_c).sum()), int(pg.batch_size_per_image * pg.positive_fraction))
            if len(ia2[i]) > 0:
                assert int((ls == 1).sum()) == n_pos == int((lr == 1).sum())
                assert int((ls == 0).sum()) == int((lr == 0).sum())


def test_sync_free_step_detector_forward_counting_rules_and_finite_losses():
    """Whole sync-free step_two forward + CKG update + student backward on CPU (shimmed kernels): 32 rows per image with roles
    consistent with the matcher (A / B rows overlap their target with IoU >= 0.5, nothing sampled on a private box), losses finite."""
    from coin_amd.modeling.text_encoder import CKGNet
    from coin_amd.structures import Boxes, pairwise_iou

    z, batch, rc, rp = _step_two_batch()
    with cpu_kernels():
        model = tiny_product_detector()
        load_weights(model, z)
        merge = CKGNet(32, 32, K + 1, head_num=4)
        load_weights(merge, z, "m::")
        model.set_sync_free_step(True)
        model.train()
        rec = {}
        orig = model.roi_heads.sample_packed_step
        model.roi_heads.sample_packed_step = lambda p, a, b, c: rec.setdefault("ps", orig(p, a, b, c))
        torch.manual_seed(5)
        losses = model(batch, merge, (rc, rp), branch="step_two", update_prototype=True)
        assert {"loss_text_align", "loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_rpn_distillation", "loss_distillation",
                "loss_merge_a", "loss_merge_b", "loss_merge_base", "loss_cls_b"} <= set(losses)
        assert all(torch.isfinite(v) for v in losses.values()), losses
        lg = model.roi_heads.box_predictor.merge_grad_loss()
        (lg + losses["loss_merge_base"]).backward(inputs=list(merge.parameters()), retain_graph=True)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in merge.parameters())
        skip = ("loss_merge_a", "loss_merge_b", "loss_merge_base")
        sum(v for k, v in losses.items() if k not in skip).backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        ps = rec["ps"]
        assert ps.per_image == 32 and ps.role.shape[0] == 64 and ps.has_b
        for i in range(2):
            sl = slice(i * 32, (i + 1) * 32)
            role, boxes = ps.role[sl], Boxes(ps.boxes[sl])
            a, b, c = rc[i]
            n_fg = int(((role == 0) | (role == 1)).sum())
            assert n_fg <= 8 and int((role == 2).sum()) == 32 - n_fg - int((role == -1).sum())
            if int((role == 0).sum()):
                assert bool((pairwise_iou(a.gt_boxes, boxes).max(0).values[role == 0] >= 0.5).all())
                close(ps.gt_boxes[sl][role == 0], a.gt_boxes.tensor[pairwise_iou(a.gt_boxes, boxes).argmax(0)[role == 0]], 0)
            if int((role == 1).sum()):
                j = pairwise_iou(b.gt_boxes, boxes).argmax(0)[role == 1]
                assert torch.equal(ps.gt_classes_online[sl][role == 1], b.gt_classes_online[j])
                assert torch.equal(ps.gt_classes_offline[sl][role == 1], b.gt_classes_offline[j])
            allt = Boxes.cat([a.gt_boxes, b.gt_boxes, c.gt_boxes])
            assert bool((pairwise_iou(allt, boxes).max(0).values[role == 2] < 0.5).all())  # background overlaps no target at all


def test_sample_packed_batched_over_images_equals_the_per_image_rules_with_ragged_and_empty_targets():
    """`sample_packed` (batched over the images: padded teacher blocks, one matcher call with a candidate set per image) on targets of
    different sizes, one of them EMPTY: every chosen row carries what the per-image composition (Matcher(pairwise_iou) -> class /
    box / probabilities of the matched teacher box) gives for that box; padding rows are never chosen."""
    from coin_amd.box_ops import PackedProposals
    from coin_amd.structures import Boxes, Instances, pairwise_iou

    g = torch.Generator().manual_seed(3)
    with cpu_kernels():
        heads = tiny_product_detector().roi_heads
        k, r = heads.num_classes, heads.batch_size_per_image
        n, p = 3, 40
        xy = torch.rand(n, p, 2, generator=g) * 80
        wh = torch.rand(n, p, 2, generator=g) * 40 + 4
        boxes = torch.cat([xy, xy + wh], dim=-1)
        valid = torch.rand(n, p, generator=g) < 0.9
        props = PackedProposals(boxes, torch.zeros(n, p), valid, [(128, 128)] * n)
        targets = []
        for i, cnt in enumerate((5, 0, 2)):
            t = Instances((128, 128))
            # teacher boxes = jittered copies of some proposals, so that foreground matches exist
            t.gt_boxes = Boxes(boxes[i, :cnt] + torch.rand(cnt, 4, generator=g))
            t.gt_classes_offline = torch.randint(0, k, (cnt,), generator=g)
            t.gt_probs_offline = torch.rand(cnt, k + 1, generator=g)
            targets.append(t)
        torch.manual_seed(9)
        ps = heads.sample_packed(props, targets)
        assert ps.per_image == min(r, p + (5 if heads.proposal_append_gt else 0)) and ps.boxes.shape[0] == n * ps.per_image
        for i, t in enumerate(targets):
            sl = slice(i * ps.per_image, (i + 1) * ps.per_image)
            c, b, gb, pr = ps.gt_classes[sl], ps.boxes[sl], ps.gt_boxes[sl], ps.gt_probs[sl]
            chosen = c >= 0
            assert not bool(((b[chosen, 2] - b[chosen, 0]) == 0).any())             # a zero padding box was never sampled
            if len(t) == 0:
                assert bool((c[chosen] == k).all()) and torch.equal(gb[chosen], b[chosen]) and float(pr[chosen].abs().max()) == 0.0
                continue
            iou = pairwise_iou(t.gt_boxes, Boxes(b))
            best, arg = iou.max(dim=0)
            fg = chosen & (c < k)
            assert bool((best[fg] >= 0.5).all()) and bool((best[chosen & (c == k)] < 0.5).all())
            assert torch.equal(c[fg], t.gt_classes_offline[arg[fg]])
            close(gb[fg], t.gt_boxes.tensor[arg[fg]], 0)
            close(gb[chosen & ~fg], b[chosen & ~fg], 0)
            close(pr[chosen], t.gt_probs_offline[arg[chosen]], 0)


def test_apply_deltas_on_coordinate_pairs_equals_the_column_form_bit_for_bit():
    """`Box2BoxTransform.apply_deltas` (round 4: (x, y) / (w, h) pairs, a third of the launches) against the oracle's column-by-column
    restatement of detectron2's: identical bits, class-agnostic and per-class deltas, both weight sets of the detector."""
    from coin_amd.box_ops import Box2BoxTransform
    from oracle import d2

    g = torch.Generator().manual_seed(0)
    for w in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        t, o = Box2BoxTransform(w), d2.Box2BoxTransform(w)
        for k in (1, 9):
            boxes = torch.rand(4000, 4, generator=g) * 500
            boxes[:, 2:] += boxes[:, :2]
            deltas = torch.randn(4000, 4 * k, generator=g) * 3          # incl. values beyond the scale clamp
            assert torch.equal(t.apply_deltas(deltas, boxes), o.apply_deltas(deltas, boxes))
