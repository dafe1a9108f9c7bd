"""Known-answer tests for oracle/d2.py (the restated detectron2 / torchvision / fvcore arithmetic).

The reference vendors none of these libraries and holds no test for them, so each function is pinned
here against values worked out BY HAND from the libraries' documented behaviour, plus the known answers the upstream
projects publish in their own test suites (`*_upstream_published_values`).  The expected numbers below are literals with
their derivation or source in the comment next to them, never the output of the code under test.
"""
import math

import numpy as np
import pytest
import torch

from oracle import d2


# ------------------------------------------------------------------------------------------ boxes / IoU
def test_boxes_area_clip_nonempty_scale():
    b = d2.Boxes(torch.tensor([[0.0, 0.0, 10.0, 20.0], [-5.0, -5.0, 5.0, 5.0], [8.0, 8.0, 30.0, 12.0]]))
    assert b.area().tolist() == [200.0, 100.0, 88.0]
    b.clip((10, 16))  # (h, w): x in [0,16], y in [0,10]
    assert b.tensor.tolist() == [[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 5.0, 5.0], [8.0, 8.0, 16.0, 10.0]]
    assert b.nonempty().tolist() == [True, True, True]
    assert b.nonempty(threshold=4.0).tolist() == [True, True, False]  # third box is 8 x 2
    b.scale(0.5, 2.0)
    assert b.tensor[2].tolist() == [4.0, 16.0, 8.0, 20.0]


def test_pairwise_iou_hand_values():
    a = d2.Boxes(torch.tensor([[0.0, 0.0, 2.0, 2.0], [0.0, 0.0, 1.0, 1.0]]))
    b = d2.Boxes(torch.tensor([[1.0, 1.0, 3.0, 3.0], [0.0, 0.0, 2.0, 2.0], [5.0, 5.0, 6.0, 6.0]]))
    iou = d2.pairwise_iou(a, b)
    # [0,2]^2 vs [1,3]^2: inter 1, union 4+4-1=7 ; identical -> 1 ; disjoint -> 0
    # [0,1]^2 vs [1,3]^2: touch only -> 0 ; vs [0,2]^2: inter 1, union 4 -> 0.25
    expect = torch.tensor([[1.0 / 7.0, 1.0, 0.0], [0.0, 0.25, 0.0]])
    torch.testing.assert_close(iou, expect, rtol=0, atol=1e-7)


# ------------------------------------------------------------------------------------------ Matcher truth tables
def test_matcher_roi_heads_thresholds():
    m = d2.Matcher([0.5], [0, 1], allow_low_quality_matches=False)  # ROI_HEADS.IOU_THRESHOLDS [0.5], IOU_LABELS [0,1]
    q = torch.tensor([[0.6, 0.2, 0.5, 0.0], [0.7, 0.49, 0.1, 0.0]])
    idx, lab = m(q)
    assert idx.tolist() == [1, 1, 0, 0]  # argmax over gt rows; ties -> first row
    assert lab.tolist() == [1, 0, 1, 0]  # >= 0.5 is foreground (inclusive lower bound)


def test_matcher_rpn_bands_and_low_quality():
    q = torch.tensor([[0.8, 0.29, 0.3, 0.69, 0.1], [0.1, 0.2, 0.0, 0.7, 0.25]])
    idx, lab = d2.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=False)(q)
    assert idx.tolist() == [0, 0, 0, 1, 1]
    assert lab.tolist() == [1, 0, -1, 1, 0]  # <0.3 bg, [0.3,0.7) ignore, >=0.7 fg
    # low-quality promotion: every gt's best prediction(s) become fg even below threshold.
    q2 = torch.tensor([[0.2, 0.1, 0.05], [0.0, 0.4, 0.4]])
    idx, lab = d2.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)(q2)
    assert idx.tolist() == [0, 1, 1]
    assert lab.tolist() == [1, 1, 1]  # col 0 best for gt0 (0.2); cols 1 and 2 tie as best for gt1 -> both promoted


def test_matcher_no_gt():
    idx, lab = d2.Matcher([0.5], [0, 1])(torch.zeros((0, 3)))
    assert idx.tolist() == [0, 0, 0] and lab.tolist() == [0, 0, 0]


def test_subsample_labels_counts_and_membership():
    torch.manual_seed(0)
    labels = torch.tensor([1, 0, -1, 3, 0, 0, 8, 0, -1, 2, 8, 8])  # bg label = 8
    pos, neg = d2.subsample_labels(labels, 6, 0.5, 8)
    assert len(pos) == 3 and len(neg) == 3  # int(6 * 0.5) = 3 of the 7 foreground entries (class 0 is a class), 3 of the 3 background
    assert set(pos.tolist()) <= {0, 1, 3, 4, 5, 7, 9} and set(neg.tolist()) <= {6, 10, 11}
    pos, neg = d2.subsample_labels(labels, 100, 0.25, 8)
    assert len(pos) == 7 and len(neg) == 3  # capped by availability; -1 entries never sampled
    pos, neg = d2.subsample_labels(torch.tensor([8, 8, 8, 8]), 3, 0.5, 8)
    assert len(pos) == 0 and len(neg) == 3  # no positives: negatives fill up to num_samples


# ------------------------------------------------------------------------------------------ Box2BoxTransform
def test_box2box_deltas_hand_values():
    t = d2.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    src = torch.tensor([[0.0, 0.0, 10.0, 20.0]])   # w 10, h 20, centre (5, 10)
    tgt = torch.tensor([[5.0, 10.0, 25.0, 30.0]])  # w 20, h 20, centre (15, 20)
    d = t.get_deltas(src, tgt)
    # dx = 10*(15-5)/10 = 10 ; dy = 10*(20-10)/20 = 5 ; dw = 5*ln(20/10) ; dh = 5*ln(1) = 0
    torch.testing.assert_close(d, torch.tensor([[10.0, 5.0, 5.0 * math.log(2.0), 0.0]]), rtol=0, atol=1e-6)
    back = t.apply_deltas(d, src)
    torch.testing.assert_close(back, tgt, rtol=0, atol=1e-4)


def test_box2box_scale_clamp():
    t = d2.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0))
    src = torch.tensor([[0.0, 0.0, 16.0, 16.0]])
    out = t.apply_deltas(torch.tensor([[0.0, 0.0, 100.0, -1.0]]), src)
    # dw clamped to ln(1000/16): width = 16 * 1000/16 = 1000 around centre 8 ; dh = -1: height 16/e
    hh = 16.0 / math.e
    torch.testing.assert_close(out, torch.tensor([[8.0 - 500.0, 8.0 - hh / 2, 8.0 + 500.0, 8.0 + hh / 2]]), rtol=1e-6, atol=1e-3)


def test_box2box_class_specific_layout():
    t = d2.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    src = torch.tensor([[0.0, 0.0, 10.0, 10.0]])
    out = t.apply_deltas(torch.tensor([[0.0, 0.0, 0.0, 0.0, 10.0, 0.0, 0.0, 0.0]]), src)  # two classes x 4
    torch.testing.assert_close(out, torch.tensor([[0.0, 0.0, 10.0, 10.0, 10.0, 0.0, 20.0, 10.0]]), rtol=0, atol=1e-5)


def test_smooth_l1():
    a, b = torch.tensor([0.0, 1.0, -3.0]), torch.tensor([0.5, 1.0, 1.0])
    assert d2.smooth_l1_loss(a, b, beta=0.0, reduction="sum").item() == pytest.approx(4.5)  # 0.5 + 0 + 4
    # beta = 1: |d| < 1 -> 0.5 d^2 = 0.125 ; |d| = 4 -> 4 - 0.5 = 3.5
    assert d2.smooth_l1_loss(a, b, beta=1.0, reduction="sum").item() == pytest.approx(3.625)
    assert d2.smooth_l1_loss(a[:0], b[:0], beta=0.0, reduction="mean").item() == 0.0


# ------------------------------------------------------------------------------------------ anchors
def test_cell_anchors_and_grid():
    cell = d2.generate_cell_anchors([32.0], [0.5, 1.0, 2.0])
    s = math.sqrt(32.0 * 32.0 / 0.5)  # ratio 0.5 = h/w: w = sqrt(area/0.5) = 45.2548.., h = 22.627..
    expect = torch.tensor([[-s / 2, -s / 4, s / 2, s / 4], [-16.0, -16.0, 16.0, 16.0], [-s / 4, -s / 2, s / 4, s / 2]])
    torch.testing.assert_close(cell, expect, rtol=0, atol=1e-4)
    gen = d2.DefaultAnchorGenerator(sizes=[[32.0]], aspect_ratios=[[1.0]], strides=[16], offset=0.0)
    a = gen([torch.zeros(1, 1, 2, 3)])[0].tensor
    # row-major over (y, x), anchors innermost; shift = stride * index (offset 0)
    assert a.shape == (6, 4)
    assert a[0].tolist() == [-16.0, -16.0, 16.0, 16.0]
    assert a[1].tolist() == [0.0, -16.0, 32.0, 16.0]
    assert a[3].tolist() == [-16.0, 0.0, 16.0, 32.0]
    assert a[5].tolist() == [16.0, 0.0, 48.0, 32.0]


# ------------------------------------------------------------------------------------------ NMS
def test_nms_greedy_hand_case():
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0],   # A score .9
                          [1.0, 1.0, 11.0, 11.0],   # B score .8  IoU(A,B) = 81/119 = .68 -> suppressed by A at thr .5
                          [9.0, 9.0, 19.0, 19.0],   # C score .7  IoU(A,C) = 1/199, IoU(B,C) = 4/196 -> kept
                          [9.5, 9.5, 19.5, 19.5],   # D score .95 IoU(C,D) = 90.25/109.75 = .82 -> D suppresses C
                          ])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert d2.nms(boxes, scores, 0.5).tolist() == [3, 0]
    assert d2.nms(boxes, scores, 0.7).tolist() == [3, 0, 1]      # B survives at .7 (IoU .68), C still dies (.82)
    assert d2.nms(boxes, scores, 0.9).tolist() == [3, 0, 1, 2]
    # exactly-at-threshold IoU is kept (strict >): [0,2]x[0,1] vs [1,3]x[0,1]: inter 1, union 3 -> 1/3
    bb = torch.tensor([[0.0, 0.0, 2.0, 1.0], [1.0, 0.0, 3.0, 1.0]])
    assert d2.nms(bb, torch.tensor([1.0, 0.5]), 1.0 / 3.0 + 1e-7).tolist() == [0, 1]
    assert d2.nms(bb, torch.tensor([1.0, 0.5]), 0.33).tolist() == [0]


def test_batched_nms_separates_categories():
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 10.0]])
    scores = torch.tensor([0.5, 0.9, 0.7])
    keep = d2.batched_nms(boxes, scores, torch.tensor([0, 1, 0]), 0.5)
    assert keep.tolist() == [1, 2]  # identical boxes: one survivor per category, by score
    assert d2.batched_nms(boxes[:0], scores[:0], torch.tensor([], dtype=torch.long), 0.5).numel() == 0


# ------------------------------------------------------------------------------------------ RoIAlign (aligned=True, sampling_ratio=0)
def test_roi_align_hand_computed_4x4():
    # feature f(y, x) = 4*y + x on a 4x4 map; bilinear interpolation of an affine function is exact inside the map,
    # so every sample equals 4*y + x at its (continuous) position and a bin's value is the mean over its samples.
    feat = np.arange(16, dtype=np.float64).reshape(1, 1, 4, 4)
    # RoI (x0,y0,x1,y1) = (1,1,3,3), scale 1, aligned: shifted by -0.5 -> [0.5,2.5]^2 ; 2x2 bins of size 1 ;
    # adaptive grid = ceil(2/2) = 1 sample per bin at the bin centre: (y,x) in {1.0, 2.0}^2  -> values 4y + x
    out = d2.roi_align_forward_np(feat, np.array([[0, 1.0, 1.0, 3.0, 3.0]]), (2, 2), 1.0, 0, True)
    np.testing.assert_allclose(out[0, 0], [[5.0, 6.0], [9.0, 10.0]], rtol=0, atol=1e-12)
    # RoI (0,0,4,4): shifted [-0.5,3.5]^2, bins of size 2, grid ceil(4/2)=2 samples/bin/axis at offsets .5 and 1.5:
    # y in {0, 1 | 2, 3}; means: bin(0,0) = 4*0.5 + 0.5 = 2.5 ; bin(0,1) = 2 + 2.5 = 4.5 ; bin(1,0) = 10.5 ; bin(1,1) = 12.5
    out = d2.roi_align_forward_np(feat, np.array([[0, 0.0, 0.0, 4.0, 4.0]]), (2, 2), 1.0, 0, True)
    np.testing.assert_allclose(out[0, 0], [[2.5, 4.5], [10.5, 12.5]], rtol=0, atol=1e-12)


def test_roi_align_upstream_published_values():
    """The known answers detectron2 itself tests its ROIAlign against (detectron2 v0.5, tests/layers/test_roi_align.py,
    `ROIAlignTest.test_forward_output`): a 5x5 ramp, box (1,1,3,3), 4x4 output, with and without the half-pixel correction.
    These literals come from the upstream project's test, not from this repository."""
    feat = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    rois = np.array([[0, 1.0, 1.0, 3.0, 3.0]], dtype=np.float32)
    old_results = [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]]               # aligned=False
    correct_results = [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]]  # aligned=True
    np.testing.assert_allclose(d2.roi_align_forward_np(feat, rois, (4, 4), 1.0, 0, False)[0, 0], old_results, rtol=0, atol=1e-6)
    np.testing.assert_allclose(d2.roi_align_forward_np(feat, rois, (4, 4), 1.0, 0, True)[0, 0], correct_results, rtol=0, atol=1e-6)
    got = d2.roi_align_torch(torch.from_numpy(feat), torch.from_numpy(rois), (4, 4), 1.0, 0, True)
    np.testing.assert_allclose(got[0, 0].numpy(), correct_results, rtol=0, atol=1e-6)


def test_matcher_upstream_published_values():
    """detectron2 v0.5 tests/modeling/test_matcher.py: RPN thresholds [0.3, 0.7], labels [0, -1, 1], low-quality matches on."""
    m = d2.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    q = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    matches, labels = m(q)
    assert matches.tolist() == [1, 1, 2, 0]
    assert labels.tolist() == [-1, 1, 0, 1]


def test_pairwise_iou_upstream_published_values():
    """detectron2 v0.5 tests/structures/test_boxes.py (`TestBoxIOU.test_pairwise_iou`)."""
    b1 = d2.Boxes(torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]]))
    b2 = d2.Boxes(torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0],
                                [0.5, 0.5, 1.5, 1.5]]))
    row = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]
    torch.testing.assert_close(d2.pairwise_iou(b1, b2), torch.tensor([row, row]), rtol=0, atol=1e-6)


def test_roi_align_border_rules():
    feat = np.arange(16, dtype=np.float64).reshape(1, 1, 4, 4)
    # one bin, one sample (roi 1x1 -> grid ceil(1/1)=1) at y = x = -0.75 (inside the [-1, H] band: clamped to 0 -> f(0,0) = 0 ..)
    out = d2.roi_align_forward_np(feat + 7.0, np.array([[0, -0.75, -0.75, 0.25, 0.25]]), (1, 1), 1.0, 0, True)
    # sample centre = -0.75 - 0.5 + 0.5 = -0.75 -> clamped to (0,0) -> 7.0
    np.testing.assert_allclose(out[0, 0], [[7.0]], atol=1e-12)
    # sample at -1.25 (< -1) contributes zero
    out = d2.roi_align_forward_np(feat + 7.0, np.array([[0, -1.25, -1.25, -0.25, -0.25]]), (1, 1), 1.0, 0, True)
    np.testing.assert_allclose(out[0, 0], [[0.0]], atol=1e-12)
    # sample beyond the last row/col but <= H: clamped to the edge pixel (3,3) = 15 + 7
    out = d2.roi_align_forward_np(feat + 7.0, np.array([[0, 3.4, 3.4, 4.4, 4.4]]), (1, 1), 1.0, 0, True)
    np.testing.assert_allclose(out[0, 0], [[22.0]], atol=1e-12)
    # aligned=True has no minimum RoI size: a zero-area RoI gets an adaptive grid of ceil(0/2) = 0 samples, the sample loops
    # do not run and the bin is 0 / max(0, 1) = 0 (torchvision's `count = max(grid_h * grid_w, 1)`)
    out = d2.roi_align_forward_np(feat + 7.0, np.array([[0, 2.0, 1.5, 2.0, 1.5]]), (2, 2), 1.0, 0, True)
    np.testing.assert_allclose(out[0, 0], np.zeros((2, 2)), atol=1e-12)
    # a tiny but non-empty RoI (0.5 x 0.5 at (1.5,1.0)): grid ceil(.25/... ) = 1, bins .25 wide, centres at start + {.125,.375}
    out = d2.roi_align_forward_np(feat, np.array([[0, 2.0, 1.5, 2.5, 2.0]]), (2, 2), 1.0, 0, True)
    ys, xs = np.array([1.0 + 0.125, 1.0 + 0.375]), np.array([1.5 + 0.125, 1.5 + 0.375])
    np.testing.assert_allclose(out[0, 0], 4 * ys[:, None] + xs[None, :], atol=1e-12)


def test_roi_align_scale_and_sampling_ratio():
    feat = np.arange(16, dtype=np.float64).reshape(1, 1, 4, 4)
    # spatial_scale 0.5 on image coords (2,2,6,6) == the (1,1,3,3) case above
    out = d2.roi_align_forward_np(feat, np.array([[0, 2.0, 2.0, 6.0, 6.0]]), (2, 2), 0.5, 0, True)
    np.testing.assert_allclose(out[0, 0], [[5.0, 6.0], [9.0, 10.0]], atol=1e-12)
    # explicit sampling_ratio 2 on the same RoI: samples at bin_start + {.25,.75} -> same means (affine map)
    out = d2.roi_align_forward_np(feat, np.array([[0, 2.0, 2.0, 6.0, 6.0]]), (2, 2), 0.5, 2, True)
    np.testing.assert_allclose(out[0, 0], [[5.0, 6.0], [9.0, 10.0]], atol=1e-12)


def test_roi_align_backward_is_transpose_of_forward():
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 3, 6, 7))
    rois = np.array([[0, 1.0, 2.0, 20.0, 17.0], [1, -3.0, 0.0, 9.0, 30.0], [1, 5.0, 5.0, 5.0, 5.0]])
    out = d2.roi_align_forward_np(feat, rois, (3, 2), 0.25, 0, True)
    go = rng.standard_normal(out.shape)
    gi = d2.roi_align_backward_np(go, rois, feat.shape, 0.25, 0, True)
    # <forward(feat), go> == <feat, backward(go)> for a linear operator
    assert float((out * go).sum()) == pytest.approx(float((feat * gi).sum()), rel=1e-12, abs=1e-10)


def test_roi_align_torch_matches_loops():
    rng = np.random.default_rng(4)
    feat = rng.standard_normal((2, 5, 9, 11))
    rois = np.array([[0, 3.0, 2.0, 120.0, 90.0], [1, -8.0, -8.0, 40.0, 200.0], [0, 50.0, 60.0, 51.0, 61.0], [1, 0.0, 0.0, 176.0, 144.0]])
    ref = d2.roi_align_forward_np(feat, rois, (7, 7), 1 / 16, 0, True)
    ft = torch.from_numpy(feat).requires_grad_(True)
    out = d2.roi_align_torch(ft, torch.from_numpy(rois), (7, 7), 1 / 16, 0, True)
    np.testing.assert_allclose(out.detach().numpy(), ref, rtol=1e-10, atol=1e-10)
    go = rng.standard_normal(ref.shape)
    out.backward(torch.from_numpy(go))
    np.testing.assert_allclose(ft.grad.numpy(), d2.roi_align_backward_np(go, rois, feat.shape, 1 / 16, 0, True), rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------------------------------ misc
def test_frozen_bn_affine():
    bn = d2.FrozenBatchNorm2d(2, eps=1e-5)
    bn.weight.copy_(torch.tensor([2.0, 0.5]))
    bn.bias.copy_(torch.tensor([1.0, -1.0]))
    bn.running_mean.copy_(torch.tensor([0.5, 0.0]))
    bn.running_var.copy_(torch.tensor([4.0 - 1e-5, 1.0 - 1e-5]))
    x = torch.tensor([[[[2.5]], [[3.0]]]])
    # (2.5-0.5)/2*2+1 = 3 ; (3-0)/1*0.5-1 = 0.5
    torch.testing.assert_close(bn(x).flatten(), torch.tensor([3.0, 0.5]), rtol=0, atol=1e-6)


def test_image_list_padding():
    a, b = torch.ones(3, 4, 6), 2 * torch.ones(3, 5, 3)
    il = d2.ImageList.from_tensors([a, b], size_divisibility=0)
    assert il.tensor.shape == (2, 3, 5, 6) and il.image_sizes == [(4, 6), (5, 3)]
    assert il.tensor[0, :, 4].abs().sum() == 0 and il.tensor[1, :, :, 3:].abs().sum() == 0  # zero pad bottom/right
    il = d2.ImageList.from_tensors([a, b], size_divisibility=4)
    assert il.tensor.shape == (2, 3, 8, 8)


def test_warmup_factor():
    assert d2.get_warmup_factor_at_iter("linear", 0, 100, 0.001) == pytest.approx(0.001)
    assert d2.get_warmup_factor_at_iter("linear", 50, 100, 0.001) == pytest.approx(0.5005)  # .001*.5 + .5
    assert d2.get_warmup_factor_at_iter("linear", 100, 100, 0.001) == 1.0
    assert d2.get_warmup_factor_at_iter("constant", 7, 100, 0.3) == 0.3


def test_add_ground_truth_to_proposals_logit():
    gt = d2.Instances((10, 10))
    gt.gt_boxes = d2.Boxes(torch.tensor([[1.0, 1.0, 5.0, 5.0]]))
    gt.gt_classes = torch.tensor([2])
    p = d2.Instances((10, 10))
    p.proposal_boxes = d2.Boxes(torch.tensor([[0.0, 0.0, 3.0, 3.0], [2.0, 2.0, 9.0, 9.0]]))
    p.objectness_logits = torch.tensor([0.3, -1.0])
    out = d2.add_ground_truth_to_proposals([gt], [p])[0]
    assert len(out) == 3 and out.proposal_boxes.tensor[2].tolist() == [1.0, 1.0, 5.0, 5.0]
    # logit of probability 1 - 1e-10: ln((1-1e-10)/1e-10) = 23.0258...
    assert out.objectness_logits[2].item() == pytest.approx(math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10))), rel=1e-6)


def test_detector_postprocess_rescale_clip():
    r = d2.Instances((100, 200))
    r.pred_boxes = d2.Boxes(torch.tensor([[10.0, 10.0, 50.0, 50.0], [190.0, 90.0, 230.0, 120.0], [300.0, 300.0, 310.0, 310.0]]))
    r.scores = torch.tensor([0.9, 0.8, 0.7])
    out = d2.detector_postprocess(r, 50, 400)  # sx = 2, sy = .5
    assert out.image_size == (50, 400)
    assert out.pred_boxes.tensor.tolist() == [[20.0, 5.0, 100.0, 25.0], [380.0, 45.0, 400.0, 50.0]]  # third clips to empty
    assert out.scores.tolist() == pytest.approx([0.9, 0.8])
