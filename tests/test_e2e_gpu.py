"""End-to-end parity of the HIP product on the MI355X (``-m gpu``): loss values and gradients of a whole
detector step against (a) the golden values captured from the reference and (b) the CPU oracle, at parity
boundary P (identical weights, images, sampled anchors and sampled RoIs).  fp32 tolerance 1e-4 (north_star);
the bf16 throughput mode is checked against the same goldens with a bf16 tolerance."""
import numpy as np
import pytest
import torch

from e2e_util import golden_pretrain_case, golden_step_case, run_oracle_pretrain, run_product_pretrain, run_product_step_two
from golden_util import close

pytestmark = pytest.mark.gpu


def check_grads(grads, ref_grads, g64, what):
    """Row P, gradients relative to the tensor's scale.  The heads (RPN head, box predictor, prompt vectors) are held to 1e-4 of the
    exact (fp64) gradient.  Everything behind train-mode BatchNorm over the few hundred samples of this tiny net (backbone / res5) is
    not representable to 1e-4 in fp32 at all: the REFERENCE's own fp32 values are 1e-3..2e-3 away from the exact gradient
    (profiles/r2_grad_precision_study.md).  Those tensors are held to 2 x the largest error the reference itself shows on any of
    them -- a measured floor of the net, not a per-tensor excuse."""
    from real_width import rel_err

    rows = [(k, rel_err(grads[k], g64[k]), rel_err(ref, g64[k]), rel_err(grads[k], ref)) for k, ref in ref_grads.items()]
    print("\n".join(f"{what} {k:62s} product vs fp64 {a:.2e}   reference vs fp64 {b:.2e}   product vs reference {c:.2e}" for k, a, b, c in rows))
    floor = max([b for k, a, b, c in rows if k.startswith("backbone.")] + [0.0])
    for k, a, b, c in rows:
        bound = max(1e-4, 2.0 * floor) if k.startswith("backbone.") else 1e-4
        assert a <= bound, f"{what} {k}: {a:.2e} from the exact gradient (bound {bound:.2e}; the reference's fp32 floor on this net {floor:.2e})"


def _check(losses, grads, case, tol_loss, exact, what):
    assert set(losses) == set(case["ref_losses"])
    for k, ref in case["ref_losses"].items():
        assert abs(float(losses[k]) - ref) < tol_loss * max(1.0, abs(ref)), (k, float(losses[k]), ref)
    check_grads(grads, case["ref_grads"], exact[1], what)


def test_pretrain_step_fp32_vs_reference_golden_and_oracle():
    from e2e_util import run_oracle_pretrain

    case = golden_pretrain_case()
    losses, grads = run_product_pretrain(case, "cuda:0", torch.float32)
    _check(losses, grads, case, 1e-4, run_oracle_pretrain(case, dtype=torch.float64), "pre_train")
    ora_losses, ora_grads = run_oracle_pretrain(case)
    for k in losses:
        assert abs(float(losses[k]) - float(ora_losses[k])) < 1e-4 * max(1.0, abs(float(ora_losses[k]))), k


def test_step_two_fp32_vs_reference_golden():
    from e2e_util import run_oracle_step_two

    case = golden_step_case()
    losses, grads = run_product_step_two(case, "cuda:0", torch.float32)
    _check(losses, grads, case, 1e-4, run_oracle_step_two(case, dtype=torch.float64), "step_two")


def test_pretrain_step_bf16_close_to_golden():
    case = golden_pretrain_case()
    losses, grads = run_product_pretrain(case, "cuda:0", torch.bfloat16)
    print({k: (round(float(losses[k]), 5), round(ref, 5), f"{abs(float(losses[k]) - ref) / max(1.0, abs(ref)):.1e}") for k, ref in case["ref_losses"].items()})
    # round 6: 3e-2 (was 5 %).  This tiny net's channel counts keep its convolutions on the LIBRARY, whose bf16 solvers differ among
    # themselves: 7e-5 ... 3.4e-3 (loss_cls) with the exhaustive solver search, 2.0e-2 on loss_cls with the fast find mode the test processes
    # use (tests/conftest.py).  The bf16 claims that bite are made where the hand-written kernels run: block by block against the storage-
    # rounding oracle (tests/test_parity_gpu.py: outputs 97-99 % identical stored values, every gradient within 2e-2 relative L2).
    for k, ref in case["ref_losses"].items():
        assert np.isfinite(float(losses[k])) and abs(float(losses[k]) - ref) < 3e-2 * max(1.0, abs(ref)), (k, float(losses[k]), ref)


def test_full_size_trainer_steps_bf16_and_fp32_agree():
    """BASELINE shape (800x1333, 512 RoIs/view, RN50): PRETrainer steps run, losses are finite, and the bf16 mode
    tracks the fp32 mode on the same weights, inputs and device RNG stream."""
    import os

    from coin_amd.config import get_cfg
    from coin_amd.engine import PRETrainer

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = {}
    for dt in ("fp32", "bf16"):
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(root, "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml"))
        cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 1, "AMD.SYNTHETIC.NUM_IMAGES", 1, "AMD.COMPUTE_DTYPE", dt, "AMD.TEXT_TEMPLATES", 2,
                             "MODEL.DEVICE", "cuda:0"])
        torch.manual_seed(7)
        tr = PRETrainer(cfg)
        # non-zero bn3 so that the residual branches carry signal in the comparison
        with torch.no_grad():
            for n, p in tr.model.named_parameters():
                if n.endswith("bn3.weight"):
                    p.fill_(0.5)
        torch.manual_seed(8)
        rec = tr.run_step()
        out[dt] = {k: float(v) for k, v in rec.items()}
        assert all(np.isfinite(v) for v in out[dt].values()), out[dt]
        rec2 = tr.run_step()
        assert all(np.isfinite(float(v)) for v in rec2.values())
    assert set(out["fp32"]) == {"loss_text_align", "loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
    for k in out["fp32"]:
        assert abs(out["fp32"][k] - out["bf16"][k]) < 0.05 * max(1.0, abs(out["fp32"][k])), (k, out)


@pytest.mark.parametrize("tag", ["one", "two", "two_nobg_noC", "two_noB", "one_noproto"])
def test_box_predictor_step_and_ckg_update_fp32_vs_reference_golden(tag):
    """FastRCNNOutputLayers.losses(step_one / step_two) on the HIP kernels + the CKG update through `merge_grad_loss`
    (gradient_discrepancy_loss, trainer.py:192-197) vs values captured from the reference modules."""
    from e2e_util import run_product_box_predictor_step
    from golden_util import close

    out = run_product_box_predictor_step(tag, device="cuda:0")
    z = out["z"]
    assert set(out["losses"]) == set(out["ref"])
    for k, v in out["losses"].items():
        assert abs(v - out["ref"][k]) < 1e-4 * max(1.0, abs(out["ref"][k])), (k, v, out["ref"][k])
    for n, g in out.get("merge_grads", {}).items():
        close(g, z["mg::" + n], 2e-4, n)
    close(out["gx"], z["gx"], 1e-4, "gx")
    for k, g in out["grads"].items():
        close(g, z["g::" + k], 1e-4, k)


@pytest.mark.parametrize("burned_up,sync_free_step", [(False, False), (True, False), (True, True)])
def test_cointrainer_full_size_steps(burned_up, sync_free_step):
    """BASELINE configs[2] shape (Foggy-Cityscapes-shaped 667x1333 views, RN50, 512 RoIs, teacher inference with 1000 RoIs):
    CoinTrainer steps run in bf16 on the HIP kernels; losses finite; the CKG module, the student and (after burn-up) the EMA
    teacher move."""
    import os
    import time

    from coin_amd.config import get_cfg
    from coin_amd.data.synthetic import synthetic_offline_detections
    from coin_amd.engine import CoinTrainer

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(root, "configs", "coin", "GDINO", "foggy_synthetic.yaml"))
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0",
                         "CLOUD.BURN_UP_STEP", 0 if burned_up else 100, "CLOUD.PROTOTYPE_UPDATE_START", 0, "CLOUD.CLS_B_THRESH", 0.2,
                         "AMD.SYNC_FREE_STEP", sync_free_step])
    torch.manual_seed(11)
    tr = CoinTrainer(cfg)
    real_forward, g_det = tr.offline_teacher.forward, torch.Generator().manual_seed(7)

    def teacher(batched_inputs, branch=None, **kw):  # the real inference runs; the matcher gets CLIPDET-like detections
        out = real_forward(batched_inputs, branch=branch, **kw)
        assert len(out) == len(batched_inputs)
        return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g_det, device="cuda:0") for d in batched_inputs]

    tr.offline_teacher.forward = teacher
    merge_before = [p.detach().clone() for p in tr.merge.parameters()]
    teacher_w = tr.offline_teacher.roi_heads.box_predictor.cls_score.weight
    tw_before = teacher_w.detach().clone()
    recs = []
    for i in range(3):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        recs.append(tr.run_step())
    torch.cuda.synchronize()
    print(f"CoinTrainer step ({'step_two' if burned_up else 'step_one'}): {(time.perf_counter() - t0) * 1e3:.1f} ms for 2 views")
    for rec in recs:
        vals = {k: float(v) for k, v in rec.items()}
        assert all(np.isfinite(v) for v in vals.values()), vals
        assert {"loss_text_align", "loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_rpn_distillation", "loss_distillation",
                "loss_merge_a", "loss_merge_b", "loss_merge_base", "loss_merge_grad"} <= set(vals), sorted(vals)
    assert any(not torch.equal(a, b) for a, b in zip(merge_before, tr.merge.parameters()))
    assert (not torch.equal(tw_before, teacher_w)) == burned_up


def test_cointrainer_rn101_bdd100k_full_size_steps():
    """BASELINE configs[3] (configs/coin/GDINO/bdd100k_rn101_synthetic.yaml): CLIP-RN101 detector (layer3 = 23 blocks, text dim 512,
    CKG at MERGE_DIM 512, 7 classes), BDD100K-shaped 750x1333 views, one GPU's share of the global batch of 64 (8 images): teacher
    inference on 8 x 1000 RoIs, student step_one on 8 x 512 RoIs, CKG + student updates, bf16.  Its pieces are pinned to the
    reference in tests/test_parity_gpu.py::test_rn101_trunk_and_ckg512_on_device_vs_reference_and_fp64; this is the full-size run."""
    import os
    import time

    from coin_amd.config import get_cfg
    from coin_amd.data.synthetic import synthetic_offline_detections
    from coin_amd.engine import CoinTrainer

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(root, "configs", "coin", "GDINO", "bdd100k_rn101_synthetic.yaml"))
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 8, "AMD.SYNTHETIC.NUM_IMAGES", 8, "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0",
                         "CLOUD.BURN_UP_STEP", 100, "CLOUD.PROTOTYPE_UPDATE_START", 0, "CLOUD.CLS_B_THRESH", 0.2])
    torch.manual_seed(13)
    tr = CoinTrainer(cfg)
    assert len(tr.model.backbone.encoder.visual.layer3) == 23 and tr.model.roi_heads.box_predictor.cls_score.weight.shape[0] == 512
    assert tr.model.roi_heads.num_classes == 7 and all(p.shape[-1] == 512 for p in tr.merge.parameters() if p.dim() == 2)
    real_forward, g_det = tr.offline_teacher.forward, torch.Generator().manual_seed(7)

    def teacher(batched_inputs, branch=None, **kw):  # the real inference runs; the matcher gets CLIPDET-like detections
        out = real_forward(batched_inputs, branch=branch, **kw)
        assert len(out) == len(batched_inputs)
        return [synthetic_offline_detections(tr.model_CLOUD.entry(d["file_name"]), g_det, device="cuda:0") for d in batched_inputs]

    tr.offline_teacher.forward = teacher
    merge_before = [p.detach().clone() for p in tr.merge.parameters()]
    recs = []
    for i in range(3):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        recs.append(tr.run_step())
    torch.cuda.synchronize()
    print(f"CoinTrainer RN101 / BDD100K-shape step_one: {(time.perf_counter() - t0) * 1e3:.1f} ms for 8 views")
    for rec in recs:
        vals = {k: float(v) for k, v in rec.items()}
        assert all(np.isfinite(v) for v in vals.values()), vals
        assert {"loss_text_align", "loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc", "loss_rpn_distillation", "loss_distillation",
                "loss_merge_a", "loss_merge_b", "loss_merge_base", "loss_merge_grad"} <= set(vals), sorted(vals)
    assert any(not torch.equal(a, b) for a, b in zip(merge_before, tr.merge.parameters()))


def test_inference_fp32_vs_reference_golden():
    """Teacher / evaluation path (OpenVocabularyRCNN.inference, clip_rcnn.py:381-426; fast_rcnn_inference, fast_rcnn.py:116-175) on
    the device: eval-mode BatchNorm through the fused apply kernel, RPN top-k + device NMS, RoIAlign, class-wise NMS, top-100."""
    from e2e_util import tiny_product_detector
    from golden_util import T, load, load_weights

    z = load("inference")
    model = tiny_product_detector()
    load_weights(model, z)
    model.to("cuda:0")
    model.eval()
    batch = [{"image": T(z[f"img{i}"]).cuda(), "height": int(z[f"hw{i}"][0]), "width": int(z[f"hw{i}"][1])} for i in range(2)]
    def check(inst, rb, rc, rsc, box_tol, score_tol, what):
        # Pair the rows: the order among tied scores (one box surviving under several classes, neighbouring anchors with the same
        # score) is an artefact of the sort, so every detection is matched to the nearest unused reference row of its class.
        ob, oc, osc = inst.pred_boxes.tensor.double().numpy(), inst.pred_classes.numpy(), inst.scores.double().numpy()
        assert len(ob) == len(rb), (what, len(ob), len(rb))
        used = np.zeros(len(rb), dtype=bool)
        for j in range(len(ob)):
            d = np.abs(rb - ob[j]).max(axis=1) + 1e3 * (rc != oc[j]) + 1e6 * used
            k = int(d.argmin())
            used[k] = True
            assert rc[k] == oc[j], (what, j)
            assert d[k] <= box_tol, (what, j, d[k])
            assert abs(rsc[k] - osc[j]) <= score_tol, (what, j, rsc[k], osc[j])
        assert used.all()

    res = model(batch, branch="test")
    # the same pass in two halves (what CoinTrainer does for the EMA teacher): everything up to the score filter enqueued with fixed
    # shapes and no host round trip, then the per-image filter / class-wise NMS / top-k
    assert model.inference_begin(batch, branch="test")
    res2 = model(batch, branch="test")
    assert model._begun is None
    for i, (a, b) in enumerate(zip(res, res2)):
        ia, ib = a["instances"].to("cpu"), b["instances"].to("cpu")
        check(ib, ia.pred_boxes.tensor.double().numpy(), ia.pred_classes.numpy(), ia.scores.double().numpy(), 1e-4, 1e-5, f"two halves vs one piece, image {i}")
        # fp32 decode: boxes to 1e-3 px (measured 3e-5 on 256-px images), scores 1e-5
        check(ib, z[f"det{i}.pred_boxes"].astype(np.float64), z[f"det{i}.pred_classes"], z[f"det{i}.scores"].astype(np.float64), 1e-3, 1e-5, f"vs golden, image {i}")
        check(ia, z[f"det{i}.pred_boxes"].astype(np.float64), z[f"det{i}.pred_classes"], z[f"det{i}.scores"].astype(np.float64), 1e-3, 1e-5, f"one piece vs golden, image {i}")
    # ... and with the fixed-shape half replayed as ONE HIP graph (captured at the third call of a shape); the weights are then
    # changed in place the way the EMA kernel does (no version bump) and the replay must see them
    for rep in range(5):
        assert model.inference_begin(batch, branch="test", graph=True)
        res3 = model(batch, branch="test")
    assert model._graphs and not model.graph_failed, "the graph path did not capture"
    for i, b in enumerate(res3):
        ib = b["instances"].to("cpu")
        check(ib, z[f"det{i}.pred_boxes"].astype(np.float64), z[f"det{i}.pred_classes"], z[f"det{i}.scores"].astype(np.float64), 1e-3, 1e-5, f"graph vs golden, image {i}")
    from coin_amd import layers as L

    with torch.no_grad():
        w = model.roi_heads.box_predictor.bbox_pred.weight
        w.data.mul_(1.5)          # (a .data write: like the raw-pointer EMA, invisible to the version counter the shadows watch)
        L.invalidate_shadows([w])
    ref = model(batch, branch="test")                                   # eager, one piece
    assert model.inference_begin(batch, branch="test", graph=True)
    got = model(batch, branch="test")                                   # graph replay
    moved = False
    for i, (a, b) in enumerate(zip(ref, got)):
        ia, ib = a["instances"].to("cpu"), b["instances"].to("cpu")
        check(ib, ia.pred_boxes.tensor.double().numpy(), ia.pred_classes.numpy(), ia.scores.double().numpy(), 1e-4, 1e-5, f"graph after a weight change, image {i}")
        moved = moved or len(ia) != len(res3[i]["instances"]) or not np.allclose(np.sort(ia.pred_boxes.tensor.numpy().ravel()), np.sort(res3[i]["instances"].pred_boxes.tensor.cpu().numpy().ravel()), atol=1e-3)
    assert moved, "the weight change did not change the detections: the test would not notice a stale graph"


def test_cointrainer_constructor_teacher_stream_equals_the_synchronous_run_over_ema_iterations():
    """The real `CoinTrainer(cfg)` constructor, three consecutive `run_step` + `prepare_next` iterations in step_two with an EMA due
    at EVERY iteration: with `AMD.TEACHER_STREAM` on (the next iteration's EMA / teacher inference / matching are issued on the
    teacher's stream right after the optimizer step, `prepare_next`) against the same trainer run synchronously (teacher on the
    main stream, no `prepare_next`).  What must hold for the ordering to be right (trainer.py:149-218, ts_ensemble.py:39-69): the
    EMA of iteration i+1 reads the weights the optimizer of iteration i wrote, and the optimizer of iteration i+1 does not
    overtake it -- then the teacher's parameters after every iteration and the number of (A, B, C) targets agree between the two
    runs (parameters to the run-to-run spread of the library convolutions)."""
    import os
    import random

    from coin_amd.config import get_cfg
    from coin_amd.engine import CoinTrainer

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "GDINO", "foggy_synthetic.yaml")

    def run(stream: bool):
        cfg = get_cfg()
        cfg.merge_from_file(root)
        cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.SYNTHETIC.HEIGHT", 256, "AMD.SYNTHETIC.WIDTH", 384,
                             "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0", "AMD.COMPUTE_DTYPE", "fp32", "CLOUD.BURN_UP_STEP", 0,
                             "CLOUD.OFFLINE_TEACHER_UPDATE_ITER", 1, "CLOUD.EMA_KEEP_RATE_OFFLINE", 0.5, "CLOUD.PROTOTYPE_UPDATE_START", 0,
                             "AMD.TEACHER_STREAM", stream, "AMD.TEACHER_GRAPH", False, "SEED", 5])   # (with the graph on, EMA-due passes run on the default stream)
        torch.manual_seed(5)
        np.random.seed(5)
        random.seed(5)
        tr = CoinTrainer(cfg)
        tr.max_iter = 10
        seen, match = [], tr.match_boxes
        cnt = lambda x: 0 if x is None else len(x)
        tr.match_boxes = lambda b, o, **kw: (lambda t: (seen.append([[cnt(x[0]), cnt(x[1]), cnt(x[2])] for x in t[0]]), t)[1])(match(b, o, **kw))
        teacher_sums, losses = [], []
        for _ in range(3):
            rec = tr.run_step()
            if stream:
                tr.prepare_next()          # EMA + teacher pass of the NEXT iteration, on the teacher stream, beside this backward
            torch.cuda.synchronize()
            losses.append({k: float(v) for k, v in rec.items()})
            teacher_sums.append(torch.stack([p.detach().double().abs().sum() for p in tr.offline_teacher.parameters()]).cpu())
        assert (tr._teacher_stream is not None) == stream
        return seen, teacher_sums, losses

    seen_a, teach_a, loss_a = run(True)
    seen_b, teach_b, loss_b = run(False)
    # stream run: the teacher has already taken the EMA of the iteration to come (prepare_next), i.e. it is one EMA AHEAD of the
    # synchronous run after every step: its state after step i equals the synchronous teacher's state after step i+1's EMA, which the
    # synchronous run exposes at the end of step i+1 (the EMA is the first thing a step does)
    for i in range(2):
        torch.testing.assert_close(teach_a[i], teach_b[i + 1], rtol=5e-3, atol=1e-6)   # |sum| of tiny (zero-initialised) tensors: absolute floor
    assert seen_a[:3] == seen_b[:3], (seen_a, seen_b)     # the same targets reach the student in both runs
    for la, lb in zip(loss_a, loss_b):
        assert set(la) == set(lb)
        for k in la:
            assert abs(la[k] - lb[k]) <= 5e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])


def test_cointrainer_teacher_prefetch_equals_the_unprefetched_run_in_step_one():
    """step_one (iter < BURN_UP_STEP: the teacher is frozen, trainer.py:170-172): with `AMD.TEACHER_PREFETCH` the teacher pass of batch
    i+1 is enqueued BEFORE the student's step i (first half, fixed shapes, no host round trip) and read back after it; without it the
    pass follows the step.  Same weights, same batches in the same order -> the same (A, B, C) targets reach the student and the
    losses agree to the run-to-run spread of the library convolutions; also with a caller that never calls prepare_next()."""
    import os
    import random

    from coin_amd.config import get_cfg
    from coin_amd.engine import CoinTrainer

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "GDINO", "foggy_synthetic.yaml")

    def run(prefetch: bool, use_prepare_next: bool = True):
        cfg = get_cfg()
        cfg.merge_from_file(root)
        cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 4, "AMD.SYNTHETIC.HEIGHT", 256, "AMD.SYNTHETIC.WIDTH", 384,
                             "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0", "AMD.COMPUTE_DTYPE", "fp32", "CLOUD.BURN_UP_STEP", 10 ** 6,
                             "CLOUD.PROTOTYPE_UPDATE_START", 0, "AMD.TEACHER_PREFETCH", prefetch, "SEED", 7])
        torch.manual_seed(7)
        np.random.seed(7)
        random.seed(7)
        tr = CoinTrainer(cfg)
        tr.max_iter = 10
        seen, match = [], tr.match_boxes
        cnt = lambda x: 0 if x is None else len(x)
        tr.match_boxes = lambda b, o, **kw: (lambda t: (seen.append([[cnt(x[0]), cnt(x[1]), cnt(x[2])] for x in t[0]]), t)[1])(match(b, o, **kw))
        begun = []
        begin = tr.offline_teacher.inference_begin
        tr.offline_teacher.inference_begin = lambda *a, **k: (begun.append(tr.iter), begin(*a, **k))[1]
        losses = []
        for _ in range(4):
            rec = tr.run_step()
            if use_prepare_next:
                tr.prepare_next()
            torch.cuda.synchronize()
            losses.append({k: float(v) for k, v in rec.items()})
        return seen, losses, begun

    seen_a, loss_a, begun_a = run(True)
    seen_b, loss_b, begun_b = run(False)
    seen_c, loss_c, _ = run(True, use_prepare_next=False)
    # with the prefetch the first half of batch i+1 is issued while `tr.iter` is still i (before the step), without it after the step
    assert begun_a[1:4] == [0, 1, 2] and begun_b[1:4] == [1, 2, 3], (begun_a, begun_b)
    assert seen_a[:4] == seen_b[:4] == seen_c[:4], (seen_a, seen_b, seen_c)
    for la, lb, lc in zip(loss_a, loss_b, loss_c):
        assert set(la) == set(lb) == set(lc)
        for k in la:
            assert abs(la[k] - lb[k]) <= 5e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
            assert abs(lc[k] - lb[k]) <= 5e-3 * max(1.0, abs(lb[k])), (k, lc[k], lb[k])


def test_teacher_bf16_shadows_and_frozen_constants_follow_the_ema_kernel():
    """Round-3 ADVICE (high): `coin_ema_update` writes the teacher's fp32 masters through raw pointers taken from `state_dict()`
    tensors (detached aliases: another id(), no `_version` bump), so everything DERIVED from them -- bf16 weight shadows, folded
    frozen convolutions, frozen-norm constants -- has to be marked stale by storage, or every eager bf16 teacher pass after an EMA
    reads the pre-EMA weights (ts_ensemble.py:39-69, trainer.py:170-177).  bf16 trainer, TEACHER_GRAPH off (the eager pass is the one
    at risk): after an EMA with a student that differs strongly from the teacher, every bf16 shadow of the teacher must equal a fresh
    cast of its master, and the eager teacher pass must equal the pass of the same model after ALL derived state was dropped."""
    import os
    import random

    from coin_amd import layers as L
    from coin_amd.config import get_cfg
    from coin_amd.engine import CoinTrainer
    from coin_amd.modeling import backbone as B

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "GDINO", "foggy_synthetic.yaml")
    cfg = get_cfg()
    cfg.merge_from_file(root)
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.SYNTHETIC.HEIGHT", 256, "AMD.SYNTHETIC.WIDTH", 384,
                         "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0", "AMD.COMPUTE_DTYPE", "bf16", "CLOUD.BURN_UP_STEP", 0,
                         "CLOUD.OFFLINE_TEACHER_UPDATE_ITER", 1, "CLOUD.EMA_KEEP_RATE_OFFLINE", 0.5, "AMD.TEACHER_STREAM", False,
                         "AMD.TEACHER_GRAPH", False, "AMD.TEACHER_PREFETCH", False, "SEED", 11])
    torch.manual_seed(11)
    np.random.seed(11)
    random.seed(11)
    tr = CoinTrainer(cfg)
    teacher = tr.offline_teacher
    _strong, weak = next(tr._data_loader_iter)

    def detect():
        tr._teacher_mode(False)
        try:
            with torch.no_grad():
                out = teacher(weak, branch="test")   # (the model opens its own bf16 autocast region, AMD.COMPUTE_DTYPE)
        finally:
            tr._teacher_mode(True)
        return [(o["instances"].pred_boxes.tensor.float().cpu(), o["instances"].scores.float().cpu()) for o in out]

    detect()   # builds every shadow / folded weight / frozen constant of the teacher from the PRE-EMA weights
    n_shadows = sum(1 for p in teacher.parameters() if L.shadow_of(p) is not None)
    assert n_shadows > 20, "the teacher pass did not run on bf16 shadows: the test would prove nothing"
    with torch.no_grad():
        for p in tr.model.parameters():   # a student far from the teacher, so that a stale read is visible
            p.mul_(1.0 + 0.5 * torch.rand_like(p))
        for m in tr.model.modules():
            if hasattr(m, "running_var") and m.running_var is not None:
                m.running_var.mul_(1.7)
                m.running_mean.add_(0.05)
    tr.update_teacher(0.5)
    got = detect()
    stale = [n for n, p in teacher.named_parameters() if L.shadow_of(p) is not None and not torch.equal(L.shadow_of(p), p.detach().to(L.shadow_of(p).dtype))]
    assert not stale, f"bf16 shadows still hold the pre-EMA weights: {stale[:5]} (+{max(0, len(stale) - 5)})"
    # reference: drop ALL derived state by hand, run again
    L._SHADOWS.clear()
    L._FROZEN_CONSTS.clear()
    B._FOLDED.clear()
    for m in teacher.modules():
        if hasattr(m, "invalidate_text_cache"):
            m.invalidate_text_cache()
    want = detect()
    for (gb, gs), (wb, ws) in zip(got, want):
        assert gb.shape == wb.shape, (gb.shape, wb.shape)
        torch.testing.assert_close(gs, ws, rtol=0, atol=2e-3)
        torch.testing.assert_close(gb, wb, rtol=0, atol=0.5)
