"""The two-view augmentation kernels (coin_aug_*, csrc/augment.hip) and the GPU mapper against the oracle (oracle/augment.py, pinned
to Pillow bit for bit by tests/test_oracle_augment.py): byte work, so every comparison is EXACT."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from oracle import augment as A  # noqa: E402


@pytest.fixture(scope="module")
def K():
    from coin_amd import kernels

    return kernels


def _img(seed, h, w):
    a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[0, :8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 10, 9], [200, 100, 100], [1, 2, 3]]
    return a


dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
same = lambda t, a: np.array_equal(t.cpu().numpy(), a)


@pytest.mark.parametrize("h,w,oh,ow", [(60, 97, 40, 65), (50, 80, 75, 120), (33, 47, 33, 20), (64, 128, 100, 128), (375, 500, 600, 800),
                                       (1024, 2048, 600, 1200), (300, 1000, 400, 1333), (96, 131, 96, 131)])
def test_resize_bilinear_and_flip_equal_the_oracle(K, h, w, oh, ow):
    a = _img(1, h, w)
    ref = A.resize_bilinear(a, oh, ow)
    assert same(K.aug_resize_bilinear(dev(a), oh, ow), ref)
    assert same(K.aug_resize_bilinear(dev(a), oh, ow, flip_h=True), A.hflip(ref))


@pytest.mark.parametrize("f", [0.6, 0.73, 0.999, 1.0, 1.0001, 1.27, 1.4])
def test_enhance_point_ops_equal_the_oracle(K, f):
    a = _img(2, 301, 517)
    d = dev(a)
    assert same(K.aug_point_op(d, K.AUG_BRIGHTNESS, fparam=f), A.adjust_brightness(a, f))
    assert same(K.aug_point_op(d, K.AUG_CONTRAST, fparam=f), A.adjust_contrast(a, f))
    assert same(K.aug_point_op(d, K.AUG_SATURATION, fparam=f), A.adjust_saturation(a, f))


def test_hue_gray_solarize_and_layout_equal_the_oracle(K):
    a = _img(3, 512, 512)                                            # 262 144 pixels through the float / double HSV rows
    d = dev(a)
    for f in (-0.1, -0.0371, 0.0, 0.02, 0.1):
        assert same(K.aug_point_op(d, K.AUG_HUE, iparam=A.hue_shift_of(f)), A.adjust_hue(a, f)), f
    assert same(K.aug_point_op(d, K.AUG_GRAYSCALE), A.rgb_to_grayscale3(a))
    assert same(K.aug_point_op(d, K.AUG_SOLARIZE, iparam=128), A.solarize(a, 128))
    assert same(K.aug_point_op(d, K.AUG_COPY, out_chw=True), a.transpose(2, 0, 1))
    # every colour on the grey axis and the primaries' neighbourhoods (hue sector borders)
    grid = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"), -1).reshape(-1, 52, 3).astype(np.uint8)
    assert same(K.aug_point_op(dev(grid), K.AUG_HUE, iparam=13), A.adjust_hue(grid, 13 / 255 + 1e-9))


@pytest.mark.parametrize("shape", [(40, 60), (7, 9), (128, 200), (600, 800)])
def test_gaussian_blur_equals_the_oracle(K, shape):
    a = _img(4, *shape)
    for r in (0.1, 0.25, 0.5, 0.77, 1.0, 1.3, 1.999, 2.0):
        assert same(K.aug_gaussian_blur(dev(a), r), A.gaussian_blur(a, r)), r


def test_mapper_two_views_equal_the_oracle_chain():
    """DatasetMapperUnsupervised on the GPU vs oracle.two_views with the same drawn parameters (and the draws themselves against the
    oracle's on identical generators), over enough calls to see every operation; VOC- and Cityscapes-shaped inputs."""
    from coin_amd.config import get_cfg
    from coin_amd.data import DatasetMapperUnsupervised

    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (600,), "INPUT.MAX_SIZE_TRAIN", 1333, "INPUT.FORMAT", "RGB", "MODEL.DEVICE", "cuda:0"])
    m = DatasetMapperUnsupervised(cfg, np_rng=np.random.RandomState(5), torch_generator=torch.Generator().manual_seed(5), py_rng=random.Random(5))
    np_rng, gen, py = np.random.RandomState(5), torch.Generator().manual_seed(5), random.Random(5)
    seen = set()
    for i in range(10):
        h, w = (375, 500) if i % 2 else (512, 1024)
        a = _img(10 + i, h, w)
        prm = A.draw_view_params(h, w, (600,), 1333, "choice", 0.5, np_rng, gen, py)
        ref_strong, ref_weak = A.two_views(a, prm)
        strong, weak = m({"file_name": f"mem://{i}", "image_id": str(i), "height": h, "width": w,
                          "annotations": [{"bbox": [10.0, 20.0, 110.0, 220.0], "category_id": 1}]}, image=a)
        assert strong["random_flip"] == weak["random_flip"] == ("horizontal" if prm["flip"] else "no")
        assert strong["image"].is_cuda and strong["image"].dtype == torch.uint8 and strong["image"].shape == weak["image"].shape
        assert same(weak["image"], ref_weak), i
        assert same(strong["image"], ref_strong), (i, prm["strong_ops"])
        assert len(strong["instances"]) == 1 and strong["instances"].image_size == prm["size"]
        seen |= {n for n, _ in prm["strong_ops"]}
    assert {"brightness", "contrast", "saturation", "hue", "blur"} <= seen


def test_mapper_reads_a_file_and_feeds_the_detector_preprocessing(tmp_path):
    """End to end from a PNG on disk: read -> two views on the device -> coin_normalize_pad (the detector's preprocessing)."""
    from PIL import Image

    from coin_amd import kernels as K
    from coin_amd.config import get_cfg
    from coin_amd.data import DatasetMapperUnsupervised

    a = _img(30, 120, 160)
    Image.fromarray(a, "RGB").save(tmp_path / "x.png")
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (96,), "INPUT.MAX_SIZE_TRAIN", 160, "INPUT.FORMAT", "RGB", "MODEL.DEVICE", "cuda:0"])
    m = DatasetMapperUnsupervised(cfg, np_rng=np.random.RandomState(1), torch_generator=torch.Generator().manual_seed(1), py_rng=random.Random(1))
    strong, weak = m({"file_name": str(tmp_path / "x.png"), "image_id": "x", "height": 120, "width": 160})
    assert tuple(weak["image"].shape) == (3, 96, 128)
    batch, sizes = K.normalize_pad([strong["image"], weak["image"]], [0.5, 0.5, 0.5], [0.25, 0.25, 0.25], 32, K.COIN_NHWC, torch.float32)
    assert sizes == [(96, 128), (96, 128)] and tuple(batch.shape) == (2, 96, 128, 3)
    ref = (weak["image"].float() / 255.0 - 0.5) / 0.25
    torch.testing.assert_close(batch[1].permute(2, 0, 1), ref, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        m({"file_name": str(tmp_path / "x.png"), "image_id": "x", "height": 121, "width": 160})


def test_pretrainer_trains_from_image_files_through_the_gpu_mapper(tmp_path):
    """The whole row (f)-3 in place: PNG files -> dataset dicts -> TrainingSampler -> decode threads -> two views on the GPU -> two-crop
    batches -> PRETrainer.run_step with the cached teacher boxes mirrored / rescaled by `random_flip` and the view size (base.py:80-126)."""
    from PIL import Image

    from coin_amd.config import get_cfg
    from coin_amd.data import build_detection_unsupervised_train_loader
    from coin_amd.data.synthetic import SyntheticTeacherCache, synthetic_teacher_result
    from coin_amd.engine import PRETrainer
    import os

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "PRETRAINS", "CLIPDET_synthetic.yaml")
    cfg = get_cfg()
    cfg.merge_from_file(root)
    cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.ENABLED", False, "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0",
                         "INPUT.MIN_SIZE_TRAIN", (256,), "INPUT.MAX_SIZE_TRAIN", 448, "INPUT.FORMAT", "RGB", "DATALOADER.NUM_WORKERS", 2, "SEED", 3])
    g = torch.Generator().manual_seed(0)
    cache, dicts = SyntheticTeacherCache(), []
    for i, (h, w) in enumerate([(300, 500), (320, 480), (375, 500), (333, 500)]):
        a = _img(40 + i, h, w)
        fn = str(tmp_path / f"{i:06d}.png")
        Image.fromarray(a, "RGB").save(fn)
        dicts.append({"file_name": fn, "image_id": f"{i:06d}", "height": h, "width": w})
        cache.add(synthetic_teacher_result(fn, f"{i:06d}", h, w, 12, len(cfg.AMD.CLASS_NAMES), g))
    torch.manual_seed(3)
    np.random.seed(3)
    random.seed(3)
    loader = build_detection_unsupervised_train_loader(cfg, dicts)
    tr = PRETrainer(cfg, data_loader=loader, collect_model=cache)
    recs = [{k: float(v) for k, v in tr.run_step().items()} for _ in range(3)]
    assert all(np.isfinite(v) for r in recs for v in r.values()), recs
    assert recs[0] != recs[2]


def test_cointrainer_teacher_stream_sees_the_views_the_loader_wrote(tmp_path):
    """Round-2 ADVICE (high): with real files the loader's upload + coin_aug_* kernels WRITE the weak / strong views; drawn on the main
    stream (behind the student's backward) while the teacher ran on its own stream, the teacher read unwritten buffers.  `_fetch`
    now draws the batch on the teacher stream.  Here the main stream is kept busy before every fetch; the checksums of the weak views
    the teacher receives (taken on ITS stream) and the strong views the student receives must equal those of a synchronous run."""
    from PIL import Image

    from coin_amd.config import get_cfg
    from coin_amd.data import build_detection_unsupervised_train_loader
    from coin_amd.data.synthetic import SyntheticTeacherCache, synthetic_teacher_result
    from coin_amd.engine import CoinTrainer
    import os

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "coin", "GDINO", "foggy_synthetic.yaml")
    g = torch.Generator().manual_seed(0)
    files = []
    for i, (h, w) in enumerate([(300, 500), (320, 480), (375, 500), (333, 500)]):
        fn = str(tmp_path / f"{i:06d}.png")
        Image.fromarray(_img(60 + i, h, w), "RGB").save(fn)
        files.append((fn, f"{i:06d}", h, w))

    def run(teacher_stream: bool):
        cfg = get_cfg()
        cfg.merge_from_file(root)
        cfg.merge_from_list(["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.ENABLED", False, "AMD.TEXT_TEMPLATES", 2, "MODEL.DEVICE", "cuda:0",
                             "INPUT.MIN_SIZE_TRAIN", (256,), "INPUT.MAX_SIZE_TRAIN", 448, "INPUT.FORMAT", "RGB", "DATALOADER.NUM_WORKERS", 2, "SEED", 3,
                             "AMD.TEACHER_STREAM", teacher_stream])
        gg = torch.Generator().manual_seed(0)
        cache, dicts = SyntheticTeacherCache(), []
        for fn, iid, h, w in files:
            dicts.append({"file_name": fn, "image_id": iid, "height": h, "width": w})
            cache.add(synthetic_teacher_result(fn, iid, h, w, 12, len(cfg.AMD.CLASS_NAMES), gg))
        torch.manual_seed(3)
        np.random.seed(3)
        random.seed(3)
        loader = build_detection_unsupervised_train_loader(cfg, dicts)
        tr = CoinTrainer(cfg, data_loader=loader, cloud_results=cache)
        seen = []
        real = tr.offline_teacher.forward

        def spy(batch, *a, **k):
            seen.append(torch.stack([d["image"].double().sum() for d in batch]))  # on the stream the teacher runs on
            return real(batch, *a, **k)

        tr.offline_teacher.forward = spy
        busy = torch.randn(4096, 4096, device="cuda")
        sums = []
        for _ in range(3):
            for _ in range(40):   # the "student's backward" still queued on the main stream when the next batch is drawn
                busy = (busy @ busy).clamp_(-1, 1)
            strong, _targets = tr._fetch()
            if tr._teacher_stream is not None:
                torch.cuda.current_stream().wait_stream(tr._teacher_stream)
            sums.append(torch.stack([d["image"].double().sum() for d in strong]))
        torch.cuda.synchronize()
        assert (tr._teacher_stream is not None) == teacher_stream
        return torch.stack(seen).cpu(), torch.stack(sums).cpu()

    weak_a, strong_a = run(True)
    weak_b, strong_b = run(False)
    assert torch.equal(weak_a, weak_b), (weak_a, weak_b)
    assert torch.equal(strong_a, strong_b)
    assert float(weak_a.min()) > 0
