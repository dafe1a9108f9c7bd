"""Host logic of the two-view input pipeline: the random decisions, annotations and batching of
coin_amd/data/dataset_mapper.py against the oracle's restatement (oracle/augment.py) -- no GPU needed; the pixels are
tests/test_input_pipeline_gpu.py."""
import random

import numpy as np
import torch

from coin_amd.config import get_cfg
from coin_amd.data import AspectRatioGroupedDatasetTwoCrop, DatasetMapperUnsupervised
from coin_amd.data.dataset_mapper import shortest_edge_size
from oracle import augment as A


def _cfg(**kw):
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (600,), "INPUT.MAX_SIZE_TRAIN", 1333, "INPUT.FORMAT", "RGB", "MODEL.DEVICE", "cpu"] +
                        [x for k, v in kw.items() for x in (k, v)])
    return cfg


def test_draws_follow_the_oracles_order_on_the_same_generators():
    m = DatasetMapperUnsupervised(_cfg(), np_rng=np.random.RandomState(11), torch_generator=torch.Generator().manual_seed(11), py_rng=random.Random(11))
    np_rng, gen, py = np.random.RandomState(11), torch.Generator().manual_seed(11), random.Random(11)
    names = set()
    for i in range(40):
        h, w = (375, 500) if i % 2 else (1024, 2048)
        got = m.draw_params(h, w)
        ref = A.draw_view_params(h, w, (600,), 1333, "choice", 0.5, np_rng, gen, py)
        assert got["size"] == ref["size"] and got["flip"] == ref["flip"]
        assert got["strong_ops"] == ref["strong_ops"]
        names |= {n for n, _ in got["strong_ops"]}
    assert names == {"brightness", "contrast", "saturation", "hue", "grayscale", "blur", "solarize"}


def test_range_sampling_and_size_rule():
    m = DatasetMapperUnsupervised(_cfg(**{"INPUT.MIN_SIZE_TRAIN": (480, 800), "INPUT.MIN_SIZE_TRAIN_SAMPLING": "range"}), np_rng=np.random.RandomState(0),
                                  torch_generator=torch.Generator().manual_seed(0), py_rng=random.Random(0))
    sizes = {m.draw_params(375, 500)["size"][0] for _ in range(200)}
    assert min(sizes) >= 480 and max(sizes) <= 800 and len(sizes) > 50
    assert shortest_edge_size(375, 500, 600, 1333) == (600, 800) and shortest_edge_size(300, 1000, 600, 1333) == (400, 1333)


def test_annotations_follow_resize_and_flip():
    # 100x200 image -> 300x600 (scale 3); box (10, 20, 50, 60) -> (30, 60, 150, 180); flipped: x -> 600 - x: (450, 60, 570, 180)
    annos = [{"bbox": [10.0, 20.0, 50.0, 60.0], "category_id": 3}, {"bbox": [0.0, 0.0, 0.0, 5.0], "category_id": 1},
             {"bbox": [150.0, 90.0, 260.0, 140.0], "category_id": 2}, {"bbox": [1.0, 1.0, 9.0, 9.0], "category_id": 0, "iscrowd": 1}]
    inst = DatasetMapperUnsupervised.transform_annotations(annos, 100, 200, {"size": (300, 600), "flip": False})
    assert inst.gt_boxes.tensor.tolist() == [[30.0, 60.0, 150.0, 180.0], [450.0, 270.0, 600.0, 300.0]] and inst.gt_classes.tolist() == [3, 2]
    inst = DatasetMapperUnsupervised.transform_annotations(annos, 100, 200, {"size": (300, 600), "flip": True})
    assert inst.gt_boxes.tensor.tolist() == [[450.0, 60.0, 570.0, 180.0], [0.0, 270.0, 150.0, 300.0]]
    assert inst.image_size == (300, 600)


def test_two_crop_batches_by_aspect_ratio_group():
    wide = lambda i: ({"width": 20, "height": 10, "id": i}, {"width": 20, "height": 10, "id": -i})
    tall = lambda i: ({"width": 10, "height": 20, "id": i}, {"width": 10, "height": 20, "id": -i})
    stream = [wide(1), tall(2), wide(3), tall(4), tall(5), wide(6), wide(7)]
    out = list(AspectRatioGroupedDatasetTwoCrop(stream, 2))
    assert [[d["id"] for d in s] for s, _ in out] == [[1, 3], [2, 4], [6, 7]]
    assert [[d["id"] for d in w] for _, w in out] == [[-1, -3], [-2, -4], [-6, -7]]


def test_training_sampler_shards_one_shuffled_stream():
    from coin_amd.data import TrainingSampler
    import itertools

    full = list(itertools.islice(iter(TrainingSampler(5, seed=3)), 20))
    assert sorted(full[:5]) == sorted(full[5:10]) == [0, 1, 2, 3, 4] and full[:5] != full[5:10]       # permutations, reshuffled each epoch
    r0 = list(itertools.islice(iter(TrainingSampler(5, seed=3, rank=0, world_size=2)), 10))
    r1 = list(itertools.islice(iter(TrainingSampler(5, seed=3, rank=1, world_size=2)), 10))
    assert r0 == full[0::2] and r1 == full[1::2]


def test_loader_end_to_end_from_files_with_decode_threads(tmp_path):
    """VOC-style files -> dataset dicts -> sampler -> mapper (kernels shimmed by the oracle on the CPU) -> two-crop batches; the result does
    not depend on the number of decode threads."""
    from PIL import Image

    from cpu_shim import cpu_kernels
    from coin_amd.data import build_detection_unsupervised_train_loader

    dicts = []
    for i, (h, w) in enumerate([(60, 90), (90, 60), (64, 96), (50, 100), (100, 50)]):
        a = np.random.default_rng(i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(a, "RGB").save(tmp_path / f"{i}.png")
        dicts.append({"file_name": str(tmp_path / f"{i}.png"), "image_id": str(i), "height": h, "width": w})
    outs = []
    for workers in (0, 3):
        cfg = _cfg(**{"INPUT.MIN_SIZE_TRAIN": (48,), "INPUT.MAX_SIZE_TRAIN": 80, "SOLVER.IMG_PER_BATCH_UNLABEL": 2, "DATALOADER.NUM_WORKERS": workers, "SEED": 7})
        m = DatasetMapperUnsupervised(cfg, np_rng=np.random.RandomState(2), torch_generator=torch.Generator().manual_seed(2), py_rng=random.Random(2))
        with cpu_kernels():
            it = iter(build_detection_unsupervised_train_loader(cfg, dicts, mapper=m))
            batches = [next(it) for _ in range(4)]
        outs.append(batches)
        for strong, weak in batches:
            assert len(strong) == len(weak) == 2
            assert len({d["width"] > d["height"] for d in strong}) == 1                      # one aspect-ratio group per batch
            for s, w_ in zip(strong, weak):
                assert s["image_id"] == w_["image_id"] and s["image"].shape == w_["image"].shape and s["image"].dtype == torch.uint8
                assert min(s["image"].shape[1:]) <= 48 and max(s["image"].shape[1:]) <= 80
    for (s0, w0), (s1, w1) in zip(*outs):
        assert [d["image_id"] for d in s0] == [d["image_id"] for d in s1]
        assert all(torch.equal(a["image"], b["image"]) for a, b in zip(s0 + w0, s1 + w1))


def test_test_loader_maps_every_image_once_in_shards(tmp_path):
    from PIL import Image

    from cpu_shim import cpu_kernels
    from coin_amd.data import InferenceSampler, build_detection_test_loader
    from oracle import augment as A

    assert [list(InferenceSampler(7, r, 3)) for r in range(3)] == [[0, 1, 2], [3, 4], [5, 6]]
    dicts, imgs = [], []
    for i, (h, w) in enumerate([(60, 90), (90, 60), (64, 96)]):
        a = np.random.default_rng(20 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(a, "RGB").save(tmp_path / f"{i}.png")
        imgs.append(a)
        dicts.append({"file_name": str(tmp_path / f"{i}.png"), "image_id": str(i), "height": h, "width": w, "annotations": [{"bbox": [1, 2, 3, 4], "category_id": 0}]})
    cfg = _cfg(**{"INPUT.MIN_SIZE_TEST": 48, "INPUT.MAX_SIZE_TEST": 64})
    with cpu_kernels():
        got = [b for r in range(2) for b in build_detection_test_loader(cfg, dicts, rank=r, world_size=2)]
    assert [b[0]["image_id"] for b in got] == ["0", "1", "2"] and all(len(b) == 1 and "annotations" not in b[0] for b in got)
    for b, a in zip(got, imgs):
        oh, ow = A.shortest_edge_size(a.shape[0], a.shape[1], 48, 64)
        assert np.array_equal(b[0]["image"].numpy(), A.resize_bilinear(a, oh, ow).transpose(2, 0, 1)) and b[0]["random_flip"] == "no"


def test_catalog_names_and_errors(tmp_path):
    import pytest

    from coin_amd.data.catalog import get_detection_dataset_dicts, thing_classes

    assert thing_classes("foggytrain_0.02") == ("truck", "car", "rider", "person", "train", "motorcycle", "bicycle", "bus")
    assert thing_classes("BDD100Ktrain")[0] == "person" and len(thing_classes("cliparttrain")) == 20
    with pytest.raises(KeyError, match="not registered"):
        get_detection_dataset_dicts(["no_such_set"], root=str(tmp_path))


def test_read_image_formats(tmp_path):
    from PIL import Image

    from coin_amd.data.dataset_mapper import read_image

    a = np.random.default_rng(9).integers(0, 256, (12, 17, 3), dtype=np.uint8)
    Image.fromarray(a, "RGB").save(tmp_path / "a.png")
    assert np.array_equal(read_image(str(tmp_path / "a.png"), "RGB"), a)
    assert np.array_equal(read_image(str(tmp_path / "a.png"), "BGR"), a[:, :, ::-1])
    Image.fromarray(a[:, :, 0], "L").save(tmp_path / "g.png")                      # a grey file is expanded to three channels
    g = read_image(str(tmp_path / "g.png"), "RGB")
    assert g.shape == (12, 17, 3) and np.array_equal(g[..., 0], a[:, :, 0]) and np.array_equal(g[..., 0], g[..., 2])
