"""oracle/augment.py against Pillow itself (the C library the reference's input pipeline calls through torchvision / detectron2;
Pillow is installed in this image, torchvision is not): every restated operation must reproduce Pillow BIT FOR BIT."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as A

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter, ImageOps  # noqa: E402


def _img(seed, h=96, w=131):
    a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[0, :8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 10, 9], [200, 100, 100], [1, 2, 3]]
    return a


P = lambda a: Image.fromarray(a, "RGB")
FACTORS = [0.6, 0.73, 0.999, 1.0, 1.0001, 1.27, 1.4]


@pytest.mark.parametrize("f", FACTORS)
def test_enhance_ops_equal_pillow(f):
    a = _img(1)
    assert np.array_equal(A.adjust_brightness(a, f), np.array(ImageEnhance.Brightness(P(a)).enhance(f)))
    assert np.array_equal(A.adjust_contrast(a, f), np.array(ImageEnhance.Contrast(P(a)).enhance(f)))
    assert np.array_equal(A.adjust_saturation(a, f), np.array(ImageEnhance.Color(P(a)).enhance(f)))


def test_gray_solarize_equal_pillow():
    a = _img(2)
    assert np.array_equal(A.to_gray(a), np.array(P(a).convert("L")))
    assert np.array_equal(A.rgb_to_grayscale3(a), np.array(P(a).convert("L").convert("RGB")))
    assert np.array_equal(A.solarize(a, 128), np.array(ImageOps.solarize(P(a), 128)))


def test_hsv_round_trip_equals_pillow_on_262144_pixels():
    a = _img(3, 512, 512)
    hsv = np.array(P(a).convert("HSV"))
    assert np.array_equal(A.rgb_to_hsv(a), hsv)
    assert np.array_equal(A.hsv_to_rgb(hsv), np.array(Image.fromarray(hsv, "HSV").convert("RGB")))


@pytest.mark.parametrize("f", [-0.1, -0.0371, 0.0, 0.02, 0.1])
def test_adjust_hue_equals_the_published_torchvision_recipe_on_pillow(f):
    a = _img(4)
    h, s, v = P(a).convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = ((np_h.astype(np.int32) + int(f * 255)) % 256).astype(np.uint8)   # `np_h += np.uint8(hue_factor * 255)` with wrap-around
    ref = np.array(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))
    assert np.array_equal(A.adjust_hue(a, f), ref)


@pytest.mark.parametrize("shape", [(40, 60), (7, 9), (128, 200)])
def test_gaussian_blur_equals_pillow(shape):
    a = _img(5, *shape)
    for r in [0.1, 0.25, 0.5, 0.77, 1.0, 1.3, 1.999, 2.0]:
        assert np.array_equal(A.gaussian_blur(a, r), np.array(P(a).filter(ImageFilter.GaussianBlur(radius=r)))), r


@pytest.mark.parametrize("h,w,oh,ow", [(60, 97, 40, 65), (50, 80, 75, 120), (33, 47, 33, 20), (64, 128, 100, 128), (300, 600, 167, 333),
                                       (375, 500, 600, 800)])
def test_resize_bilinear_equals_pillow(h, w, oh, ow):
    a = _img(6, h, w)
    assert np.array_equal(A.resize_bilinear(a, oh, ow), np.array(P(a).resize((ow, oh), Image.BILINEAR)))


def test_shortest_edge_sizes_hand_values():
    # 375x500 VOC image, short edge 600, cap 1333: scale 1.6 -> 600 x 800;  1024x2048 Cityscapes: 600 x 1200;
    # a 300x1000 panorama: 600 x 2000 exceeds the cap -> scale 1333/2000: 399.9 -> 400 x 1333
    assert A.shortest_edge_size(375, 500, 600, 1333) == (600, 800)
    assert A.shortest_edge_size(1024, 2048, 600, 1333) == (600, 1200)
    assert A.shortest_edge_size(300, 1000, 600, 1333) == (400, 1333)


def test_two_views_chain_equals_the_same_chain_on_pillow():
    """The whole chain (resize -> flip -> strong ops in the drawn order) against the same operations executed by Pillow."""
    a = _img(7, 120, 180)
    np_rng, gen, py = np.random.RandomState(3), torch.Generator().manual_seed(3), random.Random(3)
    seen = set()
    for _ in range(12):
        prm = A.draw_view_params(120, 180, (96,), 160, "choice", 0.5, np_rng, gen, py)
        strong, weak = A.two_views(a, prm)
        im = P(a).resize((prm["size"][1], prm["size"][0]), Image.BILINEAR)
        if prm["flip"]:
            im = Image.fromarray(np.ascontiguousarray(np.flip(np.asarray(im), axis=1)), "RGB")
        assert np.array_equal(weak, np.asarray(im).transpose(2, 0, 1))
        for name, p in prm["strong_ops"]:
            seen.add(name)
            if name == "brightness":
                im = ImageEnhance.Brightness(im).enhance(p)
            elif name == "contrast":
                im = ImageEnhance.Contrast(im).enhance(p)
            elif name == "saturation":
                im = ImageEnhance.Color(im).enhance(p)
            elif name == "hue":
                h, s, v = im.convert("HSV").split()
                nh = ((np.array(h, dtype=np.uint8).astype(np.int32) + int(p * 255)) % 256).astype(np.uint8)
                im = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
            elif name == "grayscale":
                im = im.convert("L").convert("RGB")
            elif name == "blur":
                im = im.filter(ImageFilter.GaussianBlur(radius=p))
            elif name == "solarize":
                im = ImageOps.solarize(im, int(p))
        assert np.array_equal(strong, np.asarray(im).transpose(2, 0, 1))
    assert {"brightness", "contrast", "saturation", "hue", "blur"} <= seen
