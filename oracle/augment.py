"""TEST INFRASTRUCTURE (oracle): CPU restatement of the image operations behind COIN's two-view input pipeline
(SURVEY.md §8(f)-3).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference call sites:
  * weak view: detectron2 `ResizeShortestEdge` + `RandomFlip` (coin/data/dataset_mapper.py:326,386-393 via
    `utils.build_augmentation`): `PIL.Image.resize(..., BILINEAR)` on the uint8 array, then `np.flip(img, axis=1)`;
  * strong view on top of the weak one (coin/data/detection_utils.py:22-45, dataset_mapper.py:433-440): torchvision
    `ColorJitter(0.4, 0.4, 0.4, 0.1)` (p = 0.8), `RandomGrayscale` (p = 0.2), `GaussianBlur([0.1, 2.0])` (p = 0.5;
    coin/data/transforms/augmentation_impl.py:64-79 = `PIL.ImageFilter.GaussianBlur`), `Solarize(0.5)` (p = 0.2; :82-88 =
    `PIL.ImageOps.solarize(img, 128)`), all on PIL images.

The arithmetic lives in third-party code that is absent from /root/reference: Pillow (C library; Pillow 12.2 is installed in this
image and is what the restatement is PINNED against, bit for bit: tests/test_oracle_augment.py) and torchvision 0.10.1
(`docs/Environment.md`; NOT installed).  torchvision's PIL code paths are thin wrappers that are restated here from their published
behaviour -- `adjust_brightness/contrast/saturation` = `ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f)`, `adjust_hue` =
HSV round trip with the H channel shifted by `uint8(f * 255)`, `rgb_to_grayscale(3 channels)` = `convert("L")` replicated -- and the
ORDER of random draws follows `ColorJitter.get_params` / `RandomApply.forward` as published: **parity unpinned** for that layer
(the Pillow layer under it is pinned).

Every function takes / returns uint8 arrays [H, W, 3] (RGB, interleaved: the PIL / numpy layout the reference uses).
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Tuple

import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------ Pillow: Image.blend / ImageEnhance
def blend(degenerate: np.ndarray, img: np.ndarray, alpha: float) -> np.ndarray:
    """`Image.blend(degenerate, img, alpha)` (Pillow src/libImaging/Blend.c): out = in1 + alpha * (in2 - in1) in C `float`, truncated
    to uint8; outside [0, 1] the result is clipped to [0, 255] first."""
    a = f32(alpha)
    d = (img.astype(np.int32) - degenerate.astype(np.int32)).astype(f32)
    r = degenerate.astype(f32) + a * d
    if 0.0 <= alpha <= 1.0:
        return r.astype(np.uint8)
    return np.where(r <= 0, 0, np.where(r >= 255, 255, r)).astype(np.uint8)


def to_gray(img: np.ndarray) -> np.ndarray:
    """`convert("L")` (Convert.c, ITU-R 601-2 in 16.16 fixed point): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16 -> [H, W]."""
    r, g, b = (img[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def adjust_brightness(img, factor):  # ImageEnhance.Brightness: degenerate = black
    return blend(np.zeros_like(img), img, factor)


def gray_mean(img) -> int:  # ImageEnhance.Contrast: int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5)
    g = to_gray(img)
    return int(int(g.astype(np.uint64).sum()) / g.size + 0.5)


def adjust_contrast(img, factor):
    return blend(np.full_like(img, gray_mean(img)), img, factor)


def adjust_saturation(img, factor):  # ImageEnhance.Color: degenerate = convert("L").convert("RGB")
    return blend(np.repeat(to_gray(img)[..., None], 3, axis=2), img, factor)


def rgb_to_grayscale3(img):  # torchvision F_pil.to_grayscale(img, 3): convert("L") replicated
    return np.repeat(to_gray(img)[..., None], 3, axis=2)


def solarize(img, threshold: int = 128):  # ImageOps.solarize: lut[i] = i if i < threshold else 255 - i
    return np.where(img < threshold, img, 255 - img).astype(np.uint8)


# ------------------------------------------------------------------------------------------ Pillow: RGB <-> HSV (Convert.c)
def rgb_to_hsv(img: np.ndarray) -> np.ndarray:
    """rgb2hsv_row: the channel ratios are C `float`; the branch expression is evaluated in double (the literals 2.0 / 4.0) and
    stored to a `float`, `fmod(h / 6.0 + 1.0, 1.0)` likewise, then `(int)(h * 255.0)` -- the combination that reproduces Pillow bit for
    bit on every pixel tried (262 144 random ones in the test)."""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    gray = maxc == minc
    with np.errstate(all="ignore"):
        s = cr / maxc.astype(f32)
        rc, gc, bc = ((maxc - c).astype(f32) / cr for c in (r, g, b))
        h = np.where(r == maxc, (bc - gc).astype(np.float64),
                     np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64), 4.0 + gc.astype(np.float64) - rc.astype(np.float64)))
        h = h.astype(f32).astype(np.float64)
        h = np.fmod(h / 6.0 + 1.0, 1.0).astype(f32).astype(np.float64)
        uh = np.clip(np.nan_to_num(h * 255.0).astype(np.int64), 0, 255)
        us = np.clip(np.nan_to_num(s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    uh, us = np.where(gray, 0, uh), np.where(gray, 0, us)
    return np.stack([uh, us, maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    """hsv2rgb_row: i = floor(h * 6 / 255), f = frac, p / q / t rounded to nearest and clipped."""
    h, s, v = (hsv[..., i].astype(np.float64) for i in range(3))
    hh = h * 6.0 / 255.0
    i = np.floor(hh)
    f, fs = hh - i, s / 255.0
    rnd = lambda x: np.clip(np.floor(x + 0.5), 0, 255)
    p, q, t = rnd(v * (1.0 - fs)), rnd(v * (1.0 - fs * f)), rnd(v * (1.0 - fs * (1.0 - f)))
    i = i.astype(np.int64) % 6
    out = np.stack([np.choose(i, [v, q, p, p, t, v]), np.choose(i, [t, v, v, q, p, p]), np.choose(i, [p, p, t, v, v, q])], axis=-1)
    return np.where((hsv[..., 1] == 0)[..., None], hsv[..., 2:3].astype(np.float64), out).astype(np.uint8)


def hue_shift_of(factor: float) -> int:
    """torchvision F_pil.adjust_hue: `np_h += np.uint8(hue_factor * 255)` with wrap-around: truncation toward zero, modulo 256."""
    return int(factor * 255) % 256


def adjust_hue(img, factor):
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_of(factor)) % 256
    return hsv_to_rgb(hsv)


# ------------------------------------------------------------------------------------------ Pillow: GaussianBlur (BoxBlur.c)
def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    """_gaussian_blur_radius: the box radius whose `passes`-fold box blur has the Gaussian's variance (float arithmetic, sqrt / floor
    in double)."""
    radius = f32(radius)
    sigma2 = f32(f32(radius * radius) / f32(passes))
    big_l = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(big_l) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    return f32(l + a)


def box_weights(fr: np.float32) -> Tuple[int, int, int]:
    """ImagingHorizontalBoxBlur: (integer radius, weight of a full pixel, weight of the two fractional edge pixels), 8.24 fixed point."""
    radius = int(fr)
    ww = int(f32(f32(1 << 24) / f32(f32(fr * f32(2)) + f32(1))))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def box_blur_h(img: np.ndarray, fr: np.float32) -> np.ndarray:
    """One horizontal extended-box-blur pass (ImagingLineBoxBlur32 in closed form): out[x] = (ww * sum_{|d| <= r} in[clamp(x + d)] +
    fw * (in[clamp(x - r - 1)] + in[clamp(x + r + 1)]) + 2^23) >> 24 in uint32 arithmetic."""
    h, w, _ = img.shape
    radius, ww, fw = box_weights(fr)
    x = np.arange(w)
    src = img.astype(np.uint64)
    acc = np.zeros_like(src)
    for d in range(-radius, radius + 1):
        acc += src[:, np.clip(x + d, 0, w - 1)]
    far = src[:, np.clip(x - radius - 1, 0, w - 1)] + src[:, np.clip(x + radius + 1, 0, w - 1)]
    bulk = (acc * np.uint64(ww) + far * np.uint64(fw)) & np.uint64(0xFFFFFFFF)
    return (((bulk + np.uint64(1 << 23)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)).astype(np.uint8)


def gaussian_blur(img: np.ndarray, radius: float, passes: int = 3) -> np.ndarray:
    """ImageFilter.GaussianBlur(radius): `passes` horizontal box passes, then `passes` vertical ones."""
    fr = gaussian_box_radius(radius, passes)
    out = img
    if fr != 0:
        for _ in range(passes):
            out = box_blur_h(out, fr)
        t = np.ascontiguousarray(out.transpose(1, 0, 2))
        for _ in range(passes):
            t = box_blur_h(t, fr)
        out = np.ascontiguousarray(t.transpose(1, 0, 2))
    return out


# ------------------------------------------------------------------------------------------ Pillow: resize(BILINEAR) (Resample.c)
PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, box = the whole axis.
    -> (kk [out, ksize] int, xmin [out], xmax [out])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    lo, cnt = np.zeros(out_size, dtype=np.int64), np.zeros(out_size, dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0.0 else v
            w.append(1.0 - v if v < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        lo[xx], cnt[xx] = xmin, xmax
    return kk, lo, cnt


def resample_h(img: np.ndarray, out_w: int) -> np.ndarray:
    """ImagingResampleHorizontal_8bpc: 22-bit fixed-point taps, rounding constant 2^21, clip to [0, 255]."""
    h, w, c = img.shape
    kk, lo, cnt = resample_coeffs(w, out_w)
    src = img.astype(np.int64)
    out = np.zeros((h, out_w, c), dtype=np.uint8)
    for xx in range(out_w):
        acc = np.full((h, c), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(int(cnt[xx])):
            acc += src[:, lo[xx] + x] * kk[xx, x]
        out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`Image.resize((out_w, out_h), BILINEAR)`: horizontal pass, then vertical pass on the uint8 intermediate."""
    h, w, _ = img.shape
    t = resample_h(img, out_w) if out_w != w else img
    if out_h != h:
        t = np.ascontiguousarray(resample_h(np.ascontiguousarray(t.transpose(1, 0, 2)), out_h).transpose(1, 0, 2))
    return t


def hflip(img):  # detectron2 HFlipTransform.apply_image: np.flip(img, axis=1)
    return np.ascontiguousarray(img[:, ::-1])


# ------------------------------------------------------------------------------------------ parameter draws
def shortest_edge_size(h: int, w: int, size: int, max_size: int) -> Tuple[int, int]:
    """detectron2 ResizeShortestEdge.get_output_shape (v0.5: inside get_transform)."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def draw_view_params(h: int, w: int, min_sizes, max_size: int, sample_style: str, flip_prob: float, np_rng, torch_gen, py_rng: random.Random) -> Dict:
    """All random decisions of one DatasetMapperUnsupervised call, in the reference's order of draws:
    numpy (detectron2 augmentations): short-edge size, flip;  torch (torchvision transforms): RandomApply(ColorJitter) -> get_params
    (randperm(4), then brightness / contrast / saturation / hue factors), RandomGrayscale, RandomApply(blur), RandomApply(solarize);
    python `random`: the blur radius (augmentation_impl.py:77), drawn only when the blur is applied."""
    import torch

    size = int(np_rng.randint(min_sizes[0], min_sizes[1] + 1)) if sample_style == "range" else int(np_rng.choice(min_sizes))
    out = {"size": shortest_edge_size(h, w, size, max_size), "flip": bool(np_rng.uniform() < flip_prob)}
    u = lambda: float(torch.rand(1, generator=torch_gen))
    ops: List[Tuple[str, float]] = []
    if not (0.8 < u()):  # RandomApply.forward: `if self.p < torch.rand(1): return img`
        order = torch.randperm(4, generator=torch_gen).tolist()
        fac = [float(torch.empty(1).uniform_(lo, hi, generator=torch_gen)) for lo, hi in ((0.6, 1.4), (0.6, 1.4), (0.6, 1.4), (-0.1, 0.1))]
        names = ("brightness", "contrast", "saturation", "hue")
        ops += [(names[i], fac[i]) for i in order]
    if u() < 0.2:        # RandomGrayscale.forward: `if torch.rand(1) < self.p`
        ops.append(("grayscale", 0.0))
    if not (0.5 < u()):
        ops.append(("blur", py_rng.uniform(0.1, 2.0)))
    if not (0.2 < u()):
        ops.append(("solarize", 128.0))
    out["strong_ops"] = ops
    return out


def apply_strong(img: np.ndarray, ops) -> np.ndarray:
    fn = {"brightness": adjust_brightness, "contrast": adjust_contrast, "saturation": adjust_saturation, "hue": adjust_hue,
          "grayscale": lambda im, _: rgb_to_grayscale3(im), "blur": gaussian_blur, "solarize": lambda im, t: solarize(im, int(t))}
    for name, p in ops:
        img = fn[name](img, p)
    return img


def two_views(img: np.ndarray, params: Dict) -> Tuple[np.ndarray, np.ndarray]:
    """(strong, weak) uint8 [3, h, w] as dataset_mapper.py:433-447 hands them on (CHW)."""
    oh, ow = params["size"]
    weak = resize_bilinear(img, oh, ow)
    if params["flip"]:
        weak = hflip(weak)
    strong = apply_strong(weak, params["strong_ops"])
    return np.ascontiguousarray(strong.transpose(2, 0, 1)), np.ascontiguousarray(weak.transpose(2, 0, 1))
