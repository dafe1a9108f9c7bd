"""Parity of every C-ABI kernel against the CPU oracle (run with ``-m gpu`` on the MI355X box).

fp32 paths: tolerance 1e-4 (north_star); integer/index outputs exact; bf16 paths: bf16 rounding
tolerance stated per test.  Inputs are seeded; sizes are what the oracle finishes in seconds; the
full BASELINE sizes are covered through size-independent properties (constant-map identity,
linearity, adjointness <Ax,y> = <x,A^T y>).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from coin_amd import kernels

    return kernels


def dev(t):
    return t.to("cuda")


def make_rois(n_img, r, h_img, w_img, g, extremes=True):
    bw = torch.rand(r, generator=g) * (w_img * 0.6) + 8
    bh = torch.rand(r, generator=g) * (h_img * 0.6) + 8
    x0 = torch.rand(r, generator=g) * (w_img - 8)
    y0 = torch.rand(r, generator=g) * (h_img - 8)
    rois = torch.stack([torch.randint(0, n_img, (r,), generator=g).float(), x0, y0, x0 + bw, y0 + bh], dim=1)
    if extremes and r >= 8:
        rois[0, 1:] = torch.tensor([-40.0, -30.0, 20.0, 25.0])                 # partly outside (top-left)
        rois[1, 1:] = torch.tensor([w_img - 10.0, h_img - 10.0, w_img + 60.0, h_img + 50.0])  # partly outside
        rois[2, 1:] = torch.tensor([0.0, 0.0, float(w_img), float(h_img)])      # whole image
        rois[3, 1:] = torch.tensor([50.0, 40.0, 50.5, 40.25])                   # sub-pixel box
        rois[4, 1:] = torch.tensor([30.0, 30.0, 20.0, 20.0])                    # inverted (negative size)
        rois[5, 1:] = torch.tensor([-500.0, -500.0, -400.0, -300.0])            # entirely outside
        rois[6, 1:] = torch.tensor([-300.0, -200.0, w_img + 300.0, h_img + 200.0])  # much larger than the image
    return rois


# ------------------------------------------------------------------------------------------ RoIAlign
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("sr,aligned", [(0, True), (2, True), (0, False)])
def test_roi_align_fwd_bwd_f32_vs_loops(K, layout, sr, aligned):
    from oracle import d2

    g = torch.Generator().manual_seed(5)
    n, c, h, w = 2, 8, 13, 17
    feat = torch.randn(n, c, h, w, generator=g)
    rois = make_rois(n, 12, h * 16, w * 16, g)
    ref = d2.roi_align_forward_np(feat.numpy(), rois.numpy(), (7, 7), 1 / 16.0, sr, aligned)
    lay = K.COIN_NHWC if layout == "nhwc" else K.COIN_NCHW
    fd = dev(feat.permute(0, 2, 3, 1).contiguous() if layout == "nhwc" else feat)
    out = K.roi_align_fwd(fd, dev(rois), (7, 7), 1 / 16.0, sr, aligned, lay)
    out = out.permute(0, 3, 1, 2) if layout == "nhwc" else out
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    go = torch.randn(ref.shape, generator=g)
    gref = d2.roi_align_backward_np(go.numpy(), rois.numpy(), (n, c, h, w), 1 / 16.0, sr, aligned)
    gd = dev(go.permute(0, 2, 3, 1).contiguous() if layout == "nhwc" else go)
    shape = (n, h, w, c) if layout == "nhwc" else (n, c, h, w)
    gin = K.roi_align_bwd(gd, dev(rois), shape, 1 / 16.0, sr, aligned, lay)
    gin = gin.permute(0, 3, 1, 2) if layout == "nhwc" else gin
    np.testing.assert_allclose(gin.cpu().numpy(), gref, rtol=1e-4, atol=2e-5)


def test_roi_align_empty_and_errors(K):
    from coin_amd._lib import CoinHipError

    feat = torch.zeros(1, 4, 4, 8, device="cuda")
    out = K.roi_align_fwd(feat, torch.zeros(0, 5, device="cuda"), (14, 14), 1 / 16.0)
    assert out.shape == (0, 14, 14, 8)
    gin = K.roi_align_bwd(out, torch.zeros(0, 5, device="cuda"), (1, 4, 4, 8), 1 / 16.0)
    assert float(gin.abs().sum()) == 0.0
    with pytest.raises(CoinHipError):  # C not a multiple of the 16-byte vector
        K.roi_align_fwd(torch.zeros(1, 4, 4, 6, device="cuda"), torch.zeros(1, 5, device="cuda"), (2, 2), 1.0)
    with pytest.raises(CoinHipError):  # CPU tensor: no fallback
        K.roi_align_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(1, 5), (2, 2), 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_res4_shape_vs_oracle(K, dtype):
    """res4-shaped map (C=256 slice of channels, 50x83), 14x14, 1/16: vectorised oracle in fp64."""
    from oracle import d2

    g = torch.Generator().manual_seed(6)
    n, c, h, w = 2, 256, 50, 83
    feat = torch.randn(n, h, w, c, generator=g)
    rois = make_rois(n, 96, 800, 1333, g)
    fq = feat.to(dtype)
    ref = d2.roi_align_torch(fq.double().permute(0, 3, 1, 2), rois.double(), (14, 14), 1 / 16.0, 0, True)
    out = K.roi_align_fwd(dev(fq), dev(rois), (14, 14), 1 / 16.0).float().permute(0, 3, 1, 2).cpu()
    tol = 1e-4 if dtype == torch.float32 else 2e-2  # bf16: one rounding of the output (2^-8 relative)
    torch.testing.assert_close(out.double(), ref, rtol=tol, atol=tol)
    go = torch.randn(96, 14, 14, c, generator=g).to(dtype)
    fd = fq.double().permute(0, 3, 1, 2).requires_grad_(True)
    d2.roi_align_torch(fd, rois.double(), (14, 14), 1 / 16.0, 0, True).backward(go.double().permute(0, 3, 1, 2))
    gin = K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0).permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(gin.double(), fd.grad, rtol=2e-4, atol=2e-4)


def test_roi_align_bwd_many_rois_large_bins_and_reproducibility(K):
    """More RoIs than one LDS list chunk (4096) on the tiled path; 20x20 bins take the zero-fill + atomics fallback;
    the tiled path sums RoIs in index order, so two runs agree bit for bit."""
    from oracle import d2

    g = torch.Generator().manual_seed(8)
    n, c, h, w, r = 2, 8, 21, 19, 4500
    rois = make_rois(n, r, h * 16, w * 16, g)
    for ph in (14, 20):
        go = torch.randn(r, ph, ph, c, generator=g)
        fd = torch.zeros(n, c, h, w, dtype=torch.float64, requires_grad=True)
        d2.roi_align_torch(fd, rois.double(), (ph, ph), 1 / 16.0, 0, True).backward(go.double().permute(0, 3, 1, 2))
        stale = torch.full((n, h, w, c), 7.0, device="cuda")  # the output buffer is overwritten, not accumulated into
        gin = K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0, grad_feat=stale)
        torch.testing.assert_close(gin.permute(0, 3, 1, 2).cpu().double(), fd.grad, rtol=2e-4, atol=2e-4)
        if ph == 14:
            again = K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0)
            assert torch.equal(gin, again)


@pytest.mark.parametrize("c,bins", [(520, 14), (1024, 7), (8, 14)])
def test_roi_align_bwd_bf16_tiny_and_huge_boxes_ragged_channel_slab(K, c, bins):
    """The bf16 backward combines ALL bin rows that put weight on a map row before it fans out to the tile columns, two bin rows x
    seven bin columns per round: boxes smaller than a map cell (all 14 bin rows land on one or two map rows -> seven rounds per bin
    column), boxes as large as the image, boxes hanging over the border; C = 520 leaves the second 512-channel slab with 8 live
    channels.  Against the fp64 oracle on the same bf16 values; twice, bit for bit."""
    from oracle import d2

    g = torch.Generator().manual_seed(61)
    n, h, w = 2, 23, 37
    per = 40
    tiny = torch.rand(per, 2, generator=g) * torch.tensor([w * 16.0 - 20, h * 16.0 - 20])
    tiny = torch.cat([tiny, tiny + torch.rand(per, 2, generator=g) * 14 + 1], 1)                      # 1-15 px: under one map cell
    huge = torch.tensor([[0.0, 0.0, w * 16.0, h * 16.0], [-40.0, -30.0, w * 16.0 + 50, h * 16.0 + 20], [100.0, -64.0, 180.0, h * 16.0 + 64]])
    mid = make_rois(1, 30, h * 16, w * 16, g)[:, 1:]
    boxes = torch.cat([tiny, huge, mid])
    rois = torch.cat([torch.cat([torch.full((len(boxes), 1), float(i)), boxes], 1) for i in range(n)])
    r = len(rois)
    go = torch.randn(r, bins, bins, c, generator=g).to(torch.bfloat16)
    fd = torch.zeros(n, c, h, w, dtype=torch.float64, requires_grad=True)
    d2.roi_align_torch(fd, rois.double(), (bins, bins), 1 / 16.0, 0, True).backward(go.double().permute(0, 3, 1, 2))
    gin = K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0)
    torch.testing.assert_close(gin.permute(0, 3, 1, 2).cpu().double(), fd.grad, rtol=2e-4, atol=2e-4)
    assert torch.equal(gin, K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0))
    # the f32 kernel on the same values (another summation order): agreement at f32 rounding
    g32 = K.roi_align_bwd(dev(go.float()), dev(rois), (n, h, w, c), 1 / 16.0)
    torch.testing.assert_close(gin, g32, rtol=1e-4, atol=1e-4)


def test_roi_align_full_size_properties(K):
    """BASELINE size (4 views x 512 RoIs, C=1024, bf16): constant map -> 1 inside the image; adjointness."""
    g = torch.Generator().manual_seed(7)
    n, c, h, w, r = 4, 1024, 50, 83, 2048
    rois = make_rois(n, r, 800, 1333, g, extremes=False)
    rois[:, 3] = rois[:, 3].clamp(max=1320.0)
    rois[:, 4] = rois[:, 4].clamp(max=790.0)
    rois[:, 1] = torch.minimum(rois[:, 1], rois[:, 3] - 16.0)
    rois[:, 2] = torch.minimum(rois[:, 2], rois[:, 4] - 16.0)
    ones = torch.ones(n, h, w, c, device="cuda", dtype=torch.bfloat16)
    out = K.roi_align_fwd(ones, dev(rois), (14, 14), 1 / 16.0)
    assert out.shape == (r, 14, 14, c)
    assert float((out.float() - 1).abs().max()) < 1e-2  # every sample lies inside -> weights sum to 1
    x = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16).cuda()
    y = torch.randn(r, 14, 14, c, generator=g).to(torch.bfloat16).cuda()
    ax = K.roi_align_fwd(x, dev(rois), (14, 14), 1 / 16.0)
    aty = K.roi_align_bwd(y, dev(rois), (n, h, w, c), 1 / 16.0)
    lhs = (ax.double() * y.double()).sum()
    rhs = (x.double() * aty.double()).sum()
    assert abs(float(lhs - rhs)) / abs(float(lhs)) < 5e-3  # bf16 rounding of A x
    # linearity: A(2x) = 2 A(x) exactly in bf16 (power-of-two scale)
    ax2 = K.roi_align_fwd(x * 2, dev(rois), (14, 14), 1 / 16.0)
    assert torch.equal(ax2, ax * 2)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("m,n,k", [(512, 1024, 2048), (200, 136, 64), (1, 4, 2048), (2048, 2048, 1024), (77, 1024, 512)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_nt_bf16(K, m, n, k, act):
    g = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, generator=g)
    ref = a.double() @ b.double().t() + bias.double()
    if act == 1:
        ref = F.leaky_relu(ref, 0.01)
    out = K.gemm_nt(dev(a), dev(b), dev(bias), act, 0.01, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=2e-3, atol=2e-3)  # fp32 accumulate of exact bf16 products
    out16 = K.gemm_nt(dev(a), dev(b), dev(bias), act, 0.01).cpu()
    torch.testing.assert_close(out16.double(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("m,n,k", [(130, 70, 48), (512, 256, 256), (3, 4, 16)])
def test_gemm_nt_f32_exact_path(K, m, n, k):
    g = torch.Generator().manual_seed(m * n)
    a, b, bias = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g), torch.randn(n, generator=g)
    ref = (a.double() @ b.double().t() + bias.double()).float()
    out = K.gemm_nt(dev(a), dev(b), dev(bias)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)  # fp32 FMA chain vs fp64 reference, |values| ~ sqrt(K)
    # asymmetric-B / A = I check of the MFMA C layout (cdna_hip_programming.md §3)
    eye = torch.eye(16, 16)
    basym = torch.arange(16 * 16, dtype=torch.float32).reshape(16, 16)
    out = K.gemm_nt(dev(eye), dev(basym)).cpu()
    assert torch.equal(out, basym.t())


@pytest.mark.parametrize("m,n,k", [(256, 128, 64), (1000, 200, 192), (4096, 512, 1024), (300, 2048, 512), (37, 8, 64)])
def test_conv_gemm_1x1_and_stats_vs_fp64(K, m, n, k):
    """coin_conv_gemm_bf16 mode 0 (1x1 convolution / linear) + the fused BatchNorm statistics of the stored outputs."""
    g = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * 0.1 + 0.02).to(torch.bfloat16)
    rows = m - m // 7
    out, part = K.conv_gemm(dev(a), dev(w), stats_rows=rows)
    ref = a.double() @ w.double().t()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()))  # one bf16 rounding of the output
    bn_rm, bn_rv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
    mean, rstd = K.conv_stats_finalize(part, m, n, rows, 1e-5, 0.1, bn_rm, bn_rv)
    y = out[:rows].double()                                  # statistics of the STORED values over the first `rows` rows
    mu, var = y.mean(0), y.var(0, unbiased=False)
    torch.testing.assert_close(mean.double(), mu, rtol=1e-5, atol=1e-5 * float(y.abs().max()))
    torch.testing.assert_close(rstd.double(), (var + 1e-5).rsqrt(), rtol=1e-4, atol=0)
    torch.testing.assert_close(bn_rm.double(), 0.1 * mu, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_rv.double(), 0.9 + 0.1 * y.var(0, unbiased=True), rtol=1e-4, atol=1e-6)
    out2, none = K.conv_gemm(dev(a), dev(w))
    assert none is None and torch.equal(out, out2)
    # R operand: C = bf16(bf16(A.B^T) + R), i.e. exactly what a bf16 product followed by a bf16 add stores
    r = (torch.randn(m, n, generator=g) * 2).to(torch.bfloat16)
    out3, _ = K.conv_gemm(dev(a), dev(w), residual=dev(r))
    assert torch.equal(out3, out2 + dev(r))


@pytest.mark.parametrize("nb,h,w,cin,cout", [(3, 7, 7, 64, 128), (2, 14, 14, 128, 64), (1, 5, 9, 192, 40), (5, 7, 7, 512, 512)])
def test_conv_gemm_3x3_vs_torch(K, nb, h, w, cin, cout):
    """coin_conv_gemm_bf16 mode 1 = F.conv2d(x, W, padding=1) on NHWC bf16 (fp64 reference on the same bf16 inputs), and the
    data-gradient as the same contraction with the weight re-laid [Cin][flipped tap][Cout]."""
    g = torch.Generator().manual_seed(nb * h + cin)
    x = torch.randn(nb, cin, h, w, generator=g).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(torch.bfloat16)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    xa = dev(x.permute(0, 2, 3, 1).contiguous().reshape(-1, cin))
    wk = dev(wt.permute(0, 2, 3, 1).contiguous().reshape(cout, 9 * cin))
    out, _ = K.conv_gemm(xa, wk, spatial=(h, w, cin))
    got = out.reshape(nb, h, w, cout).permute(0, 3, 1, 2).cpu().double()
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()))
    if cout % 64 == 0:  # dgrad: contraction over Cout must be a multiple of the K-step
        gy = torch.randn(nb, cout, h, w, generator=g).to(torch.bfloat16)
        gref = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=1)
        wd = dev(wt.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(cin, 9 * cout))
        gx, _ = K.conv_gemm(dev(gy.permute(0, 2, 3, 1).contiguous().reshape(-1, cout)), wd, spatial=(h, w, cout))
        torch.testing.assert_close(gx.reshape(nb, h, w, cin).permute(0, 3, 1, 2).cpu().double(), gref, rtol=1e-2, atol=1e-2 * float(gref.abs().max()))


@pytest.mark.parametrize("nb,h,w,cin,cout,ks", [(2048, 14, 14, 1024, 512, 1), (2048, 14, 14, 512, 512, 3), (2048, 7, 7, 512, 2048, 1),
                                                (2048, 7, 7, 2048, 512, 1), (2048, 7, 7, 512, 512, 3),
                                                # round 6: the backbone's maps of the benchmark (res4 50 x 83, res3 100 x 167, layer2's first block at
                                                # 200 x 333) and of the targetDET step (41 x 83): the launches of the captured backbone stretch --
                                                # most of them on the 128 x 128 small-map core (128-row statistics tiles), the long-K ones on the
                                                # persistent kernel
                                                (4, 50, 83, 1024, 256, 1), (4, 50, 83, 256, 256, 3), (4, 50, 83, 256, 1024, 1), (4, 50, 83, 512, 1024, 1),
                                                (4, 100, 167, 512, 128, 1), (4, 100, 167, 128, 128, 3), (4, 100, 167, 128, 512, 1), (4, 100, 167, 256, 256, 3),
                                                (4, 100, 167, 512, 256, 1), (2, 200, 333, 128, 128, 3), (2, 200, 333, 256, 128, 1), (3, 41, 83, 256, 256, 3),
                                                (3, 41, 83, 1024, 256, 1), (4, 50, 83, 1024, 1024, 3)])
def test_conv_gemm_at_the_timed_shapes_vs_fp64_on_sampled_pixels(K, nb, h, w, cin, cout, ks):
    """coin_conv_gemm_bf16 at the benchmark's own launch shapes (res5: 2048 RoIs, M = 401 408 / 100 352 pixels; the trainable backbone
    stages: M = 16 600 ... 266 400): 4096 random output pixels against an fp64 evaluation of the same bf16 operands (forward and data
    gradient), and the fused BatchNorm statistics against fp64 statistics of the stored output."""
    g = torch.Generator(device="cuda").manual_seed(nb + cin + ks)
    x = (torch.randn((nb, h, w, cin), generator=g, device="cuda") * 0.7).to(torch.bfloat16)
    wt = (torch.randn((cout, ks, ks, cin), generator=g, device="cuda") * (2.0 / (cin * ks * ks)) ** 0.5).to(torch.bfloat16)
    m = nb * h * w
    out, part = K.conv_gemm(x.reshape(m, cin), wt.reshape(cout, ks * ks * cin), spatial=(h, w, cin) if ks == 3 else None, stats_rows=m)
    pix = torch.randint(0, m, (4096,), generator=g, device="cuda")

    def gather_rows(src, c):   # im2col rows of the sampled pixels: [4096, ks*ks*c] in (ky, kx, c) order, zeros outside the image
        n_, r = pix // (h * w), pix % (h * w)
        oy, ox = r // w, r % w
        cols = []
        for ky in range(ks):
            for kx in range(ks):
                yy, xx = oy + ky - ks // 2, ox + kx - ks // 2
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                v = src[n_, yy.clamp(0, h - 1), xx.clamp(0, w - 1)].double()
                cols.append(v * ok.unsqueeze(1))
        return torch.cat(cols, dim=1)

    ref = gather_rows(x, cin) @ wt.reshape(cout, -1).double().t()
    got = out[pix].double()
    assert float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-3     # one bf16 rounding of the stored value
    mean, rstd = K.conv_stats_finalize(part, m, cout, m, 1e-5, 0.1, None, None)
    y = out.double()
    torch.testing.assert_close(mean.double(), y.mean(0), rtol=1e-5, atol=1e-5 * float(y.abs().max()))
    torch.testing.assert_close(rstd.double(), (y.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=0)
    # data gradient = the same contraction with the re-laid weight
    gy = (torch.randn((nb, h, w, cout), generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    wd = wt.flip(1, 2).permute(3, 1, 2, 0).contiguous()                                     # [cin, ky', kx', cout]
    gx, _ = K.conv_gemm(gy.reshape(m, cout), wd.reshape(cin, ks * ks * cout), spatial=(h, w, cout) if ks == 3 else None)
    gref = gather_rows(gy, cout) @ wd.reshape(cin, -1).double().t()
    assert float((gx[pix].double() - gref).abs().max()) <= 2.0 ** -8 * float(gref.abs().max()) + 1e-3


@pytest.mark.parametrize("nb,h,w,cin,cout,ks", [(40, 7, 7, 256, 256, 1), (9, 14, 14, 512, 256, 1), (33, 7, 7, 256, 512, 3), (6, 14, 14, 256, 256, 3), (700, 7, 7, 512, 256, 1),
                                                # odd multiples of 128 (the backbone's layer2): half-valid edge tiles, a 3x3 K-tile that spans two taps
                                                (2, 50, 83, 512, 128, 1), (2, 50, 83, 128, 512, 1), (2, 50, 83, 128, 128, 3), (3, 23, 31, 128, 384, 3),
                                                (3, 23, 31, 384, 128, 3), (1, 9, 11, 128, 128, 1),
                                                # round 6: the backbone's maps at the benchmark's size -- the 128 x 128 small-map kernel (every
                                                # launch whose operands are below 160 MB), incl. the RPN head's width and the box head's rows
                                                (4, 50, 83, 256, 1024, 1), (4, 50, 83, 1024, 256, 1), (4, 50, 83, 256, 256, 3), (4, 100, 167, 128, 128, 3),
                                                (1, 25, 40, 1024, 1024, 3), (2048, 1, 1, 2048, 1024, 1)])
def test_conv_wgrad_vs_fp64(K, nb, h, w, cin, cout, ks):
    """coin_conv_wgrad_bf16 (transposed-LDS-read MFMA contraction over the pixels, sliced, slabs summed in order) vs the fp64 weight
    gradient of F.conv2d on the same bf16 tensors; two runs agree bit for bit."""
    g = torch.Generator().manual_seed(nb + cin + ks)
    x = torch.randn(nb, cin, h, w, generator=g).to(torch.bfloat16)
    gy = torch.randn(nb, cout, h, w, generator=g).to(torch.bfloat16)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, ks, ks), gy.double(), padding=ks // 2)
    xa = dev(x.permute(0, 2, 3, 1).contiguous().reshape(-1, cin))
    ga = dev(gy.permute(0, 2, 3, 1).contiguous().reshape(-1, cout))
    dw = K.conv_wgrad(ga, xa, spatial=(h, w, cin) if ks == 3 else None)
    got = dw.view(cout, ks, ks, cin).permute(0, 3, 1, 2).cpu().double()
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))  # fp32 accumulation of exact bf16 products
    assert torch.equal(dw, K.conv_wgrad(ga, xa, spatial=(h, w, cin) if ks == 3 else None))


@pytest.mark.parametrize("inplanes,planes,stride", [(256, 64, 1), (128, 64, 2), (256, 64, 2), (512, 256, 2)])
def test_bottleneck_gradient_fan_in_fused_in_dgrad_epilogue(monkeypatch, inplanes, planes, stride):
    """Trainable CLIP Bottleneck (coin/modeling/utils.py:60-90) in the bf16 mode: the block input's two gradients (conv1's dgrad and
    the identity / downsample branch) are summed in conv1's dgrad epilogue.  Same bits as autograd's separate add, and one
    elementwise launch fewer.  Stride 2: the downsample branch's AvgPool2d backward is folded into the same epilogue
    (coin_conv_gemm_bf16_rpool; an odd map size exercises the floor-pooled border) -- same bits as the separate pool-backward kernel."""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    import seeded
    from coin_amd import layers as L
    from coin_amd.modeling.backbone import Bottleneck

    monkeypatch.setitem(L.CONV_GEMM, "enabled", True)
    monkeypatch.setitem(L.CONV_GEMM, "min_rows", 0)
    blk = seeded.fill_module(Bottleneck(inplanes, planes, stride), 77).cuda().train()
    hw = 15 if (stride == 2 and inplanes == 256) else 14   # 15: floor pooling leaves the last row / column without a pool gradient
    x0 = seeded.randn((4, inplanes, hw, hw), 78).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = None
    res = []
    for fork in (True, False):
        if not fork:
            real = L.conv_bn_act
            # the un-fused reference: a plain tap (average-pooled by the separate kernel pair for the stride-2 block's fork="pool")
            monkeypatch.setattr(L, "conv_bn_act", lambda *a, fork=False, **k: ((real(*a, **k), L.avg_pool2(a[0]) if fork == "pool" else a[0]) if fork else real(*a, **k)))
        x = (x0 * 1).requires_grad_(True)   # non-leaf copy: its gradient is what the previous block would receive
        x.retain_grad()
        blk.zero_grad()
        y = blk(x)
        gy = seeded.randn(tuple(y.shape), 79).cuda().to(torch.bfloat16) if gy is None else gy
        y.backward(gy)
        res.append((y.detach().clone(), x.grad.clone(), blk.conv1.weight.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]), "forward differs"
    assert torch.equal(res[0][1], res[1][1]), float((res[0][1].float() - res[1][1].float()).abs().max())
    # the weight gradient comes from the library's split-K contraction (atomic accumulation, bf16 result: not bit-reproducible run
    # to run, one bf16 ulp = 0.4 %)
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-2, atol=1e-2 * float(res[1][2].abs().max()))


@pytest.mark.parametrize("nhw", [(2, 200, 336), (16, 200, 334), (8, 160, 160), (3, 15, 15)])
def test_pooled_residual_epilogue_on_large_maps_equals_the_materialised_pool_gradient(nhw):
    """coin_conv_gemm_bf16_rpool splits an output row into (image, h, w) with multiply-high divisions; round-4 ADVICE: without a fix-up the
    quotient is one too large from row 134 399 of a [2, 200, 336] map on (the pooled quarter of the last pixel of an image was lost and the
    next image's row was read).  Same bits as the plain GEMM with the materialised avg-pool gradient as its residual."""
    from coin_amd import kernels as K

    n, h, w = nhw
    m, nn, k = n * h * w, 256, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(nn, k, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    r = torch.randn(n, h // 2, w // 2, nn, device="cuda", generator=g).to(torch.bfloat16)
    full = K.avgpool2_bwd(r, (n, h, w, nn)).reshape(m, nn)
    want, _ = K.conv_gemm(a, b, residual=full)
    got = K.conv_gemm(a, b, residual=r.reshape(-1, nn), residual_pool=(h, w))
    assert got is not None, "the persistent kernel must serve this shape"
    assert torch.equal(got[0], want), float((got[0].float() - want.float()).abs().max())


@pytest.mark.parametrize("shape", [(2, 256, 40, 56, 256), (4, 1024, 50, 83, 1024)])   # small; the RPN head at the timed shape
def test_rpn_head_conv_bias_relu_on_the_gemm_path_vs_fp64(monkeypatch, shape):
    """relu(conv3x3(x) + bias) of the RPN head (StandardRPNHead, called at rpn.py:65) on coin_conv_gemm_bf16 + the streaming bias/clamp pass
    (layers.conv_bias_relu) against the fp64 convolution of the same bf16 operands: output to bf16 rounding, and the three
    gradients (input through the masked dgrad, weight through coin_conv_wgrad_bf16, bias) to the bf16 noise of their inputs."""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    import seeded
    from coin_amd import layers as L

    n, c, h, w, co = shape
    monkeypatch.setitem(L.CONV_GEMM, "enabled", True)
    conv = torch.nn.Conv2d(c, co, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(seeded.randn(tuple(conv.weight.shape), 5) * (2.0 / (9 * c)) ** 0.5)
        conv.bias.copy_(seeded.randn((co,), 6) * 0.2)
    conv = conv.cuda()
    x = (seeded.randn((n, c, h, w), 7).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)).requires_grad_(True)
    calls = []
    real = L._ConvGemmBiasRelu.apply
    monkeypatch.setattr(L._ConvGemmBiasRelu, "apply", lambda *a: (calls.append(1), real(*a))[1])
    y = L.conv_bias_relu(x, conv, min_rows=0)
    assert y.dtype == torch.bfloat16 and calls == [1], "the hand-written path did not run"
    gy = seeded.randn(tuple(y.shape), 8).cuda().to(torch.bfloat16)
    y.backward(gy)
    # fp64 reference on the same bf16 operands (weights as the bf16 shadow the kernel reads)
    xr = x.detach().double().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).double().requires_grad_(True)
    br = conv.bias.detach().double().requires_grad_(True)
    # the product rounds conv(x) to bf16 before the bias: take the mask / values from the same intermediate
    z = F.conv2d(xr, wr, None, padding=1)
    yr = torch.relu(z.to(torch.bfloat16).double().detach() + (z - z.detach()) + br.view(1, -1, 1, 1))
    yr.backward(gy.double())
    scale = float(yr.detach().abs().max())
    assert float((y.detach().double() - yr.detach()).abs().max()) <= 2.0 ** -7 * scale           # two bf16 roundings
    for got, ref, what in ((x.grad, xr.grad, "dx"), (conv.weight.grad, wr.grad, "dw"), (conv.bias.grad, br.grad, "dbias")):
        err = float((got.double() - ref).abs().max())
        ref_scale = float(ref.abs().max())
        # dx is stored in bf16 (2^-8 relative per element, measured against the tensor's scale as everywhere in this file); dw / dbias are
        # fp32 sums of bf16 products, their error is that of the bf16 dz rows (mask + rounding of gy already in the reference)
        assert err <= (2.0 ** -7 if what == "dx" else 2e-3) * ref_scale, (what, err, ref_scale)


def test_gemm_rejects_bad_shapes(K):
    from coin_amd._lib import CoinHipError

    a = torch.zeros(8, 40, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(CoinHipError):
        K.gemm_nt(a, a)  # K % 64 != 0 on the bf16 path


def test_transpose_and_bias_act_bwd(K):
    g = torch.Generator().manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(130, 77, generator=g).to(dt)
        assert torch.equal(K.transpose2d(dev(x)).cpu(), x.t().contiguous())
    z = torch.randn(300, 96, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(96, generator=g, dtype=torch.float64, requires_grad=True)
    c = F.leaky_relu(z + bias, 0.01)
    dc = torch.randn(300, 96, generator=g, dtype=torch.float64)
    c.backward(dc)
    dz, db = K.bias_act_bwd(dev(dc.float()), dev(c.detach().float()), 1, 0.01)
    torch.testing.assert_close(dz.cpu().double(), z.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(db.cpu().double(), bias.grad, rtol=1e-4, atol=1e-4)
    # several 256-row blocks, ragged rows / columns, both access widths (N % 8 != 0 -> scalar lanes), bf16: the bias gradient is the sum of
    # per-block partials in block order (no float atomics) -> identical bits on every run, and it accumulates into `dbias`
    for m, n, dt in ((2048, 1024, torch.bfloat16), (777, 72, torch.bfloat16), (600, 9, torch.bfloat16), (1030, 100, torch.float32)):
        dc2 = dev(torch.randn(m, n, generator=g).to(dt))
        c2 = dev(torch.randn(m, n, generator=g).to(dt))
        dz2, db2 = K.bias_act_bwd(dc2, c2, 2, 0.0)
        ref = dc2.double() * (c2.double() > 0)
        torch.testing.assert_close(dz2.double(), ref, rtol=0, atol=0)
        torch.testing.assert_close(db2.double(), ref.sum(0), rtol=1e-5, atol=1e-4)
        for _ in range(3):
            assert torch.equal(K.bias_act_bwd(dc2, c2, 2, 0.0)[1], db2)
        acc = torch.ones(n, device="cuda")
        K.bias_act_bwd(dc2, c2, 2, 0.0, dbias=acc)
        torch.testing.assert_close(acc, db2 + 1.0, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("r,d,kc", [(70, 128, 9), (2048, 1024, 9), (33, 512, 21)])
def test_cosine_logits(K, dtype, r, d, kc):
    g = torch.Generator().manual_seed(r + kc)
    f = torch.randn(r, d, generator=g).to(dtype)
    t = torch.randn(kc, d, generator=g)
    fd, td = f.double().requires_grad_(True), t.double().requires_grad_(True)
    s_ref = (fd / fd.norm(dim=1, keepdim=True)) @ (td / td.norm(dim=1, keepdim=True)).t() / 0.01
    ds = torch.randn(r, kc, generator=g)
    s_ref.backward(ds.double())
    s, inv = K.cosine_logits_fwd(dev(f), dev(t), 100.0)
    torch.testing.assert_close(s.cpu().double(), s_ref.detach(), rtol=1e-4, atol=1e-4)
    df, dtx = K.cosine_logits_bwd(dev(ds), dev(f), dev(t), s, inv, 100.0)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(df.cpu().double(), fd.grad, rtol=tol, atol=tol * float(fd.grad.abs().max()))
    torch.testing.assert_close(dtx.cpu().double(), td.grad, rtol=1e-3, atol=1e-3 * float(td.grad.abs().max()))
    for _ in range(2):   # the text gradient is joined over the row blocks in block order (no float atomics): identical bits on every run
        assert torch.equal(K.cosine_logits_bwd(dev(ds), dev(f), dev(t), s, inv, 100.0)[1], dtx)


# ------------------------------------------------------------------------------------------ losses
def test_mil_ce_golden_and_grad(K):
    z = load_golden("mil_losses")
    x, hard, soft, w = (torch.from_numpy(z[k]) for k in ("x", "hard", "soft", "weights"))
    l, gr = K.mil_ce(dev(x), target=dev(hard), weights=dev(w), avg_positives=True)
    assert abs(float(l) - float(z["ce_hard_avg_w_mean"])) < 1e-4
    np.testing.assert_allclose(gr.cpu().numpy(), z["ce_hard_avg_w_mean_grad"], rtol=1e-4, atol=1e-6)
    l, _ = K.mil_ce(dev(x), labels=dev(hard.argmax(1)), weights=dev(w), avg_positives=True)
    assert abs(float(l) - float(z["ce_hard_avg_w_mean"])) < 1e-4
    l, _ = K.mil_ce(dev(x), target=dev(hard), avg_positives=False)
    assert abs(float(l) - float(z["ce_hard_noavg_mean"])) < 1e-4
    l, _ = K.mil_ce(dev(x), target=dev(soft + 1e-3), weights=dev(w), avg_positives=True, reduction="sum")
    assert abs(float(l) - float(z["ce_soft_avg_sum"])) < 1e-4 * max(1.0, abs(float(z["ce_soft_avg_sum"])))
    l, _ = K.mil_ce(dev(x), target=dev(soft + 1e-3), weights=dev(w), avg_positives=False)
    assert abs(float(l) - float(z["ce_soft_noavg_w_mean"])) < 1e-4
    l, gr = K.mil_ce(dev(x[:0]), target=dev(hard[:0]), weights=dev(w[:0]), avg_positives=True)
    assert float(l) == 0.0 == float(z["ce_empty"])


def test_mil_focal_golden_and_grad(K):
    """coin_mil_focal_fwd_bwd vs the values captured from the reference's MILFocalLoss (losses.py:36-73) and the oracle's autograd."""
    from oracle import losses as OL

    z = load_golden("mil_losses")
    x, hard, soft, alpha = (torch.from_numpy(z[k]) for k in ("x", "hard", "soft", "focal_alpha"))
    l, _ = K.mil_focal(dev(x), dev(alpha), target=dev(hard), avg_positives=True)
    assert abs(float(l) - float(z["focal_hard_avg"])) < 1e-5 * max(1.0, abs(float(z["focal_hard_avg"])))
    l, _ = K.mil_focal(dev(x), dev(alpha), labels=dev(hard.argmax(1)), avg_positives=True)
    assert abs(float(l) - float(z["focal_hard_avg"])) < 1e-5 * max(1.0, abs(float(z["focal_hard_avg"])))
    l, gr = K.mil_focal(dev(x), dev(alpha), target=dev(soft + 1e-3), avg_positives=False)
    assert abs(float(l) - float(z["focal_soft_noavg"])) < 1e-5 * max(1.0, abs(float(z["focal_soft_noavg"])))
    xr = x.clone().double().requires_grad_(True)
    OL.mil_focal_loss(xr, (soft + 1e-3).double(), alpha.double(), avg_positives=False).backward()
    torch.testing.assert_close(gr.cpu().double(), xr.grad, rtol=1e-4, atol=1e-7)
    # row weights / sum reduction (the fixed-shape sampler's validity mask) and the empty input
    w = (torch.arange(40) % 3 != 0).float()
    l, gr = K.mil_focal(dev(x), dev(alpha), target=dev(hard), weights=dev(w), reduction="sum")
    xr = x.clone().double().requires_grad_(True)
    ref = sum(OL.mil_focal_loss(xr[i:i + 1], hard[i:i + 1].double(), alpha.double()) * w[i] for i in range(40))
    ref.backward()
    assert abs(float(l) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    torch.testing.assert_close(gr.cpu().double(), xr.grad, rtol=1e-4, atol=1e-7)
    l, _ = K.mil_focal(dev(x[:0]), dev(alpha), target=dev(hard[:0]))
    assert np.isnan(float(l))  # torch's mean over zero rows, as the reference


def test_mil_ce_vs_oracle_large(K):
    from oracle import losses as OL

    g = torch.Generator().manual_seed(9)
    x = (torch.randn(2048, 9, generator=g) * 20).requires_grad_(True)  # logits = cos/0.01 are O(10)
    t = torch.rand(2048, 9, generator=g)
    w = torch.rand(2048, generator=g)
    ref = OL.mil_cross_entropy(x, t, w, avg_positives=True)
    ref.backward()
    l, gr = K.mil_ce(dev(x.detach()), target=dev(t), weights=dev(w), avg_positives=True)
    assert abs(float(l) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    torch.testing.assert_close(gr.cpu(), x.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_kl_div(K, mode):
    from oracle import losses as OL

    g = torch.Generator().manual_seed(10 + mode)
    r, c = 333, 9
    q = torch.softmax(torch.randn(r, c, generator=g) * 2, dim=1)
    q[5] = F.one_hot(torch.tensor(2), c).float()  # exact zeros in the target (xlogy branch)
    mask = torch.rand(r, generator=g) > 0.3
    if mode == 0:
        x = (torch.randn(r, c, generator=g) * 5).requires_grad_(True)
        ref = OL.kl_div_mean(torch.softmax(x[mask], dim=1), q[mask])
    elif mode == 1:
        x = torch.softmax(torch.randn(r, c, generator=g), dim=1).requires_grad_(True)
        ref = OL.kl_div_mean(x[mask], q[mask])
    else:
        x = (torch.randn(r, generator=g) * 3).requires_grad_(True)
        q = torch.rand(r, generator=g)
        p = torch.sigmoid(x[mask])
        ref = OL.kl_div_mean(torch.stack((p, 1 - p), 1), torch.stack((q[mask], 1 - q[mask]), 1))
    ref.backward()
    l, gr = K.kl_div(dev(x.detach()), dev(q), mode, row_mask=dev(mask))
    assert abs(float(l) - float(ref)) < 1e-5 + 1e-4 * abs(float(ref))
    torch.testing.assert_close(gr.cpu(), x.grad, rtol=2e-4, atol=1e-7)
    # unmasked + empty selection
    l2, _ = K.kl_div(dev(x.detach()), dev(q), mode, row_mask=dev(torch.zeros(r, dtype=torch.bool)))
    assert float(l2) == 0.0


def test_box_reg_and_l1(K):
    from oracle import losses as OL

    g = torch.Generator().manual_seed(12)
    r = 500
    x0 = torch.rand(r, 2, generator=g) * 500
    p = torch.cat([x0, x0 + torch.rand(r, 2, generator=g) * 200 + 4], 1)
    gt = p + torch.randn(r, 4, generator=g) * 3
    cls = torch.randint(-1, 10, (r,), generator=g)
    pred = torch.randn(r, 4, generator=g, requires_grad=True)
    ref = OL.box_reg_loss(p, gt, pred, cls, 8)
    ref.backward()
    l, gr = K.box_reg_l1(dev(p), dev(gt), dev(pred.detach()), dev(cls), 8, (10, 10, 5, 5), float(r))
    assert abs(float(l) - float(ref)) < 1e-4 * max(1.0, float(ref))
    torch.testing.assert_close(gr.cpu(), pred.grad)
    a = torch.randn(9, 1024, generator=g, requires_grad=True)
    b = torch.randn(9, 1024, generator=g)
    ref = F.l1_loss(a, b)
    ref.backward()
    l, gr = K.l1_mean(dev(a.detach()), dev(b))
    assert abs(float(l) - float(ref)) < 1e-5
    torch.testing.assert_close(gr.cpu(), a.grad)


def test_rpn_losses_vs_golden_inputs(K):
    """Same labels / matched boxes as the reference run in tests/golden/rpn.npz (sampling is RNG: an input)."""
    from oracle import d2

    z = load_golden("rpn")
    feat = torch.from_numpy(z["feat"])
    head = d2.StandardRPNHead(128, 9)
    head.load_state_dict({k[len("w::rpn_head."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::rpn_head.")})
    lg, dl = head([feat])
    logits = lg[0].permute(0, 2, 3, 1).flatten(1).detach()
    deltas = dl[0].view(2, -1, 4, 6, 8).permute(0, 3, 4, 1, 2).flatten(1, -2).detach().contiguous()
    labels = torch.from_numpy(z["labels"]).to(torch.int8)
    matched = torch.from_numpy(z["matched_boxes"])
    anchors = torch.from_numpy(z["anchors"])
    cls, loc, g_l, g_d = K.rpn_losses(dev(logits.contiguous()), dev(labels), dev(deltas), dev(anchors), dev(matched))
    for _ in range(2):   # block partials joined in block order (no float atomics): identical totals on every run
        c2, l2, _, _ = K.rpn_losses(dev(logits.contiguous()), dev(labels), dev(deltas), dev(anchors), dev(matched))
        assert torch.equal(c2, cls) and torch.equal(l2, loc)
    norm = 64 * 2
    assert abs(float(cls) / norm - float(z["loss::loss_rpn_cls"])) < 1e-4
    assert abs(float(loc) / norm - float(z["loss::loss_rpn_loc"])) < 1e-4
    lx = logits.clone().requires_grad_(True)
    dx = deltas.clone().requires_grad_(True)
    from oracle import losses as OL

    c2, l2 = OL.rpn_losses(anchors, lx, labels.long(), dx, matched, 64)
    (c2 * norm).backward(retain_graph=True)
    torch.testing.assert_close(g_l.cpu(), lx.grad, rtol=1e-4, atol=1e-6)
    (l2 * norm).backward()
    torch.testing.assert_close(g_d.cpu(), dx.grad, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------ streams
def test_normalize_pad(K):
    g = torch.Generator().manual_seed(13)
    imgs = [torch.randint(0, 256, (3, 37, 50), generator=g, dtype=torch.uint8),
            torch.randint(0, 256, (3, 40, 41), generator=g, dtype=torch.uint8)]
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    ref = torch.zeros(2, 3, 40, 50)
    for i, im in enumerate(imgs):
        v = (im.float() / 255 - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
        ref[i, :, : im.shape[1], : im.shape[2]] = v
    out, sizes = K.normalize_pad([dev(i) for i in imgs], mean, std)
    assert sizes == [(37, 50), (40, 41)]
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    out, _ = K.normalize_pad([dev(i) for i in imgs], mean, std, layout=K.COIN_NHWC, dtype=torch.bfloat16)
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=1e-2, atol=1e-2)


def test_sgd_table_matches_torch_sgd(K):
    g = torch.Generator().manual_seed(14)
    shapes = [(1024, 2048), (1024,), (7,), (3, 3, 5, 2), (128, 129)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    lrs = [0.01, 0.001, 0.01, 0.0001, 0.02]
    wds = [1e-4, 0.0, 1e-4, 1e-4, 0.0]
    opt = torch.optim.SGD([{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(ref, lrs, wds)], lr=0.01, momentum=0.9)
    dp = [dev(p.clone()) for p in ps]
    shadows = [torch.empty_like(p, dtype=torch.bfloat16) for p in dp]
    table = K.SgdTable(dp, lrs, wds, shadows)
    for step in range(5):
        grads = [torch.randn(s, generator=g) for s in shapes]
        if step in (0, 3):
            grads[2] = None  # a tensor without a gradient is skipped (torch.optim.SGD semantics), also on its first step
        for p, gr in zip(ref, grads):
            p.grad = gr.clone() if gr is not None else None
        opt.step()
        table.step([dev(gr) if gr is not None else None for gr in grads], momentum=0.9)  # new pointers every step: async re-upload
    for p, r, s in zip(dp, ref, shadows):
        torch.testing.assert_close(p.cpu(), r.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(s.cpu(), p.cpu().to(torch.bfloat16))


def test_sgd_gate_skips_the_update_on_the_device(K):
    """coin_sgd_step's gate (the data-parallel CKG update, trainer.py:192-197 under DDP): a zero in device memory turns the launch
    into a no-op -- parameters, momentum buffers and shadows untouched -- without the host ever reading it; a non-zero count lets
    it through.  The reducer's flag slot (parallel.GradReducer.flag) is what the trainer passes."""
    from coin_amd.parallel import GradReducer

    g = torch.Generator().manual_seed(15)
    ps = [dev(torch.randn(s, generator=g)) for s in [(300, 17), (64,)]]
    table = K.SgdTable(ps, [0.1, 0.1], [0.0, 0.0], [torch.empty_like(p, dtype=torch.bfloat16) for p in ps])
    before = [p.clone() for p in ps]
    grads = [torch.ones_like(p) for p in ps]
    gate = torch.zeros(1, device="cuda")
    table.step(grads, momentum=0.9, gate=gate)
    assert all(torch.equal(a, b) for a, b in zip(ps, before)) and all(float(b.abs().max()) == 0.0 for b in table.bufs)
    gate.fill_(3.0)
    table.step(grads, momentum=0.9, gate=gate)
    for a, b in zip(ps, before):
        torch.testing.assert_close(a, b - 0.1, rtol=0, atol=1e-6)
    # the reducer's flag slot: one fp32 behind the last slice, zero until set, usable as the gate as it stands
    params = [torch.nn.Parameter(p.clone()) for p in ps]
    red = GradReducer(params)
    assert red.flag.numel() == 1 and red.flag.dtype == torch.float32 and float(red.flag) == 0.0
    red.set_flag(1.0)
    for p in params:
        p.grad = None
    (params[0].sum() * 2 + params[1].sum()).backward()
    assert red.finalize() == 1.0 and float(red.flag) == 1.0
    torch.testing.assert_close(params[0].grad, torch.full_like(params[0], 2.0))
    red.remove()


def test_ema_golden(K):
    z = load_golden("ema")
    keys = [k[3:] for k in z.files if k.startswith("t::") and z[k].dtype == np.float32]
    t = [dev(torch.from_numpy(z["t::" + k]).clone()) for k in keys]
    s = [dev(torch.from_numpy(z["s::" + k]).clone()) for k in keys]
    K.EmaTable(t, s).update(0.9996)
    for k, tt in zip(keys, t):
        np.testing.assert_allclose(tt.cpu().numpy(), z["after::" + k], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------ fused BatchNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,res,pool,shape", [
    (True, False, 1, (6, 16, 14, 14)), (True, False, 2, (5, 32, 14, 14)), (True, True, 1, (4, 64, 7, 7)),
    (False, False, 1, (3, 2048, 7, 7)), (True, False, 2, (2, 8, 9, 11)), (True, True, 1, (2, 4096, 3, 3)),
    (True, True, 0, (5, 2048, 7, 7)), (True, False, 0, (3, 64, 5, 3)), (False, True, 0, (2, 16, 7, 7)),  # global-mean epilogue
])
def test_bn_act_vs_torch(dtype, relu, res, pool, shape):
    from coin_amd import layers as L

    g = torch.Generator().manual_seed(sum(shape) + pool)
    n, c, h, w = shape
    x = (torch.randn(shape, generator=g) * 2 + torch.randn(1, c, 1, 1, generator=g) * 3).to(dtype)
    r = torch.randn(shape, generator=g).to(dtype) if res else None
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.normal_(0, 0.3, generator=g)
        bn.running_mean.normal_(0, 1, generator=g)
        bn.running_var.uniform_(0.5, 2, generator=g)
    ref_bn = torch.nn.BatchNorm2d(c).double()
    ref_bn.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    y = ref_bn(xr)
    if res:
        y = y + rr
    if relu:
        y = F.relu(y)
    if pool == 2:
        y = F.avg_pool2d(y, 2)
    if pool == 0:
        y = y.mean(dim=[2, 3], keepdim=True)
    dy = torch.randn(y.shape, generator=g).to(dtype)
    y.backward(dy.double())
    bn = bn.cuda()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = r.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    out = L.bn_act(xd, bn, relu, rd, pool)
    out.backward(dy.cuda())
    # fp32: measured 6e-7 (outputs) / 1e-6 (gradients) of the tensor's scale at the largest launch shape
    # (tests/test_parity_gpu.py::test_bn_train_at_the_timed_shape_vs_fp64); bf16: output rounding (2^-9 relative per element)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(out.detach().cpu().double(), y.detach(), rtol=tol, atol=tol)
    gtol = 1e-4 if dtype == torch.float32 else 5e-2
    scale = float(xr.grad.abs().max())
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=gtol, atol=gtol * scale)
    torch.testing.assert_close(bn.weight.grad.cpu().double(), ref_bn.weight.grad, rtol=gtol, atol=gtol * float(ref_bn.weight.grad.abs().max()))
    torch.testing.assert_close(bn.bias.grad.cpu().double(), ref_bn.bias.grad, rtol=gtol, atol=gtol * float(ref_bn.bias.grad.abs().max()))
    if res:
        torch.testing.assert_close(rd.grad.cpu().double(), rr.grad, rtol=gtol, atol=gtol)
    rtol = 1e-5 if dtype == torch.float32 else 1e-3
    torch.testing.assert_close(bn.running_mean.cpu().double(), ref_bn.running_mean, rtol=rtol, atol=rtol)
    torch.testing.assert_close(bn.running_var.cpu().double(), ref_bn.running_var, rtol=rtol, atol=rtol)
    assert int(bn.num_batches_tracked) == 1


def test_avg_pool2_vs_torch():
    from coin_amd import layers as L

    g = torch.Generator().manual_seed(3)
    for shape in [(2, 16, 14, 14), (3, 8, 9, 11), (1, 1024, 100, 167)]:
        x = torch.randn(shape, generator=g)
        xr = x.double().requires_grad_(True)
        y = F.avg_pool2d(xr, 2)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy.double())
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        out = L.avg_pool2(xd)
        out.backward(dy.cuda())
        torch.testing.assert_close(out.detach().cpu().double(), y.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-6, atol=1e-6)


def test_frozen_stage_fused_tail_vs_fp64(monkeypatch):
    """Frozen stem + layer1 (FREEZE_AT=2; detectron2 FrozenBatchNorm2d after the reference's convert, utils.py:243-284) in the bf16
    throughput mode: every norm + identity + ReLU (+ the stem's AvgPool2d) tail is ONE coin_bn_apply_fwd pass.  Against the same
    modules evaluated in fp64 (bf16 activations: relative L2), and the fused path must be the one that ran."""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    import seeded
    from coin_amd import layers as L
    from coin_amd.modeling.backbone import ModifiedResNet

    net = seeded.fill_module(ModifiedResNet((2, 1, 1, 1), width=64, freeze_at=0), 41)
    net.freeze(2)
    x = seeded.randn((2, 3, 70, 90), 42)
    with torch.no_grad():
        ref = net.double().frozen_forward(x.double())
    net = net.float().cuda()
    calls = []
    real = L.frozen_bn_act
    monkeypatch.setattr(L, "frozen_bn_act", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    got = net.frozen_forward(x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert len(calls) == 3 + 2 * 3 + 1, calls          # stem 3, two bottlenecks x 3, one downsample
    assert got.shape == ref.shape
    e = float((got.double().cpu() - ref).norm() / ref.norm())
    assert e < 1.5e-2, e                                   # 10 bf16 convolutions deep
    # and the library-only composition of the same stage (norm folded into the convolution) agrees to the same level
    monkeypatch.setattr(L, "frozen_bn_fusable", lambda *a, **k: False)
    old = net.frozen_forward(x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert float((old.double().cpu() - ref).norm() / ref.norm()) < 1.5e-2


# ------------------------------------------------------------------------------------------ anchor labelling / subsampling
@pytest.mark.parametrize("low_quality,thresholds,labels", [(True, (0.3, 0.7), (0, -1, 1)), (False, (0.5,), (0, 1))])
def test_anchor_match_bit_exact_vs_oracle_matcher(K, low_quality, thresholds, labels):
    """coin_anchor_match = detectron2 Matcher(pairwise_iou(gt, anchors)) (oracle/d2.py) per image: indices and labels bit-exact,
    including equal maxima (lowest index), IoUs that sit exactly on a threshold, the low-quality rule's quirk for a box that
    overlaps nothing, an image without boxes and duplicated boxes; at the benchmark's 62 250 anchors."""
    from oracle import d2

    from coin_amd.box_ops import Matcher, cell_anchors, grid_anchors

    g = torch.Generator().manual_seed(11)
    cells = cell_anchors((32, 64, 128, 256, 512), (0.5, 1.0, 2.0))
    anchors = grid_anchors(cells, (50, 83), 16, 0.0, "cpu")
    a = anchors.shape[0]
    assert a == 50 * 83 * 15
    def boxes(n):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([1100.0, 600.0])
        wh = torch.rand(n, 2, generator=g) * 300 + 8
        return torch.cat([xy, xy + wh], dim=1)
    imgs = [boxes(32), torch.zeros(0, 4), boxes(1), boxes(200)]
    imgs[0][5] = anchors[12345]                      # IoU exactly 1 with one anchor
    imgs[0][6] = imgs[0][5]                          # duplicate box: equal maxima -> lowest index
    imgs[0][7] = torch.tensor([5000.0, 5000.0, 5100.0, 5100.0])   # overlaps no anchor: best IoU 0
    imgs[3][9] = anchors[777] + torch.tensor([0.0, 0.0, 16.0, 0.0])
    m = Matcher(list(thresholds), list(labels), allow_low_quality_matches=low_quality)
    idx, lab, mb = m.match_boxes([dev(b) for b in imgs], dev(anchors))
    ref = d2.Matcher(list(thresholds), list(labels), allow_low_quality_matches=low_quality)
    for i, b in enumerate(imgs):
        if b.shape[0] == 0:
            assert int(idx[i].abs().max()) == 0 and bool((lab[i] == labels[0]).all()) and float(mb[i].abs().max()) == 0.0
            continue
        ri, rl = ref(d2.pairwise_iou(d2.Boxes(b), d2.Boxes(anchors)))
        assert torch.equal(idx[i].cpu(), ri), (i, int((idx[i].cpu() != ri).sum()))
        assert torch.equal(lab[i].cpu(), rl.to(torch.int8)), (i, int((lab[i].cpu() != rl).sum()))
        assert torch.equal(mb[i].cpu(), b[ri])
    if low_quality:
        assert bool((lab[0] == 1).sum() > a // 2)    # the quirk: the box that overlaps nothing marks every IoU-0 anchor positive
    _, lab2, none = m.match_boxes([dev(b) for b in imgs], dev(anchors), empty_label=-1, want_boxes=False)
    assert none is None and bool((lab2[1] == -1).all()) and torch.equal(lab2[0], lab[0])


def test_sample_labels_is_the_stable_sort_definition(K):
    """coin_sample_labels (radix select per class) picks exactly the elements an ascending stable sort by key ranks first
    (tests/cpu_shim.py:_sample_labels), with repeated keys, a class smaller / larger than its quota, empty classes, int8 and int64."""
    from cpu_shim import _sample_labels

    g = torch.Generator().manual_seed(3)
    n, m = 7, 62250
    cls = torch.randint(-1, 3, (n, m), generator=g)              # bg label = 0
    cls[0] = -1
    cls[1] = 0
    cls[2, :100] = 2
    cls[2, 100:] = -1
    cls[3] = torch.where(torch.rand(m, generator=g) < 0.001, 1, 0)
    keys = torch.rand(n, m, generator=g)
    keys[4] = (keys[4] * 64).floor() / 64                        # heavy ties: 64 distinct keys
    keys[5, ::2] = 0.0
    for dt in (torch.int64, torch.int8):
        out = K.sample_labels(dev(cls.to(dt)), dev(keys), 0, 256, 128)
        ref = _sample_labels(cls, keys, 0, 256, 128)
        assert torch.equal(out.cpu(), ref), [int((out[i].cpu() != ref[i]).sum()) for i in range(n)]
    out = K.sample_labels(dev(cls[:, :300].contiguous()), dev(keys[:, :300].contiguous()), 0, 64, 16)
    assert torch.equal(out.cpu(), _sample_labels(cls[:, :300], keys[:, :300], 0, 64, 16))
    # keys finer than 2^-24 (what the DEVICE generator's torch.rand draws below 0.5): distinct floats that share a 24-bit radix bin
    # must still be taken in key order, not index order
    fine = torch.randperm(m, generator=g).float() * 2.0 ** -30          # 64 distinct keys per 2^-24 bin, shuffled over the indices
    keys2 = torch.stack([fine, fine.flip(0), (fine * 3.0).clamp(max=0.99)])
    cls2 = torch.randint(-1, 3, (3, m), generator=g)
    for quota, cap in ((256, 128), (1000, 37), (64, 63)):
        out = K.sample_labels(dev(cls2), dev(keys2), 0, quota, cap)
        ref = _sample_labels(cls2, keys2, 0, quota, cap)
        assert torch.equal(out.cpu(), ref), (quota, cap, [int((out[i].cpu() != ref[i]).sum()) for i in range(3)])


# ------------------------------------------------------------------------------------------ NMS
def test_nms_batched_vs_oracle(K):
    from oracle import d2

    g = torch.Generator().manual_seed(21)
    counts = [1000, 0, 777, 3000, 64, 65]
    n_max = 3000
    boxes = torch.zeros(len(counts), n_max, 4)
    for i, n in enumerate(counts):
        centers = torch.rand(max(n // 6, 1), 2, generator=g) * 800
        c = centers[torch.randint(0, centers.shape[0], (n,), generator=g)] + torch.randn(n, 2, generator=g) * 12
        wh = torch.rand(n, 2, generator=g) * 120 + 16
        boxes[i, :n] = torch.cat([c - wh / 2, c + wh / 2], dim=1)
    for thr, max_keep in ((0.7, 2000), (0.5, 50)):
        keep, num = K.nms_batched(dev(boxes), dev(torch.tensor(counts, dtype=torch.int32)), thr, max_keep)
        keep, num = keep.cpu(), num.cpu()
        for i, n in enumerate(counts):
            ref = d2.nms(boxes[i, :n], -torch.arange(n, dtype=torch.float32), thr)[:max_keep]
            assert int(num[i]) == len(ref), (i, int(num[i]), len(ref))
            assert torch.equal(keep[i, : len(ref)].long(), ref)


def test_nms_large_and_class_aware(K):
    from coin_amd import box_ops
    from oracle import d2

    g = torch.Generator().manual_seed(22)
    n = 12000
    c = torch.rand(n, 2, generator=g) * torch.tensor([1333.0, 800.0])
    wh = torch.rand(n, 2, generator=g) * 200 + 20
    b = torch.cat([c - wh / 2, c + wh / 2], dim=1)
    s = torch.rand(n, generator=g)
    got = box_ops.nms(dev(b), dev(s), 0.7).cpu()
    ref = d2.nms(b, s, 0.7)
    assert torch.equal(got, ref)
    idx = torch.randint(0, 8, (n,), generator=g)
    got = box_ops.batched_nms(dev(b[:3000]), dev(s[:3000]), dev(idx[:3000]), 0.5).cpu()
    ref = d2.batched_nms(b[:3000], s[:3000], idx[:3000], 0.5)
    assert torch.equal(got, ref)


def test_bn_act_valid_rows_padding_is_exact():
    """Shape padding (layers.valid_rows): statistics, outputs and gradients of the real rows are those of the un-padded batch;
    the filler rows get zero gradient."""
    from coin_amd import layers as L

    g = torch.Generator().manual_seed(21)
    n, pad, c, h, w = 5, 3, 64, 7, 7
    x = torch.randn(n, c, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    filler = (torch.randn(pad, c, h, w, generator=g) * 50 + 9).cuda().contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, c, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    for pool, res in ((1, True), (1, False), (2, False), (0, True)):
        bn_a, bn_b = torch.nn.BatchNorm2d(c).cuda(), torch.nn.BatchNorm2d(c).cuda()
        xa = x.clone().requires_grad_(True)
        ya = L.bn_act(xa, bn_a, True, r if res else None, pool)
        dy = torch.randn(ya.shape, generator=g).cuda()
        ya.backward(dy)
        xb = torch.cat([x, filler]).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        rb = torch.cat([r, filler]).contiguous(memory_format=torch.channels_last) if res else None
        with L.valid_rows(n):
            yb = L.bn_act(xb, bn_b, True, rb, pool)
        yb[:n].backward(dy)
        assert torch.equal(ya, yb[:n]) and torch.equal(xa.grad, xb.grad[:n])
        assert float(xb.grad[n:].abs().max()) == 0.0
        assert torch.equal(bn_a.weight.grad, bn_b.weight.grad) and torch.equal(bn_a.running_var, bn_b.running_var)


def test_roi_align_at_the_bench_shape_vs_oracle_on_sampled_rois(K):
    """The timed shape itself ([4, 50, 83, 1024] bf16 map, 2048 RoIs, 14 x 14 bins) against `oracle.d2.roi_align_torch` (fp64) on a
    sample of 64 RoIs: forward rows of the sampled RoIs out of the full launch; backward of the full launch with the gradient
    of every other RoI zero (the map gradient is a sum over RoIs, so it equals the oracle's backward over the 64)."""
    from oracle import d2

    g = torch.Generator().manual_seed(11)
    n, c, h, w, r = 4, 1024, 50, 83, 2048
    feat = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16)
    rois = make_rois(n, r, 800, 1333, g)
    pick = torch.randperm(r, generator=g)[:64].sort().values
    out = K.roi_align_fwd(dev(feat), dev(rois), (14, 14), 1 / 16.0)
    assert out.shape == (r, 14, 14, c)
    ref = d2.roi_align_torch(feat.double().permute(0, 3, 1, 2), rois[pick].double(), (14, 14), 1 / 16.0, 0, True)
    torch.testing.assert_close(out[pick.cuda()].float().permute(0, 3, 1, 2).cpu().double(), ref, rtol=2e-2, atol=2e-2)  # one bf16 rounding of the output
    go = torch.zeros(r, 14, 14, c, dtype=torch.bfloat16)
    go[pick] = torch.randn(64, 14, 14, c, generator=g).to(torch.bfloat16)
    fd = torch.zeros(n, c, h, w, dtype=torch.float64, requires_grad=True)
    d2.roi_align_torch(fd, rois[pick].double(), (14, 14), 1 / 16.0, 0, True).backward(go[pick].double().permute(0, 3, 1, 2))
    gin = K.roi_align_bwd(dev(go), dev(rois), (n, h, w, c), 1 / 16.0)   # fp32 map: sums of exactly representable bf16 x fp32 weights
    torch.testing.assert_close(gin.permute(0, 3, 1, 2).cpu().double(), fd.grad, rtol=2e-4, atol=2e-4)


# ------------------------------------------------------------------------------------------ round 4: launch-count reductions
@pytest.mark.parametrize("co,ci,ks", [(512, 1024, 1), (2048, 512, 1), (512, 512, 3), (72, 40, 3), (1024, 2048, 1), (8, 8, 1)])
def test_weight_dgrad_layout_one_launch_vs_torch(K, co, ci, ks):
    """coin_weight_dgrad_layout: dst[ci][ks-1-ky][ks-1-kx][co] = src[co][ky][kx][ci] for a whole table in one launch (byte moves: exact),
    against flip + permute of the same weights; ragged 64-tiles (72 x 40), and two entries of different size in one table."""
    g = torch.Generator().manual_seed(co + ci + ks)
    w_a = torch.randn(co, ci, ks, ks, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    w_b = torch.randn(64, 128, generator=g).to(torch.bfloat16).cuda()                       # a linear weight [N, K]
    d_a = torch.full((ci, ks * ks * co), 7.0, dtype=torch.bfloat16, device="cuda")
    d_b = torch.full((128, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    K.WdTable([(w_a, d_a, co, ci, ks), (w_b, d_b, 64, 128, 1)]).run()
    assert torch.equal(d_a, w_a.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(ci, ks * ks * co))
    assert torch.equal(d_b, w_b.t().contiguous())


def test_dgrad_weight_cache_follows_the_optimizer_and_foreign_writes(monkeypatch):
    """layers.dgrad_weight: the persistent data-gradient layout of a conv weight is refreshed by ONE launch after `build_optimizer(...).step()`
    (raw-pointer update of master + bf16 shadow), and recomputed when the master is written by anything else (version bump).  The input
    gradient of the GEMM convolution must equal the one computed with a freshly re-laid weight after each kind of update."""
    from coin_amd import layers as L

    monkeypatch.setitem(L.CONV_GEMM, "enabled", True)
    monkeypatch.setitem(L.CONV_GEMM, "min_rows", 0)
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x0 = torch.randn(4, 256, 14, 14, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, 256, 14, 14, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def dx_now():
        x = x0.clone().requires_grad_(True)
        conv.weight.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, _ = L.conv2d_gemm(x, conv)
        y.backward(gy)
        return x.grad.clone()

    def dx_fresh():
        wq = conv.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wd = wq.flip(2, 3).permute(1, 2, 3, 0).contiguous().reshape(256, 9 * 256)
        out, _ = __import__("coin_amd.kernels", fromlist=["x"]).conv_gemm(gy.permute(0, 2, 3, 1).reshape(-1, 256), wd, spatial=(14, 14, 256))
        return out.view(4, 14, 14, 256).permute(0, 3, 1, 2)

    assert torch.equal(dx_now(), dx_fresh())                       # first use: computed with torch ops
    assert len(L._DGRAD) >= 1
    # (1) the product's optimizer: fused SGD writes master + shadow through raw pointers, `weights_updated` marks the layouts stale
    from coin_amd.config import get_cfg
    from coin_amd.solver import build_optimizer

    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.BASE_LR", 0.5, "SOLVER.MOMENTUM", 0.9, "SOLVER.WEIGHT_DECAY", 0.0])
    opt = build_optimizer(cfg, conv)
    dx_now()
    before = conv.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, conv.weight.detach())
    assert L._DGRAD_STATE["dirty"]
    assert torch.equal(dx_now(), dx_fresh()) and not L._DGRAD_STATE["dirty"]
    # (2) a foreign in-place write (version bump): recomputed lazily
    with torch.no_grad():
        conv.weight.mul_(-1.5)
    assert torch.equal(dx_now(), dx_fresh())


def test_bn_counter_is_incremented_by_the_statistics_launch(monkeypatch):
    """nn.BatchNorm2d.num_batches_tracked (utils.py:77-90 under train()): incremented by coin_bn_stats / coin_conv_gemm_stats_finalize
    themselves -- once per training forward on either statistics path, not at all in eval mode."""
    from coin_amd import layers as L

    torch.manual_seed(6)
    bn = torch.nn.BatchNorm2d(256).cuda().train()
    x = torch.randn(4, 256, 14, 14, device="cuda").contiguous(memory_format=torch.channels_last)
    for i in range(3):
        L.bn_act(x, bn, True)
        assert int(bn.num_batches_tracked) == i + 1
    monkeypatch.setitem(L.CONV_GEMM, "enabled", True)
    monkeypatch.setitem(L.CONV_GEMM, "min_rows", 0)
    conv = torch.nn.Conv2d(256, 256, 1, bias=False).cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        L.conv_bn_act(x.to(torch.bfloat16), conv, bn, True)         # statistics from the GEMM epilogue -> the finalize kernel counts
    assert int(bn.num_batches_tracked) == 4
    rm = bn.running_mean.clone()
    bn.eval()
    with torch.no_grad():
        L.bn_act(x, bn, True)
    assert int(bn.num_batches_tracked) == 4 and torch.equal(rm, bn.running_mean)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pool", [1, 0])
def test_bn_relu_bit_mask_gives_the_backward_of_the_saved_output(K, dtype, pool):
    """coin_bn_apply_fwd's ReLU bit mask (one bit per element, written by the apply pass) given to coin_bn_bwd in place of the saved
    output (pool 1) / the residual input (pool 0): every gradient is BIT-identical, and the mask is the sign of the output."""
    n, h, w, c = 37, 7, 7, 256
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((n, h, w, c), generator=g, device="cuda").to(dtype)
    r = torch.randn((n, h, w, c), generator=g, device="cuda").to(dtype)
    gam, bet = torch.rand(c, generator=g, device="cuda") + 0.5, torch.randn(c, generator=g, device="cuda")
    mean, rstd = K.bn_stats(x, 1e-5, 0.1)
    y0 = K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool)
    y1, mask = K.bn_apply_fwd(x, mean, rstd, gam, bet, r, True, pool, want_mask=True)
    assert torch.equal(y0, y1)
    v = 8 if dtype == torch.bfloat16 else 4
    assert mask.shape == (n * h * w, c // v) and mask.dtype == torch.uint8
    bits = ((mask.view(n, h, w, c // v, 1).int() >> torch.arange(v, device="cuda").view(1, 1, 1, 1, v)) & 1).bool().view(n, h, w, c)
    if pool == 1:
        assert torch.equal(bits, y1 > 0)
    else:
        pre = (x.float() - mean) * (rstd * gam) + bet + r.float()
        assert float((bits != (pre > 0)).float().mean()) < 1e-4   # fp32 evaluation order at |pre| ~ 1e-7
    dy = torch.randn(y0.shape, generator=g, device="cuda").to(dtype)
    a = K.bn_bwd(x, dy, y0 if pool == 1 else r, mean, rstd, gam, bet, True, pool, True)
    b = K.bn_bwd(x, dy, None, mean, rstd, gam, bet, True, pool, True, mask=mask)
    if pool == 1:
        for u, t in zip(a, b):
            assert torch.equal(u, t)
    else:  # pool 0: the reference path recomputes the pre-activation (same expression as the forward), so it is also exact
        for u, t in zip(a, b):
            assert torch.equal(u, t)
    with pytest.raises(K.CoinHipError):
        K.bn_apply_fwd(x, mean, rstd, gam, bet, None, True, 1, want_mask=True)
