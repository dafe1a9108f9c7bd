"""FPN / Swin extension on the device (PARITY UNPINNED -- no counterpart in the reference, see tests/test_fpn_cpu.py):
the MFMA window-attention kernel against the fp32 formulation, and the full-size BASELINE.json configs[3] / configs[4] shapes as
property tests."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("heads,nw,batch,with_mask", [(3, 1, 5, False), (6, 6, 2, True), (12, 4, 3, True), (24, 1, 1, False)])
def test_window_attention_kernel_vs_fp32_reference(heads, nw, batch, with_mask):
    """coin_window_attn_fwd (QK^T, bias + shift mask, softmax, PV on MFMA; bf16 operands, fp32 scores) vs the fp64 oracle loops on the
    same bf16 inputs.  Tolerance: the probabilities enter the PV product as bf16 (2^-9 relative), the output is stored as bf16."""
    from coin_amd import kernels as K
    from coin_amd.modeling.swin import _pad64, shift_mask
    from oracle import fpn as O

    g = torch.Generator().manual_seed(heads)
    b, t = batch * nw, 49
    qkv = (torch.randn(b, t, 3 * heads * 32, generator=g) * 1.5).to(torch.bfloat16)
    bias = torch.randn(heads, t, t, generator=g)
    mask = shift_mask(14, 7 * nw // 2 if nw % 2 == 0 else 7, 7, 3, "cpu") if with_mask else None
    if with_mask:
        assert mask.shape[0] == nw, mask.shape
    out = K.window_attn_fwd(qkv.cuda(), _pad64(bias, -1e30).cuda(), _pad64(mask, 0.0).cuda() if with_mask else None, heads, 32 ** -0.5)
    ref = O.window_attention(qkv.float().numpy(), bias.numpy(), None if mask is None else mask.numpy(), heads, 32 ** -0.5)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2e-2, atol=2e-2)


def test_window_attention_function_backward_matches_the_differentiable_formulation():
    from coin_amd.modeling.swin import _pad64, shift_mask, window_attention, window_attention_reference

    torch.manual_seed(0)
    heads, nw = 3, 6
    qkv = torch.randn(2 * nw, 49, 3 * heads * 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    bias = torch.randn(heads, 49, 49, device="cuda", requires_grad=True)
    mask = shift_mask(14, 21, 7, 3, "cuda")
    dout = torch.randn(2 * nw, 49, heads * 32, device="cuda").to(torch.bfloat16)
    window_attention(qkv, bias, mask, _pad64(mask, 0.0), heads, 32 ** -0.5).backward(dout)
    g1, b1 = qkv.grad.clone(), bias.grad.clone()
    qkv.grad = bias.grad = None
    window_attention_reference(qkv, bias, mask, heads, 32 ** -0.5).backward(dout)
    torch.testing.assert_close(g1.float(), qkv.grad.float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(b1, bias.grad, rtol=1e-3, atol=1e-3)


def _cfg(yaml_name, extra=()):
    from coin_amd.config import get_cfg

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coin", "FPN", yaml_name))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda:0", "AMD.TEXT_TEMPLATES", 2] + list(extra))
    return cfg


def test_swint_fpn_student_full_size_targetdet_steps():
    """BASELINE.json configs[4] shape: Swin-T-FPN student + EMA teacher, Cityscapes-shaped 667x1333 views, bf16, window attention on the
    HIP kernel: two `CoinTrainer.run_step`s give finite losses, and the Swin parameters (incl. the relative-position-bias tables) move."""
    from coin_amd.engine import CoinTrainer

    cfg = _cfg("targetdet_swint_fpn_synthetic.yaml", ["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "CLOUD.BURN_UP_STEP", 10 ** 6,
                                                      "CLOUD.PROTOTYPE_UPDATE_START", 0])
    torch.manual_seed(1)
    tr = CoinTrainer(cfg)
    before = {n: p.detach().clone() for n, p in tr.model.backbone.bottom_up.named_parameters() if "stages.2.1" in n}
    recs = [{k: float(v) for k, v in tr.run_step().items()} for _ in range(2)]
    assert all(math.isfinite(v) for r in recs for v in r.values()), recs
    moved = [n for n, p in tr.model.backbone.bottom_up.named_parameters() if n in before and not torch.equal(p.detach(), before[n])]
    assert any("relative_position_bias_table" in n for n in moved) and any("qkv.weight" in n for n in moved), moved


def test_rn101_fpn_student_full_size_targetdet_step():
    """BASELINE.json configs[3] names a ResNet-101-FPN student: CLIP-RN101 bottom-up + FPN, BDD100K-shaped 750x1333 views, 2 images."""
    from coin_amd.engine import CoinTrainer

    cfg = _cfg("targetdet_rn101_fpn_synthetic.yaml", ["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "CLOUD.BURN_UP_STEP", 10 ** 6,
                                                      "CLOUD.PROTOTYPE_UPDATE_START", 0])
    torch.manual_seed(2)
    tr = CoinTrainer(cfg)
    rec = {k: float(v) for k, v in tr.run_step().items()}
    assert all(math.isfinite(v) for v in rec.values()), rec
    assert {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"} <= set(rec)


def test_rn_fpn_backbone_bf16_device_vs_fp32_cpu_restated_kernels():
    """The CLIP-ResNet-FPN backbone (eval mode: frozen / running-statistics norms) on the device in bf16 (fused BN kernels, channels-last)
    against the same weights on the CPU in fp32 through tests/cpu_shim (torch formulations of the kernels): p2..p6 agree to bf16 accuracy."""
    from cpu_shim import cpu_kernels
    from coin_amd.modeling import build_model

    extra = ["AMD.ARCH.LAYERS", (1, 2, 1, 1), "AMD.ARCH.WIDTH", 16, "AMD.TEXT_TEMPLATES", 1]
    torch.manual_seed(11)
    mg = build_model(_cfg("CLIPDET_rn50_fpn_synthetic.yaml", extra))
    mg = (mg[0] if isinstance(mg, tuple) else mg).eval()
    with cpu_kernels():
        mc = build_model(_cfg("CLIPDET_rn50_fpn_synthetic.yaml", extra + ["MODEL.DEVICE", "cpu", "AMD.COMPUTE_DTYPE", "fp32"]))
        mc = (mc[0] if isinstance(mc, tuple) else mc).eval()
        mc.load_state_dict({k: v.cpu() for k, v in mg.state_dict().items()})
        img = torch.randint(0, 256, (3, 250, 331), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            fc = mc.backbone(mc.preprocess_image([{"image": img}]).tensor)
    with torch.no_grad():
        fg = mg.backbone(mg.preprocess_image([{"image": img.cuda()}]).tensor)
    assert set(fc) == set(fg) == {"p2", "p3", "p4", "p5", "p6"}
    for k in fc:
        a, b = fg[k].float().cpu(), fc[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert float((a - b).abs().max()) <= 0.05 * float(b.abs().max()) + 0.02, (k, float((a - b).abs().max()), float(b.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------------------
# Device path against the INDEPENDENT restatement (oracle/fpn.py, float64, explicit loops) at real widths, forward AND gradients
# (round-3 VERDICT item 4).  Parity unpinned: the reference has none of these modules.
def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _oracle_grads(loss, tensors):
    return torch.autograd.grad(loss, tensors, allow_unused=False)


@pytest.mark.parametrize("heads,nw,batch,with_mask", [(3, 1, 5, False), (6, 6, 2, True), (12, 4, 3, True), (24, 1, 2, False)])
def test_window_attention_backward_kernel_vs_fp64_autograd(heads, nw, batch, with_mask):
    """coin_window_attn_bwd (P recomputed, dP / dV / dQ / dK on MFMA, d bias summed over the windows in chunk order) against autograd
    through the float64 loops of oracle.fpn.window_attention_t on the same bf16 inputs; run twice: bit-identical (no atomics)."""
    from coin_amd import kernels as K
    from coin_amd.modeling.swin import _pad64, shift_mask
    from oracle import fpn as O

    g = torch.Generator().manual_seed(100 + heads)
    b, t = batch * nw, 49
    qkv = (torch.randn(b, t, 3 * heads * 32, generator=g) * 1.2).to(torch.bfloat16)
    bias = torch.randn(heads, t, t, generator=g)
    dout = torch.randn(b, t, heads * 32, generator=g).to(torch.bfloat16)
    mask = shift_mask(14, 7 * nw // 2 if nw % 2 == 0 else 7, 7, 3, "cpu") if with_mask else None
    m64 = _pad64(mask, 0.0).cuda() if with_mask else None
    dq, db = K.window_attn_bwd(qkv.cuda(), _pad64(bias, -1e30).cuda(), m64, dout.cuda(), heads, 32 ** -0.5)
    dq2, db2 = K.window_attn_bwd(qkv.cuda(), _pad64(bias, -1e30).cuda(), m64, dout.cuda(), heads, 32 ** -0.5)
    assert torch.equal(dq, dq2) and torch.equal(db, db2)
    q64 = qkv.double().requires_grad_(True)
    b64 = bias.double().requires_grad_(True)
    out = O.window_attention_t(q64, b64, None if mask is None else mask.double(), heads, 32 ** -0.5)
    gq, gb = torch.autograd.grad(out, (q64, b64), dout.double())
    assert dq.shape == gq.shape and db.shape == gb.shape
    # P and dS enter the MFMA products as bf16 (2^-9 relative), the outputs are stored as bf16
    assert _rel(dq.float(), gq) < 2e-2, _rel(dq.float(), gq)
    assert _rel(db, gb) < 1e-2, _rel(db, gb)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fpn_neck_on_device_vs_oracle_forward_and_gradients(dtype):
    """FPN neck at the real widths (CLIP-RN50 stages 256 / 512 / 1024 / 2048 -> 256-channel pyramid), odd map sizes."""
    from coin_amd.modeling.fpn import FPN
    from oracle import fpn as O

    torch.manual_seed(21)
    chans, sizes = [256, 512, 1024, 2048], [(29, 41), (15, 21), (8, 11), (4, 6)]
    neck = FPN(["res2", "res3", "res4", "res5"], chans, out_channels=256)
    feats = {f"res{i + 2}": torch.randn(2, c, *s) * 0.5 for i, (c, s) in enumerate(zip(chans, sizes))}
    wts = {k: torch.randn(2, 256, *s) for k, s in zip(["p2", "p3", "p4", "p5"], sizes)}
    wts["p6"] = torch.randn(2, 256, (sizes[3][0] + 1) // 2, (sizes[3][1] + 1) // 2)
    # oracle, float64
    f64 = {k: v.double().requires_grad_(True) for k, v in feats.items()}
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in neck.state_dict().items()}
    ref = O.fpn_forward(f64, sd64)
    ref_loss = sum((ref[k] * wts[k].double()).sum() for k in ref)
    names = sorted(sd64)
    gref = _oracle_grads(ref_loss, [f64[k] for k in sorted(f64)] + [sd64[k] for k in names])
    # device
    neck = neck.cuda()
    cd = torch.bfloat16 if dtype == "bf16" else torch.float32
    fd = {k: v.cuda().to(cd).contiguous(memory_format=torch.channels_last).requires_grad_(True) for k, v in feats.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        out = neck(fd)
    loss = sum((out[k].float() * wts[k].cuda()).sum() for k in out)
    loss.backward()
    ftol, gtol = (2e-5, 2e-4) if dtype == "fp32" else (2e-2, 4e-2)
    for k in ref:
        assert out[k].shape == ref[k].shape
        assert _rel(out[k].float(), ref[k]) < ftol, (k, _rel(out[k].float(), ref[k]))
    got = [fd[k].grad for k in sorted(fd)] + [dict(neck.named_parameters())[k].grad for k in names]
    for name, a, b in zip(sorted(fd) + names, got, gref):
        assert a is not None and _rel(a.float(), b) < gtol, (name, _rel(a.float(), b))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_multilevel_pooler_and_two_fc_on_device_vs_oracle_forward_and_gradients(dtype):
    """coin_roi_align_fwd_levels (level index per RoI inside ONE launch) + coin_roi_align_bwd_level + the 2-FC head on the MFMA GEMM, 256-channel
    pyramid, RoIs of every size class, against oracle.fpn.multilevel_roi_align (one RoI at a time on its level) + two_fc."""
    from coin_amd.modeling.fpn import MultiLevelROIPooler, TwoFCHead, assign_levels
    from oracle import fpn as O

    torch.manual_seed(22)
    strides = [4, 8, 16, 32]
    feats = [torch.randn(2, 256, 200 // s + 1, 320 // s + 1) * 0.5 for s in strides]
    g = torch.Generator().manual_seed(3)
    n = 96
    xy = torch.rand(n, 2, generator=g) * torch.tensor([250.0, 150.0])
    wh = torch.exp(torch.rand(n, 1, generator=g) * 5.0 + 2.0) * (0.6 + 0.8 * torch.rand(n, 2, generator=g))
    rois = torch.cat([torch.randint(0, 2, (n, 1), generator=g).float(), xy, xy + wh], dim=1)
    assert set(assign_levels(rois[:, 1:]).tolist()) == {0, 1, 2, 3}, "the test must exercise every level"
    pool = MultiLevelROIPooler(7, [1.0 / s for s in strides], 0, min_level=2)
    head = TwoFCHead(256 * 49, 1024)
    w_out = torch.randn(n, 1024, generator=g)
    f64 = [f.double().requires_grad_(True) for f in feats]
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in head.state_dict().items()}
    pooled_ref = O.multilevel_roi_align(f64, strides, rois, 7)
    ref = O.two_fc(pooled_ref, sd64)
    names = sorted(sd64)
    gref = _oracle_grads((ref * w_out.double()).sum(), f64 + [sd64[k] for k in names])
    head = head.cuda()
    cd = torch.bfloat16 if dtype == "bf16" else torch.float32
    fd = [f.cuda().to(cd).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        pooled = pool(fd, rois.cuda())
        out = head(pooled)
    (out.float() * w_out.cuda()).sum().backward()
    ftol, gtol = (2e-5, 3e-4) if dtype == "fp32" else (2e-2, 4e-2)
    assert _rel(pooled.float(), pooled_ref) < (1e-5 if dtype == "fp32" else 1e-2), _rel(pooled.float(), pooled_ref)
    assert _rel(out.float(), ref) < ftol, _rel(out.float(), ref)
    got = [f.grad for f in fd] + [dict(head.named_parameters())[k].grad for k in names]
    # bf16: a hidden unit whose pre-activation (a 12 544-term bf16 dot product) lies within rounding of zero takes the other ReLU branch
    # than the float64 oracle; ~3 of the ~500 active units of a row flip, i.e. sqrt(3 / 500) ~ 8 % of the gradient's L2 norm is inherent to
    # the precision, not to the kernels -- so the bf16 run holds the gradients THROUGH THE HEAD to 15 % in L2 and checks the pooler's own
    # backward (no ReLU behind it) separately and tightly below; the fp32 run holds everything to 3e-4 in the max norm
    for name, a, b in zip([f"p{i + 2}" for i in range(4)] + names, got, gref):
        e = _rel(a.float(), b) if dtype == "fp32" else _rel_l2(a.float(), b)
        assert a is not None and e < (gtol if dtype == "fp32" else 0.15), (name, e)
    if dtype == "bf16":
        w_pool = torch.randn(n, 256, 7, 7, generator=g)
        gref_p = _oracle_grads((O.multilevel_roi_align(f64, strides, rois, 7) * w_pool.double()).sum(), f64)
        for f in fd:
            f.grad = None
        (pool(fd, rois.cuda()).float() * w_pool.cuda()).sum().backward()
        for i, (f, b) in enumerate(zip(fd, gref_p)):
            assert _rel(f.grad.float(), b) < 1e-2, (f"pooler only, p{i + 2}", _rel(f.grad.float(), b))


@pytest.mark.parametrize("dim,heads,shift,dtype", [(96, 3, 0, "bf16"), (96, 3, 3, "bf16"), (192, 6, 3, "bf16"), (96, 3, 3, "fp32")])
def test_swin_block_on_device_vs_oracle_forward_and_gradients(dim, heads, shift, dtype):
    """One Swin block at the Swin-T stage widths (96 / 192), un-shifted and shifted, on a map that needs window padding; bf16 = the MFMA
    window-attention forward AND backward kernels inside the block; against oracle.fpn.swin_block (windows cut one by one, float64)."""
    from coin_amd.modeling.swin import SwinBlock
    from oracle import fpn as O

    torch.manual_seed(23 + dim + shift)
    blk = SwinBlock(dim, heads, 7, shift)
    with torch.no_grad():
        blk.relative_position_bias_table.mul_(20.0)       # (the 0.02 initialisation would make the bias gradient check vacuous)
    x = torch.randn(2, 20, 27, dim)
    w_out = torch.randn(2, 20, 27, dim)
    x64 = x.double().requires_grad_(True)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in blk.state_dict().items()}
    ref = O.swin_block(x64, sd64, heads, 7, shift, blk.relative_position_index)
    names = sorted(sd64)
    gref = _oracle_grads((ref * w_out.double()).sum(), [x64] + [sd64[k] for k in names])
    blk = blk.cuda()
    cd = torch.bfloat16 if dtype == "bf16" else torch.float32
    xd = x.cuda().to(cd).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        out = blk(xd)
    (out.float() * w_out.cuda()).sum().backward()
    ftol, gtol = (3e-5, 3e-4) if dtype == "fp32" else (2e-2, 5e-2)
    assert _rel(out.float(), ref) < ftol, _rel(out.float(), ref)
    params = dict(blk.named_parameters())
    for name, a, b in zip(["x"] + names, [xd.grad] + [params[k].grad for k in names], gref):
        assert a is not None and _rel(a.float(), b) < gtol, (name, _rel(a.float(), b))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_patch_merging_on_device_vs_oracle_forward_and_gradients(dtype):
    from coin_amd.modeling.swin import PatchMerging
    from oracle import fpn as O

    torch.manual_seed(24)
    pm = PatchMerging(96)
    x = torch.randn(2, 21, 27, 96)                          # odd height and width: zero padding before the 2 x 2 gather
    w_out = torch.randn(2, 11, 14, 192)
    x64 = x.double().requires_grad_(True)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in pm.state_dict().items()}
    ref = O.patch_merging(x64, sd64)
    names = sorted(sd64)
    gref = _oracle_grads((ref * w_out.double()).sum(), [x64] + [sd64[k] for k in names])
    pm = pm.cuda()
    xd = x.cuda().to(torch.bfloat16 if dtype == "bf16" else torch.float32).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        out = pm(xd)
    (out.float() * w_out.cuda()).sum().backward()
    ftol, gtol = (3e-5, 3e-4) if dtype == "fp32" else (2e-2, 4e-2)
    assert _rel(out.float(), ref) < ftol, _rel(out.float(), ref)
    params = dict(pm.named_parameters())
    for name, a, b in zip(["x"] + names, [xd.grad] + [params[k].grad for k in names], gref):
        assert a is not None and _rel(a.float(), b) < gtol, (name, _rel(a.float(), b))


def test_rn50_fpn_pretrain_full_size_steps():
    """BASELINE.json configs[0] / [1] name a ResNet-50-FPN CLIPDET pre-training step: CLIP-RN50 bottom-up + FPN, 800x1333 views, 512 RoIs
    per view, bf16 -- two `PRETrainer.run_step`s give finite losses and move the FPN / 2-FC / backbone parameters."""
    from coin_amd.engine import PRETrainer

    cfg = _cfg("CLIPDET_rn50_fpn_synthetic.yaml", ["SOLVER.IMG_PER_BATCH_UNLABEL", 2, "AMD.SYNTHETIC.NUM_IMAGES", 2, "AMD.SYNTHETIC.HEIGHT", 800,
                                                   "AMD.SYNTHETIC.WIDTH", 1333, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 512, "SOLVER.WARMUP_FACTOR", 1.0])
    torch.manual_seed(3)
    tr = PRETrainer(cfg)
    watch = {n: p.detach().clone() for n, p in tr.model.named_parameters()
             if n in ("backbone.fpn.fpn_output3.weight", "roi_heads.box_head.fc1.weight", "backbone.bottom_up.layer3.0.conv1.weight")}
    assert len(watch) == 3, list(watch)
    recs = [{k: float(v) for k, v in tr.run_step().items()} for _ in range(2)]
    assert all(math.isfinite(v) for r in recs for v in r.values()), recs
    params = dict(tr.model.named_parameters())
    assert all(not torch.equal(params[n].detach(), w) for n, w in watch.items()), [n for n, w in watch.items() if torch.equal(params[n].detach(), w)]
