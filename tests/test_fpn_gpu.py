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
