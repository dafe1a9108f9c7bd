#!/usr/bin/env python3
"""Where does the bf16 product first leave the storage-rounding oracle?  Stage-by-stage comparison on one res5 block (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import seeded
from coin_amd import layers as L
from coin_amd.modeling.backbone import Bottleneck
from oracle import coin as OC

L.CONV_GEMM.update(enabled=True, min_rows=0, wgrad=True)
x = seeded.randn((64, 1024, 14, 14), 502)
blk = seeded.fill_module(Bottleneck(1024, 512, 2), 7).cuda().to(memory_format=torch.channels_last).train()
ob = seeded.fill_module(OC.Bottleneck(1024, 512, 2), 7).double().train()
l2 = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
mx = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().cpu().abs().max())
R = lambda t: t.to(torch.bfloat16).to(torch.float64)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    xb = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xo = R(x.double())
    print("input", l2(xb, xo))
    # conv1
    z1, part = L.conv2d_gemm(xb, blk.conv1, stats_rows=64 * 196)
    zo = R(torch.nn.functional.conv2d(xo, R(ob.conv1.weight)))
    print("conv1 out", l2(z1, zo), mx(z1, zo), "exact-equal fraction", float((z1.double().cpu() == zo).double().mean()))
    y1 = L.bn_act(z1, blk.bn1, True, None, 1, stats_part=(part, 64 * 196))
    yo_pre = ob.bn1(zo)
    yo = R(torch.relu(yo_pre))
    print("bn1+relu", l2(y1, yo), mx(y1, yo), "equal fraction", float((y1.double().cpu() == yo).double().mean()))
    # same BN fed with the PRODUCT's conv output (isolates the BN stage)
    yo2 = R(torch.relu(seeded.fill_module(OC.Bottleneck(1024, 512, 2), 7).double().train().bn1(z1.double().cpu())))
    print("bn1+relu given the product's conv1 output", l2(y1, yo2), "equal fraction", float((y1.double().cpu() == yo2).double().mean()))
    z2, part2 = L.conv2d_gemm(y1, blk.conv2, stats_rows=64 * 196)
    zo2 = R(torch.nn.functional.conv2d(y1.double().cpu(), R(ob.conv2.weight), padding=1))
    print("conv2 (3x3) given the product's input", l2(z2, zo2), "equal fraction", float((z2.double().cpu() == zo2).double().mean()))
    y2 = L.bn_act(z2, blk.bn2, True, None, 2, stats_part=(part2, 64 * 196))
    yo3 = R(torch.nn.functional.avg_pool2d(torch.relu(ob.bn2(z2.double().cpu())), 2))
    print("bn2+relu+pool given the product's conv2 output", l2(y2, yo3), "equal fraction", float((y2.double().cpu() == yo3).double().mean()))
    xp = L.avg_pool2(xb)
    print("avgpool(x)", l2(xp, R(torch.nn.functional.avg_pool2d(xo, 2))))
    zd, partd = L.conv2d_gemm(xp, blk.downsample[1], stats_rows=64 * 49)
    sx = L.bn_act(zd, blk.downsample[2], False, None, 1, stats_part=(partd, 64 * 49))
    sxo = R(ob.downsample[2](zd.double().cpu()))
    print("downsample bn given the product's conv", l2(sx, sxo), "equal fraction", float((sx.double().cpu() == sxo).double().mean()))
    z3, part3 = L.conv2d_gemm(y2, blk.conv3, stats_rows=64 * 49)
    out = L.bn_act(z3, blk.bn3, True, sx, 1, stats_part=(part3, 64 * 49))
    oo = R(torch.relu(ob.bn3(z3.double().cpu()) + sx.double().cpu()))
    print("bn3+res+relu given the product's inputs", l2(out, oo), "equal fraction", float((out.double().cpu() == oo).double().mean()))
    outm = L.bn_act(z3, seeded.fill_module(Bottleneck(1024, 512, 2), 7).cuda().train().bn3, True, sx, 0, stats_part=(part3, 64 * 49))
    om = R(torch.relu(ob.bn3(z3.double().cpu()) + sx.double().cpu()).mean(dim=[2, 3], keepdim=True))
    print("mean-pool tail given the product's inputs", l2(outm, om), "equal fraction", float((outm.double().cpu() == om).double().mean()))

# ---- whole blocks: the product's Bottleneck.forward against the oracle's emulated forward on the same (bf16) input
def whole(inpl, planes, stride, hw, seed, mean_pool=False):
    pb = seeded.fill_module(Bottleneck(inpl, planes, stride), seed).cuda().to(memory_format=torch.channels_last).train()
    oo = seeded.fill_module(OC.Bottleneck(inpl, planes, stride), seed).double().train()
    xi = seeded.randn((64, inpl, hw, hw), seed + 100)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        yp = pb(xi.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last), mean_pool=mean_pool)
    with torch.no_grad(), OC.emulate_rounding(torch.bfloat16):
        yo = oo(xi.double(), mean_pool=mean_pool)
    with torch.no_grad():
        yplain = seeded.fill_module(OC.Bottleneck(inpl, planes, stride), seed).double().train()(xi.double(), mean_pool=mean_pool)
    print(f"whole block {inpl}->{planes} stride {stride} mean_pool={mean_pool}: product vs emulated oracle L2 {l2(yp, yo):.2e} (equal fraction {float((yp.double().cpu() == yo).double().mean()):.4f}); vs plain fp64 {l2(yp, yplain):.2e}")

whole(1024, 512, 2, 14, 7)
whole(2048, 512, 1, 7, 8)
whole(2048, 512, 1, 7, 9, mean_pool=True)
