"""CPU fp32 restatement of the reference's hot-path modules (TEST INFRASTRUCTURE - see oracle/__init__.py).

Module tree and parameter names equal the reference's state-dict keys, so one set of weights
drives the reference (golden capture), this oracle and the HIP product.  Each class cites the
reference lines it restates.  Pinned by tests/golden/*.npz via tests/test_oracle_golden.py.

Differences from the reference that do not change values on CPU fp32:
  * the frozen text encoder is kept in fp32 (reference: fp16 weights, clip_text.py:137) and the final
    normalisation is out-of-place (reference: in-place `x /= norm`, clip_text.py:204);
  * no logging / event storage / visualisation; no mask head.
"""
from __future__ import annotations

import copy
import math
from bisect import bisect_right
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import d2
from . import losses as L
from .d2 import Boxes, Instances, cat

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]   # coin/config.py:59
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]   # coin/config.py:60


# --------------------------------------------------------------------------- storage-rounding emulation (bf16 throughput mode)
# The product's bf16 mode keeps fp32 masters and accumulates in fp32, but STORES activations, weights' compute copies and the outputs
# of its GEMM convolutions in bf16.  `emulate_rounding(torch.bfloat16)` makes this oracle -- run in fp64 -- round at exactly those
# points (coin_amd/layers.py: conv inputs / weight shadows / outputs, the fused BatchNorm + residual + ReLU (+ pool / mean) stores, the
# box head's linear layers), with a straight-through gradient: the forward is then the function the bf16 kernels compute up to
# accumulation order, and an output that differs from it by more than a few bf16 ulps is a kernel error, not "bf16 noise".
# (The backward's own roundings -- gradients are stored in bf16 too -- are not emulated: gradients agree to ~1e-2, not to an ulp.)
ROUND = {"dtype": None}


class _RoundSTE(torch.autograd.Function):
    """x rounded to the storage dtype (value kept in x's own dtype).  Backward: the gradient passes straight through -- or, with
    `round_grad`, is rounded to the storage dtype as well: the bf16 mode STORES the gradient of every stored activation in bf16 too
    (the data-gradient GEMM's output rows, coin_bn_bwd's dx / d_residual; weight gradients and the norm's dgamma / dbeta stay fp32)."""

    @staticmethod
    def forward(ctx, x, dtype, round_grad):
        ctx.gdtype = dtype if round_grad else None
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        if ctx.gdtype is not None:
            g = g.to(ctx.gdtype).to(g.dtype)
        return g, None, None


def rnd(x, grad=True):
    """grad=False: an edge whose gradient the product never materialises (a weight: its gradient is fp32; the un-pooled activation
    inside the fused BatchNorm + ReLU + pool store)."""
    return x if ROUND["dtype"] is None else _RoundSTE.apply(x, ROUND["dtype"], bool(ROUND.get("grads")) and grad)


class emulate_rounding:
    """Inside: the oracle rounds where the bf16 mode stores (forward); with grads=True also where its BACKWARD stores."""

    def __init__(self, dtype, grads=False):
        self.dtype, self.grads = dtype, grads

    def __enter__(self):
        self.prev = (ROUND["dtype"], ROUND.get("grads", False))
        ROUND["dtype"], ROUND["grads"] = self.dtype, self.grads

    def __exit__(self, *a):
        ROUND["dtype"], ROUND["grads"] = self.prev
        return False


def _conv(conv: nn.Conv2d, x):
    """conv(x); under `emulate_rounding`: operands and result rounded (the input normally already is)."""
    if ROUND["dtype"] is None:
        return conv(x)
    return rnd(F.conv2d(rnd(x), rnd(conv.weight, grad=False), None if conv.bias is None else conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups))


def _linear(lin: nn.Linear, x, round_out=True):
    if ROUND["dtype"] is None:
        return lin(x)
    y = F.linear(rnd(x), rnd(lin.weight, grad=False), lin.bias)      # bias added in the fp32 accumulator, before the store
    return rnd(y) if round_out else y


# --------------------------------------------------------------------------- backbone (A2)
class Bottleneck(nn.Module):
    """coin/modeling/utils.py:26-90: 1x1 -> 3x3 -> avgpool(stride) -> 1x1, anti-aliased shortcut."""

    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=False)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != out:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)),
                ("0", nn.Conv2d(inplanes, out, 1, stride=1, bias=False)),
                ("1", nn.BatchNorm2d(out)),
            ]))

    def forward(self, x, mean_pool: bool = False):
        """mean_pool (the RoI head's last block, clip_roi_heads.py:207-208): the spatial mean [N, C, 1, 1] of the output."""
        if ROUND["dtype"] is None:
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.relu(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(self.avgpool(y)))
            sc = x if self.downsample is None else self.downsample(x)
            out = self.relu(y + sc)
            return out.mean(dim=[2, 3], keepdim=True) if mean_pool else out
        # the product's stores (coin_amd/modeling/backbone.py:Bottleneck.forward + layers.conv_bn_act): every convolution output; BatchNorm +
        # ReLU fused with the anti-aliasing pool (the kernel rounds the un-pooled value "as nn.AvgPool2d sees it", then the pooled one --
        # found with tools/round_debug.py (round 5; in the git history): every other stage matched to 2e-5, this one to 2e-3 until the oracle rounded twice as well);
        # the downsample branch's pooled input, convolution and norm;
        # bn3 + identity + ReLU in one store -- or, for the last block of the RoI head, only the spatial mean of it
        x = rnd(x)
        y = rnd(self.relu(self.bn1(_conv(self.conv1, x))))
        y = rnd(self.avgpool(rnd(self.relu(self.bn2(_conv(self.conv2, y))), grad=False)))   # coin_bn_apply_fwd rounds the un-pooled activation, then pools (backward: coin_bn_bwd reads the POOLED gradient, the un-pooled one is never stored)
        z = _conv(self.conv3, y)
        if self.downsample is None:
            sc = x
        else:
            sc = rnd(self.downsample[2](_conv(self.downsample[1], rnd(self.downsample[0](x)))))
        out = self.relu(self.bn3(z) + sc)
        return rnd(out.mean(dim=[2, 3], keepdim=True)) if mean_pool else rnd(out)


class ModifiedResNet(nn.Module):
    """coin/modeling/utils.py:129-292 (C4 use: res4 out, layer4 kept for the RoI head)."""

    def __init__(self, layers, width=64, freeze_at=0):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self.relu = nn.ReLU(inplace=False)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        self.out_channels = width * 16  # res4
        self.freeze(freeze_at)

    def _make_layer(self, planes, blocks, stride=1):
        mods = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * Bottleneck.expansion
        mods += [Bottleneck(self._inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def freeze(self, freeze_at: int):
        """utils.py:243-284: stages < freeze_at get requires_grad=False and FrozenBatchNorm."""
        def fz(m):
            for p in m.parameters():
                p.requires_grad = False
            return d2.FrozenBatchNorm2d.convert_frozen_batchnorm(m)

        if freeze_at >= 1:
            for name in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3"):
                setattr(self, name, fz(getattr(self, name)))
        for idx, stage in enumerate([self.layer1, self.layer2, self.layer3, self.layer4], start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    fz(block)
        return self

    def forward(self, x):
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            x = self.relu(bn(conv(x)))
        x = self.avgpool(x)
        x = self.layer3(self.layer2(self.layer1(x)))
        return {"res4": x}

    def output_shape(self):
        return {"res4": d2.ShapeSpec(channels=self.out_channels, stride=16)}


class _Encoder(nn.Module):
    def __init__(self, visual):
        super().__init__()
        self.visual = visual
        self.attnpool = None


class ClipImageBackbone(nn.Module):
    """coin/modeling/backbone/clip_backbone.py:150-283 (CLIP_IMAGE) with random init instead of a download.
    `bn3.weight` of every block is zero-initialised as CLIP does (clip_backbone.py:56-61)."""

    size_divisibility = 0

    def __init__(self, layers=(3, 4, 6, 3), width=64, freeze_at=2, update_backbone=True, zero_init_bn3=True):
        super().__init__()
        self.encoder = _Encoder(ModifiedResNet(layers, width, freeze_at))
        self.update_backbone = update_backbone
        if zero_init_bn3:
            for n, p in self.encoder.visual.named_parameters():
                if n.endswith("bn3.weight") and "layer" in n:
                    nn.init.zeros_(p)
        if not update_backbone:
            for n, p in self.encoder.visual.named_parameters():
                if "layer4" not in n:
                    p.requires_grad = False

    layer4 = property(lambda self: self.encoder.visual.layer4)
    attnpool = property(lambda self: self.encoder.attnpool)

    def output_shape(self):
        return self.encoder.visual.output_shape()

    def train(self, mode: bool = True):  # clip_backbone.py:223-234
        if self.update_backbone:
            return super().train(mode)
        self.training = False
        self.encoder.training = False
        for m in self.encoder.children():
            m.eval()
        self.encoder.visual.layer4.train(mode)
        return self

    def forward(self, x):
        return self.encoder.visual(x)


# --------------------------------------------------------------------------- text encoder (A8)
class _Attn(nn.Module):
    """Parameter layout of nn.MultiheadAttention (in_proj_weight/in_proj_bias/out_proj) with explicit math."""

    def __init__(self, d, heads):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        self.heads = heads

    def forward(self, x, mask):  # x [L, N, D]
        l, n, d = x.shape
        hd = d // self.heads
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)
        q, k, v = qkv.chunk(3, dim=-1)

        def split(t):
            return t.reshape(l, n * self.heads, hd).transpose(0, 1)  # [N*h, L, hd]

        q, k, v = split(q), split(k), split(v)
        a = torch.baddbmm(mask.to(q.dtype), q * (hd ** -0.5), k.transpose(1, 2))
        a = torch.softmax(a, dim=-1)
        o = torch.bmm(a, v).transpose(0, 1).reshape(l, n, d)
        return self.out_proj(o)


class _Block(nn.Module):
    """coin/modeling/utils.py:309-330 ResidualAttentionBlock (QuickGELU MLP)."""

    def __init__(self, d, heads):
        super().__init__()
        self.attn = _Attn(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, d * 4)), ("gelu", nn.Identity()), ("c_proj", nn.Linear(d * 4, d))]))
        self.ln_2 = nn.LayerNorm(d)

    def forward(self, x, mask):
        x = x + self.attn(self.ln_1(x), mask)
        h = self.mlp.c_fc(self.ln_2(x))
        h = h * torch.sigmoid(1.702 * h)
        return x + self.mlp.c_proj(h)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])

    def forward(self, x, mask):
        for b in self.resblocks:
            x = b(x, mask)
        return x


class TextEncoder(nn.Module):
    """coin/modeling/text_encoder/clip_text.py:31-205 (TEXT_ENCODER): frozen CLIP text transformer with a
    learnable prompt 'SOS | embedding_tmp | add_in_embedding | class token | EOS...'."""

    def __init__(self, embed_dim, context_length, vocab_size, width, heads, layers, tokenized_prompts, prompt_tmp_len, add_prompt_num):
        super().__init__()
        self.context_length = context_length
        self.transformer = _Transformer(width, layers, heads)
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.tokenized_prompts = tokenized_prompts
        self.prompt_tmp_len, self.add_prompt_num = prompt_tmp_len, add_prompt_num
        self._init()
        self.load_embedding(width)
        self.freeze_encoder()

    def _init(self):  # clip_text.py:65-79
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, nl = self.transformer.width, self.transformer.layers
        proj_std, attn_std, fc_std = (w ** -0.5) * ((2 * nl) ** -0.5), w ** -0.5, (2 * w) ** -0.5
        for b in self.transformer.resblocks:
            nn.init.normal_(b.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(b.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(b.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(b.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    def load_embedding(self, width):  # clip_text.py:148-159
        with torch.no_grad():
            emb = self.token_embedding(self.tokenized_prompts.long())
        t, a = self.prompt_tmp_len, self.add_prompt_num
        self.sos = nn.Parameter(emb[0, :1, :].clone(), requires_grad=False)
        self.embedding_tmp = nn.Parameter(emb[0, 1:1 + t, :].clone().float(), requires_grad=True)
        self.register_buffer("embedding_class", emb[:, 1 + t + a:2 + t + a, :].clone())
        self.eos = nn.Parameter(emb[0, 2 + t + a:, :].clone(), requires_grad=False)
        v = torch.empty(a, width)
        nn.init.normal_(v, std=0.02)
        self.add_in_embedding = nn.Parameter(v, requires_grad=True)

    def freeze_encoder(self):  # clip_text.py:89-98
        for p in self.token_embedding.parameters():
            p.requires_grad = False
        for p in self.ln_final.parameters():
            p.requires_grad = False
        for p in self.transformer.parameters():
            p.requires_grad = False
        self.positional_embedding.requires_grad = False
        self.text_projection.requires_grad = False
        self.logit_scale.requires_grad = False

    def _mask(self, device):
        m = torch.full((self.context_length, self.context_length), float("-inf"), device=device)
        return m.triu_(1)

    def forward(self, text, add: bool):  # clip_text.py:165-205
        if add:
            n = self.embedding_class.size(0)
            ex = lambda p: p.unsqueeze(0).expand(n, -1, -1)
            x = torch.cat([ex(self.sos), ex(self.embedding_tmp), ex(self.add_in_embedding), self.embedding_class, ex(self.eos)], dim=1)
            eot = self.tokenized_prompts.argmax(dim=-1)
        else:
            x = self.token_embedding(text.long())
            eot = text.argmax(dim=-1)
        x = x + self.positional_embedding
        x = self.transformer(x.permute(1, 0, 2), self._mask(x.device)).permute(1, 0, 2)
        x = self.ln_final(x)
        x = x[torch.arange(x.shape[0]), eot.to(x.device).long()] @ self.text_projection
        return x / torch.norm(x, dim=-1, keepdim=True)


class ClipText(nn.Module):
    """coin/modeling/text_encoder/clip_text.py:209-327 (CLIP_TEXT): encoder + fixed class embeddings + prototypes."""

    def __init__(self, encoder: TextEncoder, classes: Sequence[str], per_class_feat: torch.Tensor):
        super().__init__()
        self.encoder = encoder
        self.classes = list(classes)
        f = per_class_feat / per_class_feat.norm(dim=1, keepdim=True)
        self.register_buffer("per_class_feat", f)
        self.register_buffer("prototype_b_online", f.clone())
        self.register_buffer("prototype_b_offline", f.clone())

    num_classes = property(lambda self: len(self.classes))
    prototype = property(lambda self: self.per_class_feat)

    def train(self, mode: bool = True):  # clip_text.py:296-302: always eval
        self.training = False
        for m in self.children():
            m.eval()
        return self

    def forward(self, added: bool):
        return self.encoder(None, add=True) if added else self.per_class_feat


# --------------------------------------------------------------------------- CKG merge net (A12)
class _CrossAttention(nn.Module):
    """coin/modeling/merge/ckg.py:36-82."""

    def __init__(self, hidden, all_head, num_classes, heads=8):
        super().__init__()
        self.num_heads, self.h_size = heads, all_head // heads
        self.linear_q = nn.Linear(hidden, all_head, bias=False)
        self.linear_k = nn.Linear(hidden, all_head, bias=False)
        self.linear_v = nn.Linear(hidden, all_head, bias=False)
        self.linear_output = nn.Linear(all_head, num_classes)
        for l in (self.linear_q, self.linear_k, self.linear_v, self.linear_output):
            nn.init.xavier_normal_(l.weight)
        nn.init.constant_(self.linear_output.bias, 0)

    def forward(self, x, y):
        sp = lambda t: t.view(1, -1, self.num_heads, self.h_size).transpose(1, 2)
        q, k, v = sp(self.linear_q(x)), sp(self.linear_k(y)), sp(self.linear_v(y))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.size(-1)), dim=-1) @ v
        return self.linear_output(a.transpose(1, 2).contiguous().view(-1, self.num_heads * self.h_size))


class CKGNet(nn.Module):
    """coin/modeling/merge/ckg.py:85-115."""

    def __init__(self, hidden_size, all_head_size, num_classes, head_num=8):
        super().__init__()
        self.cross_offline = _CrossAttention(hidden_size, all_head_size, num_classes, head_num)
        self.cross_online = _CrossAttention(hidden_size, all_head_size, num_classes, head_num)

    def forward(self, x, proto_off, proto_on, probs_off, probs_on):
        w_off = self.cross_offline(x, proto_off)
        w_on = self.cross_online(x, proto_on)
        return F.softmax(w_off * probs_off + w_on * probs_on, dim=1)


# --------------------------------------------------------------------------- box predictor (A7, A9, A10, A19)
def _xavier(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        nn.init.constant_(m.bias, 0)


def _prototype_ema(proto: torch.Tensor, feats: torch.Tensor, one_hot: torch.Tensor, rate: float) -> torch.Tensor:
    """fast_rcnn.py:405-412: per-class mean of unit features, lerp into the buffer for classes that occur."""
    dt = proto.dtype if proto.dtype == torch.float64 else torch.float32  # the reference computes this in fp32; fp64 only for the precision study
    new = proto.clone().to(dt)
    one_hot = one_hot.to(dt)
    cnt = one_hot.sum(0)
    present = cnt != 0
    new[present] = (one_hot.T @ feats.to(dt) / cnt.unsqueeze(1))[present]
    return proto * rate + (1 - rate) * new


class BoxPredictor(nn.Module):
    """coin/modeling/roi_heads/fast_rcnn.py:182-752 (FastRCNNOutputLayers), meanpool + class-agnostic boxes."""

    def __init__(self, input_size, text_encoder: ClipText, text_dim, classes_weight, loss_weight, batch_size_per_image,
                 cls_b_thresh=0.7, dataset=("foggytrain_0.02",), prototype_update_rate=0.9996, loss_type="MILCrossEntropy",
                 bbox_reg_weights=(10.0, 10.0, 5.0, 5.0), test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100):
        super().__init__()
        h = input_size // 2
        self.trans = nn.Sequential(nn.Linear(input_size, h), nn.LeakyReLU(), nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, input_size))
        self.cls_score = nn.Linear(input_size, text_dim)
        self.logit_scale = nn.Parameter(torch.FloatTensor([0.01]), requires_grad=False)
        self.bbox_pred = nn.Linear(input_size, 4)
        self.trans.apply(_xavier)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.cls_score.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.text_encoder = text_encoder
        self.num_classes = text_encoder.num_classes - 1
        self.box2box_transform = d2.Box2BoxTransform(bbox_reg_weights)
        self.classes_weight, self.loss_weight = list(classes_weight), dict(loss_weight)
        self.batch_size_per_image, self.cls_b_thresh = batch_size_per_image, cls_b_thresh
        self.dataset, self.prototype_update_rate, self.loss_type = tuple(dataset), prototype_update_rate, loss_type
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image

    # ---- forward (fast_rcnn.py:318-353)
    def forward(self, x, branch, return_feats=True):
        if ROUND["dtype"] is None:
            x = self.trans(torch.flatten(x, start_dim=1))
            feats = self.cls_score(x)
            deltas = self.bbox_pred(x)
        else:   # coin_amd/modeling/fast_rcnn.py:forward in the bf16 mode: five linear layers on coin_gemm_nt, bf16 stores except the fp32 deltas
            t = self.trans
            h = rnd(F.leaky_relu(_linear(t[0], torch.flatten(x, start_dim=1), round_out=False), 0.01))
            h = rnd(F.leaky_relu(_linear(t[2], h, round_out=False), 0.01))
            x = _linear(t[4], h)
            feats = _linear(self.cls_score, x)
            deltas = _linear(self.bbox_pred, x, round_out=False)
        scores = self.do_classify(feats, branch)
        if return_feats and self.training and branch != "test":
            return scores, deltas, feats
        return scores, deltas

    def do_classify(self, image_features, branch):
        t = self.text_encoder(added=True)
        t = t / t.norm(dim=1, keepdim=True)
        f = image_features / image_features.norm(dim=1, keepdim=True)
        scores = (f @ t.t()) / self.logit_scale
        if self.training and branch != "test":
            fixed = self.text_encoder(added=False).detach()
            fixed = fixed / fixed.norm(dim=1, keepdim=True)
            return scores, F.l1_loss(t, fixed)
        return scores

    # ---- helpers
    def _cls_loss(self, scores, target, n_fg, n_bg, avg_positives=True):
        if self.loss_type == "MILCrossEntropy":
            w = torch.cat([torch.ones(n_fg), torch.full((n_bg,), self.classes_weight[-1])]).to(scores.device)
            # the reference builds the weights with ones_like(int64 labels)*0.9 -> int64 truncation happens only for
            # `ones_like(classes_bg)*0.9`, which torch promotes to float; so the bg weight is 0.9 (fast_rcnn.py:462,579)
            return L.mil_cross_entropy(scores, target, weights=w, avg_positives=avg_positives)
        if self.loss_type == "MILFocalLoss":
            return L.mil_focal_loss(scores, target, torch.tensor(self.classes_weight), avg_positives=True)
        raise NotImplementedError

    def box_reg_loss(self, proposal_boxes, gt_boxes, pred_deltas, gt_classes, normalizer=None):
        return L.box_reg_loss(proposal_boxes, gt_boxes, pred_deltas, gt_classes, self.num_classes,
                              self.box2box_transform.weights, normalizer)

    # ---- losses (fast_rcnn.py:355-571)
    def losses(self, predictions, proposals, merge_module, branch, update_prototype=False):
        kc = self.num_classes + 1
        te = self.text_encoder
        if branch == "pre_train":
            (scores, lta), deltas, feats = predictions
            losses = {"loss_text_align": lta}
            nfg = [len(p[0]) for p in proposals]
            nbg = [len(p[1]) for p in proposals]
            off = [0]
            for a, b in zip(nfg, nbg):
                off.append(off[-1] + a + b)
            # an image with fg but no bg trips the reference's own assert (fast_rcnn.py:383-385)
            assert all(b > 0 or a == 0 for a, b in zip(nfg, nbg)), "image with foreground but no background RoIs"
            fg_idx = torch.cat([torch.arange(off[i], off[i] + nfg[i]) for i in range(len(proposals))]).long()
            bg_idx = torch.cat([torch.arange(off[i] + nfg[i], off[i + 1]) for i in range(len(proposals))]).long()
            cls_fg = cat([p[0].gt_classes_offline for p in proposals])
            probs_fg = cat([p[0].gt_probs_offline for p in proposals])
            cls_bg = cat([p[1].gt_classes for p in proposals])
            any_fg = sum(nfg) != 0
            if any_fg:
                s = torch.cat([scores[fg_idx], scores[bg_idx]])
                if self.dataset != ("cliparttrain",):
                    tgt = torch.cat([F.one_hot(cls_fg, kc), F.one_hot(cls_bg, kc)]).to(s.dtype)
                    losses["loss_cls"] = self._cls_loss(s, tgt, len(fg_idx), len(bg_idx), True)
                else:  # class_cross_loss1, fast_rcnn.py:587-599: fg target scaled by the teacher's max prob, no averaging
                    tgt = torch.cat([F.one_hot(cls_fg, kc) * probs_fg.max(1)[0].unsqueeze(1), F.one_hot(cls_bg, kc)]).to(s.dtype)
                    losses["loss_cls"] = self._cls_loss(s, tgt, len(fg_idx), len(bg_idx), False)
            else:
                losses["loss_cls"] = torch.zeros_like(lta)
            if update_prototype and any_fg:
                fn = feats / feats.norm(dim=1, keepdim=True)
                f = torch.cat([fn[fg_idx], fn[bg_idx]])
                oh = torch.cat([F.one_hot(cls_fg, kc), F.one_hot(cls_bg, kc)]).float()
                te.per_class_feat.data = _prototype_ema(te.per_class_feat.data, f.detach(), oh, self.prototype_update_rate)
            cls_all = cat([cat([p[0].gt_classes_offline, p[1].gt_classes]) for p in proposals])
            pboxes = cat([cat([p[0].proposal_boxes.tensor, p[1].proposal_boxes.tensor]) for p in proposals])
            gboxes = cat([cat([p[0].gt_boxes.tensor, p[1].proposal_boxes.tensor]) for p in proposals])
            losses["loss_box_reg"] = self.box_reg_loss(pboxes, gboxes, deltas, cls_all)
            return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

        assert branch in ("step_one", "step_two")
        ((scores, lta), deltas, feats), ((scores_c, _), _) = predictions
        proposals, inst_c = proposals
        na = [len(p[0]) for p in proposals]
        nb = [len(p[1]) for p in proposals]
        ng = [len(p[2]) for p in proposals]
        off = [0]
        for a, b, g in zip(na, nb, ng):
            off.append(off[-1] + a + b + g)
        rng = lambda lo, hi: torch.arange(lo, hi)
        ia = torch.cat([rng(off[i], off[i] + na[i]) for i in range(len(proposals))]).long()
        ib = torch.cat([rng(off[i] + na[i], off[i] + na[i] + nb[i]) for i in range(len(proposals))]).long()
        ig = torch.cat([rng(off[i] + na[i] + nb[i], off[i + 1]) for i in range(len(proposals))]).long()
        calc_bg = sum(ng) != 0
        losses = {"loss_text_align": lta}
        cls_a = cat([p[0].gt_classes for p in proposals])
        cls_g = cat([p[2].gt_classes for p in proposals])
        oh_a, oh_g = F.one_hot(cls_a, kc), F.one_hot(cls_g, kc)
        s_a, s_g = scores[ia], scores[ig]
        losses["loss_cls"] = self._cls_loss(torch.cat([s_a, s_g]), torch.cat([oh_a, oh_g]).to(scores.dtype), len(ia), len(ig), True)
        if update_prototype:
            fn = (feats / feats.norm(dim=1, keepdim=True)).detach()
            f_a, f_b, f_g = fn[ia], fn[ib], fn[ig]
            rate = self.prototype_update_rate
            te.per_class_feat.data = _prototype_ema(te.per_class_feat.data, torch.cat([f_a, f_g]), torch.cat([oh_a, oh_g]).float(), rate)
            if sum(nb) != 0:
                pb_on = cat([p[1].gt_probs_online for p in proposals])
                pb_off = cat([p[1].gt_probs_offline for p in proposals])
                oh_b_on = F.one_hot(cat([p[1].gt_classes_online for p in proposals]), kc)
                oh_b_off = F.one_hot(cat([p[1].gt_classes_offline for p in proposals]), kc)
                f_abg = torch.cat([f_a, f_b, f_g])
                te.prototype_b_online.data = _prototype_ema(te.prototype_b_online.data, f_abg, torch.cat([oh_a, oh_b_on, oh_g]).float(), rate)
                te.prototype_b_offline.data = _prototype_ema(te.prototype_b_offline.data, f_abg, torch.cat([oh_a, oh_b_off, oh_g]).float(), rate)
                pa_on = cat([p[0].gt_probs_online for p in proposals])
                pa_off = cat([p[0].gt_probs_offline for p in proposals])
                m_a = merge_module(f_a, te.prototype_b_offline.data, te.prototype_b_online.data, pa_off, pa_on)
                losses["loss_merge_base"] = L.kl_div_mean(m_a, oh_a.to(m_a.dtype))
                m_b = merge_module(f_b, te.prototype_b_offline.data, te.prototype_b_online.data, pb_off, pb_on)
                p_b = F.softmax(scores[ib], dim=1)
                p_a = F.softmax(s_a, dim=1)
                losses["loss_merge_b"] = F.mse_loss(p_b, m_b)
                losses["loss_merge_a"] = F.mse_loss(p_a, oh_a.to(p_a.dtype))
                if branch == "step_two":
                    keep = (m_b.max(1)[0] >= self.cls_b_thresh).detach()
                    if keep.sum() > 0:
                        losses["loss_cls_b"] = L.kl_div_mean(p_b[keep], m_b[keep].detach())
        if scores_c is not None:
            q = cat([c.gt_probs for c in inst_c])
            losses["loss_distillation"] = L.kl_div_mean(F.softmax(scores_c, dim=1), q)
        cls_on = cat([cat([p[0].gt_classes, p[1].gt_classes_online, p[2].gt_classes]) for p in proposals])
        pboxes = cat([cat([p[0].proposal_boxes.tensor, p[1].proposal_boxes.tensor, p[2].proposal_boxes.tensor]) for p in proposals])
        gboxes = cat([cat([p[0].gt_boxes.tensor, p[1].gt_boxes.tensor, p[2].proposal_boxes.tensor]) for p in proposals])
        norm = None if calc_bg else self.batch_size_per_image * len(proposals)
        losses["loss_box_reg"] = self.box_reg_loss(pboxes, gboxes, deltas, cls_on, normalizer=norm)
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    # ---- inference (fast_rcnn.py:116-175, 648-671)
    def inference(self, predictions, proposals: List[Instances]):
        scores, deltas = predictions
        n = [len(p) for p in proposals]
        pb = cat([p.proposal_boxes.tensor for p in proposals])
        boxes = self.box2box_transform.apply_deltas(deltas, pb).split(n)
        probs = F.softmax(scores, dim=-1).split(n)
        out = []
        for b, s, p in zip(boxes, probs, proposals):
            out.append(fast_rcnn_inference_single_image(b, s, p.image_size, self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image))
        return [o[0] for o in out], [o[1] for o in out]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """fast_rcnn.py:116-175 (class-agnostic boxes): threshold, per-class NMS, top-k; also returns `probs`."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes, scores = boxes[valid], scores[valid]
    probs = scores.clone()
    scores = scores[:, :-1]
    b = Boxes(boxes.reshape(-1, 4))
    b.clip(image_shape)
    boxes = b.tensor.view(-1, 1, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes = boxes[inds[:, 0], 0]
    scores = scores[mask]
    probs = probs[inds[:, 0]]
    keep = d2.batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    res = Instances(image_shape)
    res.pred_boxes = Boxes(boxes[keep])
    res.scores = scores[keep]
    res.probs = probs[keep]
    res.pred_classes = inds[keep][:, 1]
    return res, inds[keep][:, 0]


# --------------------------------------------------------------------------- RoI heads (A4-A6)
class Res5ROIHeads(d2.ROIHeads):
    """coin/modeling/roi_heads/clip_roi_heads.py:90-399 (OpenVocabularyRes5ROIHeads), meanpool."""

    def __init__(self, box_predictor: BoxPredictor, num_classes, batch_size_per_image=512, positive_fraction=0.25,
                 pooler_resolution=14, scale=1.0 / 16, sampling_ratio=0, bg_train=True, proposal_append_gt=True):
        super().__init__(num_classes=num_classes, batch_size_per_image=batch_size_per_image, positive_fraction=positive_fraction,
                         proposal_matcher=d2.Matcher([0.5], [0, 1], allow_low_quality_matches=False),
                         proposal_append_gt=proposal_append_gt)
        self.pooler = d2.ROIPooler(pooler_resolution, (scale,), sampling_ratio, "ROIAlignV2")
        self.box_predictor = box_predictor
        self.BG_TRAIN = bg_train
        self.in_features = ["res4"]

    def _pool(self, features, boxes, res5):
        return res5(self.pooler([features[f] for f in self.in_features], boxes)).mean(dim=[2, 3])

    def forward(self, images, features, proposals, res5, attnpool, branch, merge_module=None, targets=None, update_prototype=False):
        train = self.training and branch != "test"
        if train and branch == "pre_train":
            proposals = self.label_and_sample_proposals(proposals, targets, branch)
            boxes = [Boxes.cat([p[0].proposal_boxes, p[1].proposal_boxes]) for p in proposals]
        elif train:
            ta, tb, tc = [t[0] for t in targets], [t[1] for t in targets], [t[2] for t in targets]
            proposals = self.label_and_sample_proposals(proposals, [ta, tb, tc], branch)
            boxes = [Boxes.cat([p[0].proposal_boxes, p[1].proposal_boxes, p[2].proposal_boxes]) for p in proposals]
        else:
            boxes = [p.proposal_boxes for p in proposals]
        predictions = self.box_predictor(self._pool(features, boxes, res5), branch=branch)
        if not train:
            return self.box_predictor.inference(predictions, proposals)[0], {}
        if branch != "pre_train":
            if sum(len(c) for c in tc) != 0:
                cpred = self.box_predictor(self._pool(features, [c.gt_boxes for c in tc], res5), branch=branch, return_feats=False)
                predictions, proposals = (predictions, cpred), (proposals, tc)
            else:
                predictions, proposals = (predictions, ((None, None), None)), (proposals, None)
        return [], self.box_predictor.losses(predictions, proposals, merge_module, branch=branch, update_prototype=update_prototype)

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, branch):
        """clip_roi_heads.py:282-399.  RNG: two randperm draws per image inside subsample_labels."""
        out = []
        if branch == "pre_train":
            if self.proposal_append_gt:
                proposals = d2.add_ground_truth_to_proposals(targets, proposals)
            for p, t in zip(proposals, targets):
                idx, lab = self.proposal_matcher(d2.pairwise_iou(t.gt_boxes, p.proposal_boxes))
                sampled, cls = self._sample_proposals(idx, lab, t.gt_classes_offline)
                idx = idx[sampled]
                is_bg = cls == self.num_classes
                fg, bg = p[sampled[~is_bg]], p[sampled[is_bg]]
                bg.gt_classes = cls[is_bg]
                for name, val in t.get_fields().items():
                    if name.startswith("gt_") and not fg.has(name):
                        fg.set(name, val[idx[~is_bg]])
                out.append((fg, bg))
            return out
        ta, tb, tc = targets
        if self.proposal_append_gt:
            proposals = d2.add_ground_truth_to_proposals(ta, proposals)
            proposals = d2.add_ground_truth_to_proposals(tb, proposals)
        for p, a, b, c in zip(proposals, ta, tb, tc):
            la, lb, lc = len(a), len(b), len(c)
            idx, lab = self.proposal_matcher(d2.pairwise_iou(Boxes.cat([a.gt_boxes, b.gt_boxes, c.gt_boxes]), p.proposal_boxes))
            in_c = (idx >= la + lb) & (idx < la + lb + lc)
            lab[in_c & (lab != 0)] = -1  # proposals matched to a private (C) box are ignored
            sampled, cls = self._sample_proposals(idx, lab, torch.cat([a.gt_classes, b.gt_classes_online, c.gt_classes]))
            idx = idx[sampled]
            is_bg = cls == self.num_classes
            m_a = (idx >= 0) & (idx < la) & ~is_bg
            m_b = (idx >= la) & (idx < la + lb) & ~is_bg
            pa, pb, pg = p[sampled[m_a]], p[sampled[m_b]], p[sampled[is_bg]]
            pg.gt_classes = cls[is_bg]
            if not self.BG_TRAIN:
                pg = pg[0:0]
            for name, val in a.get_fields().items():
                if name.startswith("gt_") and not pa.has(name):
                    pa.set(name, val[idx[m_a]])
            for name, val in b.get_fields().items():
                if name.startswith("gt_") and not pb.has(name):
                    pb.set(name, val[idx[m_b] - la])
            out.append((pa, pb, pg))
        return out


# --------------------------------------------------------------------------- RPN (A3)
class DualTeacherRPN(d2.RPN):
    """coin/modeling/proposal_generator/rpn.py:16-345."""

    def __init__(self, in_channels, anchor_sizes=((32, 64, 128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),), stride=16,
                 batch_size_per_image=256, positive_fraction=0.5, pre_nms_topk=(12000, 6000), post_nms_topk=(2000, 1000),
                 nms_thresh=0.7, loss_weight=None, bg_train=True):
        ag = d2.DefaultAnchorGenerator([list(s) for s in anchor_sizes], [list(a) for a in aspect_ratios], [stride])
        super().__init__(in_features=["res4"], head=d2.StandardRPNHead(in_channels, ag.num_anchors[0]), anchor_generator=ag,
                         anchor_matcher=d2.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
                         box2box_transform=d2.Box2BoxTransform((1.0, 1.0, 1.0, 1.0)), batch_size_per_image=batch_size_per_image,
                         positive_fraction=positive_fraction, pre_nms_topk=pre_nms_topk, post_nms_topk=post_nms_topk,
                         nms_thresh=nms_thresh, min_box_size=0.0, anchor_boundary_thresh=-1.0,
                         loss_weight=loss_weight or {"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0, "loss_rpn_distillation": 0.1})
        self.BG_TRAIN = bg_train

    def head_outputs(self, features):
        feats = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(feats)
        lg, dl = self.rpn_head(feats)
        logits = [s.permute(0, 2, 3, 1).flatten(1) for s in lg]
        deltas = [x.view(x.shape[0], -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2) for x in dl]
        return anchors, logits, deltas

    def forward(self, images, features, gt_instances=None, branch=None):
        anchors, logits, deltas = self.head_outputs(features)
        losses = {}
        if self.training and branch != "test":
            if branch == "pre_train":
                labels, gt_boxes = self.label_and_sample_anchors(anchors, gt_instances, branch)
                losses = self.losses(anchors, logits, labels, deltas, gt_boxes)
            else:
                ia, ic = [g[0] for g in gt_instances], [g[2] for g in gt_instances]
                labels, gt_boxes, midx, dlabels = self.label_and_sample_anchors(anchors, [ia, ic], branch)
                teacher = [c.gt_probs[:, :-1].sum(1)[m] if len(c) != 0 else torch.zeros_like(m) for c, m in zip(ic, midx)]
                losses = self.losses(anchors, logits, labels, deltas, gt_boxes, calc_bg=self.BG_TRAIN)
                losses.update(self.losses(anchors, logits, dlabels, None, None, teacher_probs=teacher, only_distillation=True))
        proposals = self.predict_proposals(anchors, logits, deltas, images.image_sizes)
        return proposals, losses

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors, gt_instances, branch):
        """rpn.py:118-254 (no `no_thresh_boxes`: the trainer never sets them, base.py:119-121)."""
        anchors = Boxes.cat(anchors)
        labels_out, boxes_out = [], []
        if branch == "pre_train":
            for g in gt_instances:
                gb = g.gt_boxes
                idx, lab = self.anchor_matcher(d2.pairwise_iou(gb, anchors))
                lab = self._subsample_labels(lab)
                if len(gb) == 0:
                    mb = torch.zeros_like(anchors.tensor)
                    lab[:] = -1
                else:
                    mb = gb[idx].tensor
                labels_out.append(lab)
                boxes_out.append(mb)
            return labels_out, boxes_out
        ga, gc = gt_instances
        midx_out, dist_out = [], []
        for a, c in zip(ga, gc):
            ba, bc = a.gt_boxes, c.gt_boxes
            both = Boxes.cat([ba, bc])
            idx, lab = self.anchor_matcher(d2.pairwise_iou(both, anchors))
            in_c = (idx >= len(ba)) & (idx < len(both))
            is_bg = lab == 0
            fg_c = in_c & ~is_bg
            didx = idx - len(ba)
            didx[~fg_c] = 0
            lab[fg_c] = -1
            idx = idx.clone()
            idx[in_c] = 0
            dlab = torch.zeros_like(lab)
            dlab[fg_c] = 1
            lab = self._subsample_labels(lab)
            if len(ba) == 0:
                mb = torch.zeros_like(anchors.tensor)
                lab[~(in_c & is_bg)] = -1
            else:
                mb = ba[idx].tensor
            labels_out.append(lab)
            boxes_out.append(mb)
            midx_out.append(didx)
            dist_out.append(dlab)
        return labels_out, boxes_out, midx_out, dist_out

    def losses(self, anchors, logits, gt_labels, deltas, gt_boxes, teacher_probs=None, only_distillation=False, calc_bg=True):
        labels = torch.stack(gt_labels)
        lg = cat(logits, dim=1)
        if not only_distillation:
            cls, loc = L.rpn_losses(Boxes.cat(anchors).tensor, lg, labels, cat(deltas, dim=1), torch.stack(gt_boxes),
                                    self.batch_size_per_image, calc_bg)
            out = {"loss_rpn_cls": cls, "loss_rpn_loc": loc}
        else:
            kl, n = L.rpn_distillation(lg, labels, torch.stack(teacher_probs))
            out = {"loss_rpn_distillation": kl} if n != 0 else {}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}


# --------------------------------------------------------------------------- detector (A1, A19)
class OpenVocabularyRCNN(nn.Module):
    """coin/modeling/meta_arch/clip_rcnn.py:187-426."""

    def __init__(self, backbone, proposal_generator, roi_heads, pixel_mean=CLIP_MEAN, pixel_std=CLIP_STD):
        super().__init__()
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std), False)

    def preprocess_image(self, batched_inputs):
        """clip_rcnn.py:287-298: ToTensor (u8/255) -> Normalize -> zero-pad to the batch maximum."""
        m, s = self.pixel_mean.view(3, 1, 1), self.pixel_std.view(3, 1, 1)
        imgs = [(x["image"].to(m.dtype).div(255) - m) / s for x in batched_inputs]
        return d2.ImageList.from_tensors(imgs, self.backbone.size_divisibility)

    def forward(self, batched_inputs, merge_module=None, dual_teacher_instances=None, branch=None, update_prototype=False):
        if not self.training or branch == "test":
            return self.inference(batched_inputs, branch=branch)
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        if branch == "pre_train":
            rcnn = [x["RCNN"] for x in batched_inputs]
            rpn = [x["RPN"] for x in batched_inputs]
        else:
            rcnn, rpn = dual_teacher_instances
        proposals, proposal_losses = self.proposal_generator(images, features, rpn, branch=branch)
        _, det_losses = self.roi_heads(images, features, proposals, self.backbone.layer4, self.backbone.attnpool, branch=branch,
                                       merge_module=merge_module, targets=rcnn, update_prototype=update_prototype)
        out = dict(det_losses)
        out.update(proposal_losses)
        return out

    def inference(self, batched_inputs, branch=None):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        proposals, _ = self.proposal_generator(images, features, None, branch)
        results, _ = self.roi_heads(images, features, proposals, self.backbone.layer4, self.backbone.attnpool, branch=branch)
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            out.append({"instances": d2.detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out


# --------------------------------------------------------------------------- trainer-side arithmetic
def gradient_discrepancy_loss(box_predictor: BoxPredictor, loss_a, loss_b):
    """coin/utils/losses.py:75-96: 1 - mean cosine between d(loss_a)/d(theta) (detached) and d(loss_b)/d(theta),
    per `trans` parameter, double-differentiable in loss_b."""
    cos = []
    for _, p in box_predictor.trans.named_parameters():
        if not p.requires_grad:
            continue
        ga = torch.autograd.grad([loss_a], [p], create_graph=True)[0]
        gb = torch.autograd.grad([loss_b], [p], create_graph=True)[0]
        if p.dim() > 1:
            cos.append(F.cosine_similarity(ga.detach(), gb, dim=1).mean())
        else:
            cos.append(F.cosine_similarity(ga.detach(), gb, dim=0))
    return (1.0 - torch.stack(cos)).mean()


@torch.no_grad()
def ema_update(teacher: nn.Module, student: nn.Module, keep_rate: float):
    """coin/modeling/meta_arch/ts_ensemble.py:39-69: state-dict lerp (buffers included), then load_state_dict."""
    s = student.state_dict()
    new = OrderedDict((k, s[k] * (1 - keep_rate) + v * keep_rate) for k, v in teacher.state_dict().items())
    teacher.load_state_dict(new)


def lr_at_iter(base_lr, it, milestones, factor_list, warmup_iters, warmup_factor=0.001, warmup_method="linear"):
    """coin/solver/lr_scheduler.py:51-62."""
    return base_lr * d2.get_warmup_factor_at_iter(warmup_method, it, warmup_iters, warmup_factor) * factor_list[bisect_right(list(milestones), it)]


def optimizer_param_groups(model: nn.Module, base_lr, overrides: Dict[str, float], weight_decay_norm=0.0, weight_decay_bias=1e-4):
    """coin/solver/build.py:106-201: one group per tensor; lr = base_lr * last matching substring multiplier
    (later keys win); norm layers get weight_decay_norm; params literally named 'bias' get weight_decay_bias."""
    norm_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
                  nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm)
    seen, groups = set(), []
    for mname, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in seen:
                continue
            seen.add(p)
            hp = {"lr": base_lr}
            for key, mult in overrides.items():
                if key in mname or key in pname:
                    hp["lr"] = mult * base_lr
            if isinstance(module, norm_types) and weight_decay_norm is not None:
                hp["weight_decay"] = weight_decay_norm
            if pname == "bias" and weight_decay_bias is not None:
                hp["weight_decay"] = weight_decay_bias
            groups.append({"params": [p], "name": (mname + "." if mname else "") + pname, **hp})
    return groups


def weighted_box_fusion_split(box_a, box_b, score_a, score_b):
    """coin/layers/nms.py:24-31."""
    s = torch.stack((score_a, score_b), dim=1)
    w = s / s.sum(dim=1, keepdim=True)
    return box_a * w[:, 0:1] + box_b * w[:, 1:]


def rescale_flip_boxes(boxes: torch.Tensor, old_size, new_size, random_flip: str) -> torch.Tensor:
    """coin/engine/base.py:80-103: scale cached teacher boxes to the network input size, mirror if the view is flipped."""
    (ih, iw), (nh, nw) = old_size, new_size
    b = boxes.clone()
    b[:, 0::2] *= nw / iw
    b[:, 1::2] *= nh / ih
    if random_flip == "horizontal":
        b[:, 0], b[:, 2] = nw - b[:, 2].clone(), nw - b[:, 0].clone()
    elif random_flip == "vertical":
        b[:, 1], b[:, 3] = nh - b[:, 3].clone(), nh - b[:, 1].clone()
    elif random_flip != "no":
        raise NotImplementedError
    return b


def build_detector(num_classes=8, layers=(3, 4, 6, 3), width=64, text_dim=1024, text_width=512, text_layers=12, text_heads=8,
                   context_length=77, vocab_size=49408, tokenized_prompts=None, classes_weight=None, loss_weight=None,
                   roi_batch=512, rpn_batch=256, anchor_sizes=((32, 64, 128, 256, 512),), pre_nms_topk=(12000, 6000),
                   post_nms_topk=(2000, 1000), freeze_at=2, zero_init_bn3=True, cls_b_thresh=0.7) -> OpenVocabularyRCNN:
    """Random-init detector of the reference's architecture (SURVEY §8d synthetic weights)."""
    k1 = num_classes + 1
    if tokenized_prompts is None:
        tokenized_prompts = synthetic_prompt_tokens(k1, context_length, vocab_size)
    enc = TextEncoder(text_dim, context_length, vocab_size, text_width, text_heads, text_layers, tokenized_prompts, 4, 4)
    te = ClipText(enc, [f"class{i}" for i in range(num_classes)] + ["backgroud"], torch.randn(k1, text_dim))
    bb = ClipImageBackbone(layers, width, freeze_at, True, zero_init_bn3)
    res5_ch = width * 32
    lw = loss_weight or {"loss_box_reg": 1.0, "loss_cls": 1.0, "loss_text_align": 10.0, "loss_distillation": 0.1, "loss_cls_b": 0.1}
    bp = BoxPredictor(res5_ch, te, text_dim, classes_weight or [1.0] * num_classes + [0.9], lw, roi_batch, cls_b_thresh=cls_b_thresh)
    rh = Res5ROIHeads(bp, num_classes, roi_batch)
    pg = DualTeacherRPN(width * 16, anchor_sizes, batch_size_per_image=rpn_batch, pre_nms_topk=pre_nms_topk, post_nms_topk=post_nms_topk)
    return OpenVocabularyRCNN(bb, pg, rh)


def synthetic_prompt_tokens(n_classes: int, context_length=77, vocab_size=49408) -> torch.Tensor:
    """'SOS a photo of a X X X X {cls} . EOT' token layout (clip_text.py:281-291) with CLIP's ids for the template
    (tests/golden/clip_tokens.npz holds the real tokenizer's output for the Cityscapes names)."""
    sos, eot = vocab_size - 2, vocab_size - 1
    t = torch.zeros(n_classes, context_length, dtype=torch.int)
    for i in range(n_classes):
        seq = [sos, 320 % (vocab_size - 2), 1125 % (vocab_size - 2), 539 % (vocab_size - 2), 320 % (vocab_size - 2)] + [343 % (vocab_size - 2)] * 4 + [(1000 + i) % (vocab_size - 2), 269 % (vocab_size - 2), eot]
        t[i, : len(seq)] = torch.tensor(seq)
    return t
