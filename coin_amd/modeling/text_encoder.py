"""Frozen CLIP text transformer with a learnable prompt, and the CKG merge network.

Mirrors coin/modeling/text_encoder/clip_text.py:31-327 (``TEXT_ENCODER`` / registered ``CLIP_TEXT``) and
coin/modeling/merge/ckg.py:36-115 (registered ``CKGNet``): same state-dict keys, same maths.  Both are
tiny (9 x 77 tokens x 512; tens of RoIs) and run as plain device tensor ops in fp32: the reference keeps
the encoder in fp16 under CUDA autocast (clip_text.py:137); fp32 is >= that precision.

Without network access neither CLIP weights nor the BPE vocabulary can be fetched: the encoder is
CLIP-initialised at random (clip_text.py:65-79) and prompts use CLIP's token ids for the template
'a photo of a X X X X {cls}.' (ids pinned by tests/golden/clip_tokens.npz) with synthetic class ids.
"""
from __future__ import annotations

import math
import os
import weakref
from collections import OrderedDict
from typing import List, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from .. import layers as L
from ..registry import MERGE_REGISTRY, TEXT_ENCODER_REGISTRY

TEXT_DIMS = {"RN50": 1024, "RN101": 512, "RN50x4": 640, "RN50x16": 768}  # fast_rcnn.py:283
SOS, EOT = 49406, 49407
TEMPLATE_IDS = [320, 1125, 539, 320]  # "a photo of a"
X_ID, DOT_ID = 343, 269
N_TEMPLATES = 81  # len(MODIFIED_REGION_CLIP_TEMPLATES), coin/modeling/utils.py:415-497


class _Attention(nn.Module):
    """nn.MultiheadAttention parameter layout; causal self-attention via SDPA."""

    def __init__(self, d: int, heads: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        self.heads = heads

    def forward(self, x):  # [N, L, D]
        n, l, d = x.shape
        q, k, v = L.linear(x, self.in_proj_weight, self.in_proj_bias).view(n, l, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        return L.linear(o.transpose(1, 2).reshape(n, l, d), self.out_proj.weight, self.out_proj.bias)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d: int, heads: int):
        super().__init__()
        self.attn = _Attention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, d * 4)), ("gelu", nn.Identity()), ("c_proj", nn.Linear(d * 4, d))]))
        self.ln_2 = nn.LayerNorm(d)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        h = L.linear(self.ln_2(x), self.mlp.c_fc.weight, self.mlp.c_fc.bias)
        return x + L.linear(h * torch.sigmoid(1.702 * h), self.mlp.c_proj.weight, self.mlp.c_proj.bias)  # QuickGELU


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


def prompt_tokens(n_classes: int, context_length: int = 77, add_prompt_num: int = 4, class_ids: Sequence[int] = None,
                  vocab_size: int = 49408) -> torch.Tensor:
    sos, eot = (SOS, EOT) if vocab_size >= 49408 else (vocab_size - 2, vocab_size - 1)
    cap = vocab_size - 2
    toks = torch.zeros(n_classes, context_length, dtype=torch.int)
    for i in range(n_classes):
        cid = class_ids[i] if class_ids is not None else 1000 + i
        seq = [sos] + [t % cap for t in TEMPLATE_IDS] + [X_ID % cap] * add_prompt_num + [cid % cap, DOT_ID % cap, eot]
        toks[i, : len(seq)] = torch.tensor(seq)
    return toks


class TEXT_ENCODER(nn.Module):
    def __init__(self, embed_dim, context_length, vocab_size, transformer_width, transformer_heads, transformer_layers, prompt_info):
        super().__init__()
        self.context_length = context_length
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        toks, self.prompt_tmp_len, self.add_prompt_num = prompt_info
        self.register_buffer("tokenized_prompts", toks.clone(), persistent=False)
        self.initialize_parameters()
        self.load_embedding(transformer_width)
        self.freeze_encoder()

    def initialize_parameters(self):
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

    def load_embedding(self, width):
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

    def freeze_encoder(self):
        for m in (self.token_embedding, self.ln_final, self.transformer):
            for p in m.parameters():
                p.requires_grad = False
        self.positional_embedding.requires_grad = False
        self.text_projection.requires_grad = False
        self.logit_scale.requires_grad = False

    # ---- the prompt-conditioned pass (clip_text.py:165-205 with `add=True`) ------------------------------------------------------
    # Shapes are static (classes x 77 x width) and the transformer is frozen: the ~300 forward and ~300 backward launches of the
    # 12 blocks are captured ONCE as two HIP graphs (torch.cuda.make_graphed_callables; the prompt vectors are the graph's inputs
    # and receive its gradients) and replayed with two launches per step; without a gradient (the EMA teacher's inference) the result
    # is cached until the prompt vectors change (`invalidate_text_cache`, called after an EMA; in-place loads bump `_version`).
    use_graph = True           # cfg.AMD.TEXT_GRAPH
    _graphs = None             # (autocast dtype or None) -> graphed callable
    _nograd_cache = None       # (key, tensor)
    _graph_failed = False

    def _prompted(self, embedding_tmp, add_in_embedding):
        n = self.embedding_class.size(0)
        ex = lambda p: p.unsqueeze(0).expand(n, -1, -1)
        x = torch.cat([ex(self.sos), ex(embedding_tmp), ex(add_in_embedding), self.embedding_class, ex(self.eos)], dim=1)
        return self._encode(x, self.tokenized_prompts.argmax(dim=-1))

    def _encode(self, x, eot):
        x = x.float() + self.positional_embedding
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            # bf16 throughput mode: keep the residual stream in the compute dtype, as the reference's fp16 encoder does
            # (clip_text.py:137,194 `x.type(self.dtype)`); an fp32 stream turns every residual add into a mixed-dtype kernel
            x = x.to(torch.get_autocast_dtype("cuda"))
        x = self.ln_final(self.transformer(x))
        x = x[torch.arange(x.shape[0], device=x.device), eot.long()]
        x = x @ L.compute_weight(self.text_projection, L.compute_dtype_of(x))
        return x / torch.norm(x, dim=-1, keepdim=True)

    def invalidate_text_cache(self):
        self._nograd_cache = None

    def _prompt_key(self):
        a, b = self.embedding_tmp, self.add_in_embedding
        ac = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        return (a.data_ptr(), a._version, b.data_ptr(), b._version, ac)

    def _graphed(self):
        """-> graphed callable for the current autocast mode (built on first use), or None if capture is not possible here."""
        ac = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        if self._graphs is None:
            self._graphs = {}
        if ac not in self._graphs:
            def fn(tmp, add):
                with torch.autocast("cuda", dtype=ac or torch.bfloat16, enabled=ac is not None, cache_enabled=False):
                    return self._prompted(tmp, add)
            sample = tuple(t.detach().clone().requires_grad_(True) for t in (self.embedding_tmp, self.add_in_embedding))
            try:
                with torch.autocast("cuda", enabled=False):
                    self._graphs[ac] = torch.cuda.make_graphed_callables(fn, sample)
            except Exception as e:  # same device path either way: eager launches of the same kernels
                import warnings

                warnings.warn(f"text encoder: HIP graph capture failed ({type(e).__name__}: {e}); running it eagerly")
                self._graph_failed = True
                return None
        return self._graphs[ac]

    def forward(self, text, add: bool):
        if not add:
            return self._encode(self.token_embedding(text.long()), text.argmax(dim=-1))
        tmp, addv = self.embedding_tmp, self.add_in_embedding
        needs_grad = torch.is_grad_enabled() and (tmp.requires_grad or addv.requires_grad)
        if not tmp.is_cuda:
            return self._prompted(tmp, addv)
        if not needs_grad:
            if tmp.requires_grad or addv.requires_grad:
                # a trainable copy evaluated without a gradient (student in eval mode): the fused SGD kernel rewrites the prompt
                # vectors through raw pointers without a version bump, so nothing may be cached here
                return self._prompted(tmp, addv)
            key = self._prompt_key()   # frozen copy (the EMA teacher): changes only through update_teacher / load_state_dict
            if self._nograd_cache is None or self._nograd_cache[0] != key:
                with torch.no_grad():
                    self._nograd_cache = (key, self._prompted(tmp, addv))
            return self._nograd_cache[1]
        # only on the default stream: replayed from the pre-train step's side stream (beside the backbone) the two graphs cost 30 ms per
        # step on this runtime (bench.py 37.1 -> 67.9 ms, round-3 A/B), on the main stream of the targetDET step they save 0-4 ms
        if (self.use_graph and not self._graph_failed and not torch.cuda.is_current_stream_capturing() and not self._graph_busy()
                and torch.cuda.current_stream(tmp.device) == torch.cuda.default_stream(tmp.device)):
            g = self._graphed()
            if g is not None:
                out = g(tmp, addv)
                # the graphs own ONE set of activation buffers: until this output has been back-propagated (or dropped) a second
                # pass must not replay them (the step branches classify twice per forward when `shared_text` is off)
                self._inflight = weakref.ref(out)
                out.register_hook(self._graph_done)
                return out
        return self._prompted(tmp, addv)

    _inflight = None

    def _graph_done(self, grad):
        self._inflight = None
        return grad

    def _graph_busy(self) -> bool:
        return self._inflight is not None and self._inflight() is not None


@TEXT_ENCODER_REGISTRY.register()
class CLIP_TEXT(nn.Module):
    def __init__(self, type: str, classes: List[str], add_prompt_num: int = 4, dataset_style: str = "", embed_dim=None,
                 context_length=77, vocab_size=49408, width=512, heads=8, layers=12, tokenized_prompts=None,
                 n_templates: int = N_TEMPLATES, graph: bool = True):
        super().__init__()
        self.type, self.classes, self.dataset_style, self.add_prompt_num = type, list(classes), dataset_style, add_prompt_num
        embed_dim = embed_dim or TEXT_DIMS[type]
        toks = tokenized_prompts if tokenized_prompts is not None else prompt_tokens(len(classes), context_length, add_prompt_num, vocab_size=vocab_size)
        self.encoder = TEXT_ENCODER(embed_dim, context_length, vocab_size, width, heads, layers, (toks, len(TEMPLATE_IDS), add_prompt_num))
        self.encoder.use_graph = bool(graph)
        self.n_templates = n_templates
        self.load_embedding()

    @classmethod
    def from_config(cls, cfg, backgroud=False):
        classes = list(cfg.AMD.CLASS_NAMES)
        if backgroud:
            classes.append("backgroud")
        a = cfg.AMD.ARCH
        return cls(type=cfg.MODEL.TEACHER_OFFLINE.TYPE or "RN50", classes=classes, add_prompt_num=cfg.CLOUD.ADD_PROMPT_NUM,
                   dataset_style=cfg.DATASETS.STYLE_NAME, n_templates=cfg.AMD.TEXT_TEMPLATES, embed_dim=a.TEXT_DIM or None,
                   context_length=a.CONTEXT_LENGTH, vocab_size=a.VOCAB_SIZE, width=a.TEXT_WIDTH or 512, heads=a.TEXT_HEADS or 8,
                   layers=a.TEXT_LAYERS or 12,
                   graph=bool(getattr(cfg.AMD, "TEXT_GRAPH", True)) and os.environ.get("COIN_TEXT_GRAPH", "1") != "0")

    @torch.no_grad()
    def load_embedding(self):
        """clip_text.py:262-279: mean of the 81 template embeddings per class, L2-normalised.  Template wording needs
        the BPE vocabulary; synthetic token sequences of the same lengths stand in for it."""
        enc = self.encoder
        g = torch.Generator().manual_seed(1234)
        feats = []
        for ci in range(len(self.classes)):
            toks = torch.zeros(self.n_templates, enc.context_length, dtype=torch.int)
            for t in range(self.n_templates):
                length = int(torch.randint(6, min(14, enc.context_length - 2), (1,), generator=g, device="cpu"))
                hi = min(enc.vocab_size - 2, 40000)
                body = torch.randint(min(300, hi // 4), hi, (length,), generator=g, device="cpu")
                body[-2] = (1000 + ci) % (enc.vocab_size - 2)
                sos, eot = (SOS, EOT) if enc.vocab_size >= 49408 else (enc.vocab_size - 2, enc.vocab_size - 1)
                seq = torch.cat([torch.tensor([sos]), body, torch.tensor([eot])])
                toks[t, : len(seq)] = seq.int()
            feats.append(enc(toks.to(enc.positional_embedding.device), add=False).mean(0, keepdim=True))
        f = torch.cat(feats, dim=0)
        f = f / f.norm(dim=1, keepdim=True)
        self.register_buffer("per_class_feat", f)
        self.register_buffer("prototype_b_online", f.clone())
        self.register_buffer("prototype_b_offline", f.clone())

    def train(self, mode: bool = True):  # clip_text.py:296-302: the text encoder always stays in eval mode
        self.training = False
        for m in self.children():
            m.eval()
        return self

    num_classes = property(lambda self: len(self.classes))
    logit_scale = property(lambda self: self.encoder.logit_scale)
    prototype = property(lambda self: self.per_class_feat)

    def forward(self, added: bool):
        return self.encoder(None, add=True) if added else self.per_class_feat


def text_dim_of(cfg) -> int:
    return cfg.AMD.ARCH.TEXT_DIM or TEXT_DIMS[cfg.MODEL.TEACHER_OFFLINE.TYPE or "RN50"]


def build_text_encoder(cfg, backgroud):
    return TEXT_ENCODER_REGISTRY.get(cfg.MODEL.TEACHER_OFFLINE.TEXT_ENCODER).from_config(cfg, backgroud)


class CROSS_ATTENTION(nn.Module):
    def __init__(self, hidden_size, all_head_size, num_classes, head_num=8):
        super().__init__()
        assert all_head_size % head_num == 0
        self.num_heads, self.h_size = head_num, all_head_size // head_num
        self.linear_q = nn.Linear(hidden_size, all_head_size, bias=False)
        self.linear_k = nn.Linear(hidden_size, all_head_size, bias=False)
        self.linear_v = nn.Linear(hidden_size, all_head_size, bias=False)
        self.linear_output = nn.Linear(all_head_size, num_classes)
        for l in (self.linear_q, self.linear_k, self.linear_v, self.linear_output):
            nn.init.xavier_normal_(l.weight)
        nn.init.constant_(self.linear_output.bias, 0)

    def forward(self, x, y):
        sp = lambda t: t.view(1, -1, self.num_heads, self.h_size).transpose(1, 2)
        q, k, v = sp(self.linear_q(x)), sp(self.linear_k(y)), sp(self.linear_v(y))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.size(-1)), dim=-1) @ v
        return self.linear_output(a.transpose(1, 2).contiguous().view(-1, self.num_heads * self.h_size))


@MERGE_REGISTRY.register()
class CKGNet(nn.Module):
    def __init__(self, hidden_size, all_head_size, num_classes, head_num=8):
        super().__init__()
        self.cross_offline = CROSS_ATTENTION(hidden_size, all_head_size, num_classes, head_num)
        self.cross_online = CROSS_ATTENTION(hidden_size, all_head_size, num_classes, head_num)

    @classmethod
    def from_config(cls, cfg):
        return cls(cfg.MODEL.MERGE_DIM, cfg.MODEL.MERGE_DIM, len(cfg.AMD.CLASS_NAMES) + 1)

    def forward(self, x, prototype_offline, prototype_online, probs_offline, probs_online):
        w_off = self.cross_offline(x, prototype_offline)
        w_on = self.cross_online(x, prototype_online)
        return F.softmax(w_off * probs_offline + w_on * probs_online, dim=1)


def build_merge(cfg):
    return MERGE_REGISTRY.get(cfg.MODEL.MERGE).from_config(cfg)
