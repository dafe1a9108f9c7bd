"""SGD parameter groups + the one-launch fused SGD step.

Mirrors coin/solver/build.py:24-201: one param group PER TENSOR, lr = BASE_LR * (last matching substring
multiplier of SOLVER.PER_MODULE_PARAM_WEIGHT[0]), norm layers use WEIGHT_DECAY_NORM, parameters literally
named ``bias`` use WEIGHT_DECAY_BIAS.  The reference then loops torch.optim.SGD over ~170 groups (one small
kernel chain per tensor); here all groups are updated by ONE ``coin_sgd_step`` launch over a device table.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from .. import kernels as K
from .. import layers as L
from .lr_scheduler import WarmupTwoStageMultiStepLR

_NORM_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm, torch.nn.GroupNorm,
               torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d, torch.nn.LayerNorm, torch.nn.LocalResponseNorm)


def get_default_optimizer_params(model, base_lr, weight_decay_norm=None, bias_lr_factor=1.0, weight_decay_bias=None, overrides=None,
                                 only_text_encoder=None) -> List[Dict[str, Any]]:
    mults = dict(overrides[0]) if overrides else {}
    bias_over = {}
    if bias_lr_factor is not None and bias_lr_factor != 1.0:
        bias_over["lr"] = base_lr * bias_lr_factor
    if weight_decay_bias is not None:
        bias_over["weight_decay"] = weight_decay_bias
    params, seen = [], set()
    for mname, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if only_text_encoder is True and "text_encoder" not in mname:
                continue
            if only_text_encoder is False and "text_encoder" in mname:
                continue
            if not p.requires_grad or p in seen:
                continue
            seen.add(p)
            hp = {"lr": base_lr}
            for key, m in mults.items():  # later keys overwrite earlier ones (build.py:191-194)
                if key in mname or key in pname:
                    hp["lr"] = m * base_lr
            if isinstance(module, _NORM_TYPES) and weight_decay_norm is not None:
                hp["weight_decay"] = weight_decay_norm
            if pname == "bias":
                hp.update(bias_over)
            params.append({"params": [p], "name": (mname + "." if mname else "") + pname, **hp})
    return params


class FusedSGD:
    """torch.optim.SGD(momentum, weight_decay) semantics over single-tensor groups, one HIP launch per step."""

    def __init__(self, param_groups: List[Dict[str, Any]], lr: float, momentum: float = 0.9, weight_decay: float = 0.0, nesterov: bool = False):
        assert not nesterov, "SOLVER.NESTEROV is False in every COIN config"
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g["base_lr"] = g["lr"]  # the schedule multiplies every group by one common factor
            g["momentum"] = momentum
            assert len(g["params"]) == 1
            self.param_groups.append(g)
        self.momentum = momentum
        self._table: Optional[K.SgdTable] = None

    @property
    def params(self):
        return [g["params"][0] for g in self.param_groups]

    def zero_grad(self, set_to_none: bool = True):
        grads = [p.grad for p in self.params if p.grad is not None]
        if set_to_none:
            for p in self.params:
                p.grad = None
        elif grads:
            torch._foreach_zero_(grads)

    @torch.no_grad()
    def step(self, inv_loss_scale: float = 1.0, gate: Optional[torch.Tensor] = None):
        """gate: one fp32 element on the parameters' device; the whole update is skipped ON THE DEVICE when it holds 0 (no host sync)."""
        params = self.params
        if self._table is None:
            self._table = K.SgdTable(params, [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups])
        grads = [p.grad for p in params]  # None -> the tensor is skipped this step, as torch.optim.SGD does
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is not None and (g.stride() != p.stride() or g.dtype != p.dtype):
                # the kernel is elementwise on storage: bring a gradient that autograd / DDP laid out differently to the
                # parameter's (dense) layout -- normally never taken (both honour the "gradient layout contract")
                p.grad = grads[i] = torch.empty_like(p).copy_(g)
        # lr_i(t) = base_lr_i * factor(t): the factor travels as a kernel argument, the table keeps the base rates
        factor, uniform = None, True
        for g in self.param_groups:
            if g["base_lr"] != 0.0:
                f = g["lr"] / g["base_lr"]
                if factor is None:
                    factor = f
                elif abs(f - factor) > 1e-9 * max(1.0, abs(factor)):
                    uniform = False
                    break
        # compute-dtype copies of the weights (bf16 mode) are rewritten by the same launch as their fp32 masters
        shadows = [L.shadow_of(p) for p in params]
        if uniform:
            self._table.step(grads, self.momentum, inv_loss_scale, lrs=[g["base_lr"] for g in self.param_groups],
                             lr_scale=1.0 if factor is None else factor, shadows=shadows, gate=gate)
        else:  # groups were edited independently: fall back to uploading the absolute rates
            self._table.step(grads, self.momentum, inv_loss_scale, lrs=[g["lr"] for g in self.param_groups], shadows=shadows, gate=gate)
        L.weights_updated()   # the persistent data-gradient layouts are re-derived by ONE launch at the next backward

    def load_momentum(self, bufs):
        """Momentum buffers from a checkpoint (None entries = that tensor has not been stepped yet)."""
        if all(b is None for b in bufs):
            self._table = None
            return
        self._table = K.SgdTable(self.params, [g["base_lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups])
        for dst, src in zip(self._table.bufs, bufs):
            if src is not None:
                dst.copy_(src.reshape(dst.shape))
        self._table.first = False

    def state_dict(self):
        bufs = self._table.bufs if self._table is not None else None
        return {"momentum_buffers": bufs, "first": self._table.first if self._table else True,
                "groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd["groups"]):
            g.update(s)
        if sd.get("momentum_buffers") is not None:
            self._table = K.SgdTable(self.params, [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups])
            for b, s in zip(self._table.bufs, sd["momentum_buffers"]):
                b.copy_(s)
            self._table.first = sd.get("first", False)


def build_optimizer(cfg, model, name="all") -> FusedSGD:
    only = {"backbone": False, "cls": True, "all": None}[name]
    params = get_default_optimizer_params(model, base_lr=cfg.SOLVER.BASE_LR, weight_decay_norm=cfg.SOLVER.WEIGHT_DECAY_NORM,
                                          bias_lr_factor=cfg.SOLVER.BIAS_LR_FACTOR, weight_decay_bias=cfg.SOLVER.WEIGHT_DECAY_BIAS,
                                          overrides=cfg.SOLVER.PER_MODULE_PARAM_WEIGHT, only_text_encoder=only)
    return FusedSGD(params, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, weight_decay=cfg.SOLVER.WEIGHT_DECAY, nesterov=cfg.SOLVER.NESTEROV)


def build_lr_scheduler(cfg, optimizer):
    name = cfg.SOLVER.LR_SCHEDULER_NAME
    if name == "WarmupTwoStageMultiStepLR":
        return WarmupTwoStageMultiStepLR(optimizer, cfg.SOLVER.STEPS, factor_list=cfg.SOLVER.FACTOR_LIST, gamma=cfg.SOLVER.GAMMA,
                                         warmup_factor=cfg.SOLVER.WARMUP_FACTOR, warmup_iters=cfg.SOLVER.WARMUP_ITERS,
                                         warmup_method=cfg.SOLVER.WARMUP_METHOD)
    if name == "WarmupMultiStepLR":
        steps = list(cfg.SOLVER.STEPS)
        return WarmupTwoStageMultiStepLR(optimizer, steps, factor_list=[cfg.SOLVER.GAMMA ** i for i in range(len(steps) + 1)],
                                         warmup_factor=cfg.SOLVER.WARMUP_FACTOR, warmup_iters=cfg.SOLVER.WARMUP_ITERS,
                                         warmup_method=cfg.SOLVER.WARMUP_METHOD)
    raise ValueError(f"Unknown LR scheduler: {name}")
