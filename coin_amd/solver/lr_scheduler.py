"""coin/solver/lr_scheduler.py:22-66 (WarmupTwoStageMultiStepLR) as a closed-form schedule.

lr(it) = base_lr * warmup(it) * factor_list[bisect_right(milestones, it)]; warmup is detectron2's
`_get_warmup_factor_at_iter` (linear: warmup_factor*(1-a) + a, a = it/warmup_iters; 1 afterwards)."""
from bisect import bisect_right
from typing import List, Sequence


def warmup_factor_at_iter(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError(f"Unknown warmup method: {method}")


class WarmupTwoStageMultiStepLR:
    def __init__(self, optimizer, milestones: Sequence[int], factor_list: Sequence[float], gamma: float = 0.1,
                 warmup_factor: float = 0.001, warmup_iters: int = 1000, warmup_method: str = "linear", last_epoch: int = -1):
        if list(milestones) != sorted(milestones):
            raise ValueError(f"Milestones should be a list of increasing integers. Got {milestones}")
        if len(milestones) + 1 != len(factor_list):
            raise ValueError("Length of milestones should match length of factor_list.")
        self.optimizer, self.milestones, self.factor_list = optimizer, list(milestones), list(factor_list)
        self.gamma, self.warmup_factor, self.warmup_iters, self.warmup_method = gamma, warmup_factor, warmup_iters, warmup_method
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def factor(self, it: int) -> float:
        return warmup_factor_at_iter(self.warmup_method, it, self.warmup_iters, self.warmup_factor) * \
            self.factor_list[bisect_right(self.milestones, it)]

    def get_lr(self) -> List[float]:
        f = self.factor(self.last_epoch)
        return [b * f for b in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        if sd.get("base_lrs") is not None and len(sd["base_lrs"]) == len(self.base_lrs):
            self.base_lrs = list(sd["base_lrs"])
        self.last_epoch = sd["last_epoch"] - 1
        self.step()
