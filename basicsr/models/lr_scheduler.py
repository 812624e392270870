"""The two iteration-based schedules the reference's option files name (reference basicsr/models/lr_scheduler.py):
``MultiStepRestartLR`` (:7-53) and ``CosineAnnealingRestartLR`` (:75-131).  Both are written here as closed forms of the
iteration count, so a schedule can be evaluated (and resumed) at any iteration without replaying the earlier ones."""
from __future__ import annotations

import math
from bisect import bisect_left

from torch.optim.lr_scheduler import LRScheduler


class MultiStepRestartLR(LRScheduler):
    """lr = initial_lr * w_r * gamma^(number of milestones in (r, t]) where r is the latest restart <= t and w_r its weight."""

    def __init__(self, optimizer, milestones, gamma=0.1, restarts=(0,), restart_weights=(1,), last_epoch=-1):
        if len(restarts) != len(restart_weights):
            raise AssertionError("restarts and their weights do not match.")
        self.milestones = sorted(int(m) for m in milestones)
        self.gamma = gamma
        self.restarts = [int(r) for r in restarts]
        self.restart_weights = list(restart_weights)
        super().__init__(optimizer, last_epoch)

    def _factor(self, t):
        start, weight = 0, 1.0
        for r, w in sorted(zip(self.restarts, self.restart_weights)):
            if r <= t:
                start, weight = r, w
        decays = sum(1 for m in self.milestones if start < m <= t)
        return weight * self.gamma ** decays

    def get_lr(self):
        f = self._factor(self.last_epoch)
        return [g["initial_lr"] * f for g in self.optimizer.param_groups]


class CosineAnnealingRestartLR(LRScheduler):
    """Cycle i spans iterations (c_{i-1}, c_i] with c_i = periods[0] + ... + periods[i]:
    lr = eta_min_i + w_i * (base_lr - eta_min_i) * (1 + cos(pi * (t - c_{i-1}) / periods[i])) / 2."""

    def __init__(self, optimizer, periods, restart_weights=(1,), eta_min=0, last_epoch=-1):
        if len(periods) != len(restart_weights):
            raise AssertionError("periods and restart_weights should have the same length.")
        self.periods = [int(p) for p in periods]
        self.restart_weights = list(restart_weights)
        self.ends, total = [], 0
        for p in self.periods:
            total += p
            self.ends.append(total)
        self.eta_min = list(eta_min) if isinstance(eta_min, (list, tuple)) else [eta_min]
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t = self.last_epoch
        i = bisect_left(self.ends, t)          # first cycle whose end is >= t
        if i >= len(self.periods):
            raise IndexError(f"iteration {t} is past the last period of the schedule ({self.ends[-1]})")
        begin = self.ends[i - 1] if i > 0 else 0
        eta = self.eta_min[i] if len(self.eta_min) > 1 else self.eta_min[0]
        shape = 0.5 * (1.0 + math.cos(math.pi * (t - begin) / self.periods[i]))
        return [eta + self.restart_weights[i] * (base - eta) * shape for base in self.base_lrs]
