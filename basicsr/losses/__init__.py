"""LOSS_REGISTRY entries the path's callers use (reference basicsr/losses/basic_loss.py:39-86,338-363;
plain torch expressions whose gradients seed the hot path -- SURVEY section 2 #17 keeps them in torch)."""
from copy import deepcopy

import torch
from torch import nn
from torch.nn import functional as F

from basicsr.utils import get_root_logger
from basicsr.utils.registry import LOSS_REGISTRY

__all__ = ["build_loss", "L1Loss", "MSELoss", "CharbonnierLoss", "PSNRLoss", "CrossEntropyLoss"]


def _reduce(loss, reduction):
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise ValueError(f"Unsupported reduction mode: {reduction}. Supported ones are: none | mean | sum")


@LOSS_REGISTRY.register()
class L1Loss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, weight=None, **kwargs):
        loss = (pred - target).abs()
        if weight is not None:
            loss = loss * weight
        return self.loss_weight * _reduce(loss, self.reduction)


@LOSS_REGISTRY.register()
class MSELoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, weight=None, **kwargs):
        loss = (pred - target) ** 2
        if weight is not None:
            loss = loss * weight
        return self.loss_weight * _reduce(loss, self.reduction)


@LOSS_REGISTRY.register()
class CharbonnierLoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction="mean", eps=1e-12):
        super().__init__()
        self.loss_weight, self.reduction, self.eps = loss_weight, reduction, eps

    def forward(self, pred, target, weight=None, **kwargs):
        loss = torch.sqrt((pred - target) ** 2 + self.eps)
        if weight is not None:
            loss = loss * weight
        return self.loss_weight * _reduce(loss, self.reduction)


@LOSS_REGISTRY.register()
class PSNRLoss(nn.Module):
    """10/ln(10) * log(mse + 1e-8) per image, averaged (reference basic_loss.py PSNRLoss)."""

    def __init__(self, loss_weight=1.0, reduction="mean", toY=False):
        super().__init__()
        assert reduction == "mean"
        self.loss_weight = loss_weight
        self.scale = 10 / torch.log(torch.tensor(10.0)).item()
        self.toY = toY

    def forward(self, pred, target):
        if self.toY:
            coef = torch.tensor([65.481, 128.553, 24.966], device=pred.device).reshape(1, 3, 1, 1)
            pred = ((pred * coef).sum(dim=1, keepdim=True) + 16.0) / 255.0
            target = ((target * coef).sum(dim=1, keepdim=True) + 16.0) / 255.0
        return self.loss_weight * self.scale * torch.log(((pred - target) ** 2).mean(dim=(1, 2, 3)) + 1e-8).mean()


@LOSS_REGISTRY.register()
class CrossEntropyLoss(nn.Module):
    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, **kwargs):
        return self.loss_weight * F.cross_entropy(pred, target, reduction=self.reduction)


def build_loss(opt):
    opt = deepcopy(opt)
    loss = LOSS_REGISTRY.get(opt.pop("type"))(**opt)
    get_root_logger().info(f"Loss [{loss.__class__.__name__}] is created.")
    return loss
