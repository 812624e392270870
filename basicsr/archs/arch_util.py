"""TLSC test-time local pooling support (reference basicsr/archs/arch_util.py:313-455: ``AvgPool2d``,
``replace_layers``, ``Local_Base``).  In this implementation the pooling itself runs inside the fused NAFBlock op
(``dcpt_nafblock_local_fwd``: separable box-mean kernels + MFMA 1x1 conv for the per-pixel attention map); the module
below records the window geometry exactly as the reference derives it."""
from __future__ import annotations

import torch
import torch.nn as nn


class AvgPool2d(nn.Module):
    def __init__(self, kernel_size=None, base_size=None, auto_pad=True, fast_imp=False, train_size=None):
        super().__init__()
        if fast_imp:
            raise NotImplementedError("fast_imp=True is a non-equivalent approximation in the reference (arch_util.py:355) and is not built")
        self.kernel_size = kernel_size
        self.base_size = base_size
        self.auto_pad = auto_pad
        self.fast_imp = fast_imp
        self.train_size = train_size

    def extra_repr(self):
        return f"kernel_size={self.kernel_size}, base_size={self.base_size}, stride={self.kernel_size}, fast_imp={self.fast_imp}"

    def set_kernel_from_feature(self, h, w):
        """reference :339-346: fixed on the first (dummy, train-size) forward of ``Local_Base.convert``"""
        if self.kernel_size is None and self.base_size:
            base = (self.base_size, self.base_size) if isinstance(self.base_size, int) else tuple(self.base_size)
            self.kernel_size = [h * base[0] // self.train_size[-2], w * base[1] // self.train_size[-1]]

    def forward(self, x):
        raise RuntimeError("AvgPool2d is a geometry holder here; it is applied inside the fused NAFBlock op")


def replace_layers(model, base_size, train_size, fast_imp, **kwargs):
    for name, child in model.named_children():
        if len(list(child.children())) > 0:
            replace_layers(child, base_size, train_size, fast_imp, **kwargs)
        if isinstance(child, nn.AdaptiveAvgPool2d):
            assert child.output_size == 1
            setattr(model, name, AvgPool2d(base_size=base_size, fast_imp=fast_imp, train_size=train_size))


class Local_Base:
    def convert(self, *args, train_size, **kwargs):
        """reference :450-455 replaces the pools and runs a dummy forward on ``torch.rand(train_size)`` only to let
        every pool see its feature-map size; the sizes are known in closed form, so no forward is needed."""
        replace_layers(self, *args, train_size=train_size, **kwargs)
        self._assign_local_kernels(train_size)
