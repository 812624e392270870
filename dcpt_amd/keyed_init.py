"""Keyed deterministic parameter generator.

The reference ships no weights offline and zero-initialises NAFBlock ``beta`` /
``gamma`` (reference basicsr/archs/nafnet_arch.py:162-163), which turns every block into
the identity.  Parity fixtures and the benchmark therefore fill every state-dict entry
from a generator keyed by the entry's *name*: the same key + seed gives the same values
in this container (where the reference is importable) and on the GPU box (where it is
not), without storing full-size weights.

values(key) = scale(key) * N(0,1) [+ offset(key)], drawn from
``numpy.random.RandomState(crc32(key) ^ seed)``.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _scale_offset(key: str, shape) -> tuple[float, float]:
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("beta", "gamma"):
        return 0.5, 0.0  # non-trivial residual scales (reference inits them to 0)
    if leaf == "temperature":
        return 0.25, 1.0
    if leaf == "mixing_weights":
        return 0.5, 0.0
    parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    if leaf == "weight" and len(shape) == 1:
        return 0.25, 1.0  # norm scales around 1
    if leaf == "bias" and ("norm" in parent or parent in ("body",)):
        return 0.1, 0.0
    if leaf == "bias":
        return 0.05, 0.0
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= int(s)
        return 1.0 / max(1.0, float(fan_in)) ** 0.5, 0.0
    return 0.1, 0.0


def keyed_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed & 0xFFFFFFFF)) & 0xFFFFFFFF)
    scale, offset = _scale_offset(key, tuple(shape))
    arr = rs.standard_normal(size=tuple(shape)).astype(np.float32) * np.float32(scale) + np.float32(offset)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


def keyed_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {key: shape}.  Returns {key: float32 tensor} in the given key order."""
    return {k: keyed_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """Overwrite every float parameter/buffer of ``module`` with its keyed values."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            new[k] = keyed_tensor(k, tuple(v.shape), seed).to(dtype=v.dtype)
        else:
            new[k] = v
    module.load_state_dict(new, strict=True)
    return module


def keyed_input(tag: str, shape, seed: int = 0, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    rs = np.random.RandomState((zlib.crc32(("input:" + tag).encode()) ^ (seed & 0xFFFFFFFF)) & 0xFFFFFFFF)
    arr = rs.uniform(lo, hi, size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(arr)
