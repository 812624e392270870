"""AdamW on the library's multi-tensor kernel (include/dcpt_hip.h dcpt_adamw_step, csrc/optim.hip).

The reference builds ``torch.optim.AdamW`` from the YAML (basicsr/models/base_model.py:70-93) and calls ``optimizer.step()`` once per
training step (sr_model.py:118, degradation_classification_pretrain_model.py:170-173).  ``FusedAdamW`` is that optimizer for fp32 CUDA
parameters: same update rule and rounding order as torch's fused kernel, same ``param_groups`` keys and the same ``state_dict`` layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so a training state written by one loads into the other; what differs is the
launch structure (a handful of launches for the 664 tensors of NAFNet-64 instead of 19-23: 1.7 -> ~0.5 ms per step).

Not supported (ValueError / NotImplementedError at construction or at the first step): amsgrad, capturable, differentiable, sparse
gradients, non-fp32 or non-CUDA parameters, gradients whose layout differs from their parameter's.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from . import _lib


class _HParams(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("bias_correction1", C.c_double), ("bias_correction2", C.c_double), ("maximize", C.c_int)]


def _dense(t) -> bool:
    """non-overlapping and dense: the elements are exactly the first numel() entries of the storage behind data_ptr()"""
    want = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != want:
            return False
        want *= size
    return True


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=True):
        if amsgrad or capturable or differentiable:
            raise NotImplementedError("FusedAdamW: amsgrad / capturable / differentiable are not supported (use torch.optim.AdamW)")
        if isinstance(lr, torch.Tensor):
            raise ValueError("FusedAdamW: lr must be a Python number")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError(f"FusedAdamW: invalid hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        # the keys torch.optim.AdamW keeps in a param group (a state dict of either loads into the other)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize, foreach=None,
                        capturable=False, differentiable=False, fused=True)
        super().__init__(params, defaults)
        for group in self.param_groups:
            for p in group["params"]:
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise ValueError("FusedAdamW: fp32 CUDA parameters only (the library has no CPU path)")

    # ---- state: torch's layout; `step` kept as a Python float between steps (664 scalar tensor updates per step would cost more host
    #      time than the whole update takes on the device) and written back as a tensor by state_dict() -----------------------------
    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0.0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif isinstance(st["step"], torch.Tensor):   # loaded from a torch.optim.AdamW checkpoint
            st["step"] = float(st["step"])
        return st

    def state_dict(self):
        sd = super().state_dict()
        sd["state"] = {k: {kk: (torch.tensor(float(vv), dtype=torch.float32) if kk == "step" and not isinstance(vv, torch.Tensor) else vv)
                           for kk, vv in v.items()} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}

    def _plan(self, gi, group):
        """the per-group constants of a step -- pointer arrays of the parameters and their moments, element counts, strides -- built once
        (and again when the group's parameter list or a state tensor changed): a step then only collects the gradient pointers"""
        plans = self.__dict__.setdefault("_plans", {})
        ps = [p for p in group["params"] if p.grad is not None]
        key = tuple(id(p) for p in ps)
        plan = plans.get(gi)
        if plan is not None and plan["key"] == key and all(self.state[p]["exp_avg"] is m for p, m in zip(ps, plan["m"])):
            return plan
        for p in ps:
            if not _dense(p):
                raise RuntimeError("FusedAdamW: parameters must be dense (contiguous in some dimension order)")
            st = self._init_state(p)
            if st["exp_avg"].stride() != p.stride() or st["exp_avg_sq"].stride() != p.stride():   # (a checkpoint with another layout)
                st["exp_avg"] = torch.empty_like(p).copy_(st["exp_avg"])
                st["exp_avg_sq"] = torch.empty_like(p).copy_(st["exp_avg_sq"])
        n = len(ps)
        arr = C.c_void_p * n
        ms = [self.state[p]["exp_avg"] for p in ps]
        vs = [self.state[p]["exp_avg_sq"] for p in ps]
        plan = dict(key=key, ps=ps, m=ms, v=vs, n=n, arr=arr, strides=[p.stride() for p in ps], states=[self.state[p] for p in ps],
                    m_ptr=arr(*[t.data_ptr() for t in ms]), v_ptr=arr(*[t.data_ptr() for t in vs]),
                    numel=(C.c_int64 * n)(*[p.numel() for p in ps]), dev=ps[0].device.index if ps else None,
                    one_device=len({p.device for p in ps}) <= 1)
        plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            if plan["n"] == 0:
                continue
            if not plan["one_device"]:
                raise RuntimeError("FusedAdamW: the parameters of a group must live on one device")
            ps, strides = plan["ps"], plan["strides"]
            gs = []
            for p, strd in zip(ps, strides):
                g = p.grad
                if g.dtype != torch.float32 or g.stride() != strd or g.device != p.device or g.is_sparse:   # (the update walks the four storages in step)
                    if g.is_sparse:
                        raise RuntimeError("FusedAdamW does not support sparse gradients")
                    g2 = torch.empty_like(p)
                    g2.copy_(g)
                    g = p.grad = g2
                gs.append(g)
            # parameters that take the same step number share a call (normally: all of them)
            steps = {st["step"] for st in plan["states"]}
            b1, b2 = group["betas"]
            dev = plan["dev"]
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                for t0 in steps:
                    t = t0 + 1.0
                    h = _HParams(float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                 1.0 - float(b1) ** t, 1.0 - float(b2) ** t, int(bool(group["maximize"])))
                    if len(steps) == 1:
                        # (parameter pointers are read every step: ``p.data = ...`` / ``module.to()`` swap the storage behind the same object)
                        rc = lib.dcpt_adamw_step(plan["n"], plan["arr"](*[p.data_ptr() for p in ps]), plan["arr"](*[g.data_ptr() for g in gs]), plan["m_ptr"], plan["v_ptr"],
                                                 plan["numel"], C.byref(h), stream)
                    else:
                        idx = [i for i, st in enumerate(plan["states"]) if st["step"] == t0]
                        arr = C.c_void_p * len(idx)
                        rc = lib.dcpt_adamw_step(len(idx), arr(*[ps[i].data_ptr() for i in idx]), arr(*[gs[i].data_ptr() for i in idx]),
                                                 arr(*[plan["m"][i].data_ptr() for i in idx]), arr(*[plan["v"][i].data_ptr() for i in idx]),
                                                 (C.c_int64 * len(idx))(*[ps[i].numel() for i in idx]), C.byref(h), stream)
                    _lib.check(rc, "dcpt_adamw_step")
            for st in plan["states"]:
                st["step"] += 1.0
            # the kernel wrote through raw pointers: tell autograd (a step between a forward and its backward trips the saved-tensor
            # version check as with torch's optimizers) and every bf16 weight pack keyed on the parameters' versions
            torch.autograd.graph.increment_version(ps)
        return loss
