"""Data-parallel gradients without the per-parameter copies (reference basicsr/models/base_model.py:108-115 wraps the network in
torch's DistributedDataParallel; ``basicsr/`` here keeps that mechanism -- this module only removes what the wrap costs).

With ``gradient_as_bucket_view=True`` DDP wants every ``param.grad`` to BE a view of its bucket buffer; a gradient that arrives in its
own allocation is copied into the bucket (one small kernel per parameter: 664 for NAFNet-64) and, without a communication hook, divided
by the world size there (another one).  Two things remove both:

* ``prepare(ddp)`` registers torch's built-in all-reduce communication hook (divide the whole bucket once, then all-reduce it) and marks
  the network's parameters as DDP-managed;
* after every optimizer step (a global post-step hook in ``dcpt_amd.functional``) the bucket view that DDP left in ``param.grad`` is
  remembered on the parameter; the next backward pass of a fused block (``functional._grad_buffers``) lets its kernels write the
  parameter gradients STRAIGHT INTO those views and hands autograd an alias of each, which ``AccumulateGrad`` adopts as ``param.grad``
  -- DDP then finds the gradient already in place (``grad.is_alias_of(bucket_view)``) and copies nothing.

Safe by construction: a view is handed out at most once between two optimizer steps (a second use of the same parameter in one graph,
gradient accumulation under ``no_sync()``, a parameter whose ``.grad`` is still set -- all get an ordinary fresh tensor), and a stale view
(DDP re-buckets once, after its first iteration) merely brings the copy back."""
from __future__ import annotations

import torch


def prepare(ddp: torch.nn.parallel.DistributedDataParallel, hook=None):
    """call once, right after wrapping the network (``hook``: a Python communication hook to register instead -- the tests record the
    bucket order with one).  The default is torch's BUILT-IN C++ all-reduce hook: the same divide-then-all-reduce per bucket as the Python
    ``allreduce_hook``, but the Python one costs 12-13 ms per step on this workload (it runs on the autograd thread and fights the
    launching thread for the interpreter lock; tools/ddp_probe.py: 130 ms vs 118 ms)."""
    import torch.distributed as dist

    if hook is not None:
        ddp.register_comm_hook(None, hook)
    elif hasattr(ddp, "_register_builtin_comm_hook"):
        ddp._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
    else:
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

        ddp.register_comm_hook(None, default_hooks.allreduce_hook)
    for p in ddp.module.parameters():
        p._dcpt_ddp = True
    return ddp


def refresh_views(params) -> int:
    """remember the bucket views DDP left in ``.grad`` (called from the optimizer post-step hook); returns how many were found"""
    n = 0
    for p in params:
        if not getattr(p, "_dcpt_ddp", False):
            continue
        g = p.grad
        p._dcpt_view_busy = False
        if g is not None and g.shape == p.shape and g.is_contiguous() and g.dtype == p.dtype and g.device == p.device:
            p._dcpt_grad_view = g
            n += 1
    return n
