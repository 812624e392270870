"""torch.autograd bindings over the C ABI (include/dcpt_hip.h).

PyTorch is plumbing here: it owns device memory (activations, saved tensors, workspaces), the
stream, and autograd's graph.  All arithmetic happens in libdcpt_hip.so.

Layout: feature maps are ordinary torch tensors of logical shape (N, C, H, W) in
``torch.channels_last`` memory (= NHWC), so forward hooks, DDP and user code see the reference's
shapes while the kernels see contiguous channel rows.  The 3-channel image stays NCHW-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os as _os
from typing import Dict, Tuple

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _lib
from ._lib import NafBlockGrads, NafBlockParams, NafBlockSaved, PARAM_FIELDS, check

CL = torch.channels_last
_ws_cache: Dict[Tuple[int, int], torch.Tensor] = {}


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream); kernels using it are stream-ordered."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def release_workspaces():
    _ws_cache.clear()


def _require_gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DcptHipError(
                "dcpt_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise _lib.DcptHipError(f"dcpt_amd kernels are fp32 (got {t.dtype})")


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) tensor whose memory is dense NHWC."""
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got {tuple(x.shape)}")
    n, c, h, w = x.shape
    if x.stride() == (h * w * c, 1, w * c, c):
        return x
    return x.contiguous(memory_format=CL) if (c > 1 and h * w > 1) else _force_nhwc(x)


def _force_nhwc(x):
    n, c, h, w = x.shape
    out = torch.empty_strided((n, c, h, w), (h * w * c, 1, w * c, c), dtype=x.dtype, device=x.device)
    out.copy_(x)
    return out


def _empty_nhwc(n, c, h, w, dev) -> torch.Tensor:
    return torch.empty_strided((n, c, h, w), (h * w * c, 1, w * c, c), dtype=torch.float32, device=dev)


def _p(t):
    return 0 if t is None else t.data_ptr()


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# Opt-in GEMM precision mode (include/dcpt_hip.h dcpt_set_gemm_x3): "fp32" = the exact fp32 MFMA kernels (default, the reference's
# arithmetic); "bf16x3" = the wide fp32 GEMMs on the bf16 matrix pipe with operands split into three bf16 pieces (fp32-class results).
# The library's switch is process-wide; here it is driven per NETWORK (``network_g.gemm_precision`` in the options): a network's forward
# runs inside ``gemm_precision(mode)``, every autograd node built there remembers the mode and its backward runs under it again.
_X3_SCRATCH: Dict[int, torch.Tensor] = {}       # device index -> scratch for the split weight images
_GEMM_DEFAULT = ("fp32", 0)                     # the process default: (mode, min_tiles)
_GEMM_ACTIVE = ("fp32", 0)                      # what the library is switched to right now
_GEMM_ACTIVE_BUF = (None, 0)                    # (device index, bytes) of the scratch registered with the library while bf16x3 is on
_GEMM_SCRATCH_MB = 256
_SIDE_BEFORE_X3 = None                          # the side-stream setting the mode found when it was switched on


def _activate_gemm_mode(mode: str, min_tiles: int = 0, device=None):
    global _GEMM_ACTIVE, _GEMM_ACTIVE_BUF, _SIDE_BEFORE_X3
    if (mode, min_tiles) == _GEMM_ACTIVE:
        if mode == "fp32":
            return
        # already on: still re-register when the scratch has to grow (a later set_gemm_precision(scratch_mb=...)) or to move to
        # another device -- otherwise large launches would silently fall back to the fp32 kernels (round-4 advisor finding)
        want = torch.device(device).index if device is not None else _GEMM_ACTIVE_BUF[0]
        if want is None:
            want = torch.cuda.current_device()
        if want == _GEMM_ACTIVE_BUF[0] and _GEMM_ACTIVE_BUF[1] >= (_GEMM_SCRATCH_MB << 20):
            return
    lib = _lib.load()
    if mode == "fp32":
        check(lib.dcpt_set_gemm_x3(None, 0, 0), "dcpt_set_gemm_x3")
        if _SIDE_BEFORE_X3 is not None:
            lib.dcpt_set_side_stream(_SIDE_BEFORE_X3)   # what the caller had, not a hard-wired "on"
            _SIDE_BEFORE_X3 = None
    elif mode == "bf16x3":
        idx = torch.device(device).index if device is not None else torch.cuda.current_device()   # (one process per GPU: its own device)
        if idx is None:
            idx = torch.cuda.current_device()
        buf = _X3_SCRATCH.get(idx)
        if buf is None or buf.numel() < (_GEMM_SCRATCH_MB << 20):
            buf = _X3_SCRATCH[idx] = torch.empty(_GEMM_SCRATCH_MB << 20, dtype=torch.uint8, device=torch.device("cuda", idx))
        check(lib.dcpt_set_gemm_x3(buf.data_ptr(), buf.numel(), int(min_tiles)), "dcpt_set_gemm_x3")
        _GEMM_ACTIVE_BUF = (idx, buf.numel())
        # the split-operand kernels take the whole CU (160 KB of LDS, 8 waves): a weight-gradient block of the side stream cannot share a
        # CU with them (measured: 107.0 ms serialized vs 109.2 ms with it), so the side stream is off while the mode is on
        prev = lib.dcpt_set_side_stream(0)
        if _SIDE_BEFORE_X3 is None:
            _SIDE_BEFORE_X3 = prev
    else:
        raise ValueError(f"gemm precision {mode!r}: expected 'fp32' or 'bf16x3'")
    _GEMM_ACTIVE = (mode, min_tiles)


def set_gemm_precision(mode: str, device=None, scratch_mb: int = 256, min_tiles: int = 0) -> str:
    """Process default: 'fp32' (exact fp32 MFMA, the reference's arithmetic) or 'bf16x3'; returns the previous default.  ``scratch_mb``:
    room for the split images of one launch's weights (6 bytes per element; the per-image conv3 weights of a batch are the largest:
    B x C x C x 6) -- ``gemm_x3_scratch_misses()`` counts eligible launches that did not fit.  ``min_tiles`` > 0 lowers the size
    threshold (tests force the mode onto small launches with 1)."""
    global _GEMM_DEFAULT, _GEMM_SCRATCH_MB
    if mode not in ("fp32", "bf16x3"):
        raise ValueError(f"gemm precision {mode!r}: expected 'fp32' or 'bf16x3'")
    prev = _GEMM_DEFAULT[0]
    _GEMM_SCRATCH_MB = max(_GEMM_SCRATCH_MB, int(scratch_mb))
    _GEMM_DEFAULT = (mode, int(min_tiles))
    _activate_gemm_mode(mode, int(min_tiles), device)
    return prev


def get_gemm_precision() -> str:
    return _GEMM_ACTIVE[0]


def gemm_x3_scratch_misses() -> int:
    return int(_lib.load().dcpt_gemm_x3_scratch_misses())


class gemm_precision:
    """``with gemm_precision(mode):`` -- the GEMM precision for the kernels launched inside (None: leave it as it is); the autograd nodes
    created inside run their backward under the same mode."""

    def __init__(self, mode, min_tiles=None):
        self.want = None if mode is None else (mode, _GEMM_DEFAULT[1] if min_tiles is None else int(min_tiles))

    def __enter__(self):
        self.prev = _GEMM_ACTIVE
        if self.want is not None:
            _activate_gemm_mode(*self.want)
        return self

    def __exit__(self, *exc):
        if self.want is not None:
            _activate_gemm_mode(*self.prev)
        return False


def _remember_gemm_mode(cls):
    """forward notes the active mode on the node, backward re-activates it (a no-op comparison when nothing changed)"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._gemm_mode = _GEMM_ACTIVE
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        mode = getattr(ctx, "_gemm_mode", _GEMM_ACTIVE)
        if mode == _GEMM_ACTIVE:
            return bwd(ctx, *grads)
        prev = _GEMM_ACTIVE
        _activate_gemm_mode(*mode)
        try:
            return bwd(ctx, *grads)
        finally:
            _activate_gemm_mode(*prev)

    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


# ------------------------------------------------------------------------------------------------
def _saved_f32(t1, t2, y, v, stats, pooled, s, xn):
    """dcpt_nafblock_saved over the forward's buffers; v / stats / xn are None where the library does not use them (NULL pointers)."""
    st = [0, 0, 0, 0] if stats is None else [stats[i].data_ptr() for i in range(4)]
    xs = [0, 0, 0] if xn is None else [xn[i].data_ptr() for i in range(3)]
    return NafBlockSaved(t1.data_ptr(), t2.data_ptr(), y.data_ptr(), 0 if v is None else v.data_ptr(), st[0], st[1], st[2], st[3],
                         pooled.data_ptr(), s.data_ptr(), xs[0], xs[1], xs[2])


@_remember_gemm_mode
class _NAFBlockFn(torch.autograd.Function):
    """reference basicsr/archs/nafnet_arch.py:165-186 (NAFBlock.forward) -> dcpt_nafblock_fwd/bwd."""

    @staticmethod
    def forward(ctx, inp, grad_mode, *params):
        lib = _lib.load()
        _require_gpu(inp, *params)
        inp = _nhwc(inp)
        ctx.owners = params   # (the Parameter objects: their DDP bucket views, if any, receive the gradients)
        params = tuple(_contig(p.detach()) for p in params)
        B, Cc, H, W = inp.shape
        dev = inp.device
        M = B * H * W
        out = _empty_nhwc(B, Cc, H, W, dev)
        t1 = _empty_nhwc(B, 2 * Cc, H, W, dev)
        t2 = _empty_nhwc(B, Cc, H, W, dev)
        y = _empty_nhwc(B, Cc, H, W, dev)
        # where the forward 1 x 1 chains are fused (dcpt_nafblock_fused_ffn) LN1(inp), LN2(y) and the gate do not exist; with no
        # backward coming, v and the statistics are not written either
        fused = bool(lib.dcpt_nafblock_fused_ffn(Cc))
        # (needs_input_grad follows requires_grad alone -- it stays True under torch.no_grad() --, hence the caller's grad mode)
        infer = fused and not (grad_mode and any(ctx.needs_input_grad))
        _NAFBlockFn.last_infer = infer   # (read by the tests: which form the last forward took)
        v = None if infer else _empty_nhwc(B, 2 * Cc, H, W, dev)
        xn = None if fused else torch.empty((3, B, H, W, Cc), dtype=torch.float32, device=dev)   # LN1(inp), LN2(y), SimpleGate(v)
        stats = None if infer else torch.empty((4, M), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        s = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        ps = NafBlockParams(*[p.data_ptr() for p in params])
        sv = _saved_f32(t1, t2, y, v, stats, pooled, s, xn)
        nws = lib.dcpt_nafblock_fwd_ws_bytes(B, H, W, Cc)
        ws = _workspace(dev, nws)
        check(lib.dcpt_nafblock_fwd(C.byref(ps), inp.data_ptr(), out.data_ptr(), C.byref(sv), ws.data_ptr(), ws.numel(),
                                    B, H, W, Cc, _stream(dev)), "dcpt_nafblock_fwd")
        if not infer:
            ctx.has_xn = xn is not None
            ctx.save_for_backward(inp, t1, t2, y, v, stats, pooled, s, *(() if xn is None else (xn,)), *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        if ctx.has_xn:
            inp, t1, t2, y, v, stats, pooled, s, xn, *params = ctx.saved_tensors
        else:
            (inp, t1, t2, y, v, stats, pooled, s, *params), xn = ctx.saved_tensors, None
        dout = _nhwc(dout)
        B, Cc, H, W = inp.shape
        dev = inp.device
        grads = _grad_buffers(params, ctx.owners)
        dinp = _empty_nhwc(B, Cc, H, W, dev)
        ps = NafBlockParams(*[p.data_ptr() for p in params])
        gs = NafBlockGrads(*[g.data_ptr() for g in grads])
        sv = _saved_f32(t1, t2, y, v, stats, pooled, s, xn)
        nws = lib.dcpt_nafblock_bwd_ws_bytes(B, H, W, Cc)
        ws = _workspace(dev, nws)
        check(lib.dcpt_nafblock_bwd(C.byref(ps), C.byref(gs), inp.data_ptr(), C.byref(sv), dout.data_ptr(),
                                    dinp.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cc, _stream(dev)),
              "dcpt_nafblock_bwd")
        return (dinp, None, *grads)


def nafblock(inp: torch.Tensor, params: Dict[str, torch.Tensor]) -> torch.Tensor:
    """params: dict with the keys of _lib.PARAM_FIELDS (reference state-dict tensors)."""
    return _NAFBlockFn.apply(inp, torch.is_grad_enabled(), *[params[k] for k in PARAM_FIELDS])


# ------------------------------------------------------------------------------------------------
# bf16-storage path (BASELINE.json configs[2]): activations / saved tensors torch.bfloat16 (channels_last), parameters fp32
def _empty_nhwc_bf16(n, c, h, w, dev) -> torch.Tensor:
    return torch.empty_strided((n, c, h, w), (h * w * c, 1, w * c, c), dtype=torch.bfloat16, device=dev)


def _require_gpu_feat(*ts):
    """feature maps of the ops that exist in both storage types: fp32 or bf16, on the device"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DcptHipError("dcpt_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU fallback")
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.DcptHipError(f"feature maps are fp32 or bf16 (got {t.dtype})")


def _require_gpu_bf16(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.DcptHipError("dcpt_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.bfloat16:
            raise _lib.DcptHipError(f"the bf16-storage path takes torch.bfloat16 activations (got {t.dtype})")


def _saved_bf16(t1, v, acts, stats, sca, infer):
    """dcpt_nafblock_saved_bf16 over the forward's buffers.  acts / stats have 5 / 4 rows, or 3 / 2 where the fused second half keeps
    nothing but v (dcpt_nafblock_bf16_fused_ffn): the missing pointers are NULL when no backward follows (``infer``: the library then
    skips v as well) and stand-ins that the fused kernels never touch otherwise (the struct's contract asks for non-null in training)."""
    full = acts.shape[0] == 5
    if infer:
        return _lib.NafBlockSavedBf16(t1.data_ptr(), acts[0].data_ptr(), acts[1].data_ptr(), 0, stats[0].data_ptr(), stats[1].data_ptr(), 0, 0,
                                      sca[0].data_ptr(), sca[1].data_ptr(), acts[2].data_ptr(), 0, 0)
    # (v None: no backward follows at a width whose second half is three kernels -- the bias + gate epilogue then keeps the gate only)
    return _lib.NafBlockSavedBf16(t1.data_ptr(), acts[0].data_ptr(), acts[1].data_ptr(), 0 if v is None else v.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                  stats[2 if full else 0].data_ptr(), stats[3 if full else 1].data_ptr(), sca[0].data_ptr(), sca[1].data_ptr(),
                                  acts[2].data_ptr(), acts[3 if full else 2].data_ptr(), acts[4 if full else 2].data_ptr())


class _NAFBlockBf16Fn(torch.autograd.Function):
    """NAFBlock.forward (reference nafnet_arch.py:165-186) with bf16 storage -> dcpt_nafblock_fwd_bf16 / bwd_bf16."""

    @staticmethod
    def forward(ctx, inp, packed, grad_mode, *params):
        lib = _lib.load()
        _require_gpu_bf16(inp)
        _require_gpu(*params)
        inp = _nhwc(inp)
        ctx.owners = params
        params = tuple(_contig(p.detach()) for p in params)
        B, Cc, H, W = inp.shape
        if Cc % 8:
            raise _lib.DcptHipError(f"the bf16-storage path needs channel counts that are multiples of 8 (got {Cc})")
        dev = inp.device
        M = B * H * W
        out = _empty_nhwc_bf16(B, Cc, H, W, dev)
        t1 = _empty_nhwc_bf16(B, 2 * Cc, H, W, dev)
        # no backward coming (torch.no_grad / nothing requires grad) and the block's second half is one kernel at this width: the five
        # tensors only that backward reads (v, LN2(y), SimpleGate(v), LN2's statistics) are neither allocated nor written
        # where the block's second half is one kernel, its backward recomputes LN2(y), the gate and LN2's statistics: never allocated
        # (1: C = 64, ffn_bf16.hip -- never touched; 2: C = 256 / 512, chain_bf16.hip -- written for the backward pass only)
        ffn = int(lib.dcpt_nafblock_bf16_fused_ffn(Cc))
        fused = ffn == 1
        nograd = not (grad_mode and any(ctx.needs_input_grad))
        infer = ffn != 0 and nograd
        _NAFBlockBf16Fn.last_infer = infer
        v = None if nograd else _empty_nhwc_bf16(B, 2 * Cc, H, W, dev)   # (conv4's output is read by the backward only)
        slim = fused or infer
        acts = torch.empty((3 if slim else 5, B, H, W, Cc), dtype=torch.bfloat16, device=dev)   # t2, y, LN1(inp) [, LN2(y), SimpleGate(v)]
        stats = torch.empty((2 if slim else 4, M), dtype=torch.float32, device=dev)
        sca = torch.empty((2, B, Cc), dtype=torch.float32, device=dev)           # pooled, s
        ps = NafBlockParams(*[p.data_ptr() for p in params])
        sv = _saved_bf16(t1, v, acts, stats, sca, infer)
        ws = _workspace(dev, lib.dcpt_nafblock_fwd_bf16_ws_bytes(B, H, W, Cc))
        if packed is None:
            check(lib.dcpt_nafblock_fwd_bf16(C.byref(ps), inp.data_ptr(), out.data_ptr(), C.byref(sv), ws.data_ptr(), ws.numel(),
                                             B, H, W, Cc, _stream(dev)), "dcpt_nafblock_fwd_bf16")
        else:
            check(lib.dcpt_nafblock_fwd_bf16_packed(C.byref(ps), packed.data_ptr(), packed.numel(), inp.data_ptr(), out.data_ptr(), C.byref(sv),
                                                    ws.data_ptr(), ws.numel(), B, H, W, Cc, _stream(dev)), "dcpt_nafblock_fwd_bf16_packed")
        ctx.packed = packed   # (a plain byte buffer owned by the module; the backward of THIS forward reads the same pack)
        if not nograd:
            ctx.save_for_backward(inp, t1, v, acts, stats, sca, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inp, t1, v, acts, stats, sca, *params = ctx.saved_tensors
        dout = _nhwc(dout if dout.dtype == torch.bfloat16 else dout.to(torch.bfloat16))
        B, Cc, H, W = inp.shape
        dev = inp.device
        grads = _grad_buffers(params, ctx.owners)
        dinp = _empty_nhwc_bf16(B, Cc, H, W, dev)
        ps = NafBlockParams(*[p.data_ptr() for p in params])
        gs = NafBlockGrads(*[g.data_ptr() for g in grads])
        sv = _saved_bf16(t1, v, acts, stats, sca, False)
        ws = _workspace(dev, lib.dcpt_nafblock_bwd_bf16_ws_bytes(B, H, W, Cc))
        packed = ctx.packed
        if packed is None:
            check(lib.dcpt_nafblock_bwd_bf16(C.byref(ps), C.byref(gs), inp.data_ptr(), C.byref(sv), dout.data_ptr(), dinp.data_ptr(),
                                             ws.data_ptr(), ws.numel(), B, H, W, Cc, _stream(dev)), "dcpt_nafblock_bwd_bf16")
        else:
            check(lib.dcpt_nafblock_bwd_bf16_packed(C.byref(ps), packed.data_ptr(), packed.numel(), C.byref(gs), inp.data_ptr(), C.byref(sv),
                                                    dout.data_ptr(), dinp.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cc, _stream(dev)),
                  "dcpt_nafblock_bwd_bf16_packed")
        return (dinp, None, None, *grads)


# parameters whose values the packed operand copies are made from
_PACK_DEPS = ("conv1_w", "conv2_w", "conv3_w", "conv4_w", "conv5_w", "beta", "gamma")


# Generation counter of "the parameters may have changed".  torch's ``_version`` is NOT enough to detect that: fused / foreach
# optimizers (``AdamW(fused=True)``: aten::_fused_adamw_) and updates through ``p.data`` change a parameter's values without bumping
# the parameter's version counter.  Every optimizer step of the process bumps the generation (a global post-step hook), as do
# ``BaseModel.model_ema`` / ``load_network``; code that writes parameters any other way calls ``invalidate_packed_weights()``.
_PACK_GENERATION = 0


def invalidate_packed_weights() -> int:
    """Mark every cached operand copy of the weights (PackedWeightsBf16) stale; returns the new generation."""
    global _PACK_GENERATION
    _PACK_GENERATION += 1
    return _PACK_GENERATION


def _optimizer_step_post_hook(optimizer, args, kwargs):
    # only an optimizer that owns a parameter some pack was made from makes the packs stale (in the DCPT step optimizer_dc.step() must not
    # make the encoder's blocks repack -- with freeze_encoder it is the only optimizer that ever steps); parameters are tagged when packed
    if any(getattr(p, "_dcpt_packdep", False) for grp in optimizer.param_groups for p in grp["params"]):
        invalidate_packed_weights()
    # DistributedDataParallel with gradient_as_bucket_view: remember the bucket views now sitting in .grad (dcpt_amd/ddp.py)
    first = next((p for grp in optimizer.param_groups for p in grp["params"]), None)
    if first is not None and getattr(first, "_dcpt_ddp", False):
        from .ddp import refresh_views

        refresh_views(p for grp in optimizer.param_groups for p in grp["params"])


def _grad_buffers(saved, owners):
    """Output tensors for a fused block's parameter gradients: a fresh tensor per parameter, or -- where DDP's bucket view of that parameter
    is known and not yet used since the last optimizer step (dcpt_amd/ddp.py) -- an alias of the view, so that the kernels write the
    gradient where the all-reduce will read it."""
    outs = []
    for t, p in zip(saved, owners or (None,) * len(saved)):
        v = getattr(p, "_dcpt_grad_view", None) if p is not None else None
        if (v is not None and p.grad is None and not getattr(p, "_dcpt_view_busy", True) and v.shape == t.shape and v.device == t.device
                and v.dtype == t.dtype):
            p._dcpt_view_busy = True
            _grad_buffers.hits += 1
            outs.append(v.view(v.shape))   # a new tensor object on the bucket's storage: AccumulateGrad adopts it as param.grad
        else:
            outs.append(torch.empty_like(t))
    return outs


_grad_buffers.hits = 0   # (gradients written straight into a DDP bucket view so far: read by the tests and bench.py)


register_optimizer_step_post_hook(_optimizer_step_post_hook)


class PackedWeightsBf16:
    """Per-block cache of the operand copies of a NAFBlock's weights for the bf16 path (dcpt_nafblock_wpack_bf16): refreshed -- one
    launch -- when the parameters may have changed since the last pack, reused otherwise: within a training step the forward, the
    backward and, in the DCPT step, both encoder passes share one pack.  "May have changed" = a new generation (any optimizer step,
    EMA update or checkpoint load of the process, ``invalidate_packed_weights()``), a bumped ``_version`` (ordinary in-place updates:
    ``load_state_dict``, ``.copy_``) or a new address (``.to(device)``)."""

    def __init__(self):
        self.key = None
        self.buf = None
        self.stream = None   # the stream the pack was made on, and the event behind it: another stream waits before it reads the pack
        self.event = None    # (tiled inference runs its tile batches on several streams, basicsr/models/sr_model.py::test_tile)

    def __deepcopy__(self, memo):   # a copied module (EMA network, DDP replica) starts with an empty cache of its own
        return type(self)()

    def __reduce__(self):
        return (type(self), ())

    @staticmethod
    def key_of(params: Dict[str, torch.Tensor]):
        return (_PACK_GENERATION,) + tuple((params[k].data_ptr(), params[k]._version) for k in _PACK_DEPS)

    @staticmethod
    def tag(params: Dict[str, torch.Tensor]):
        """mark the parameters a pack depends on: an optimizer step over any of them invalidates the packs (_optimizer_step_post_hook)"""
        for k in _PACK_DEPS:
            params[k]._dcpt_packdep = True

    def get(self, params: Dict[str, torch.Tensor]) -> torch.Tensor:
        key = self.key_of(params)
        if key == self.key:
            if self.stream != _stream(self.buf.device):
                torch.cuda.current_stream(self.buf.device).wait_event(self.event)
            return self.buf
        if key != self.key:
            lib = _lib.load()
            ps = tuple(_contig(params[k].detach()) for k in PARAM_FIELDS)
            _require_gpu(*ps)
            dev = ps[0].device
            Cc = params["conv3_w"].shape[0]
            nbytes = lib.dcpt_nafblock_wpack_bf16_bytes(Cc)
            if self.buf is None or self.buf.numel() != nbytes or self.buf.device != dev:
                self.buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            pp = NafBlockParams(*[p.data_ptr() for p in ps])
            check(lib.dcpt_nafblock_wpack_bf16(C.byref(pp), self.buf.data_ptr(), self.buf.numel(), Cc, _stream(dev)), "dcpt_nafblock_wpack_bf16")
            self.key = key
            self.tag(params)
            self.stream = _stream(dev)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(dev))
        return self.buf


def pack_blocks_bf16(blocks) -> int:
    """Refresh the weight packs of several NAFBlocks in ceil(n / 8) launches (dcpt_nafblock_wpack_bf16_multi) instead of one per block:
    ``blocks`` = [(PackedWeightsBf16, params dict), ...] of a network, called before its forward; only the stale ones are packed, and
    ``PackedWeightsBf16.get`` then finds them current.  Returns how many were packed."""
    stale = []
    for pk, params in blocks:
        key = pk.key_of(params)
        if key != pk.key:
            stale.append((pk, params, key))
    if len(stale) < 2:
        return 0   # (a single block packs itself in get())
    lib = _lib.load()
    n = len(stale)
    ps_arr = (NafBlockParams * n)()
    bufs, nbytes, widths, keep = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_int * n)(), []
    dev = None
    for i, (pk, params, key) in enumerate(stale):
        ps = tuple(_contig(params[k].detach()) for k in PARAM_FIELDS)
        _require_gpu(*ps)
        keep.append(ps)
        dev = ps[0].device
        Cc = params["conv3_w"].shape[0]
        need = lib.dcpt_nafblock_wpack_bf16_bytes(Cc)
        if pk.buf is None or pk.buf.numel() != need or pk.buf.device != dev:
            pk.buf = torch.empty(need, dtype=torch.uint8, device=dev)
        ps_arr[i] = NafBlockParams(*[p.data_ptr() for p in ps])
        bufs[i], nbytes[i], widths[i] = pk.buf.data_ptr(), need, Cc
    check(lib.dcpt_nafblock_wpack_bf16_multi(ps_arr, bufs, nbytes, widths, n, _stream(dev)), "dcpt_nafblock_wpack_bf16_multi")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    for pk, params, key in stale:
        pk.key, pk.stream, pk.event = key, _stream(dev), ev
        pk.tag(params)
    return n


class PackedConvBf16:
    """Cache of ONE conv's two bf16 operand images (forward, data gradient) for the classifier head's bf16 path (dcpt_conv_wpack_bf16_multi,
    ABI 14): the same staleness rule as PackedWeightsBf16 (generation / ``_version`` / address).  ``pack_convs_bf16`` refreshes a whole head
    in three launches before its forward; without it every conv entry point packs its own image per call (68 launches per DCPT step)."""

    def __init__(self):
        self.key = None
        self.buf = None
        self.stream = None
        self.event = None

    def __deepcopy__(self, memo):
        return type(self)()

    def __reduce__(self):
        return (type(self), ())

    @staticmethod
    def key_of(weight: torch.Tensor):
        return (_PACK_GENERATION, weight.data_ptr(), weight._version, tuple(weight.shape))

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        if self.key_of(weight) != self.key:
            pack_convs_bf16([(self, weight)])
        elif self.stream != _stream(self.buf.device):
            torch.cuda.current_stream(self.buf.device).wait_event(self.event)
        return self.buf


CONV_PACK_CACHE = _os.environ.get("DCPT_CONV_PACK_CACHE", "1") != "0"   # False: the head's conv entry points pack per call again (A/B, the bit-identity test)


def pack_convs_bf16(convs) -> int:
    """Refresh the stale ones of ``convs`` = [(PackedConvBf16, weight [Cout][Cin][k][k] fp32), ...] in one call of dcpt_conv_wpack_bf16_multi;
    returns how many were packed."""
    if not CONV_PACK_CACHE:
        return 0
    stale = [(pk, w, pk.key_of(w)) for pk, w in convs if pk.key_of(w) != pk.key]
    if not stale:
        return 0
    lib = _lib.load()
    n = len(stale)
    ws, bufs, nbytes = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_size_t * n)()
    cin, cout, ksz, keep = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)(), []
    dev = None
    for i, (pk, w, key) in enumerate(stale):
        w_ = _contig(w.detach())
        _require_gpu(w_)
        keep.append(w_)
        dev = w_.device
        Co, Ci, ks = w_.shape[0], w_.shape[1], w_.shape[2]
        need = lib.dcpt_conv_wpack_bf16_bytes(Ci, Co, ks)
        if pk.buf is None or pk.buf.numel() != need or pk.buf.device != dev:
            pk.buf = torch.empty(need, dtype=torch.uint8, device=dev)
        ws[i], bufs[i], nbytes[i], cin[i], cout[i], ksz[i] = w_.data_ptr(), pk.buf.data_ptr(), need, Ci, Co, ks
    check(lib.dcpt_conv_wpack_bf16_multi(ws, bufs, nbytes, cin, cout, ksz, n, _stream(dev)), "dcpt_conv_wpack_bf16_multi")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    for pk, w, key in stale:
        pk.key, pk.stream, pk.event = key, _stream(dev), ev
        w._dcpt_packdep = True
    return n


def _pk(packed, weight):
    """(pointer, bytes) of a conv's cached operand images -- (None, 0) without a cache: the entry point packs in the call"""
    if packed is None or not CONV_PACK_CACHE:
        return None, 0
    buf = packed.get(weight)
    return buf.data_ptr(), buf.numel()


def nafblock_bf16(inp: torch.Tensor, params: Dict[str, torch.Tensor], packed: PackedWeightsBf16 = None) -> torch.Tensor:
    """bf16 activations in / out, fp32 parameters (dict with the keys of _lib.PARAM_FIELDS); ``packed``: the block's weight-pack cache
    (without it every call packs its own operand copies)."""
    buf = packed.get(params) if packed is not None else None
    return _NAFBlockBf16Fn.apply(inp, buf, torch.is_grad_enabled(), *[params[k] for k in PARAM_FIELDS])


def conv1x1_wgrad_bf16(dy: torch.Tensor, x: torch.Tensor, with_bias: bool = True):
    """Weight (and bias) gradient of a 1 x 1 convolution in bf16 storage (dcpt_conv1x1_wgrad_bf16; the four such products of
    NAFBlock's backward, reference nafnet_arch.py:170-186): dy [M][N], x [M][K] bf16 row-major (or NHWC feature maps) ->
    dW [N][K] fp32 (= dy^T x) and db [N] fp32 (column sums of dy)."""
    lib = _lib.load()
    _require_gpu_bf16(dy, x)
    dy2 = (_nhwc(dy).permute(0, 2, 3, 1) if dy.dim() == 4 else dy).reshape(-1, dy.shape[1] if dy.dim() == 4 else dy.shape[-1])
    x2 = (_nhwc(x).permute(0, 2, 3, 1) if x.dim() == 4 else x).reshape(-1, x.shape[1] if x.dim() == 4 else x.shape[-1])
    dy2, x2 = _contig(dy2), _contig(x2)
    M, N = dy2.shape
    K = x2.shape[1]
    if x2.shape[0] != M:
        raise ValueError(f"conv1x1_wgrad_bf16: {tuple(dy2.shape)} vs {tuple(x2.shape)}")
    dev = dy2.device
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev) if with_bias else None
    nws = lib.dcpt_conv1x1_wgrad_bf16_ws_bytes(M, N, K)
    if nws == 0:
        raise _lib.DcptHipError(f"conv1x1_wgrad_bf16: N={N}, K={K} must be positive multiples of 8")
    ws = _workspace(dev, nws)
    check(lib.dcpt_conv1x1_wgrad_bf16(dy2.data_ptr(), x2.data_ptr(), dW.data_ptr(), _p(db), ws.data_ptr(), ws.numel(), M, N, K, _stream(dev)),
          "dcpt_conv1x1_wgrad_bf16")
    return (dW, db) if with_bias else dW


class _CastFn(torch.autograd.Function):
    """edge of the bf16 path: fp32 -> bf16 (RNE) in forward, the bf16 gradient back to fp32 in backward (or the reverse)"""

    @staticmethod
    def forward(ctx, x, to_bf16):
        lib = _lib.load()
        ctx.to_bf16 = bool(to_bf16)
        return _cast(lib, x, ctx.to_bf16)

    @staticmethod
    def backward(ctx, g):
        return _cast(_lib.load(), g, not ctx.to_bf16), None


def _cast(lib, x, to_bf16):
    src, dst = (torch.float32, torch.bfloat16) if to_bf16 else (torch.bfloat16, torch.float32)
    if x.dtype == dst:
        return x
    if not x.is_cuda or x.dtype != src:
        raise _lib.DcptHipError(f"cast: expected a {src} device tensor, got {x.dtype} on {x.device}")
    x = _nhwc(x) if x.dim() == 4 else _contig(x)
    if x.numel() % 8:
        raise _lib.DcptHipError("cast: element count must be a multiple of 8")
    y = torch.empty_like(x, dtype=dst)   # preserves the (dense NHWC) strides
    fn = lib.dcpt_cast_f32_bf16 if to_bf16 else lib.dcpt_cast_bf16_f32
    check(fn(x.data_ptr(), y.data_ptr(), x.numel(), _stream(x.device)), "dcpt_cast")
    return y


def to_bf16(x):
    return x if x.dtype == torch.bfloat16 else _CastFn.apply(x, True)


def to_f32(x):
    return x if x.dtype == torch.float32 else _CastFn.apply(x, False)


def nafblock_local(inp: torch.Tensor, params: Dict[str, torch.Tensor], k1: int, k2: int) -> torch.Tensor:
    """TLSC inference forward (reference nafnet_arch.py:277-288 + arch_util.py:313-455): SCA with a k1 x k2 local mean."""
    if torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in params.values())):
        raise NotImplementedError("the TLSC NAFNet variant is inference-only: call it under torch.no_grad()")
    lib = _lib.load()
    ps = [_contig(params[k].detach()) for k in PARAM_FIELDS]
    _require_gpu(inp, *ps)
    inp = _nhwc(inp)
    B, Cc, H, W = inp.shape
    out = _empty_nhwc(B, Cc, H, W, inp.device)
    pp = NafBlockParams(*[p.data_ptr() for p in ps])
    ws = _workspace(inp.device, lib.dcpt_nafblock_local_ws_bytes(B, H, W, Cc, k1, k2))
    check(lib.dcpt_nafblock_local_fwd(C.byref(pp), inp.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cc,
                                      int(k1), int(k2), _stream(inp.device)), "dcpt_nafblock_local_fwd")
    return out


# ------------------------------------------------------------------------------------------------
class _TakeBatchFn(torch.autograd.Function):
    """x[lo:hi] along the batch axis as a view; the backward builds the zero-padded gradient WITH THE INPUT'S STRIDES (torch's own
    slice backward allocates it NCHW-contiguous, which costs two layout conversions of the whole feature map per use for NHWC
    tensors: one into the NCHW buffer and one back when the producing block reads it)."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.meta = (tuple(x.shape), tuple(x.stride()), x.dtype, x.device, int(lo), int(hi))
        return x[lo:hi]

    @staticmethod
    def backward(ctx, g):
        shape, stride, dtype, dev, lo, hi = ctx.meta
        # the input's own strides when they describe a dense, non-overlapping layout (contiguous or channels_last: sorted by
        # stride, the strides telescope over the sizes); anything else (expanded / overlapping views) gets a contiguous gradient.
        # Computed arithmetically -- no throw-away allocation of the feature map's size.
        keep, run = len(shape) == 4, 1
        for st, sz in sorted((st, sz) for st, sz in zip(stride, shape) if sz > 1):
            keep, run = keep and st == run, run * sz
        out = torch.empty_strided(shape, stride, dtype=dtype, device=dev) if keep else torch.empty(shape, dtype=dtype, device=dev)
        if lo > 0:
            out[:lo].zero_()
        if hi < shape[0]:
            out[hi:].zero_()
        out[lo:hi].copy_(g)
        return out, None, None


def _private_grad(g) -> bool:
    """True when the incoming gradient may be mutated in place: a dense tensor that owns its storage and that nothing but the autograd
    engine's input buffer (and this frame) references.  A gradient that the consumer's backward hands to several inputs (torch ``add``),
    a view / expanded gradient (``sum``, slicing) or one kept alive elsewhere shows up as a second owner of the tensor or of its storage
    (probed on torch 2.10: private = tensor use count 2, storage use count 2; ``a + b`` = 3 / 3; ``sum`` = a view).  Anything this
    cannot prove private is cloned by the caller."""
    try:
        if g._base is not None or g.requires_grad or g.grad_fn is not None:
            return False
        if g._use_count() > 2 or torch._C._storage_Use_Count(g.untyped_storage()._cdata) > 2:
            return False
        return torch._prims_common.is_non_overlapping_and_dense(g)
    except (AttributeError, RuntimeError):   # (a torch without the probes: never mutate)
        return False


class _TapSplitFn(torch.autograd.Function):
    """(x, x[lo:hi]) for a feature map with TWO consumers -- the network goes on with x, a tap (the classifier head of the DCPT step,
    reference ...pretrain_model.py:60-68,140-163) takes samples lo..hi-1.  With two separate uses autograd builds the tap's zero-padded
    gradient (a fill + a copy of the feature map's size) and then ADDS the two gradients (three more passes, 18 bf16 `add` launches per
    DCPT step at 256 x 256: 1.4 ms); here the tap's gradient is added INTO rows lo..hi-1 of the main path's gradient, in place: one read
    of the slice and a read-modify-write of the same rows -- when that gradient is provably private (_private_grad: the fresh output of
    the layer above's backward, as in the NAFNet / Restormer graphs); otherwise into a copy of it."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.meta = (tuple(x.shape), tuple(x.stride()), x.dtype, x.device, int(lo), int(hi))
        return x.view(x.shape), x[lo:hi]

    @staticmethod
    def backward(ctx, g, gs):
        shape, stride, dtype, dev, lo, hi = ctx.meta
        if g is None:
            if gs is None:
                return None, None, None
            return _TakeBatchFn.backward(ctx, gs)
        if gs is not None:
            if g.shape != tuple(shape) or not g.is_floating_point() or not _private_grad(g):
                g = g.clone()
            g[lo:hi].add_(gs.to(g.dtype))
        return g, None, None


def tap_split(x, lo, hi):
    """-> (x, x[lo:hi]) with the two gradients merged in place on the way back (one use of x downstream, one tap)"""
    return _TapSplitFn.apply(x, lo, hi)


def take_batch(x, lo, hi):
    """samples lo..hi-1 of a batch (a view in forward, a stride-preserving zero-padded gradient in backward)"""
    return _TakeBatchFn.apply(x, lo, hi)


# ------------------------------------------------------------------------------------------------
class _LayerNorm2dFn(torch.autograd.Function):
    """reference nafnet_arch.py:25-53 (LayerNormFunction)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        _require_gpu(x, weight, bias)
        x = _nhwc(x)
        B, Cc, H, W = x.shape
        M = B * H * W
        y = _empty_nhwc(B, Cc, H, W, x.device)
        stats = torch.empty((2, M), dtype=torch.float32, device=x.device)
        w_, b_ = _contig(weight.detach()), _contig(bias.detach())
        check(lib.dcpt_ln2d_fwd(x.data_ptr(), w_.data_ptr(), b_.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                stats[1].data_ptr(), M, Cc, float(eps), _stream(x.device)), "dcpt_ln2d_fwd")
        ctx.save_for_backward(x, stats, w_)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, w_ = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        M = B * H * W
        dx = _empty_nhwc(B, Cc, H, W, x.device)
        dw = torch.empty_like(w_)
        db = torch.empty_like(w_)
        nws = lib.dcpt_ln2d_bwd_ws_bytes(M, Cc)
        ws = _workspace(x.device, nws)
        check(lib.dcpt_ln2d_bwd(dy.data_ptr(), x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w_.data_ptr(),
                                dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), M, Cc,
                                _stream(x.device)), "dcpt_ln2d_bwd")
        return dx, dw, db, None


def layernorm2d(x, weight, bias, eps=1e-6):
    return _LayerNorm2dFn.apply(x, weight, bias, eps)


# ------------------------------------------------------------------------------------------------
class _IntroFn(torch.autograd.Function):
    """3x3 conv, image NCHW -> features NHWC (reference nafnet_arch.py:202-210, :252)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _require_gpu(x, weight, bias)
        x = _contig(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        y = _empty_nhwc(B, Cout, H, W, x.device)
        check(lib.dcpt_conv3x3_in_fwd(x.data_ptr(), w_.data_ptr(), _p(b_), y.data_ptr(), B, H, W, Cin, Cout,
                                      _stream(x.device)), "dcpt_conv3x3_in_fwd")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dw = torch.empty_like(w_)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        nws = lib.dcpt_conv3x3_in_bwd_ws_bytes(B, H, W, Cin, Cout)
        ws = _workspace(dev, nws)
        check(lib.dcpt_conv3x3_in_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), _p(dx), dw.data_ptr(), db.data_ptr(),
                                      ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, _stream(dev)), "dcpt_conv3x3_in_bwd")
        return dx, dw, (db if ctx.has_bias else None)


class _EndingFn(torch.autograd.Function):
    """3x3 conv, features NHWC -> image NCHW, + residual image (reference nafnet_arch.py:211-219, :271-272)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        lib = _lib.load()
        _require_gpu(x, weight, bias, res)
        x = _nhwc(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        res_ = None if res is None else _contig(res)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        check(lib.dcpt_conv3x3_out_fwd(x.data_ptr(), w_.data_ptr(), _p(b_), _p(res_), y.data_ptr(), B, H, W, Cin, Cout,
                                       _stream(x.device)), "dcpt_conv3x3_out_fwd")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _contig(dy)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dx = _empty_nhwc(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
        nws = lib.dcpt_conv3x3_out_bwd_ws_bytes(B, H, W, Cin, Cout)
        ws = _workspace(dev, nws)
        check(lib.dcpt_conv3x3_out_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                       db.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, _stream(dev)),
              "dcpt_conv3x3_out_bwd")
        return dx, dw, (db if ctx.has_bias else None), (dy if ctx.has_res and ctx.needs_input_grad[3] else None)


class _IntroBf16Fn(torch.autograd.Function):
    """_IntroFn with the feature side in bf16 storage (edge_bf16.hip): fp32 image in, bf16 NHWC features out."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _require_gpu(x, weight, bias)
        x = _contig(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        y = _empty_nhwc_bf16(B, Cout, H, W, x.device)
        check(lib.dcpt_conv3x3_in_fwd_bf16(x.data_ptr(), w_.data_ptr(), _p(b_), y.data_ptr(), B, H, W, Cin, Cout, _stream(x.device)),
              "dcpt_conv3x3_in_fwd_bf16")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        _require_gpu_bf16(dy)
        dy = _nhwc(dy)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dw = torch.empty_like(w_)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = _workspace(dev, lib.dcpt_conv3x3_in_bwd_ws_bytes(B, H, W, Cin, Cout))
        check(lib.dcpt_conv3x3_in_bwd_bf16(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), _p(dx), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                           ws.numel(), B, H, W, Cin, Cout, _stream(dev)), "dcpt_conv3x3_in_bwd_bf16")
        return dx, dw, (db if ctx.has_bias else None)


class _EndingBf16Fn(torch.autograd.Function):
    """_EndingFn with bf16 NHWC features in, fp32 image (+ residual image) out."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        lib = _lib.load()
        _require_gpu_bf16(x)
        _require_gpu(weight, bias, res)
        x = _nhwc(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        res_ = None if res is None else _contig(res)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        check(lib.dcpt_conv3x3_out_fwd_bf16(x.data_ptr(), w_.data_ptr(), _p(b_), _p(res_), y.data_ptr(), B, H, W, Cin, Cout,
                                            _stream(x.device)), "dcpt_conv3x3_out_fwd_bf16")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _contig(dy)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dx = _empty_nhwc_bf16(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_conv3x3_out_bwd_ws_bytes(B, H, W, Cin, Cout))
        check(lib.dcpt_conv3x3_out_bwd_bf16(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                            ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, _stream(dev)), "dcpt_conv3x3_out_bwd_bf16")
        return dx, dw, (db if ctx.has_bias else None), (dy if ctx.has_res and ctx.needs_input_grad[3] else None)


def conv3x3_in(x, weight, bias, out_bf16=False):
    """3x3 conv image -> features; ``out_bf16``: emit bf16-storage features (the first layer of the bf16 path)"""
    return _IntroBf16Fn.apply(x, weight, bias) if out_bf16 else _IntroFn.apply(x, weight, bias)


def conv3x3_out(x, weight, bias, res=None):
    if x.dtype == torch.bfloat16:
        return _EndingBf16Fn.apply(x, weight, bias, res)
    return _EndingFn.apply(x, weight, bias, res)


# ------------------------------------------------------------------------------------------------
@_remember_gemm_mode
class _DownFn(torch.autograd.Function):
    """Conv2d(C, 2C, 2, 2) (reference nafnet_arch.py:230)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _require_gpu(x, weight, bias)
        x = _nhwc(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        B, Cc, H, W = x.shape
        if H % 2 or W % 2:
            raise ValueError(f"down2x2: H={H}, W={W} must be even")
        y = _empty_nhwc(B, 2 * Cc, H // 2, W // 2, x.device)
        nws = lib.dcpt_down2x2_ws_bytes(B, H, W, Cc, 0)
        ws = _workspace(x.device, nws)
        check(lib.dcpt_down2x2_fwd(x.data_ptr(), w_.data_ptr(), _p(b_), y.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W,
                                   Cc, _stream(x.device)), "dcpt_down2x2_fwd")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc(B, Cc, H, W, dev)
        dw = torch.empty_like(w_)
        db = torch.empty((2 * Cc,), dtype=torch.float32, device=dev)
        nws = lib.dcpt_down2x2_ws_bytes(B, H, W, Cc, 1)
        ws = _workspace(dev, nws)
        check(lib.dcpt_down2x2_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                   ws.data_ptr(), ws.numel(), B, H, W, Cc, _stream(dev)), "dcpt_down2x2_bwd")
        return dx, dw, (db if ctx.has_bias else None)


@_remember_gemm_mode
class _UpFn(torch.autograd.Function):
    """Conv2d(C, 2C, 1, bias=False) + PixelShuffle(2) + skip add (reference nafnet_arch.py:238-242, :264-265)."""

    @staticmethod
    def forward(ctx, x, weight, skip):
        lib = _lib.load()
        _require_gpu(x, weight, skip)
        x = _nhwc(x)
        w_ = _contig(weight.detach())
        skip_ = None if skip is None else _nhwc(skip)
        B, Cc, H, W = x.shape
        y = _empty_nhwc(B, Cc // 2, 2 * H, 2 * W, x.device)
        nws = lib.dcpt_up_ps_ws_bytes(B, H, W, Cc, 0)
        ws = _workspace(x.device, nws)
        check(lib.dcpt_up_ps_fwd(x.data_ptr(), w_.data_ptr(), _p(skip_), y.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W,
                                 Cc, _stream(x.device)), "dcpt_up_ps_fwd")
        ctx.save_for_backward(x, w_)
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc(B, Cc, H, W, dev)
        dw = torch.empty_like(w_)
        nws = lib.dcpt_up_ps_ws_bytes(B, H, W, Cc, 1)
        ws = _workspace(dev, nws)
        check(lib.dcpt_up_ps_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                 ws.numel(), B, H, W, Cc, _stream(dev)), "dcpt_up_ps_bwd")
        return dx, dw, (dy if ctx.has_skip else None)


class _DownBf16Fn(torch.autograd.Function):
    """_DownFn in bf16 storage (edge_bf16.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _require_gpu_bf16(x)
        _require_gpu(weight, bias)
        x = _nhwc(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        B, Cc, H, W = x.shape
        if H % 2 or W % 2:
            raise ValueError(f"down2x2: H={H}, W={W} must be even")
        y = _empty_nhwc_bf16(B, 2 * Cc, H // 2, W // 2, x.device)
        ws = _workspace(x.device, lib.dcpt_down2x2_bf16_ws_bytes(B, H, W, Cc, 0))
        check(lib.dcpt_down2x2_fwd_bf16(x.data_ptr(), w_.data_ptr(), _p(b_), y.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cc,
                                        _stream(x.device)), "dcpt_down2x2_fwd_bf16")
        ctx.save_for_backward(x, w_)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        _require_gpu_bf16(dy)
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc_bf16(B, Cc, H, W, dev)
        dw = torch.empty_like(w_)
        db = torch.empty((2 * Cc,), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_down2x2_bf16_ws_bytes(B, H, W, Cc, 1))
        check(lib.dcpt_down2x2_bwd_bf16(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                        ws.numel(), B, H, W, Cc, _stream(dev)), "dcpt_down2x2_bwd_bf16")
        return dx, dw, (db if ctx.has_bias else None)


class _UpBf16Fn(torch.autograd.Function):
    """_UpFn in bf16 storage (edge_bf16.hip)."""

    @staticmethod
    def forward(ctx, x, weight, skip):
        lib = _lib.load()
        _require_gpu_bf16(x, *([] if skip is None else [skip]))
        _require_gpu(weight)
        x = _nhwc(x)
        w_ = _contig(weight.detach())
        skip_ = None if skip is None else _nhwc(skip)
        B, Cc, H, W = x.shape
        y = _empty_nhwc_bf16(B, Cc // 2, 2 * H, 2 * W, x.device)
        ws = _workspace(x.device, lib.dcpt_up_ps_bf16_ws_bytes(B, H, W, Cc, 0))
        check(lib.dcpt_up_ps_fwd_bf16(x.data_ptr(), w_.data_ptr(), _p(skip_), y.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cc,
                                      _stream(x.device)), "dcpt_up_ps_fwd_bf16")
        ctx.save_for_backward(x, w_)
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        _require_gpu_bf16(dy)
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc_bf16(B, Cc, H, W, dev)
        dw = torch.empty_like(w_)
        ws = _workspace(dev, lib.dcpt_up_ps_bf16_ws_bytes(B, H, W, Cc, 1))
        check(lib.dcpt_up_ps_bwd_bf16(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W,
                                      Cc, _stream(dev)), "dcpt_up_ps_bwd_bf16")
        return dx, dw, (dy if ctx.has_skip else None)


def down2x2(x, weight, bias):
    return _DownBf16Fn.apply(x, weight, bias) if x.dtype == torch.bfloat16 else _DownFn.apply(x, weight, bias)


@_remember_gemm_mode
class _DownSkipFn(torch.autograd.Function):
    """(down2x2(x), x) for an encoder group's output, which has TWO consumers: the down layer and the skip connection into the decoder
    (reference nafnet_arch.py:255-258, :264-265).  As two uses autograd sums the two gradients with a pass of its own over the feature map
    (four `add` launches per step, the level-0 one over the largest tensor of the network); here the skip's gradient -- the up layer's
    dy, untouched -- is the ``dx_add`` of the down layer's backward (dcpt_down2x2_bwd_acc*, ABI 13): summed in the scatter epilogue of
    its data-gradient GEMM.  fp32 or bf16 activations by x.dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        bf = x.dtype == torch.bfloat16
        (_require_gpu_bf16 if bf else _require_gpu)(x)
        _require_gpu(weight, bias)
        x = _nhwc(x)
        w_, b_ = _contig(weight.detach()), (None if bias is None else _contig(bias.detach()))
        B, Cc, H, W = x.shape
        if H % 2 or W % 2:
            raise ValueError(f"down2x2: H={H}, W={W} must be even")
        y = (_empty_nhwc_bf16 if bf else _empty_nhwc)(B, 2 * Cc, H // 2, W // 2, x.device)
        ws = _workspace(x.device, (lib.dcpt_down2x2_bf16_ws_bytes if bf else lib.dcpt_down2x2_ws_bytes)(B, H, W, Cc, 0))
        check((lib.dcpt_down2x2_fwd_bf16 if bf else lib.dcpt_down2x2_fwd)(x.data_ptr(), w_.data_ptr(), _p(b_), y.data_ptr(), ws.data_ptr(),
                                                                          ws.numel(), B, H, W, Cc, _stream(x.device)), "dcpt_down2x2_fwd")
        ctx.save_for_backward(x, w_)
        ctx.has_bias, ctx.bf = bias is not None, bf
        return y, x.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dskip):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        bf = ctx.bf
        B, Cc, H, W = x.shape
        dev = x.device
        if dy is None:   # (only the skip was used downstream)
            return dskip, None, None
        dy = _nhwc(dy)
        if dskip is not None:
            dskip = _nhwc(dskip if dskip.dtype == x.dtype else dskip.to(x.dtype))
        dx = (_empty_nhwc_bf16 if bf else _empty_nhwc)(B, Cc, H, W, dev)
        dw = torch.empty_like(w_)
        db = torch.empty((2 * Cc,), dtype=torch.float32, device=dev)
        ws = _workspace(dev, (lib.dcpt_down2x2_bf16_ws_bytes if bf else lib.dcpt_down2x2_ws_bytes)(B, H, W, Cc, 1))
        check((lib.dcpt_down2x2_bwd_acc_bf16 if bf else lib.dcpt_down2x2_bwd_acc)(
            dy.data_ptr(), x.data_ptr(), w_.data_ptr(), _p(dskip), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
            B, H, W, Cc, _stream(dev)), "dcpt_down2x2_bwd_acc")
        return dx, dw, (db if ctx.has_bias else None)


def down2x2_skip(x, weight, bias):
    """-> (down2x2(x), x): the second output is the skip connection's view of x; the two gradients are summed inside the down layer's
    data-gradient GEMM on the way back"""
    return _DownSkipFn.apply(x, weight, bias)


def up_ps(x, weight, skip=None):
    if x.dtype == torch.bfloat16:
        return _UpBf16Fn.apply(x, weight, skip)
    return _UpFn.apply(x, weight, skip)


# ------------------------------------------------------------------------------------------------
def fused_bias_act(x, bias, ref, act: int, grad: int, alpha: float, scale: float):
    """API parity with the reference's fused_act_ext.fused_bias_act
    (basicsr/ops/fused_act/src/fused_bias_act.cpp:14-26): NCHW input, bias over dim 1."""
    lib = _lib.load()
    _require_gpu(x)
    x = _contig(x)
    y = torch.empty_like(x)
    use_bias = bias is not None and bias.numel() > 0
    use_ref = ref is not None and ref.numel() > 0
    step_b = 1
    for d in x.shape[2:]:
        step_b *= int(d)
    b_ = _contig(bias) if use_bias else None
    r_ = _contig(ref) if use_ref else None
    check(lib.dcpt_fused_bias_act(x.data_ptr(), _p(b_), _p(r_), y.data_ptr(), x.numel(), x.shape[1] if x.dim() > 1 else 1,
                                  step_b, act, grad, float(alpha), float(scale), _stream(x.device)), "dcpt_fused_bias_act")
    return y


class _FusedLeakyReLUFn(torch.autograd.Function):
    """reference basicsr/ops/fused_act/fused_act.py:55-82 (first-order part)."""

    @staticmethod
    def forward(ctx, x, bias, negative_slope, scale):
        out = fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        gi = fused_bias_act(grad_output, None, out, 3, 1, ctx.negative_slope, ctx.scale)
        dims = [0] + list(range(2, gi.dim()))
        return gi, gi.sum(dims), None, None


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _FusedLeakyReLUFn.apply(x, bias, negative_slope, scale)


# ------------------------------------------------------------------------------------------------
# degradation-classifier head (reference basicsr/archs/degrad_classify_arch.py)
@_remember_gemm_mode
class _ConvLNFn(torch.autograd.Function):
    """conv(1x1 | 3x3, no bias) -> channels-first LayerNorm -> [+res] -> [ReLU]
    (Conv2d wrapper :69-103 with norm=LN; BottleneckBlock tail :227-243)."""

    @staticmethod
    def forward(ctx, x, weight, lnw, lnb, res, relu):
        lib = _lib.load()
        _require_gpu(x, weight, lnw, lnb, res)
        x = _nhwc(x)
        res_ = None if res is None else _nhwc(res)
        w_, lw, lb = _contig(weight.detach()), _contig(lnw.detach()), _contig(lnb.detach())
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        z = _empty_nhwc(B, Cout, H, W, dev)
        y = _empty_nhwc(B, Cout, H, W, dev)
        stats = torch.empty((2, B * H * W), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_conv_ln_ws_bytes(B, H, W, Cin, Cout, ks, 0))
        check(lib.dcpt_conv_ln_fwd(x.data_ptr(), w_.data_ptr(), lw.data_ptr(), lb.data_ptr(), _p(res_), int(bool(relu)),
                                   z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                                   ws.numel(), B, H, W, Cin, Cout, ks, _stream(dev)), "dcpt_conv_ln_fwd")
        ctx.save_for_backward(x, w_, lw, z, y, stats)
        ctx.relu, ctx.has_res = bool(relu), res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_, lw, z, y, stats = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        dx = _empty_nhwc(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        dlw, dlb = torch.empty_like(lw), torch.empty_like(lw)
        dres = _empty_nhwc(B, Cout, H, W, dev) if ctx.has_res else None
        ws = _workspace(dev, lib.dcpt_conv_ln_ws_bytes(B, H, W, Cin, Cout, ks, 1))
        check(lib.dcpt_conv_ln_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), lw.data_ptr(), z.data_ptr(), y.data_ptr(),
                                   stats[0].data_ptr(), stats[1].data_ptr(), dx.data_ptr(), dw.data_ptr(), dlw.data_ptr(),
                                   dlb.data_ptr(), _p(dres), ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, ks,
                                   int(ctx.relu), _stream(dev)), "dcpt_conv_ln_bwd")
        return dx, dw, dlw, dlb, dres, None


def conv_ln(x, weight, lnw, lnb, res=None, relu=True):
    return _ConvLNFn.apply(x, weight, lnw, lnb, res, relu)


class _ConvLNBf16Fn(torch.autograd.Function):
    """_ConvLNFn with bf16 activations (dcpt_conv_ln_fwd_bf16 / bwd_bf16): x, res, z, y and their gradients bf16; parameters fp32."""

    @staticmethod
    def forward(ctx, x, weight, lnw, lnb, res, relu, packed=None):
        lib = _lib.load()
        _require_gpu_bf16(x, *([] if res is None else [res]))
        _require_gpu(weight, lnw, lnb)
        x = _nhwc(x)
        res_ = None if res is None else _nhwc(res)
        w_, lw, lb = _contig(weight.detach()), _contig(lnw.detach()), _contig(lnb.detach())
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        z = _empty_nhwc_bf16(B, Cout, H, W, dev)
        y = _empty_nhwc_bf16(B, Cout, H, W, dev)
        stats = torch.empty((2, B * H * W), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_conv_ln_bf16_ws_bytes(B, H, W, Cin, Cout, ks, 0))
        pkp, pkn = _pk(packed, weight)
        check(lib.dcpt_conv_ln_fwd_bf16_packed(x.data_ptr(), w_.data_ptr(), pkp, pkn, lw.data_ptr(), lb.data_ptr(), _p(res_), int(bool(relu)),
                                               z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                                               ws.numel(), B, H, W, Cin, Cout, ks, _stream(dev)), "dcpt_conv_ln_fwd_bf16")
        ctx.save_for_backward(x, w_, lw, z, y, stats)
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.pk = (packed.buf if pkp is not None else None)   # (the backward of THIS forward reads the same images)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_, lw, z, y, stats = ctx.saved_tensors
        dy = _nhwc(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16))
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        dx = _empty_nhwc_bf16(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        dlw, dlb = torch.empty_like(lw), torch.empty_like(lw)
        dres = _empty_nhwc_bf16(B, Cout, H, W, dev) if ctx.has_res else None
        ws = _workspace(dev, lib.dcpt_conv_ln_bf16_ws_bytes(B, H, W, Cin, Cout, ks, 1))
        pk = ctx.pk
        check(lib.dcpt_conv_ln_bwd_acc_bf16_packed(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), _p(pk), 0 if pk is None else pk.numel(),
                                                   lw.data_ptr(), z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), None,
                                                   dx.data_ptr(), dw.data_ptr(), dlw.data_ptr(), dlb.data_ptr(), _p(dres), ws.data_ptr(),
                                                   ws.numel(), B, H, W, Cin, Cout, ks, int(ctx.relu), _stream(dev)), "dcpt_conv_ln_bwd_bf16")
        return dx, dw, dlw, dlb, dres, None, None


def conv_ln_bf16(x, weight, lnw, lnb, res=None, relu=True, packed: PackedConvBf16 = None):
    return _ConvLNBf16Fn.apply(x, weight, lnw, lnb, res, relu, packed)


@_remember_gemm_mode
class _BottleneckFn(torch.autograd.Function):
    """The whole BottleneckBlock (reference degrad_classify_arch.py:132-243 as the DCPT head instantiates it: identity shortcut) as ONE
    autograd node: relu(LN(conv1(x))) -> relu(LN(conv2(.))) -> relu(LN(conv3(.)) + x), the three conv -> LN groups through the same C
    entry points as conv_ln / conv_ln_bf16 (fp32 or bf16 activations by x.dtype).  As three nodes the block's input has two consumers
    (conv1 and the shortcut) and autograd sums their gradients with a pass of its own over the feature map -- 10 of the 18 bf16 `add`
    launches of a DCPT step; here conv3's shortcut gradient is conv1's ``dx_add`` (dcpt_conv_ln_bwd_acc*, ABI 13): summed in the epilogue
    of conv1's data-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, w1, lw1, lb1, w2, lw2, lb2, w3, lw3, lb3, packs=None):
        lib = _lib.load()
        bf = x.dtype == torch.bfloat16
        packs = packs if (bf and packs is not None) else (None, None, None)   # (PackedConvBf16 per conv: bf16 activations only)
        (_require_gpu_bf16 if bf else _require_gpu)(x)
        _require_gpu(w1, lw1, lb1, w2, lw2, lb2, w3, lw3, lb3)
        x = _nhwc(x)
        dev = x.device
        B, _, H, W = x.shape
        if bf:
            return _BottleneckFn._forward_bf16(ctx, lib, x, ((w1, lw1, lb1), (w2, lw2, lb2), (w3, lw3, lb3)), packs)
        empty = _empty_nhwc_bf16 if bf else _empty_nhwc
        ws_bytes = lib.dcpt_conv_ln_bf16_ws_bytes if bf else lib.dcpt_conv_ln_ws_bytes
        saved, cur, pkbufs = [x], x, []
        for (w, lw, lb, res, pck) in ((w1, lw1, lb1, None, packs[0]), (w2, lw2, lb2, None, packs[1]), (w3, lw3, lb3, x, packs[2])):
            w_, lw_, lb_ = _contig(w.detach()), _contig(lw.detach()), _contig(lb.detach())
            Cout, Cin, ks = w_.shape[0], w_.shape[1], w_.shape[2]
            z, y = empty(B, Cout, H, W, dev), empty(B, Cout, H, W, dev)
            stats = torch.empty((2, B * H * W), dtype=torch.float32, device=dev)
            ws = _workspace(dev, ws_bytes(B, H, W, Cin, Cout, ks, 0))
            if bf:
                pkp, pkn = _pk(pck, w)
                check(lib.dcpt_conv_ln_fwd_bf16_packed(cur.data_ptr(), w_.data_ptr(), pkp, pkn, lw_.data_ptr(), lb_.data_ptr(), _p(res), 1,
                                                       z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                                                       ws.numel(), B, H, W, Cin, Cout, ks, _stream(dev)), "dcpt_conv_ln_fwd_bf16")
                pkbufs.append(pck.buf if pkp is not None else None)
            else:
                check(lib.dcpt_conv_ln_fwd(cur.data_ptr(), w_.data_ptr(), lw_.data_ptr(), lb_.data_ptr(), _p(res), 1, z.data_ptr(),
                                           y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cin,
                                           Cout, ks, _stream(dev)), "dcpt_conv_ln_fwd")
            saved += [w_, lw_, z, y, stats]
            cur = y
        ctx.save_for_backward(*saved)
        ctx.bf = bf
        ctx.pkbufs = pkbufs   # (the backward of THIS forward reads the same operand images)
        return cur

    @staticmethod
    def _forward_bf16(ctx, lib, x, groups, packs):
        """bf16 activations: the whole block in ONE library call (dcpt_bottleneck_fwd_bf16, ABI 15: LayerNorms in the GEMM epilogues)"""
        dev = x.device
        B, C_, H, W = x.shape
        M = B * H * W
        garr = (_lib.BneckGroup * 3)()
        saved, pkbufs, keep = [x], [], []
        for k, ((w, lw, lb), pck) in enumerate(zip(groups, packs)):
            w_, lw_, lb_ = _contig(w.detach()), _contig(lw.detach()), _contig(lb.detach())
            Cout = w_.shape[0]
            z, y = _empty_nhwc_bf16(B, Cout, H, W, dev), _empty_nhwc_bf16(B, Cout, H, W, dev)
            stats = torch.empty((2, M), dtype=torch.float32, device=dev)
            pkp, pkn = _pk(pck, w)
            garr[k] = _lib.BneckGroup(w_.data_ptr(), pkp, pkn, lw_.data_ptr(), lb_.data_ptr(), z.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                      stats[1].data_ptr(), None, None, None)
            pkbufs.append(pck.buf if pkp is not None else None)
            saved += [w_, lw_, z, y, stats]
            keep.append(lb_)
        ctx.lnb = keep   # (the backward recomputes the inner ReLU masks from z: it needs the LayerNorm biases)
        ws = _workspace(dev, lib.dcpt_bottleneck_bf16_ws_bytes(B, H, W, C_, 0))
        check(lib.dcpt_bottleneck_fwd_bf16(x.data_ptr(), garr, ws.data_ptr(), ws.numel(), B, H, W, C_, _stream(dev)), "dcpt_bottleneck_fwd_bf16")
        ctx.save_for_backward(*saved)
        ctx.bf = True
        ctx.pkbufs = pkbufs   # (the backward of THIS forward reads the same operand images)
        return saved[-2]

    @staticmethod
    def _backward_bf16(ctx, lib, dy):
        sv = ctx.saved_tensors
        x = sv[0]
        dev = x.device
        B, C_, H, W = x.shape
        g = _nhwc(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16))
        garr = (_lib.BneckGroup * 3)()
        grads = []
        for k in range(3):
            w_, lw_, z, y, stats = sv[1 + 5 * k: 6 + 5 * k]
            dw, dlw, dlb = torch.empty_like(w_), torch.empty_like(lw_), torch.empty_like(lw_)
            pk = ctx.pkbufs[k]
            garr[k] = _lib.BneckGroup(w_.data_ptr(), _p(pk), 0 if pk is None else pk.numel(), lw_.data_ptr(), ctx.lnb[k].data_ptr(), z.data_ptr(), y.data_ptr(),
                                      stats[0].data_ptr(), stats[1].data_ptr(), dw.data_ptr(), dlw.data_ptr(), dlb.data_ptr())
            grads += [dw, dlw, dlb]
        dx = _empty_nhwc_bf16(B, C_, H, W, dev)
        ws = _workspace(dev, lib.dcpt_bottleneck_bf16_ws_bytes(B, H, W, C_, 1))
        check(lib.dcpt_bottleneck_bwd_bf16(g.data_ptr(), x.data_ptr(), garr, dx.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, C_, _stream(dev)),
              "dcpt_bottleneck_bwd_bf16")
        return (dx, *grads, None)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        if ctx.bf:
            return _BottleneckFn._backward_bf16(ctx, lib, dy)
        sv = ctx.saved_tensors
        x, bf = sv[0], ctx.bf
        dev = x.device
        B, _, H, W = x.shape
        empty = _empty_nhwc_bf16 if bf else _empty_nhwc
        ws_bytes = lib.dcpt_conv_ln_bf16_ws_bytes if bf else lib.dcpt_conv_ln_ws_bytes
        g = _nhwc(dy if (not bf or dy.dtype == torch.bfloat16) else dy.to(torch.bfloat16))
        grads, dshort = [None] * 9, None
        for k in (2, 1, 0):   # conv3 (its dres = the shortcut gradient), conv2, conv1 (dx = dx_add + ...)
            w_, lw_, z, y, stats = sv[1 + 5 * k: 6 + 5 * k]
            xin = x if k == 0 else sv[5 * k - 1]      # the y of group k - 1
            Cout, Cin, ks = w_.shape[0], w_.shape[1], w_.shape[2]
            dx = empty(B, Cin, H, W, dev)
            dw, dlw, dlb = torch.empty_like(w_), torch.empty_like(lw_), torch.empty_like(lw_)
            dres = empty(B, Cout, H, W, dev) if k == 2 else None
            ws = _workspace(dev, ws_bytes(B, H, W, Cin, Cout, ks, 1))
            if bf:
                pk = ctx.pkbufs[k]
                check(lib.dcpt_conv_ln_bwd_acc_bf16_packed(g.data_ptr(), xin.data_ptr(), w_.data_ptr(), _p(pk), 0 if pk is None else pk.numel(),
                                                           lw_.data_ptr(), z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                           _p(dshort if k == 0 else None), dx.data_ptr(), dw.data_ptr(), dlw.data_ptr(),
                                                           dlb.data_ptr(), _p(dres), ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, ks, 1,
                                                           _stream(dev)), "dcpt_conv_ln_bwd_acc_bf16")
            else:
                check(lib.dcpt_conv_ln_bwd_acc(g.data_ptr(), xin.data_ptr(), w_.data_ptr(), lw_.data_ptr(), z.data_ptr(), y.data_ptr(),
                                               stats[0].data_ptr(), stats[1].data_ptr(), _p(dshort if k == 0 else None), dx.data_ptr(),
                                               dw.data_ptr(), dlw.data_ptr(), dlb.data_ptr(), _p(dres), ws.data_ptr(), ws.numel(), B, H, W,
                                               Cin, Cout, ks, 1, _stream(dev)), "dcpt_conv_ln_bwd_acc")
            if k == 2:
                dshort = dres
            grads[3 * k: 3 * k + 3] = [dw, dlw, dlb]
            g = dx
        return (g, *grads, None)


def bottleneck(x, w1, lw1, lb1, w2, lw2, lb2, w3, lw3, lb3, packs=None):
    """relu(LN(conv3(relu(LN(conv2(relu(LN(conv1(x)))))))) + x) as one autograd node (fp32 or bf16 activations); ``packs``: the three convs'
    PackedConvBf16 caches (bf16 activations)"""
    return _BottleneckFn.apply(x, w1, lw1, lb1, w2, lw2, lb2, w3, lw3, lb3, packs)


class _ConvPoolReluBf16Fn(torch.autograd.Function):
    """_ConvPoolReluFn with bf16 activations."""

    @staticmethod
    def forward(ctx, x, weight, packed=None):
        lib = _lib.load()
        _require_gpu_bf16(x)
        _require_gpu(weight)
        x = _nhwc(x)
        w_ = _contig(weight.detach())
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        z = _empty_nhwc_bf16(B, Cout, H, W, dev)
        y = _empty_nhwc_bf16(B, Cout, H // 2, W // 2, dev)
        ws = _workspace(dev, lib.dcpt_conv1x1_pool_relu_bf16_ws_bytes(B, H, W, Cin, Cout, 0))
        pkp, pkn = _pk(packed, weight)
        check(lib.dcpt_conv1x1_pool_relu_fwd_bf16_packed(x.data_ptr(), w_.data_ptr(), pkp, pkn, z.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                                         ws.numel(), B, H, W, Cin, Cout, _stream(dev)), "dcpt_conv1x1_pool_relu_fwd_bf16")
        ctx.save_for_backward(x, w_, z)
        ctx.pk = (packed.buf if pkp is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_, z = ctx.saved_tensors
        dy = _nhwc(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16))
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dx = _empty_nhwc_bf16(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        ws = _workspace(dev, lib.dcpt_conv1x1_pool_relu_bf16_ws_bytes(B, H, W, Cin, Cout, 1))
        pk = ctx.pk
        check(lib.dcpt_conv1x1_pool_relu_bwd_bf16_packed(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), _p(pk), 0 if pk is None else pk.numel(),
                                                         z.data_ptr(), dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cin,
                                                         Cout, _stream(dev)), "dcpt_conv1x1_pool_relu_bwd_bf16")
        return dx, dw, None


def conv1x1_pool_relu_bf16(x, weight, packed: PackedConvBf16 = None):
    return _ConvPoolReluBf16Fn.apply(x, weight, packed)


class _PatchUnfoldFn(torch.autograd.Function):
    """NCHW image -> patch rows (B, Kp, Ho, Wo) in NHWC memory, last real column = 1 (carries the conv bias);
    the unfolding half of PromptIR_DC.conv_embed (degrad_classify_arch.py:491-494)."""

    @staticmethod
    def forward(ctx, x, ksize, stride, pad):
        lib = _lib.load()
        _require_gpu(x)
        x = _contig(x)
        B, Cin, H, W = x.shape
        Kp = (Cin * ksize * ksize + 1 + 3) // 4 * 4
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        A = _empty_nhwc(B, Kp, Ho, Wo, x.device)
        check(lib.dcpt_patch_unfold(x.data_ptr(), A.data_ptr(), B, Cin, H, W, ksize, stride, pad, Kp, _stream(x.device)),
              "dcpt_patch_unfold")
        ctx.geom = (B, Cin, H, W, ksize, stride, pad, Kp)
        return A

    @staticmethod
    def backward(ctx, dA):
        lib = _lib.load()
        B, Cin, H, W, ksize, stride, pad, Kp = ctx.geom
        dA = _nhwc(dA)
        dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dA.device)
        check(lib.dcpt_patch_fold(dA.data_ptr(), dx.data_ptr(), B, Cin, H, W, ksize, stride, pad, Kp, _stream(dA.device)),
              "dcpt_patch_fold")
        return dx, None, None, None


def conv_embed_ln(x, weight, bias, lnw, lnb, stride=2, pad=3):
    """Conv2d(Cin, dim, k, stride, pad) + bias -> channels-first LayerNorm (PromptIR_DC.conv_embed, :491-494):
    patch rows x [weight | bias | 0] on the MFMA GEMM, LayerNorm fused behind it."""
    dim, cin, ks, _ = weight.shape
    A = _PatchUnfoldFn.apply(x, ks, stride, pad)
    kp = A.shape[1]
    cols = [weight.reshape(dim, cin * ks * ks), bias.reshape(dim, 1)]
    if kp > cin * ks * ks + 1:
        cols.append(weight.new_zeros(dim, kp - cin * ks * ks - 1))
    wext = torch.cat(cols, dim=1).reshape(dim, kp, 1, 1)
    return conv_ln(A, wext, lnw, lnb, None, relu=False)


@_remember_gemm_mode
class _ConvPoolReluFn(torch.autograd.Function):
    """Conv2d(1x1, bias=False) -> MaxPool2d(2,2) -> ReLU (degrad_classify_arch.py:596-602)."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        _require_gpu(x, weight)
        x = _nhwc(x)
        w_ = _contig(weight.detach())
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        z = _empty_nhwc(B, Cout, H, W, dev)
        y = _empty_nhwc(B, Cout, H // 2, W // 2, dev)
        ws = _workspace(dev, lib.dcpt_conv1x1_pool_relu_ws_bytes(B, H, W, Cin, Cout, 0))
        check(lib.dcpt_conv1x1_pool_relu_fwd(x.data_ptr(), w_.data_ptr(), z.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(),
                                             B, H, W, Cin, Cout, _stream(dev)), "dcpt_conv1x1_pool_relu_fwd")
        ctx.save_for_backward(x, w_, z)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_, z = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cin, H, W = x.shape
        Cout = w_.shape[0]
        dev = x.device
        dx = _empty_nhwc(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        ws = _workspace(dev, lib.dcpt_conv1x1_pool_relu_ws_bytes(B, H, W, Cin, Cout, 1))
        check(lib.dcpt_conv1x1_pool_relu_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), z.data_ptr(), dx.data_ptr(),
                                             dw.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, _stream(dev)),
              "dcpt_conv1x1_pool_relu_bwd")
        return dx, dw


def conv1x1_pool_relu(x, weight):
    return _ConvPoolReluFn.apply(x, weight)


class _MixFn(torch.autograd.Function):
    """prev + softmax(mixing_weights)[idx] * feat (degrad_classify_arch.py:632-637)."""

    @staticmethod
    def forward(ctx, prev, feat, mixing_weights, idx):
        lib = _lib.load()
        _require_gpu_feat(prev, feat)
        _require_gpu(mixing_weights)
        feat = _nhwc(feat)
        prev_ = None if prev is None else _nhwc(prev)
        mw = _contig(mixing_weights.detach())
        # bf16 storage (the all-bf16 head: taps and the previous stage's output both bf16): the same arithmetic without cast passes
        bf = feat.dtype == torch.bfloat16
        if bf and prev_ is not None and prev_.dtype != torch.bfloat16:
            raise _lib.DcptHipError("mix: prev and feat must have the same storage dtype")
        out = (_empty_nhwc_bf16 if bf else _empty_nhwc)(*feat.shape, feat.device)
        fn = lib.dcpt_mix_fwd_bf16 if bf else lib.dcpt_mix_fwd
        check(fn(_p(prev_), feat.data_ptr(), mw.data_ptr(), mw.numel(), int(idx), out.data_ptr(), feat.numel(), _stream(feat.device)),
              "dcpt_mix_fwd")
        ctx.save_for_backward(feat, mw)
        ctx.idx, ctx.has_prev = int(idx), prev is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        feat, mw = ctx.saved_tensors
        bf = feat.dtype == torch.bfloat16
        dout = _nhwc(dout if dout.dtype == feat.dtype else dout.to(feat.dtype))
        dev = feat.device
        dfeat = (_empty_nhwc_bf16 if bf else _empty_nhwc)(*feat.shape, dev)
        dmix = torch.empty_like(mw)
        ws = _workspace(dev, lib.dcpt_mix_bwd_ws_bytes(feat.numel()))
        fn = lib.dcpt_mix_bwd_bf16 if bf else lib.dcpt_mix_bwd
        check(fn(dout.data_ptr(), feat.data_ptr(), mw.data_ptr(), mw.numel(), ctx.idx, dfeat.data_ptr(), dmix.data_ptr(), ws.data_ptr(),
                 ws.numel(), feat.numel(), _stream(dev)), "dcpt_mix_bwd")
        return (dout if ctx.has_prev else None), dfeat, dmix, None


def mix(prev, feat, mixing_weights, idx):
    return _MixFn.apply(prev, feat, mixing_weights, idx)


@_remember_gemm_mode
class _MeanPoolFCFn(torch.autograd.Function):
    """x.mean(dim=[-1,-2]) -> Linear (degrad_classify_arch.py:639-640)."""

    @staticmethod
    def forward(ctx, x, fw, fb):
        lib = _lib.load()
        _require_gpu_feat(x)
        _require_gpu(fw, fb)
        x = _nhwc(x)
        fw_, fb_ = _contig(fw.detach()), (None if fb is None else _contig(fb.detach()))
        B, Cc, H, W = x.shape
        NC = fw_.shape[0]
        dev = x.device
        pooled = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        logits = torch.empty((B, NC), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_meanpool_fc_ws_bytes(B, H * W, Cc))
        bf = x.dtype == torch.bfloat16   # bf16-storage feature map: pooled in fp32 straight from it, its gradient stored in bf16
        fn = lib.dcpt_meanpool_fc_fwd_bf16 if bf else lib.dcpt_meanpool_fc_fwd
        check(fn(x.data_ptr(), fw_.data_ptr(), _p(fb_), pooled.data_ptr(), logits.data_ptr(), ws.data_ptr(), ws.numel(), B, H * W, Cc, NC,
                 _stream(dev)), "dcpt_meanpool_fc_fwd")
        ctx.save_for_backward(pooled, fw_)
        ctx.shape, ctx.has_bias, ctx.bf = (B, Cc, H, W), fb is not None, bf
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        pooled, fw_ = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        NC = fw_.shape[0]
        dev = pooled.device
        dl = _contig(dlogits)
        dx = (_empty_nhwc_bf16 if ctx.bf else _empty_nhwc)(B, Cc, H, W, dev)
        dfw = torch.empty_like(fw_)
        dfb = torch.empty((NC,), dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.dcpt_meanpool_fc_ws_bytes(B, H * W, Cc))
        fn = lib.dcpt_meanpool_fc_bwd_bf16 if ctx.bf else lib.dcpt_meanpool_fc_bwd
        check(fn(dl.data_ptr(), pooled.data_ptr(), fw_.data_ptr(), dx.data_ptr(), dfw.data_ptr(), dfb.data_ptr(), ws.data_ptr(), ws.numel(),
                 B, H * W, Cc, NC, _stream(dev)), "dcpt_meanpool_fc_bwd")
        return dx, dfw, (dfb if ctx.has_bias else None)


def meanpool_fc(x, fw, fb):
    return _MeanPoolFCFn.apply(x, fw, fb)


# ------------------------------------------------------------------------------------------------
# Restormer pieces (reference basicsr/archs/restormer_arch.py)
from ._lib import GdfnParams, GdfnSaved, MdtaParams, MdtaSaved  # noqa: E402


@_remember_gemm_mode
class _ConvFn(torch.autograd.Function):
    """bias-free conv (1x1 or dense 3x3 / pad 1), NHWC -> NHWC."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        _require_gpu(x, weight)
        x = _nhwc(x)
        w_ = _contig(weight.detach())
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        y = _empty_nhwc(B, Cout, H, W, dev)
        ws = _workspace(dev, lib.dcpt_conv_ws_bytes(B, H, W, Cin, Cout, ks, 0))
        check(lib.dcpt_conv_fwd(x.data_ptr(), w_.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, Cin, Cout, ks,
                                _stream(dev)), "dcpt_conv_fwd")
        ctx.save_for_backward(x, w_)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_ = ctx.saved_tensors
        dy = _nhwc(dy)
        B, Cin, H, W = x.shape
        Cout, ks = w_.shape[0], w_.shape[2]
        dev = x.device
        dx = _empty_nhwc(B, Cin, H, W, dev)
        dw = torch.empty_like(w_)
        ws = _workspace(dev, lib.dcpt_conv_ws_bytes(B, H, W, Cin, Cout, ks, 1))
        check(lib.dcpt_conv_bwd(dy.data_ptr(), x.data_ptr(), w_.data_ptr(), dx.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                ws.numel(), B, H, W, Cin, Cout, ks, _stream(dev)), "dcpt_conv_bwd")
        return dx, dw


def conv_nobias(x, weight):
    return _ConvFn.apply(x, weight)


class _PixelUnshuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _require_gpu(x)
        x = _nhwc(x)
        B, Cc, H, W = x.shape
        y = _empty_nhwc(B, 4 * Cc, H // 2, W // 2, x.device)
        check(lib.dcpt_pixel_unshuffle(x.data_ptr(), y.data_ptr(), B, H, W, Cc, _stream(x.device)), "dcpt_pixel_unshuffle")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _nhwc(dy)
        B, C4, H, W = dy.shape
        dx = _empty_nhwc(B, C4 // 4, 2 * H, 2 * W, dy.device)
        check(lib.dcpt_pixel_shuffle(dy.data_ptr(), dx.data_ptr(), B, H, W, C4, _stream(dy.device)), "dcpt_pixel_shuffle")
        return dx


class _PixelShuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _require_gpu(x)
        x = _nhwc(x)
        B, C4, H, W = x.shape
        y = _empty_nhwc(B, C4 // 4, 2 * H, 2 * W, x.device)
        check(lib.dcpt_pixel_shuffle(x.data_ptr(), y.data_ptr(), B, H, W, C4, _stream(x.device)), "dcpt_pixel_shuffle")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _nhwc(dy)
        B, Cc, H, W = dy.shape
        dx = _empty_nhwc(B, 4 * Cc, H // 2, W // 2, dy.device)
        check(lib.dcpt_pixel_unshuffle(dy.data_ptr(), dx.data_ptr(), B, H, W, Cc, _stream(dy.device)), "dcpt_pixel_unshuffle")
        return dx


def pixel_unshuffle2(x):
    return _PixelUnshuffleFn.apply(x)


def pixel_shuffle2(x):
    return _PixelShuffleFn.apply(x)


class _ConcatFn(torch.autograd.Function):
    """torch.cat([a, b], 1) on NHWC maps."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        _require_gpu(a, b)
        a, b = _nhwc(a), _nhwc(b)
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        out = _empty_nhwc(B, Ca + Cb, H, W, a.device)
        check(lib.dcpt_concat_channels(a.data_ptr(), b.data_ptr(), out.data_ptr(), B * H * W, Ca, Cb, _stream(a.device)),
              "dcpt_concat_channels")
        ctx.dims = (Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        dout = _nhwc(dout)
        Ca, Cb = ctx.dims
        B, _, H, W = dout.shape
        da, db = _empty_nhwc(B, Ca, H, W, dout.device), _empty_nhwc(B, Cb, H, W, dout.device)
        check(lib.dcpt_split_channels(dout.data_ptr(), da.data_ptr(), db.data_ptr(), B * H * W, Ca, Cb, _stream(dout.device)),
              "dcpt_split_channels")
        return da, db


def concat_channels(a, b):
    return _ConcatFn.apply(a, b)


# What the Restormer halves keep for backward (every saved pointer of dcpt_mdta_saved / dcpt_gdfn_saved may be NULL on its own; the
# backward entry points recompute what is missing with the forward kernels, bit-identical results):
#   "full"      everything the backward pass reads: 19 [M][C] units per block (fastest; B = 64, 128 x 128: 118-125 GB);
#   "balanced"  (DEFAULT) LN(x) of both halves, attn @ v and the GDFN gate product gelu(x1) * x2 are NOT kept -- two bandwidth passes,
#               one small batched GEMM and one depthwise-gate pass per block in backward: 13.3 units (86 GB, +6 % time);
#   "lean"      additionally the qkv conv output (one more C x 3C GEMM in backward): 10.3 units (69 GB);
#   "auto"      opt-in: ONE choice per network forward from the memory the process holds on the device when the forward starts
#               ('full' below 40 % of the device memory, 'balanced' up to 70 %, 'lean' beyond).
# The mode is explicit: ``network_g.save_mode`` in the options (ctor kwarg of Restormer / Restormer_origin) or the process default
# (set_restormer_save / DCPT_RESTORMER_SAVE); the default does NOT look at free memory -- a DDP rank or a co-resident model gets the same
# schedule, step time and peak memory as a process that has the device to itself (round-4 verdict / advisor finding).
# Measured on MI355X, Restormer B = 64, 128 x 128 (profiles/r3/extra_restormer_*.json, profiles/r5/).
import contextlib as _contextlib  # noqa: E402

_RESTORMER_MODES = ("auto", "full", "balanced", "lean")
_RESTORMER_SAVE = _os.environ.get("DCPT_RESTORMER_SAVE", "balanced")
if _RESTORMER_SAVE not in _RESTORMER_MODES:
    raise ValueError(f"DCPT_RESTORMER_SAVE={_RESTORMER_SAVE!r}: expected one of {_RESTORMER_MODES}")
_RESTORMER_SCOPED = None          # the resolved mode of the network forward that is running (restormer_save), or None
_DEV_TOTAL: Dict[int, int] = {}


def set_restormer_save(mode: str) -> str:
    """Process default: 'balanced' (initially), 'full', 'lean' or 'auto'; returns the previous default."""
    global _RESTORMER_SAVE
    if mode not in _RESTORMER_MODES:
        raise ValueError(f"restormer save mode {mode!r}: expected one of {_RESTORMER_MODES}")
    prev, _RESTORMER_SAVE = _RESTORMER_SAVE, mode
    return prev


def _resolve_restormer_mode(mode, dev) -> str:
    if mode != "auto":
        return mode
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    total = _DEV_TOTAL.get(idx)
    if total is None:
        total = _DEV_TOTAL[idx] = torch.cuda.get_device_properties(idx).total_memory
    used = torch.cuda.memory_allocated(idx) / total
    return "full" if used < 0.40 else "balanced" if used < 0.70 else "lean"


@_contextlib.contextmanager
def restormer_save(mode, dev):
    """Scope of one network forward: ``mode`` (None = the process default) is resolved ONCE here -- 'auto' looks at the device memory now,
    not block by block -- and every MDTA / GDFN half inside the scope keeps the same set of tensors."""
    global _RESTORMER_SCOPED
    mode = _RESTORMER_SAVE if mode is None else mode
    if mode not in _RESTORMER_MODES:
        raise ValueError(f"restormer save mode {mode!r}: expected one of {_RESTORMER_MODES}")
    prev, _RESTORMER_SCOPED = _RESTORMER_SCOPED, _resolve_restormer_mode(mode, dev)
    try:
        yield _RESTORMER_SCOPED
    finally:
        _RESTORMER_SCOPED = prev


def _restormer_mode(dev) -> str:
    if _RESTORMER_SCOPED is not None:
        return _RESTORMER_SCOPED
    return _resolve_restormer_mode(_RESTORMER_SAVE, dev)   # a bare MDTA / GDFN call outside a network


@_remember_gemm_mode
class _MDTAFn(torch.autograd.Function):
    """x + project_out(attn(LN(x)))  (restormer_arch.py:103-145, :156-157)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, qkv_w, dw_w, proj_w, temperature, heads, biasfree):
        lib = _lib.load()
        _require_gpu(x, norm_w, norm_b, qkv_w, dw_w, proj_w, temperature)
        x = _nhwc(x)
        ps = [None if t is None else _contig(t.detach()) for t in (norm_w, norm_b, qkv_w, dw_w, proj_w, temperature)]
        B, Cc, H, W = x.shape
        dev = x.device
        M, ch = B * H * W, Cc // heads
        y = _empty_nhwc(B, Cc, H, W, dev)
        stats = torch.empty((2, M), dtype=torch.float32, device=dev)
        mode = _restormer_mode(dev)
        qkv1 = None if mode == "lean" else _empty_nhwc(B, 3 * Cc, H, W, dev)
        qkv = _empty_nhwc(B, 3 * Cc, H, W, dev)
        nrm = torch.empty((B, 2 * Cc), dtype=torch.float32, device=dev)
        att = torch.empty((3, B, heads, ch, ch), dtype=torch.float32, device=dev)
        out_att = None if mode != "full" else _empty_nhwc(B, Cc, H, W, dev)
        xn = None if mode != "full" else _empty_nhwc(B, Cc, H, W, dev)
        sv = MdtaSaved(stats[0].data_ptr(), stats[1].data_ptr(), _p(qkv1), qkv.data_ptr(), nrm.data_ptr(),
                       att[0].data_ptr(), att[1].data_ptr(), att[2].data_ptr(), _p(out_att), _p(xn))
        pp = MdtaParams(*[_p(t) for t in ps])
        ws = _workspace(dev, lib.dcpt_mdta_ws_bytes(B, H, W, Cc, heads, 0))
        check(lib.dcpt_mdta_fwd(C.byref(pp), x.data_ptr(), y.data_ptr(), C.byref(sv), ws.data_ptr(), ws.numel(), B, H, W, Cc,
                                heads, int(biasfree), _stream(dev)), "dcpt_mdta_fwd")
        kept = [t for t in (qkv1, out_att, xn) if t is not None]
        ctx.kept = (qkv1 is not None, out_att is not None, xn is not None)
        ctx.save_for_backward(x, stats, qkv, nrm, att, *kept, *[t for t in ps if t is not None])
        ctx.has_bias = ps[1] is not None
        ctx.heads, ctx.biasfree = heads, int(biasfree)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, qkv, nrm, att, *ps = ctx.saved_tensors
        ps = list(ps)
        qkv1, out_att, xn = (ps.pop(0) if k else None for k in ctx.kept)
        if ctx.has_bias:
            norm_w, norm_b, qkv_w, dw_w, proj_w, temp = ps
        else:
            norm_w, qkv_w, dw_w, proj_w, temp = ps
            norm_b = None
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc(B, Cc, H, W, dev)
        plist = [norm_w, norm_b, qkv_w, dw_w, proj_w, temp]
        grads = [None if t is None else torch.empty_like(t) for t in plist]
        sv = MdtaSaved(stats[0].data_ptr(), stats[1].data_ptr(), _p(qkv1), qkv.data_ptr(), nrm.data_ptr(),
                       att[0].data_ptr(), att[1].data_ptr(), att[2].data_ptr(), _p(out_att), _p(xn))
        pp = MdtaParams(*[_p(t) for t in plist])
        gg = MdtaParams(*[_p(t) for t in grads])
        ws = _workspace(dev, lib.dcpt_mdta_ws_bytes(B, H, W, Cc, ctx.heads, 1))
        check(lib.dcpt_mdta_bwd(C.byref(pp), C.byref(gg), x.data_ptr(), C.byref(sv), dy.data_ptr(), dx.data_ptr(), ws.data_ptr(),
                                ws.numel(), B, H, W, Cc, ctx.heads, int(ctx.biasfree), _stream(dev)), "dcpt_mdta_bwd")
        return (dx, *grads, None, None)


LN_BIASFREE, LN_EPS_1E5, ATTN_SOFTMAX = 1, 2, 4   # include/dcpt_hip.h: DCPT_LN_BIASFREE, DCPT_LN_EPS_1E5, DCPT_ATTN_SOFTMAX


def mdta(x, norm_w, norm_b, qkv_w, dw_w, proj_w, temperature, heads, biasfree, eps_1e5=False, softmax=False):
    """``eps_1e5`` / ``softmax``: the PromptIR variants (LayerNorm eps 1e-5, softmax instead of ReLU attention)."""
    flags = (LN_BIASFREE if biasfree else 0) | (LN_EPS_1E5 if eps_1e5 else 0) | (ATTN_SOFTMAX if softmax else 0)
    return _MDTAFn.apply(x, norm_w, norm_b, qkv_w, dw_w, proj_w, temperature, heads, flags)


@_remember_gemm_mode
class _GDFNFn(torch.autograd.Function):
    """x + project_out(gelu(x1) * x2)  (restormer_arch.py:75-100, :158)."""

    @staticmethod
    def forward(ctx, x, norm_w, norm_b, in_w, dw_w, out_w, biasfree):
        lib = _lib.load()
        _require_gpu(x, norm_w, norm_b, in_w, dw_w, out_w)
        x = _nhwc(x)
        ps = [None if t is None else _contig(t.detach()) for t in (norm_w, norm_b, in_w, dw_w, out_w)]
        B, Cc, H, W = x.shape
        dev = x.device
        M = B * H * W
        hidden = ps[4].shape[1]
        hp = (hidden + 3) // 4 * 4
        y = _empty_nhwc(B, Cc, H, W, dev)
        stats = torch.empty((2, M), dtype=torch.float32, device=dev)
        full = _restormer_mode(dev) == "full"
        u = _empty_nhwc(B, 2 * hp, H, W, dev)
        t = _empty_nhwc(B, hp, H, W, dev) if full else None
        xn = _empty_nhwc(B, Cc, H, W, dev) if full else None
        sv = GdfnSaved(stats[0].data_ptr(), stats[1].data_ptr(), u.data_ptr(), _p(t), _p(xn))
        pp = GdfnParams(*[_p(q) for q in ps])
        ws = _workspace(dev, lib.dcpt_gdfn_ws_bytes(B, H, W, Cc, hidden, 0))
        check(lib.dcpt_gdfn_fwd(C.byref(pp), x.data_ptr(), y.data_ptr(), C.byref(sv), ws.data_ptr(), ws.numel(), B, H, W, Cc,
                                hidden, int(biasfree), _stream(dev)), "dcpt_gdfn_fwd")
        ctx.full = full
        ctx.save_for_backward(x, stats, u, *([t, xn] if full else []), *[q for q in ps if q is not None])
        ctx.has_bias, ctx.biasfree, ctx.hidden = ps[1] is not None, int(biasfree), hidden
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, stats, u, *ps = ctx.saved_tensors
        t = xn = None
        if ctx.full:
            t, xn, *ps = ps
        if ctx.has_bias:
            norm_w, norm_b, in_w, dw_w, out_w = ps
        else:
            norm_w, in_w, dw_w, out_w = ps
            norm_b = None
        dy = _nhwc(dy)
        B, Cc, H, W = x.shape
        dev = x.device
        dx = _empty_nhwc(B, Cc, H, W, dev)
        plist = [norm_w, norm_b, in_w, dw_w, out_w]
        grads = [None if q is None else torch.empty_like(q) for q in plist]
        sv = GdfnSaved(stats[0].data_ptr(), stats[1].data_ptr(), u.data_ptr(), _p(t), _p(xn))
        pp = GdfnParams(*[_p(q) for q in plist])
        gg = GdfnParams(*[_p(q) for q in grads])
        ws = _workspace(dev, lib.dcpt_gdfn_ws_bytes(B, H, W, Cc, ctx.hidden, 1))
        check(lib.dcpt_gdfn_bwd(C.byref(pp), C.byref(gg), x.data_ptr(), C.byref(sv), dy.data_ptr(), dx.data_ptr(), ws.data_ptr(),
                                ws.numel(), B, H, W, Cc, ctx.hidden, int(ctx.biasfree), _stream(dev)), "dcpt_gdfn_bwd")
        return (dx, *grads, None)


def gdfn(x, norm_w, norm_b, in_w, dw_w, out_w, biasfree, eps_1e5=False):
    flags = (LN_BIASFREE if biasfree else 0) | (LN_EPS_1E5 if eps_1e5 else 0)
    return _GDFNFn.apply(x, norm_w, norm_b, in_w, dw_w, out_w, flags)


# ------------------------------------------------------------------------------------------------
class _PromptMixFn(torch.autograd.Function):
    """bilinear_(H,W)( sum_l softmax(logits)[b, l] * prompt_param[l] ) as an NHWC map
    (PromptGenBlock.forward, basicsr/archs/promptir_arch.py:253-259)."""

    @staticmethod
    def forward(ctx, logits, prompt_param, H, W):
        lib = _lib.load()
        _require_gpu(logits, prompt_param)
        lg, pp = _contig(logits.detach()), _contig(prompt_param.detach())
        B, L = lg.shape
        _, L2, D, S, S2 = pp.shape
        if L2 != L or S2 != S:
            raise ValueError(f"prompt_param {tuple(pp.shape)} does not match logits {tuple(lg.shape)}")
        dev = lg.device
        wsm = torch.empty((B, L), dtype=torch.float32, device=dev)
        out = _empty_nhwc(B, D, H, W, dev)
        check(lib.dcpt_prompt_mix_fwd(lg.data_ptr(), pp.data_ptr(), wsm.data_ptr(), out.data_ptr(), B, L, D, S, H, W, _stream(dev)),
              "dcpt_prompt_mix_fwd")
        ctx.save_for_backward(pp, wsm)
        ctx.geom = (B, L, D, S, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        pp, wsm = ctx.saved_tensors
        B, L, D, S, H, W = ctx.geom
        dout = _nhwc(dout)
        dev = dout.device
        dlogits = torch.empty((B, L), dtype=torch.float32, device=dev)
        dparam = torch.empty_like(pp)
        ws = _workspace(dev, lib.dcpt_prompt_mix_bwd_ws_bytes(B, D, S))
        check(lib.dcpt_prompt_mix_bwd(dout.data_ptr(), pp.data_ptr(), wsm.data_ptr(), dlogits.data_ptr(), dparam.data_ptr(),
                                      ws.data_ptr(), ws.numel(), B, L, D, S, H, W, _stream(dev)), "dcpt_prompt_mix_bwd")
        return dlogits, dparam, None, None


def prompt_mix(logits, prompt_param, H, W):
    return _PromptMixFn.apply(logits, prompt_param, H, W)
