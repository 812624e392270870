"""Host side of ``dcpt_allreduce_flat`` (include/dcpt_hip.h): the data-parallel gradient all-reduce over RCCL / xGMI as a C-ABI
call on a caller-provided communicator (reference basicsr/models/base_model.py:108-115 does the same collective through torch
DDP; ``basicsr/`` here keeps that mechanism for the training step -- this module is for hosts that drive the collective
themselves, and for the tests).

``RcclComm`` is plumbing, not product arithmetic: it creates the ``ncclComm_t`` the C ABI expects with RCCL's own bootstrap
calls (``ncclGetUniqueId`` on rank 0, the 128-byte id handed to the other ranks by any side channel, ``ncclCommInitRank``
everywhere), using the RCCL copy torch already ships."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

_rccl = None


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]   # rccl.h: NCCL_UNIQUE_ID_BYTES


def _load_rccl():
    global _rccl
    if _rccl is None:
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _rccl = C.CDLL(bundled if os.path.exists(bundled) else "librccl.so.1", mode=C.RTLD_GLOBAL)
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        _rccl.ncclGetErrorString.restype = C.c_char_p
        _rccl.ncclGetErrorString.argtypes = [C.c_int]
    return _rccl


def _ok(rc, what):
    if rc != 0:
        raise _lib.DcptHipError(f"{what}: {_load_rccl().ncclGetErrorString(rc).decode()}")


def unique_id() -> bytes:
    """rank 0: the 128-byte rendezvous id to hand to every rank"""
    uid = _UniqueId()
    _ok(_load_rccl().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(uid)


class RcclComm:
    """an ``ncclComm_t`` for (rank, world) on the CURRENT device"""

    def __init__(self, world: int, rank: int, uid: bytes):
        self.world, self.rank = world, rank
        self.handle = C.c_void_p()
        _ok(_load_rccl().ncclCommInitRank(C.byref(self.handle), world, _UniqueId.from_buffer_copy(uid), rank), "ncclCommInitRank")

    def close(self):
        if self.handle:
            _load_rccl().ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()


def allreduce_flat_(buf: torch.Tensor, comm: RcclComm, mean: bool = True) -> torch.Tensor:
    """in place: buf <- sum over ranks (/ world if ``mean``), enqueued on the current stream of buf's device"""
    if not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()):
        raise _lib.DcptHipError("allreduce_flat_: a contiguous fp32 device tensor is required (no CPU path)")
    lib = _lib.load()
    _lib.check(lib.dcpt_allreduce_flat(buf.data_ptr(), buf.numel(), comm.handle, 1.0 / comm.world if mean else 1.0,
                                       torch.cuda.current_stream(buf.device).cuda_stream), "dcpt_allreduce_flat")
    return buf
