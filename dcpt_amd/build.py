"""Build libdcpt_hip.so (gfx950 only) in-tree with hipcc.  No torch, no cmake: the library has a
plain C ABI (include/dcpt_hip.h) and links only against the HIP runtime.

    python -m dcpt_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libdcpt_hip.so")
ARCH = "gfx950"
SOURCES = ["gemm_nt.hip", "gemm_x3.hip", "gemm_tn.hip", "ln.hip", "dwconv.hip", "misc.hip", "conv3x3.hip", "nafblock.hip", "capi.hip", "prof.hip", "side.hip", "dchead.hip", "restormer.hip", "promptir.hip", "comm.hip", "gemm_bf16.hip", "gemm_bf16_256.hip", "gemm_tn_bf16_256.hip", "bf16_ops.hip", "nafblock_bf16.hip", "dwring.hip", "dchead_bf16.hip", "edge_bf16.hip", "ffn_bf16.hip", "ffn_f32.hip", "chain_bf16.hip", "optim.hip"]
# No packed-fp32 VALU instructions (v_pk_mul / fma / add_f32) in device code.  Their operand-select forms (op_sel / op_sel_hi: a result half
# taking the other half of a source pair) returned wrong values in lanes 48-63 -- a few elements per launch -- whenever bf16 MFMA GEMMs of
# ANOTHER stream shared the SIMD: found in the ending conv under two-stream tiled inference, pinned down by replacing exactly those
# instructions in its assembly (LABNOTES.md 4h; exact alone, exact with every wait state padded, exact once the 24 op_sel forms are scalar).
# The compiler offers no switch for the operand-select forms alone and they occur in most kernels of the library, so packed fp32 is off
# everywhere: +0.35 % on the fp32 step, +1.3 % on the bf16 step.  (The host pass does not know the feature and says so on stderr.)
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", *NO_PACKED_FP32]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libdcpt_hip.so for gfx950)")


def _digest() -> str:
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + ["../../include/dcpt_hip.h"]
    for n in names:
        p = os.path.join(CSRC, n)
        if os.path.isfile(p):
            h.update(n.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile + link if the sources changed.  Safe to call from several processes at once (one rank per GPU): an exclusive
    file lock serialises the builders, and whoever comes second finds the digest up to date."""
    import fcntl

    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    stamp = os.path.join(LIBDIR, "libdcpt_hip.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    tmp = LIB + ".tmp"   # link next to the target and rename: a reader never maps a half-written library
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
