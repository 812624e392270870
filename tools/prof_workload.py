import os
"""Per-GEMM-class timing (library HIP-event profiler, side stream off) of one bench_extra workload step."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
import bench
from dcpt_amd import _lib
lib = _lib.load(); lib.dcpt_set_side_stream(0)
sys.argv = ["bench_extra.py", "--workload", sys.argv[1], "--steps", "1", "--warmup", "1"]
import bench_extra
import io, contextlib
orig_timed = bench_extra.timed
def timed(fn, steps, warmup):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    lib.dcpt_prof_enable(1); fn(); torch.cuda.synchronize()
    buf = (ctypes.c_double * (8 * 512))(); n = lib.dcpt_prof_read(buf, 512); lib.dcpt_prof_enable(0)
    rows = []
    for i in range(n):
        cls, M, N, K, cnt, ms, fl, by = (buf[i * 8 + j] for j in range(8))
        rows.append((ms, bench.prof_class_name(int(cls)), int(M), int(N), int(K), int(cnt), fl / ms / 1e9, by / ms / 1e6))
    rows.sort(reverse=True)
    print("total GEMM ms %.2f" % sum(r[0] for r in rows))
    for r in rows[:24]:
        print("%8.3f ms  %-34s M=%-8d N=%-5d K=%-5d x%-3d %7.1f TF/s %7.1f GB/s(alg)" % r)
    return orig_timed(fn, 2, 0)
bench_extra.timed = timed
bench_extra.main()
