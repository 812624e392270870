// Optional per-launch timing with HIP events on the launch stream (off by default; used by bench.py
// to measure the dominant kernels' duration live, inside the timed region).
#pragma once
#include "dcpt_common.h"

void prof_begin(hipStream_t s, int cls, int64_t M, int N, int K, double flops, double bytes);
void prof_end(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    ProfScope(hipStream_t st, int cls, int64_t M, int N, int K, double flops, double bytes) : s(st) {
        prof_begin(s, cls, M, N, K, flops, bytes);
    }
    ~ProfScope() { prof_end(s); }
};

// class ids
enum { PROF_NT = 0, PROF_TN = 512, PROF_OTHER = 1024 };

// Launch trace for the tests (off by default): every dispatch decision of the library names the kernel family it launched with a static
// string; dcpt_trace_read returns "tag count" lines.  A parity test can then assert WHICH kernel produced the result it checked.
extern bool g_trace_on;
void trace_tag_slow(const char* tag);
inline void trace_tag(const char* tag) {
    if (g_trace_on) trace_tag_slow(tag);
}
