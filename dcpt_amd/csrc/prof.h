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
