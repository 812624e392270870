// Fused operand loaders shared by the NT and TN MFMA GEMMs.
// An operand is a logical [M][ncols] fp32 matrix derived on the fly from NHWC activations.
#pragma once
#include "gemm.h"

struct Operand {
    const float* ptr;
    int64_t M;
    int ncols;  // logical columns (multiple of 4)
    int ld;     // physical row stride (elements); unused for gather
    const float* mu;
    const float* rstd;
    const float* lnw;
    const float* lnb;
    const float* simg;
    int P;
    int gH, gW, gC;
};

struct RowCtx {
    int64_t off;
    float mu, rstd;
    int img;
    bool valid;
};

template <int KIND>
__device__ __forceinline__ void make_row(const Operand& o, int64_t m, RowCtx& rc) {
    rc.valid = m < o.M;
    const int64_t mm = rc.valid ? m : 0;
    rc.mu = 0.f;
    rc.rstd = 0.f;
    rc.img = 0;
    if constexpr (KIND == A_GATHER) {
        const int w = (int)(mm % o.gW);
        const int64_t t = mm / o.gW;
        const int h = (int)(t % o.gH);
        const int64_t b = t / o.gH;
        rc.off = ((b * (2 * o.gH) + 2 * h) * (int64_t)(2 * o.gW) + 2 * w) * o.gC;
    } else {
        rc.off = mm * (int64_t)o.ld;
    }
    if constexpr (KIND == A_LN) {
        rc.mu = o.mu[mm];
        rc.rstd = o.rstd[mm];
    }
    if constexpr (KIND == A_SCALE) rc.img = (int)(mm / o.P);
}

// loads logical columns [c, c+4) of the row; zero outside the matrix
template <int KIND>
__device__ __forceinline__ float4 load_op(const Operand& o, const RowCtx& rc, int c) {
    if (!rc.valid || c >= o.ncols) return f4_zero();
    if constexpr (KIND == A_PLAIN) {
        return ldg4(o.ptr + rc.off + c);
    } else if constexpr (KIND == A_LN) {
        const float4 x = ldg4(o.ptr + rc.off + c);
        const float4 w = ldg4(o.lnw + c);
        const float4 b = ldg4(o.lnb + c);
        float4 r;
        r.x = fmaf((x.x - rc.mu) * rc.rstd, w.x, b.x);
        r.y = fmaf((x.y - rc.mu) * rc.rstd, w.y, b.y);
        r.z = fmaf((x.z - rc.mu) * rc.rstd, w.z, b.z);
        r.w = fmaf((x.w - rc.mu) * rc.rstd, w.w, b.w);
        return r;
    } else if constexpr (KIND == A_SCALE) {
        const float4 x = ldg4(o.ptr + rc.off + c);
        const float4 s = ldg4(o.simg + (int64_t)rc.img * o.ncols + c);
        return f4_mul(x, s);
    } else if constexpr (KIND == A_SG) {
        const float4 x1 = ldg4(o.ptr + rc.off + c);
        const float4 x2 = ldg4(o.ptr + rc.off + o.ncols + c);
        return f4_mul(x1, x2);
    } else {  // A_GATHER
        const int ij = c / o.gC;
        const int ch = c - ij * o.gC;
        const int64_t a = rc.off + ((int64_t)(ij >> 1) * (2 * o.gW) + (ij & 1)) * o.gC + ch;
        return ldg4(o.ptr + a);
    }
}
