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
    int h, w;  // A_CONV3: pixel coordinates of the row
    bool valid;
};

template <int KIND>
__device__ __forceinline__ void make_row(const Operand& o, int64_t m, RowCtx& rc) {
    rc.valid = m < o.M;
    const int64_t mm = rc.valid ? m : 0;
    rc.mu = 0.f;
    rc.rstd = 0.f;
    rc.img = 0;
    rc.h = 0;
    rc.w = 0;
    if constexpr (KIND == A_GATHER) {
        const int w = (int)(mm % o.gW);
        const int64_t t = mm / o.gW;
        const int h = (int)(t % o.gH);
        const int64_t b = t / o.gH;
        rc.off = ((b * (2 * o.gH) + 2 * h) * (int64_t)(2 * o.gW) + 2 * w) * o.gC;
    } else if constexpr (KIND == A_CONV3) {
        rc.w = (int)(mm % o.gW);
        rc.h = (int)((mm / o.gW) % o.gH);
        rc.off = mm * (int64_t)o.gC;
    } else {
        rc.off = mm * (int64_t)o.ld;
    }
    if constexpr (KIND == A_LN) {
        rc.mu = o.mu[mm];
        rc.rstd = o.rstd[mm];
    }
    if constexpr (KIND == A_LNBF) rc.rstd = o.rstd[mm];
    if constexpr (KIND == A_SCALE) rc.img = (int)(mm / o.P);
}

// Operand access is split in two so that the global loads of the NEXT tile can stay in flight
// across the current tile's MFMAs: load_raw() only issues loads (no dependent math), finish() applies
// the fused transform right before the value is written to LDS.
struct RawVec {
    float4 a, b, c;
};

template <int KIND>
__device__ __forceinline__ void load_raw(const Operand& o, const RowCtx& rc, int c, RawVec& r) {
    const bool ok = rc.valid && c < o.ncols;
    if constexpr (KIND == A_PLAIN) {
        r.a = ok ? ldg4(o.ptr + rc.off + c) : f4_zero();
    } else if constexpr (KIND == A_LN) {
        r.a = ok ? ldg4(o.ptr + rc.off + c) : f4_zero();
        r.b = ok ? ldg4(o.lnw + c) : f4_zero();
        r.c = ok ? ldg4(o.lnb + c) : f4_zero();
    } else if constexpr (KIND == A_LNBF) {
        r.a = ok ? ldg4(o.ptr + rc.off + c) : f4_zero();
        r.b = ok ? ldg4(o.lnw + c) : f4_zero();
    } else if constexpr (KIND == A_SCALE) {
        r.a = ok ? ldg4(o.ptr + rc.off + c) : f4_zero();
        r.b = ok ? ldg4(o.simg + (int64_t)rc.img * o.ncols + c) : f4_zero();
    } else if constexpr (KIND == A_SG) {
        r.a = ok ? ldg4(o.ptr + rc.off + c) : f4_zero();
        r.b = ok ? ldg4(o.ptr + rc.off + o.ncols + c) : f4_zero();
    } else if constexpr (KIND == A_CONV3) {
        r.a = f4_zero();
        if (ok) {
            const int tap = c / o.gC;
            const int ch = c - tap * o.gC;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int hh = rc.h + ky - 1, ww = rc.w + kx - 1;
            if (hh >= 0 && hh < o.gH && ww >= 0 && ww < o.gW)
                r.a = ldg4(o.ptr + rc.off + ((int64_t)(ky - 1) * o.gW + (kx - 1)) * o.gC + ch);
        }
    } else {  // A_GATHER
        if (ok) {
            const int ij = c / o.gC;
            const int ch = c - ij * o.gC;
            const int64_t a = rc.off + ((int64_t)(ij >> 1) * (2 * o.gW) + (ij & 1)) * o.gC + ch;
            r.a = ldg4(o.ptr + a);
        } else {
            r.a = f4_zero();
        }
    }
}

template <int KIND>
__device__ __forceinline__ float4 finish(const RowCtx& rc, const RawVec& r) {
    if constexpr (KIND == A_LN) {
        // zero-padded lanes have b = c = 0, so they stay exactly 0
        float4 o;
        o.x = fmaf((r.a.x - rc.mu) * rc.rstd, r.b.x, r.c.x);
        o.y = fmaf((r.a.y - rc.mu) * rc.rstd, r.b.y, r.c.y);
        o.z = fmaf((r.a.z - rc.mu) * rc.rstd, r.b.z, r.c.z);
        o.w = fmaf((r.a.w - rc.mu) * rc.rstd, r.b.w, r.c.w);
        return o;
    } else if constexpr (KIND == A_LNBF) {
        return make_float4(r.a.x * rc.rstd * r.b.x, r.a.y * rc.rstd * r.b.y, r.a.z * rc.rstd * r.b.z, r.a.w * rc.rstd * r.b.w);
    } else if constexpr (KIND == A_SCALE || KIND == A_SG) {
        return f4_mul(r.a, r.b);
    } else {
        return r.a;
    }
}
