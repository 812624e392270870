// Fused operand loaders shared by the NT and TN MFMA GEMMs.
// An operand is a logical [M][ncols] fp32 matrix derived on the fly from NHWC activations.
//
// All global accesses of the GEMMs go through buffer resources (buffer_load/store_dwordx4 ... offen):
// a block opens a WINDOW at the first element its tile can touch (64-bit, wave-uniform base in SGPRs)
// and addresses everything with 32-bit byte offsets relative to it.  Rows past M, columns past ncols
// and zero-padding taps get an offset above the window size (ROW_SENT / COL_SENT), which the hardware
// range check turns into "load returns 0 / store is dropped" -- no exec-mask branches around the
// memory instructions, so the compiler is free to interleave them with the MFMAs.
#pragma once
#include "gemm.h"
#include "bufops.h"

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
    // window, set by open_window()
    rsrc_t rs;
    i32x4 rsd;     // the same window for LDS-DMA
    int64_t base;  // element index of the window start
    rsrc_t rs_s;   // A_SCALE: the per-image scale table
};

struct RowCtx {
    uint32_t off;   // byte offset of the row's first element inside the window, or ROW_SENT
    uint32_t soff;  // A_SCALE: byte offset of the row's image inside simg
    float mu, rstd;
    int h, w;  // A_CONV3: pixel coordinates of the row
};

template <int KIND>
__device__ __forceinline__ int64_t row_elem(const Operand& o, int64_t m) {
    if constexpr (KIND == A_GATHER) {
        const int w = (int)(m % o.gW);
        const int64_t t = m / o.gW;
        const int h = (int)(t % o.gH);
        const int64_t b = t / o.gH;
        return ((b * (2 * o.gH) + 2 * h) * (int64_t)(2 * o.gW) + 2 * w) * o.gC;
    } else if constexpr (KIND == A_CONV3) {
        return m * (int64_t)o.gC;
    } else {
        return m * (int64_t)o.ld;
    }
}

// m_first: first row this block will touch (wave-uniform)
template <int KIND>
__device__ __forceinline__ void open_window(Operand& o, int64_t m_first) {
    int64_t base = row_elem<KIND>(o, m_first);
    if constexpr (KIND == A_CONV3) {  // taps reach one image row + one pixel back
        base -= (int64_t)(o.gW + 1) * o.gC;
        if (base < 0) base = 0;
    }
    o.base = base;
    o.rs = make_rsrc(o.ptr + base);
    o.rsd = make_rsrc_dma(o.ptr + base);
    if constexpr (KIND == A_SCALE) o.rs_s = make_rsrc(o.simg);
}

template <int KIND>
__device__ __forceinline__ void make_row(const Operand& o, int64_t m, RowCtx& rc) {
    const bool valid = m < o.M;
    const int64_t mm = valid ? m : 0;
    rc.mu = 0.f;
    rc.rstd = 0.f;
    rc.soff = 0;
    rc.h = 0;
    rc.w = 0;
    rc.off = valid ? (uint32_t)((row_elem<KIND>(o, mm) - o.base) * 4) : ROW_SENT;
    if constexpr (KIND == A_CONV3) {
        rc.w = (int)(mm % o.gW);
        rc.h = (int)((mm / o.gW) % o.gH);
    }
    if constexpr (KIND == A_LN) {
        rc.mu = o.mu[mm];
        rc.rstd = o.rstd[mm];
    }
    if constexpr (KIND == A_LNBF) rc.rstd = o.rstd[mm];
    if constexpr (KIND == A_SCALE) rc.soff = (uint32_t)(mm / o.P) * (uint32_t)o.ncols * 4u;
}

// byte offset (inside the window) of the 4 floats at logical column c of the row, or a sentinel
template <int KIND>
__device__ __forceinline__ uint32_t elem_voff(const Operand& o, const RowCtx& rc, int c) {
    bool ok = c < o.ncols;
    uint32_t v;
    if constexpr (KIND == A_CONV3) {
        const int tap = c / o.gC;
        const int ch = c - tap * o.gC;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int hh = rc.h + ky - 1, ww = rc.w + kx - 1;
        ok = ok && hh >= 0 && hh < o.gH && ww >= 0 && ww < o.gW;
        v = rc.off + (uint32_t)((((ky - 1) * o.gW + (kx - 1)) * o.gC + ch) * 4);
    } else if constexpr (KIND == A_GATHER) {
        const int ij = c / o.gC;
        const int ch = c - ij * o.gC;
        v = rc.off + (uint32_t)((((ij >> 1) * (2 * o.gW) + (ij & 1)) * o.gC + ch) * 4);
    } else {
        v = rc.off + 4u * (uint32_t)c;
    }
    return ok ? v : COL_SENT;
}

// Operand access is split in two so that the global loads of the NEXT tile can stay in flight
// across the current tile's MFMAs: load_raw() only issues loads (no dependent math), finish() applies
// the fused transform right before the value is written to LDS.  (A_LN / A_LNBF: the per-column
// weight/bias in b/c are filled by the kernels, which load them once per k-tile.)
struct RawVec {
    float4 a, b, c;
};

template <int KIND>
__device__ __forceinline__ void load_raw(const Operand& o, const RowCtx& rc, int c, RawVec& r) {
    const uint32_t v = elem_voff<KIND>(o, rc, c);
    r.a = buf_ld4(o.rs, v);
    if constexpr (KIND == A_SCALE) r.b = buf_ld4(o.rs_s, c < o.ncols ? rc.soff + 4u * (uint32_t)c : COL_SENT);
    if constexpr (KIND == A_SG) r.b = buf_ld4(o.rs, v + 4u * (uint32_t)o.ncols);
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
