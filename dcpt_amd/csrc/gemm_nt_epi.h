// Epilogues of the fp32-output NT GEMMs (gemm_nt.hip: 128-row tiles, 256 threads; gemm_x3.hip: 256 x 256 tiles written as four
// 128 x 128 quadrants, 512 threads).
#pragma once
#include "gemm_operand.h"

namespace {

// element index of output row m in the fine NHWC image of the scatter epilogues (i = j = ch = 0)
__device__ __forceinline__ int64_t scatter_elem(const GemmNT& p, int64_t m) {
    const int w = (int)(m % p.gW);
    const int64_t t = m / p.gW;
    const int h = (int)(t % p.gH);
    const int64_t b = t / p.gH;
    return ((b * (2 * p.gH) + 2 * h) * (int64_t)(2 * p.gW) + 2 * w) * p.gC;
}

// Epilogue: the accumulators are first parked in LDS ([ROWS][BN] floats, reusing the operand tiles'
// space), then every thread owns one float4 column group of ROWS / RPP rows: global loads (residual / gate
// inputs) and stores are 16 B per lane and BN*4 B contiguous per row, all through buffer windows opened at
// the tile's first row (rows past M / columns past N are dropped by the range check), and all the loads
// of a thread are issued before the LDS round trip so their latency overlaps it.
// E_BIASGATE: the tile's columns are [BN/2 of the first half | the matching BN/2 of the second half]; a thread owns the
// float4 group q of both halves, writes C = acc + bias for both and their product to `gate`.
template <int ROWS, int BN, int NT_>
__device__ __forceinline__ void epilogue_gate(const GemmNT& p, const float* __restrict__ Cs, int64_t m0, int n0h, int tid) {
    constexpr int HB = BN / 2;
    constexpr int Q = HB / 4;          // float4 groups per half row
    constexpr int RPP = NT_ / Q;
    constexpr int IT = ROWS / RPP;
    const int Ch = p.N / 2;
    const int q = tid % Q, r0 = tid / Q;
    const int n = n0h + 4 * q;
    const bool nok = n < Ch;
    float4 b1 = f4_zero(), b2 = f4_zero();
    if (p.bias && nok) {
        b1 = ldg4(p.bias + n);
        b2 = ldg4(p.bias + Ch + n);
    }
    const rsrc_t rsC = make_rsrc(p.C + m0 * (int64_t)p.ldc);
    const rsrc_t rsG = make_rsrc(p.gate + m0 * (int64_t)Ch);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int rl = r0 + it * RPP;
        const bool ok = (m0 + rl < p.M) && nok;
        const float4 v1 = f4_add(*reinterpret_cast<const float4*>(&Cs[rl * BN + 4 * q]), b1);
        const float4 v2 = f4_add(*reinterpret_cast<const float4*>(&Cs[rl * BN + HB + 4 * q]), b2);
        const uint32_t o = ok ? ((uint32_t)rl * (uint32_t)p.ldc + (uint32_t)n) * 4u : ROW_SENT;
        buf_st4(rsC, o, v1);
        buf_st4(rsC, o + 4u * (uint32_t)Ch, v2);
        buf_st4(rsG, ok ? ((uint32_t)rl * (uint32_t)Ch + (uint32_t)n) * 4u : ROW_SENT, f4_mul(v1, v2));
    }
}

template <int EK, int ROWS, int BN, int NT_>
__device__ __forceinline__ void epilogue_rows(const GemmNT& p, const float* __restrict__ Cs, int64_t m0, int n0, int tid) {
    constexpr int Q = BN / 4;                        // float4 groups per row
    constexpr int QP = (NT_ % Q == 0) ? Q : 32;      // lanes per row (BN = 96: 24 of 32 lanes carry a group)
    constexpr int RPP = NT_ / QP;                    // rows per pass
    constexpr int IT = ROWS / RPP;
    static_assert(QP >= Q && ROWS % RPP == 0, "epilogue row map");
    const int ldres = p.ldres ? p.ldres : p.ldc;
    const int q = tid % QP, r0 = tid / QP;
    const int n = n0 + 4 * q;
    const bool nok = (QP == Q || q < Q) && n < p.N;
    float4 bias = f4_zero(), cs = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (EK == E_BIAS || EK == E_RESID || EK == E_MUL || EK == E_RESIDLN) {
        if (p.bias && nok) bias = ldg4(p.bias + n);
    }
    if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_RESIDLN) {
        if (p.cscale && nok) cs = ldg4(p.cscale + n);
    }
    constexpr bool SCAT = (EK == E_SCATTER || EK == E_SCATTER_ADD);
    // windows
    int64_t cbase;
    uint32_t coladd;  // byte offset of this thread's column group inside a row
    if constexpr (SCAT) {
        const int64_t mf = (m0 < p.M) ? m0 : 0;
        cbase = scatter_elem(p, mf);
        const int nn = nok ? n : 0;
        const int ij = nn / p.gC;
        const int ch = nn - ij * p.gC;
        coladd = (uint32_t)((((ij >> 1) * (2 * p.gW) + (ij & 1)) * p.gC + ch) * 4);
    } else {
        cbase = m0 * (int64_t)p.ldc;
        coladd = 4u * (uint32_t)n;
    }
    const rsrc_t rsC = make_rsrc(p.C + cbase);
    rsrc_t rsR = rsC, rsX = rsC;
    if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_MUL || EK == E_DOTCOL || EK == E_LNBWD || EK == E_RESIDLN || EK == E_LNBWD2) rsR = make_rsrc(p.res + m0 * (int64_t)ldres);
    if constexpr (EK == E_RESIDLN) rsX = make_rsrc(p.ln_out + m0 * (int64_t)p.ldc);
    if constexpr (EK == E_LNBWD || EK == E_LNBWD2) rsX = make_rsrc((p.aux ? p.aux : p.res) + m0 * (int64_t)ldres);
    if constexpr (EK == E_SCATTER_ADD) rsR = make_rsrc(p.res + cbase);
    if constexpr (EK == E_SGBWD) rsX = make_rsrc(p.aux + m0 * (2 * (int64_t)p.N));
    constexpr int HALF = (EK == E_SGBWD) ? 2 : 1;   // SGBWD needs two loads per row: do it in two halves
    constexpr int ITH = IT / HALF;
    float4 dot = f4_zero();   // E_DOTCOL / E_LNBWD: this thread's part of the column sums
    float4 dot2 = f4_zero();  // E_LNBWD: second plane (sum of g)
    float4 lnw4 = f4_zero(), lnb4 = f4_zero();
    if constexpr (EK == E_LNBWD || EK == E_RESIDLN || EK == E_LNBWD2) {
        if (nok) lnw4 = ldg4(p.lnw + n);
    }
    // E_SGBWD with row partials: this thread's entries of u / cvec for both halves of the gate
    float4 u_lo = f4_zero(), u_hi = f4_zero(), c_lo = f4_zero(), c_hi = f4_zero();
    if constexpr (EK == E_SGBWD) {
        if (p.rowpart && nok) {
            u_lo = ldg4(p.uvec + n);
            u_hi = ldg4(p.uvec + p.N + n);
            c_lo = ldg4(p.cvec + n);
            c_hi = ldg4(p.cvec + p.N + n);
        }
    }
    if constexpr (EK == E_RESIDLN) {
        if (nok && p.lnb) lnb4 = ldg4(p.lnb + n);
    }
#pragma unroll
    for (int hh = 0; hh < HALF; ++hh) {
        float4 pre1[ITH], pre2[ITH];
        uint32_t addr[ITH];
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int rl = r0 + (hh * ITH + it) * RPP;
            const int64_t m = m0 + rl;
            const bool ok = (m < p.M) && nok;
            pre1[it] = f4_zero();
            pre2[it] = f4_zero();
            if constexpr (SCAT) {
                addr[it] = ok ? (uint32_t)((scatter_elem(p, m) - cbase) * 4) + coladd : ROW_SENT;
                if constexpr (EK == E_SCATTER_ADD) pre1[it] = buf_ld4(rsR, addr[it]);
            } else {
                addr[it] = ok ? (uint32_t)rl * (uint32_t)p.ldc * 4u + coladd : ROW_SENT;
                if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_MUL || EK == E_DOTCOL || EK == E_LNBWD || EK == E_RESIDLN || EK == E_LNBWD2)
                    pre1[it] = buf_ld4(rsR, ok ? (uint32_t)rl * (uint32_t)ldres * 4u + coladd : ROW_SENT);
                if constexpr (EK == E_LNBWD || EK == E_LNBWD2) {
                    if (p.aux) pre2[it] = buf_ld4(rsX, ok ? (uint32_t)rl * (uint32_t)ldres * 4u + coladd : ROW_SENT);
                }
                if constexpr (EK == E_SGBWD) {
                    const uint32_t xo = ok ? (uint32_t)rl * (uint32_t)p.N * 8u + coladd : ROW_SENT;
                    pre1[it] = buf_ld4(rsX, xo);
                    pre2[it] = buf_ld4(rsX, xo + 4u * (uint32_t)p.N);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int rl = r0 + (hh * ITH + it) * RPP;
            const float4 v = *reinterpret_cast<const float4*>(&Cs[rl * BN + 4 * q]);
            if constexpr (EK == E_RESIDLN) {
                // y = res + (acc + bias) * gain, then LayerNorm of the row y (two-pass mean / variance like ln_fwd)
                const int64_t m = m0 + rl;
                const float4 y = f4_fma(f4_add(v, bias), cs, pre1[it]);
                buf_st4(rsC, addr[it], y);
                const float invN = 1.0f / (float)p.N;
                const float mean = group_sum(nok ? f4_sum(y) : 0.f, QP) * invN;
                const float4 dlt = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                const float var = group_sum(nok ? f4_sum(f4_mul(dlt, dlt)) : 0.f, QP) * invN;
                const float rs = 1.0f / sqrtf(var + p.ln_eps);
                buf_st4(rsX, addr[it], f4_fma(make_float4(dlt.x * rs, dlt.y * rs, dlt.z * rs, dlt.w * rs), lnw4, lnb4));
                if (q == 0 && m < p.M) {
                    p.ln_mu[m] = mean;
                    p.ln_rstd[m] = rs;
                }
            } else if constexpr (EK == E_LNBWD) {
                // LayerNorm backward of this row (the tile spans all N columns; a row lives in QP consecutive lanes)
                const int64_t m = m0 + rl;
                const bool rok = m < p.M;
                const float mean = rok ? p.mu[m] : 0.f, rs = rok ? p.rstd[m] : 0.f;
                const float4 xh = make_float4((pre1[it].x - mean) * rs, (pre1[it].y - mean) * rs, (pre1[it].z - mean) * rs,
                                              (pre1[it].w - mean) * rs);
                const float4 gw = f4_mul(v, lnw4);
                const float invN = 1.0f / (float)p.N;
                const float s1 = group_sum(nok ? f4_sum(gw) : 0.f, QP) * invN;
                const float s2 = group_sum(nok ? f4_sum(f4_mul(gw, xh)) : 0.f, QP) * invN;
                float4 d;
                d.x = rs * (gw.x - xh.x * s2 - s1);
                d.y = rs * (gw.y - xh.y * s2 - s1);
                d.z = rs * (gw.z - xh.z * s2 - s1);
                d.w = rs * (gw.w - xh.w * s2 - s1);
                buf_st4(rsC, addr[it], f4_add(d, pre2[it]));
                if (rok && nok) {
                    dot = f4_fma(v, xh, dot);
                    dot2 = f4_add(dot2, v);
                }
            } else if constexpr (EK == E_LNBWD2) {
                // the row sums come from the producer of dZ: lane q of the row's lane group fetches partial q, the group adds them
                const int64_t m = m0 + rl;
                const bool rok = m < p.M;
                const float mean = rok ? p.mu[m] : 0.f, rs = rok ? p.rstd[m] : 0.f;
                float a1 = 0.f, a2 = 0.f;
                if (rok && q < p.rowparts) {
                    const float2 pr = *reinterpret_cast<const float2*>(p.rowpart + (m * p.rowparts + q) * 2);
                    a1 = pr.x;
                    a2 = pr.y;
                }
                const float invN = 1.0f / (float)p.N;
                const float s1 = group_sum(a1, QP) * invN, s2 = group_sum(a2, QP) * invN;
                const float4 xh = make_float4((pre1[it].x - mean) * rs, (pre1[it].y - mean) * rs, (pre1[it].z - mean) * rs,
                                              (pre1[it].w - mean) * rs);
                const float4 gw = f4_mul(v, lnw4);
                float4 d;
                d.x = rs * (gw.x - xh.x * s2 - s1);
                d.y = rs * (gw.y - xh.y * s2 - s1);
                d.z = rs * (gw.z - xh.z * s2 - s1);
                d.w = rs * (gw.w - xh.w * s2 - s1);
                buf_st4(rsC, addr[it], f4_add(d, pre2[it]));
                if (rok && nok) {
                    dot = f4_fma(v, xh, dot);
                    dot2 = f4_add(dot2, v);
                }
            } else if constexpr (EK == E_DOTCOL) {
                buf_st4(rsC, addr[it], v);
                dot = f4_fma(v, pre1[it], dot);   // rows past M loaded 0
            } else if constexpr (EK == E_PLAIN || EK == E_SCATTER) {
                buf_st4(rsC, addr[it], v);
            } else if constexpr (EK == E_BIAS) {
                buf_st4(rsC, addr[it], f4_add(v, bias));
            } else if constexpr (EK == E_RESID) {
                buf_st4(rsC, addr[it], f4_fma(f4_add(v, bias), cs, pre1[it]));
            } else if constexpr (EK == E_ADDSCALED) {
                buf_st4(rsC, addr[it], f4_fma(cs, pre1[it], v));
            } else if constexpr (EK == E_MUL) {
                buf_st4(rsC, addr[it], f4_mul(f4_add(v, bias), pre1[it]));
            } else if constexpr (EK == E_SGBWD) {
                const float4 d1 = f4_mul(v, pre2[it]), d2 = f4_mul(v, pre1[it]);   // gradients of the first / second half of the gate input
                buf_st4(rsC, addr[it], d1);
                buf_st4(rsC, addr[it] + 4u * (uint32_t)p.N, d2);
                if (p.rowpart) {   // row partials of  dZ . u  and  dZ . (Z - cvec)  for the LayerNorm backward downstream (E_LNBWD2)
                    const int64_t m = m0 + rl;
                    const float4 z1 = make_float4(pre1[it].x - c_lo.x, pre1[it].y - c_lo.y, pre1[it].z - c_lo.z, pre1[it].w - c_lo.w);
                    const float4 z2 = make_float4(pre2[it].x - c_hi.x, pre2[it].y - c_hi.y, pre2[it].z - c_hi.z, pre2[it].w - c_hi.w);
                    float a1 = nok ? f4_sum(f4_mul(d1, u_lo)) + f4_sum(f4_mul(d2, u_hi)) : 0.f;
                    float a2 = nok ? f4_sum(f4_mul(d1, z1)) + f4_sum(f4_mul(d2, z2)) : 0.f;
                    a1 = group_sum(a1, QP);
                    a2 = group_sum(a2, QP);
                    if (q == 0 && m < p.M) {
                        const int np = (p.N + BN - 1) / BN;
                        *reinterpret_cast<float2*>(p.rowpart + (m * np + n0 / BN) * 2) = make_float2(a1, a2);
                    }
                }
            } else {  // E_SCATTER_ADD
                buf_st4(rsC, addr[it], f4_add(v, pre1[it]));
            }
        }
    }
    if constexpr (EK == E_LNBWD || EK == E_LNBWD2) {
        // the two column-sum planes over the tile's rows, as for E_DOTCOL
        float* sm = const_cast<float*>(Cs);
        for (int pl = 0; pl < 2; ++pl) {
            __syncthreads();
            if (QP == Q || q < Q) *reinterpret_cast<float4*>(&sm[r0 * BN + 4 * q]) = pl == 0 ? dot : dot2;
            __syncthreads();
            if (r0 == 0 && nok) {
                float4 t = *reinterpret_cast<const float4*>(&sm[4 * q]);
#pragma unroll
                for (int g = 1; g < RPP; ++g) t = f4_add(t, *reinterpret_cast<const float4*>(&sm[g * BN + 4 * q]));
                stg4(p.colpart + ((m0 / ROWS) * 2 + pl) * (int64_t)p.N + n, t);
            }
        }
    }
    if constexpr (EK == E_DOTCOL) {
        // column sums over the tile's rows: RPP row groups through LDS (the staged C tile is dead), fixed order
        float* sm = const_cast<float*>(Cs);
        __syncthreads();
        if (QP == Q || q < Q) *reinterpret_cast<float4*>(&sm[r0 * BN + 4 * q]) = dot;
        __syncthreads();
        if (r0 == 0 && nok) {
            float4 t = *reinterpret_cast<const float4*>(&sm[4 * q]);
#pragma unroll
            for (int g = 1; g < RPP; ++g) t = f4_add(t, *reinterpret_cast<const float4*>(&sm[g * BN + 4 * q]));
            stg4(p.colpart + (m0 / ROWS) * (int64_t)p.N + n, t);
        }
    }
}

}  // namespace
