// Epilogues of the bf16 NT GEMMs (gemm_bf16.hip: 128-row tiles, 256 threads; gemm_bf16_256.hip: 256 x 256 tiles written as four
// 128 x 128 quadrants, 512 threads): the fp32 accumulator tile is parked in LDS as Cs[BM][BN], a thread owns 8 consecutive columns
// of a row (16-byte bf16 accesses), residuals are prefetched through range-checked buffer windows.
#pragma once
#include "bf16.h"

namespace {

// first element of the fine pixel (2h, 2w) that coarse pixel m = (b, h, w) of a gH x gW grid maps to (fine image 2gH x 2gW x gC)
template <typename P>
__device__ __forceinline__ int64_t fine_elem(const P& p, int64_t m) {
    const int w = (int)(m % p.gW);
    const int64_t t = m / p.gW;
    const int h = (int)(t % p.gH);
    const int64_t b = t / p.gH;
    return ((b * (2 * p.gH) + 2 * h) * (int64_t)(2 * p.gW) + 2 * w) * p.gC;
}

template <int EK, int BM, int BN, int NT = 256>   // NT: threads of the block
__device__ __forceinline__ void epilogue8(const GemmNTB& p, float* __restrict__ Cs, int64_t m0, int n0, int tid) {
    constexpr bool GATE = (EK == EB_BIASGATE);
    constexpr int W = GATE ? BN / 2 : BN;   // columns a thread row spans
    constexpr int Q = W / 8;                // 8-column groups per row
    constexpr int RPP = NT / Q;             // rows per pass
    constexpr int IT = BM / RPP;
    static_assert(BM % RPP == 0, "epilogue row map");
    const int q = tid % Q, r0 = tid / Q;
    const int Ch = p.N / 2;
    const int n = n0 + 8 * q;
    const bool nok = GATE ? (n < Ch) : (n < p.N);
    const int ldres = p.ldres ? p.ldres : p.ldc;
    f8 bias = f8_zero(), bias2 = f8_zero(), cs = f8{make_float4(1.f, 1.f, 1.f, 1.f), make_float4(1.f, 1.f, 1.f, 1.f)};
    if constexpr (EK == EB_BIAS || EK == EB_RESID) {
        if (p.bias && nok) bias = f8_ld(p.bias + n);
    }
    if constexpr (GATE) {
        if (p.bias && nok) {
            bias = f8_ld(p.bias + n);
            bias2 = f8_ld(p.bias + Ch + n);
        }
    }
    if constexpr (EK == EB_RESID) {
        if (p.cscale && nok) cs = f8_ld(p.cscale + n);
    }
    constexpr bool SCAT = (EK == EB_SCATTER || EK == EB_SCATTER_ADD);
    int64_t cbase = m0 * (int64_t)p.ldc;
    uint32_t coladd = 0;   // SCAT: byte offset of this thread's 8 columns (one (i, j) cell, 8 channels) relative to the row's fine pixel
    if constexpr (SCAT) {
        cbase = fine_elem(p, m0 < p.M ? m0 : 0);
        const int nn = nok ? n : 0;
        const int ij = nn / p.gC, ch = nn - ij * p.gC;
        coladd = (uint32_t)((((ij >> 1) * (2 * p.gW) + (ij & 1)) * p.gC + ch) * 2);
    }
    const rsrc_t rsC = make_rsrc(p.C + cbase);
    rsrc_t rsR = rsC, rsX = rsC;
    if constexpr (EK == EB_SCATTER_ADD) rsR = make_rsrc(p.res + cbase);
    if constexpr (EK == EB_RESID || EK == EB_DOTCOL || EK == EB_LNBWD2) rsR = make_rsrc(p.res + m0 * (int64_t)ldres);
    if constexpr (EK == EB_LNBWD2) rsX = make_rsrc((p.aux ? p.aux : p.res) + m0 * (int64_t)ldres);
    rsrc_t rsY = rsC;   // EB_LNFWD: y2 (the LayerNorm output);  EB_LNBWDM: y2 (the masked gradient, optional) -- rows as C's
    if constexpr (EK == EB_LNFWD) {
        rsR = make_rsrc(p.res + m0 * (int64_t)p.ldc);
        rsY = make_rsrc(p.y2 + m0 * (int64_t)p.ldc);
    }
    if constexpr (EK == EB_LNBWDM) {
        rsR = make_rsrc(p.aux + m0 * (int64_t)p.ldc);     // z, the LayerNorm's input
        rsX = make_rsrc(p.ymask + m0 * (int64_t)p.ldc);   // its [ReLU'd] output
        rsY = make_rsrc(p.y2 + m0 * (int64_t)p.ldc);
    }
    f8 lnb8 = f8_zero();
    if constexpr (EK == EB_SGBWD) rsX = make_rsrc(p.aux + m0 * (2 * (int64_t)p.N));
    if constexpr (GATE) rsX = make_rsrc(p.gate + m0 * (int64_t)Ch);
    f8 dot = f8_zero(), dot2 = f8_zero(), lnw8 = f8_zero(), u_lo = f8_zero(), u_hi = f8_zero(), c_lo = f8_zero(), c_hi = f8_zero();
    if constexpr (EK == EB_LNBWD2 || EK == EB_LNFWD || EK == EB_LNBWDM) {
        if (nok) lnw8 = f8_ld(p.lnw + n);
    }
    if constexpr (EK == EB_LNFWD) {
        if (nok) lnb8 = f8_ld(p.lnb + n);
    }
    if constexpr (EK == EB_LNBWDM) {
        if (nok && p.relu && !p.ymask) lnb8 = f8_ld(p.lnb + n);
    }
    if constexpr (EK == EB_SGBWD) {
        if (p.rowpart && nok) {
            u_lo = f8_ld(p.uvec + n);
            u_hi = f8_ld(p.uvec + p.N + n);
            c_lo = f8_ld(p.cvec + n);
            c_hi = f8_ld(p.cvec + p.N + n);
        }
    }
    // two 16-byte loads per row: prefetch in two halves (registers); the masked LayerNorm backward keeps two rows in flight per thread
    constexpr int HALF = (EK == EB_SGBWD) ? 2 : (EK == EB_LNBWDM && IT >= 4) ? IT / 2 : 1;
    constexpr int ITH = IT / HALF;
#pragma unroll
    for (int hh = 0; hh < HALF; ++hh) {
        f8 pre1[ITH], pre2[ITH];
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int rl = r0 + (hh * ITH + it) * RPP;
            const bool ok = (m0 + rl < p.M) && nok;
            pre1[it] = f8_zero();
            pre2[it] = f8_zero();
            if constexpr (EK == EB_RESID || EK == EB_DOTCOL || EK == EB_LNBWD2)
                pre1[it] = bbuf_ld8(rsR, ok ? ((uint32_t)rl * (uint32_t)ldres + (uint32_t)n) * 2u : ROW_SENT);
            if constexpr (EK == EB_SCATTER_ADD) pre1[it] = bbuf_ld8(rsR, ok ? (uint32_t)((fine_elem(p, m0 + rl) - cbase) * 2) + coladd : ROW_SENT);
            if constexpr (EK == EB_LNBWD2) {
                if (p.aux) pre2[it] = bbuf_ld8(rsX, ok ? ((uint32_t)rl * (uint32_t)ldres + (uint32_t)n) * 2u : ROW_SENT);
            }
            if constexpr (EK == EB_SGBWD) {
                const uint32_t xo = ok ? ((uint32_t)rl * (uint32_t)p.N * 2u + (uint32_t)n) * 2u : ROW_SENT;
                pre1[it] = bbuf_ld8(rsX, xo);
                pre2[it] = bbuf_ld8(rsX, xo + 2u * (uint32_t)p.N);
            }
            if constexpr (EK == EB_LNFWD) {
                if (p.res) pre1[it] = bbuf_ld8(rsR, ok ? ((uint32_t)rl * (uint32_t)p.ldc + (uint32_t)n) * 2u : ROW_SENT);
            }
            if constexpr (EK == EB_LNBWDM) {
                const uint32_t xo = ok ? ((uint32_t)rl * (uint32_t)p.ldc + (uint32_t)n) * 2u : ROW_SENT;
                pre1[it] = bbuf_ld8(rsR, xo);
                if (p.ymask) pre2[it] = bbuf_ld8(rsX, xo);
            }
        }
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int rl = r0 + (hh * ITH + it) * RPP;
            const bool ok = (m0 + rl < p.M) && nok;
            uint32_t o = ok ? ((uint32_t)rl * (uint32_t)p.ldc + (uint32_t)n) * 2u : ROW_SENT;
            if constexpr (SCAT) o = ok ? (uint32_t)((fine_elem(p, m0 + rl) - cbase) * 2) + coladd : ROW_SENT;
            f8 v;
            v.lo = *reinterpret_cast<const float4*>(&Cs[rl * BN + 8 * q]);
            v.hi = *reinterpret_cast<const float4*>(&Cs[rl * BN + 8 * q + 4]);
            if constexpr (GATE) {
                f8 v2;
                v2.lo = *reinterpret_cast<const float4*>(&Cs[rl * BN + W + 8 * q]);
                v2.hi = *reinterpret_cast<const float4*>(&Cs[rl * BN + W + 8 * q + 4]);
                v = f8_add(v, bias);
                v2 = f8_add(v2, bias2);
                if (p.C) {   // (null: inference -- only the gate output is kept)
                    bbuf_st8(rsC, o, v);
                    bbuf_st8(rsC, o + 2u * (uint32_t)Ch, v2);
                }
                bbuf_st8(rsX, ok ? ((uint32_t)rl * (uint32_t)Ch + (uint32_t)n) * 2u : ROW_SENT, f8_mul(v, v2));
            } else if constexpr (EK == EB_PLAIN || EK == EB_SCATTER) {
                bbuf_st8(rsC, o, v);
            } else if constexpr (EK == EB_SCATTER_ADD) {
                bbuf_st8(rsC, o, f8_add(v, pre1[it]));
            } else if constexpr (EK == EB_BIAS) {
                bbuf_st8(rsC, o, f8_add(v, bias));
            } else if constexpr (EK == EB_RESID) {
                bbuf_st8(rsC, o, f8_fma(f8_add(v, bias), cs, pre1[it]));
            } else if constexpr (EK == EB_SGBWD) {
                const f8 d1 = f8_mul(v, pre2[it]), d2 = f8_mul(v, pre1[it]);
                bbuf_st8(rsC, o, d1);
                bbuf_st8(rsC, o + 2u * (uint32_t)p.N, d2);
                if (p.rowpart) {
                    const int64_t m = m0 + rl;
                    f8 z1, z2;
                    z1.lo = make_float4(pre1[it].lo.x - c_lo.lo.x, pre1[it].lo.y - c_lo.lo.y, pre1[it].lo.z - c_lo.lo.z, pre1[it].lo.w - c_lo.lo.w);
                    z1.hi = make_float4(pre1[it].hi.x - c_lo.hi.x, pre1[it].hi.y - c_lo.hi.y, pre1[it].hi.z - c_lo.hi.z, pre1[it].hi.w - c_lo.hi.w);
                    z2.lo = make_float4(pre2[it].lo.x - c_hi.lo.x, pre2[it].lo.y - c_hi.lo.y, pre2[it].lo.z - c_hi.lo.z, pre2[it].lo.w - c_hi.lo.w);
                    z2.hi = make_float4(pre2[it].hi.x - c_hi.hi.x, pre2[it].hi.y - c_hi.hi.y, pre2[it].hi.z - c_hi.hi.z, pre2[it].hi.w - c_hi.hi.w);
                    float a1 = nok ? f8_sum(f8_mul(d1, u_lo)) + f8_sum(f8_mul(d2, u_hi)) : 0.f;
                    float a2 = nok ? f8_sum(f8_mul(d1, z1)) + f8_sum(f8_mul(d2, z2)) : 0.f;
                    a1 = group_sum(a1, Q);
                    a2 = group_sum(a2, Q);
                    if (q == 0 && m < p.M) {
                        const int np = (p.N + BN - 1) / BN;
                        *reinterpret_cast<float2*>(p.rowpart + (m * np + n0 / BN) * 2) = make_float2(a1, a2);
                    }
                }
            } else if constexpr (EK == EB_LNFWD) {
                // (the expressions, the lane -> channel map and the reduction tree of ln_fwd_bf16_kernel: the two paths agree bit for bit)
                const int64_t m = m0 + rl;
                u32x4 zw;
                const f8 z = f8_round_bf16(v, zw);
                bbuf_st8_raw(rsC, o, zw);
                const float mean = group_sum(nok ? f8_sum(z) : 0.f, Q) / (float)p.N;
                f8 d;
                d.lo = make_float4(z.lo.x - mean, z.lo.y - mean, z.lo.z - mean, z.lo.w - mean);
                d.hi = make_float4(z.hi.x - mean, z.hi.y - mean, z.hi.z - mean, z.hi.w - mean);
                const float sq = group_sum(nok ? f8_sum(f8_mul(d, d)) : 0.f, Q);
                const float rs = 1.0f / sqrtf(sq / (float)p.N + p.eps);
                if (q == 0 && m < p.M) {
                    p.mu_out[m] = mean;
                    p.rstd_out[m] = rs;
                }
                const f8 r8 = f8{make_float4(rs, rs, rs, rs), make_float4(rs, rs, rs, rs)};
                f8 ov = f8_fma(f8_mul(d, r8), lnw8, lnb8);
                if (p.res) ov = f8_add(ov, pre1[it]);
                if (p.relu) {
                    ov.lo = make_float4(fmaxf(ov.lo.x, 0.f), fmaxf(ov.lo.y, 0.f), fmaxf(ov.lo.z, 0.f), fmaxf(ov.lo.w, 0.f));
                    ov.hi = make_float4(fmaxf(ov.hi.x, 0.f), fmaxf(ov.hi.y, 0.f), fmaxf(ov.hi.z, 0.f), fmaxf(ov.hi.w, 0.f));
                }
                bbuf_st8(rsY, o, ov);
            } else if constexpr (EK == EB_LNBWDM) {
                // (ln_bwd_bf16_kernel's arithmetic on the gradient as the unfused GEMM would have stored it: rounded to bf16 first)
                const int64_t m = m0 + rl;
                const bool rok = m < p.M;
                const float mean = rok ? p.mu[m] : 0.f, rs = rok ? p.rstd[m] : 0.f;
                u32x4 gw_raw;
                f8 g = f8_round_bf16(v, gw_raw);
                if (p.ymask) {
                    const f8 ym = pre2[it];
                    g.lo = make_float4(ym.lo.x > 0.f ? g.lo.x : 0.f, ym.lo.y > 0.f ? g.lo.y : 0.f, ym.lo.z > 0.f ? g.lo.z : 0.f, ym.lo.w > 0.f ? g.lo.w : 0.f);
                    g.hi = make_float4(ym.hi.x > 0.f ? g.hi.x : 0.f, ym.hi.y > 0.f ? g.hi.y : 0.f, ym.hi.z > 0.f ? g.hi.z : 0.f, ym.hi.w > 0.f ? g.hi.w : 0.f);
                }
                f8 xh;
                xh.lo = make_float4((pre1[it].lo.x - mean) * rs, (pre1[it].lo.y - mean) * rs, (pre1[it].lo.z - mean) * rs, (pre1[it].lo.w - mean) * rs);
                xh.hi = make_float4((pre1[it].hi.x - mean) * rs, (pre1[it].hi.y - mean) * rs, (pre1[it].hi.z - mean) * rs, (pre1[it].hi.w - mean) * rs);
                if (p.relu && !p.ymask) {
                    // the ReLU mask recomputed from z instead of read from y: the forward's own expression (EB_LNFWD / ln_fwd_bf16_kernel:
                    // fma((z - mean) rstd, w, b)), whose sign is the sign of the bf16 value it was rounded to
                    const f8 u = f8_fma(xh, lnw8, lnb8);
                    g.lo = make_float4(u.lo.x > 0.f ? g.lo.x : 0.f, u.lo.y > 0.f ? g.lo.y : 0.f, u.lo.z > 0.f ? g.lo.z : 0.f, u.lo.w > 0.f ? g.lo.w : 0.f);
                    g.hi = make_float4(u.hi.x > 0.f ? g.hi.x : 0.f, u.hi.y > 0.f ? g.hi.y : 0.f, u.hi.z > 0.f ? g.hi.z : 0.f, u.hi.w > 0.f ? g.hi.w : 0.f);
                }
                if (p.y2) bbuf_st8(rsY, o, g);
                if (!ok) xh = f8_zero();
                const f8 gw = f8_mul(g, lnw8);
                const float invN = 1.0f / (float)p.N;
                const float s1 = group_sum(f8_sum(gw), Q) * invN, s2 = group_sum(f8_sum(f8_mul(gw, xh)), Q) * invN;
                f8 d;
                d.lo = make_float4(rs * (gw.lo.x - xh.lo.x * s2 - s1), rs * (gw.lo.y - xh.lo.y * s2 - s1), rs * (gw.lo.z - xh.lo.z * s2 - s1),
                                   rs * (gw.lo.w - xh.lo.w * s2 - s1));
                d.hi = make_float4(rs * (gw.hi.x - xh.hi.x * s2 - s1), rs * (gw.hi.y - xh.hi.y * s2 - s1), rs * (gw.hi.z - xh.hi.z * s2 - s1),
                                   rs * (gw.hi.w - xh.hi.w * s2 - s1));
                bbuf_st8(rsC, o, d);
                dot = f8_fma(g, xh, dot);   // (rows past M: g = 0, xhat = 0)
                dot2 = f8_add(dot2, g);
            } else if constexpr (EK == EB_LNBWD2) {
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
                const float s1 = group_sum(a1, Q) * invN, s2 = group_sum(a2, Q) * invN;
                f8 xh, d;
                xh.lo = make_float4((pre1[it].lo.x - mean) * rs, (pre1[it].lo.y - mean) * rs, (pre1[it].lo.z - mean) * rs, (pre1[it].lo.w - mean) * rs);
                xh.hi = make_float4((pre1[it].hi.x - mean) * rs, (pre1[it].hi.y - mean) * rs, (pre1[it].hi.z - mean) * rs, (pre1[it].hi.w - mean) * rs);
                const f8 gw = f8_mul(v, lnw8);
                d.lo = make_float4(rs * (gw.lo.x - xh.lo.x * s2 - s1), rs * (gw.lo.y - xh.lo.y * s2 - s1), rs * (gw.lo.z - xh.lo.z * s2 - s1),
                                   rs * (gw.lo.w - xh.lo.w * s2 - s1));
                d.hi = make_float4(rs * (gw.hi.x - xh.hi.x * s2 - s1), rs * (gw.hi.y - xh.hi.y * s2 - s1), rs * (gw.hi.z - xh.hi.z * s2 - s1),
                                   rs * (gw.hi.w - xh.hi.w * s2 - s1));
                bbuf_st8(rsC, o, f8_add(d, pre2[it]));
                if (rok && nok) {
                    dot = f8_fma(v, xh, dot);
                    dot2 = f8_add(dot2, v);
                }
            } else {   // EB_DOTCOL
                bbuf_st8(rsC, o, v);
                dot = f8_fma(v, pre1[it], dot);   // rows past M loaded 0
            }
        }
    }
    if constexpr (EK == EB_LNBWD2 || EK == EB_LNBWDM) {
        // the two column-sum planes (-> LayerNorm weight / bias gradients) over the tile's rows, as for EB_DOTCOL
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            __syncthreads();
            *reinterpret_cast<float4*>(&Cs[r0 * BN + 8 * q]) = pl == 0 ? dot.lo : dot2.lo;
            *reinterpret_cast<float4*>(&Cs[r0 * BN + 8 * q + 4]) = pl == 0 ? dot.hi : dot2.hi;
            __syncthreads();
            if (r0 == 0 && nok) {
                float4 a = *reinterpret_cast<const float4*>(&Cs[8 * q]), b = *reinterpret_cast<const float4*>(&Cs[8 * q + 4]);
#pragma unroll
                for (int g = 1; g < RPP; ++g) {
                    a = f4_add(a, *reinterpret_cast<const float4*>(&Cs[g * BN + 8 * q]));
                    b = f4_add(b, *reinterpret_cast<const float4*>(&Cs[g * BN + 8 * q + 4]));
                }
                float* dst = p.colpart + ((m0 / BM) * 2 + pl) * (int64_t)p.N + n;
                stg4(dst, a);
                stg4(dst + 4, b);
            }
        }
    }
    if constexpr (EK == EB_DOTCOL) {
        // column sums over the tile's rows: the RPP row groups through LDS (the staged tile is dead), fixed order
        __syncthreads();
        *reinterpret_cast<float4*>(&Cs[r0 * BN + 8 * q]) = dot.lo;
        *reinterpret_cast<float4*>(&Cs[r0 * BN + 8 * q + 4]) = dot.hi;
        __syncthreads();
        if (r0 == 0 && nok) {
            float4 a = *reinterpret_cast<const float4*>(&Cs[8 * q]), b = *reinterpret_cast<const float4*>(&Cs[8 * q + 4]);
#pragma unroll
            for (int g = 1; g < RPP; ++g) {
                a = f4_add(a, *reinterpret_cast<const float4*>(&Cs[g * BN + 8 * q]));
                b = f4_add(b, *reinterpret_cast<const float4*>(&Cs[g * BN + 8 * q + 4]));
            }
            float* dst = p.colpart + (m0 / BM) * (int64_t)p.N + n;
            stg4(dst, a);
            stg4(dst + 4, b);
        }
    }
}

}  // namespace
