// Bandwidth kernels of the bf16-storage path (bf16.h): per-pixel channel LayerNorm forward / backward on bf16 rows with fp32
// statistics (reference basicsr/archs/nafnet_arch.py:25-64), fp32 <-> bf16 casts at the path's edges, the bf16 weight packs of the
// MFMA GEMMs (straight, transposed + gain-scaled, per-image SCA-scaled), and SCA's channel sums when they do not come out of a
// GEMM epilogue.  A lane owns 8 consecutive channels (one 16-byte access); a row lives in G = pow2 >= C / 8 lanes of one wave.
#include "bf16.h"
#include "prof.h"
#include "bf16_ops.h"
#include "chain_bf16.h"

namespace {

// 8 consecutive bf16 through a plain (per-lane) pointer, predicated
__device__ __forceinline__ f8 ld8(const bf16_t* p, bool ok) {
    if (!ok) return f8_zero();
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
    return f8{make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y)), make_float4(bf_lo(w.z), bf_hi(w.z), bf_lo(w.w), bf_hi(w.w))};
}
// the same load kept PACKED (4 registers instead of 8): what the LayerNorm kernels prefetch for their next row
__device__ __forceinline__ u32x4 ld8raw(const bf16_t* p, bool ok) {
    u32x4 w;
    w.x = w.y = w.z = w.w = 0u;
    if (ok) w = *reinterpret_cast<const u32x4*>(p);
    return w;
}
__device__ __forceinline__ f8 unpack8(u32x4 w) {
    return f8{make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y)), make_float4(bf_lo(w.z), bf_hi(w.z), bf_lo(w.w), bf_hi(w.w))};
}
__device__ __forceinline__ void st8(bf16_t* p, bool ok, f8 v) {
    if (!ok) return;
    u32x4 w;
    w.x = bf_pack(v.lo.x, v.lo.y);
    w.y = bf_pack(v.lo.z, v.lo.w);
    w.z = bf_pack(v.hi.x, v.hi.y);
    w.w = bf_pack(v.hi.z, v.hi.w);
    *reinterpret_cast<u32x4*>(p) = w;
}

__host__ __device__ inline int lnb_group(int C) {
    const int q = C / 8;
    int g = 1;
    while (g < q && g < 64) g <<= 1;
    return g;
}

// y = (x - mean) * rstd * w + b, statistics two-pass in fp32 over the bf16 inputs (biased variance, eps inside the sqrt)
template <int NQ>
__global__ __launch_bounds__(256) void ln_fwd_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          bf16_t* __restrict__ y, float* __restrict__ mu, float* __restrict__ rstd, int64_t M,
                                                          int C, float eps, int G, const bf16_t* __restrict__ res, int relu) {
    const int tid = threadIdx.x, gpb = 256 / G, lig = tid % G, nq = C / 8;
    f8 ww[NQ], bb[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = lig + i * G;
        ww[i] = q < nq ? f8_ld(w + 8 * q) : f8_zero();
        bb[i] = q < nq ? f8_ld(b + 8 * q) : f8_zero();
    }
    // the row of the NEXT iteration is loaded before this one's reductions (a row is one 16-byte load per lane: without the prefetch a wave has
    // 1 KB in flight and the kernel is latency-bound beyond the Infinity Cache -- 2.7 TB/s at 65 536 x 512 -- tools/ubench/stream_mix.hip)
    u32x4 nxt[NQ];
    {
        const int64_t row0 = (int64_t)blockIdx.x * gpb + tid / G;
#pragma unroll
        for (int i = 0; i < NQ; ++i) nxt[i] = ld8raw(x + (row0 < M ? row0 : 0) * (int64_t)C + 8 * (lig + i * G), row0 < M && lig + i * G < nq);
    }
    for (int64_t rb = blockIdx.x; rb * gpb < M; rb += gridDim.x) {
        const int64_t row = rb * gpb + tid / G;
        const bool valid = row < M;
        f8 v[NQ];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[i] = unpack8(nxt[i]);
            sum += f8_sum(v[i]);
        }
        {
            const int64_t row2 = (rb + gridDim.x) * gpb + tid / G;
#pragma unroll
            for (int i = 0; i < NQ; ++i) nxt[i] = ld8raw(x + (row2 < M ? row2 : 0) * (int64_t)C + 8 * (lig + i * G), row2 < M && lig + i * G < nq);
        }
        sum = group_sum(sum, G);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (lig + i * G < nq) {
                f8 d;
                d.lo = make_float4(v[i].lo.x - mean, v[i].lo.y - mean, v[i].lo.z - mean, v[i].lo.w - mean);
                d.hi = make_float4(v[i].hi.x - mean, v[i].hi.y - mean, v[i].hi.z - mean, v[i].hi.w - mean);
                sq += f8_sum(f8_mul(d, d));
                v[i] = d;
            }
        }
        sq = group_sum(sq, G);
        const float rs = 1.0f / sqrtf(sq / (float)C + eps);
        if (valid && lig == 0) {
            mu[row] = mean;
            rstd[row] = rs;
        }
        bf16_t* yr = y + (valid ? row : 0) * (int64_t)C;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            const f8 r8 = f8{make_float4(rs, rs, rs, rs), make_float4(rs, rs, rs, rs)};
            f8 o = f8_fma(f8_mul(v[i], r8), ww[i], bb[i]);
            if (res) o = f8_add(o, ld8(res + (valid ? row : 0) * (int64_t)C + 8 * q, valid && q < nq));   // channels-first LN + shortcut
            if (relu) {
                o.lo = make_float4(fmaxf(o.lo.x, 0.f), fmaxf(o.lo.y, 0.f), fmaxf(o.lo.z, 0.f), fmaxf(o.lo.w, 0.f));
                o.hi = make_float4(fmaxf(o.hi.x, 0.f), fmaxf(o.hi.y, 0.f), fmaxf(o.hi.z, 0.f), fmaxf(o.hi.w, 0.f));
            }
            st8(yr + 8 * q, valid && q < nq, o);
        }
    }
}

// dx = rstd * (g w - xhat mean_c(g w xhat) - mean_c(g w)) + dres;  part[blk][0][c] = sum_rows g xhat, part[blk][1][c] = sum_rows g
template <int NQ>
__global__ __launch_bounds__(256) void ln_bwd_bf16_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ rstd, const float* __restrict__ w,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ part,
                                                          int64_t M, int C, int G, int64_t iters, const bf16_t* __restrict__ ymask,
                                                          bf16_t* __restrict__ gmasked) {
    __shared__ float red[2][256 * 8 * NQ];
    const int tid = threadIdx.x, gpb = 256 / G, gid = tid / G, lig = tid % G, nq = C / 8;
    f8 ww[NQ], aw[NQ], ab[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = lig + i * G;
        ww[i] = q < nq ? f8_ld(w + 8 * q) : f8_zero();
        aw[i] = f8_zero();
        ab[i] = f8_zero();
    }
    const float invC = 1.0f / (float)C;
    // operands of the NEXT iteration are loaded before this one's reductions and stores (two rows in flight per wave: see ln_fwd_bf16_kernel)
    u32x4 ng[NQ], nx[NQ], nd[NQ], nm[NQ];
    float nmean, nrs;
    auto fetch = [&](int64_t it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * gpb + gid;
        const bool valid = it < iters && row < M;
        const int64_t ro = (valid ? row : 0) * (int64_t)C;
        nmean = valid ? mu[row] : 0.f;
        nrs = valid ? rstd[row] : 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            const bool ok = valid && q < nq;
            ng[i] = ld8raw(gy + ro + 8 * q, ok);
            nm[i] = ld8raw(ymask + ro + 8 * q, ok && ymask != nullptr);
            nx[i] = ld8raw(x + ro + 8 * q, ok);
            nd[i] = ld8raw(dres + ro + 8 * q, ok && dres != nullptr);
        }
    };
    fetch(0);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * gpb + gid;
        const bool valid = row < M;
        const int64_t ro = (valid ? row : 0) * (int64_t)C;
        const float mean = nmean, rs = nrs;
        f8 g[NQ], xh[NQ], dr[NQ], xraw[NQ], ymk[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            g[i] = unpack8(ng[i]);
            xraw[i] = unpack8(nx[i]);
            dr[i] = unpack8(nd[i]);
            ymk[i] = unpack8(nm[i]);
        }
        fetch(it + 1);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            const bool ok = valid && q < nq;
            if (ymask) {   // ReLU behind the LayerNorm: the gradient passes where the (post-ReLU) output is > 0
                const f8 ym = ymk[i];
                g[i].lo = make_float4(ym.lo.x > 0.f ? g[i].lo.x : 0.f, ym.lo.y > 0.f ? g[i].lo.y : 0.f, ym.lo.z > 0.f ? g[i].lo.z : 0.f,
                                      ym.lo.w > 0.f ? g[i].lo.w : 0.f);
                g[i].hi = make_float4(ym.hi.x > 0.f ? g[i].hi.x : 0.f, ym.hi.y > 0.f ? g[i].hi.y : 0.f, ym.hi.z > 0.f ? g[i].hi.z : 0.f,
                                      ym.hi.w > 0.f ? g[i].hi.w : 0.f);
            }
            if (gmasked) st8(gmasked + ro + 8 * q, ok, g[i]);   // the shortcut branch receives the masked gradient
            const f8 xv = xraw[i];
            xh[i].lo = make_float4((xv.lo.x - mean) * rs, (xv.lo.y - mean) * rs, (xv.lo.z - mean) * rs, (xv.lo.w - mean) * rs);
            xh[i].hi = make_float4((xv.hi.x - mean) * rs, (xv.hi.y - mean) * rs, (xv.hi.z - mean) * rs, (xv.hi.w - mean) * rs);
            if (!ok) xh[i] = f8_zero();
            const f8 gw = f8_mul(g[i], ww[i]);
            s1 += f8_sum(gw);
            s2 += f8_sum(f8_mul(gw, xh[i]));
            aw[i] = f8_fma(g[i], xh[i], aw[i]);
            ab[i] = f8_add(ab[i], g[i]);
        }
        s1 = group_sum(s1, G) * invC;
        s2 = group_sum(s2, G) * invC;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            const f8 gw = f8_mul(g[i], ww[i]);
            f8 d;
            d.lo = make_float4(rs * (gw.lo.x - xh[i].lo.x * s2 - s1), rs * (gw.lo.y - xh[i].lo.y * s2 - s1), rs * (gw.lo.z - xh[i].lo.z * s2 - s1),
                               rs * (gw.lo.w - xh[i].lo.w * s2 - s1));
            d.hi = make_float4(rs * (gw.hi.x - xh[i].hi.x * s2 - s1), rs * (gw.hi.y - xh[i].hi.y * s2 - s1), rs * (gw.hi.z - xh[i].hi.z * s2 - s1),
                               rs * (gw.hi.w - xh[i].hi.w * s2 - s1));
            st8(dx + ro + 8 * q, valid && q < nq, f8_add(d, dr[i]));
        }
    }
    // column partials of this block: the gpb row groups through LDS, fixed order
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float* r0 = &red[0][(tid * NQ + i) * 8];
        float* r1 = &red[1][(tid * NQ + i) * 8];
        *reinterpret_cast<float4*>(r0) = aw[i].lo;
        *reinterpret_cast<float4*>(r0 + 4) = aw[i].hi;
        *reinterpret_cast<float4*>(r1) = ab[i].lo;
        *reinterpret_cast<float4*>(r1 + 4) = ab[i].hi;
    }
    __syncthreads();
    if (gid == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            if (q >= nq) continue;
            for (int pl = 0; pl < 2; ++pl) {
                float4 a = f4_zero(), c = f4_zero();
                for (int gg = 0; gg < gpb; ++gg) {
                    const float* r = &red[pl][((gg * G + lig) * NQ + i) * 8];
                    a = f4_add(a, *reinterpret_cast<const float4*>(r));
                    c = f4_add(c, *reinterpret_cast<const float4*>(r + 4));
                }
                float* dst = part + ((int64_t)blockIdx.x * 2 + pl) * C + 8 * q;
                stg4(dst, a);
                stg4(dst + 4, c);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f8 v = f8_ld(x + 8 * i);
        u32x4 w;
        w.x = bf_pack(v.lo.x, v.lo.y);
        w.y = bf_pack(v.lo.z, v.lo.w);
        w.z = bf_pack(v.hi.x, v.hi.y);
        w.w = bf_pack(v.hi.z, v.hi.w);
        *reinterpret_cast<u32x4*>(y + 8 * i) = w;
    }
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(x + 8 * i);
        stg4(y + 8 * i, make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y)));
        stg4(y + 8 * i + 4, make_float4(bf_lo(w.z), bf_hi(w.z), bf_lo(w.w), bf_hi(w.w)));
    }
}

// weight packs: one launch, grid.y = job
template <int MAXJ>
__global__ __launch_bounds__(256) void wpack_bf16_kernel(const WpackBJobsT<MAXJ> jobs) {
    const int j = blockIdx.y;
    const float* __restrict__ in = jobs.in[j];
    bf16_t* __restrict__ out = jobs.out[j];
    const float* __restrict__ rs = jobs.rs[j];
    const float* __restrict__ ks = jobs.kscale[j];
    const int N = jobs.N[j], K = jobs.K[j], nimg = jobs.nimg[j] > 0 ? jobs.nimg[j] : 1;
    const int64_t total = (int64_t)nimg * N * K;
    if (jobs.transpose[j] == 8) {   // the depthwise conv's [N = 2C][K = 9] taps -> fp32 [9][2C] (dwconv.hip's dw_pack layout); `out` holds floats
        float* __restrict__ of = reinterpret_cast<float*>(out);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) of[(i % 9) * N + i / 9] = in[i];
        return;
    }
    if (jobs.transpose[j] == 9) {   // conv4 + conv5 of a wide NAFBlock as the chain kernel's fragment-order weight stream (chain_bf16.hip):
        // in = conv4_w [2C][C], rs = conv5_w [C][C] (the second matrix rides in the row-scale slot), N = 3 C, K = C.  Wave w of the chain
        // kernel reads FR fragments of 1 KB in the order it consumes them: conv4 pass p, k-step ks -> the v1 tile (gate channels
        // w CW + 32 p ..+31) then its v2 partner (+C); after all passes conv5, k-step ks -> output tiles t = 0 .. NT-1.  A fragment = A operand
        // of v_mfma_f32_32x32x16_bf16: lane (rho = lane & 31, kg = lane >> 5) holds W[row(rho)][16 ks + 8 kg ..+7], and the MFMA row rho
        // stands for channel 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3) of the tile, which makes the 16 accumulator registers of a lane
        // 16 CONSECUTIVE channels (32 contiguous bytes of a bf16 row).
        // (rs == null: the [2C][C] matrix alone -- conv1 for the kernel's LayerNorm1 -> conv1 form)
        const float* __restrict__ W5 = rs;
        const int C = K, CW = C / CHAIN_NW, NT = CW / 32, KS = C / 16, F4 = 2 * NT * KS, FR = W5 ? F4 + NT * KS : F4;
        const int64_t nq = (int64_t)CHAIN_NW * FR * 64;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
            const int lane = (int)(q & 63), fi = (int)((q >> 6) % FR), w = (int)((q >> 6) / FR);
            const int rho = lane & 31, kg = lane >> 5, cc = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
            const float* src;
            if (fi < F4) {
                const int rec = fi >> 1, f = fi & 1, pp = rec / KS, ks = rec % KS;
                src = in + (int64_t)(f * C + w * CW + 32 * pp + cc) * C + 16 * ks + 8 * kg;
            } else {
                const int f5 = fi - F4, ks = f5 / NT, t = f5 % NT;
                src = W5 + (int64_t)(w * CW + 32 * t + cc) * C + 16 * ks + 8 * kg;
            }
            const f8 v = f8_ld(src);
            u32x4 o;
            o.x = bf_pack(v.lo.x, v.lo.y);
            o.y = bf_pack(v.lo.z, v.lo.w);
            o.z = bf_pack(v.hi.x, v.hi.y);
            o.w = bf_pack(v.hi.z, v.hi.w);
            *reinterpret_cast<u32x4*>(out + q * 8) = o;
        }
        return;
    }
    if (jobs.transpose[j] == 11) {   // conv3 with SCA's per-image scale as the chain kernel's streams: out[img][wave][NT * KS fragments] of W3[n][k] * kscale[img][k]
        const int C = K, CW = C / CHAIN_NW, NT = CW / 32, KS = C / 16, FR = NT * KS;
        const int64_t per = (int64_t)CHAIN_NW * FR * 64, nq = per * nimg;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
            const int64_t img = q / per, qi = q - img * per;
            const int lane = (int)(qi & 63), fi = (int)((qi >> 6) % FR), w = (int)((qi >> 6) / FR);
            const int rho = lane & 31, kg = lane >> 5, cc = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
            const int kst = fi / NT, t = fi % NT, k0 = 16 * kst + 8 * kg;
            f8 v = f8_ld(in + (int64_t)(w * CW + 32 * t + cc) * C + k0);
            if (ks) v = f8_mul(v, f8_ld(ks + img * C + k0));
            u32x4 o;
            o.x = bf_pack(v.lo.x, v.lo.y);
            o.y = bf_pack(v.lo.z, v.lo.w);
            o.z = bf_pack(v.hi.x, v.hi.y);
            o.w = bf_pack(v.hi.z, v.hi.w);
            *reinterpret_cast<u32x4*>(out + q * 8) = o;
        }
        return;
    }
    if (jobs.transpose[j] == 10) {   // the chain kernels' stream of ONE transposed, row-scaled [C][C] matrix: out row r, column c = in[c][r] * rs[c]
        // (conv3 for the backward kernel: dts = dy (beta W3)^T).  Per wave NT * KS fragments: k-step ks, then output tile t (as conv5's part of mode 9).
        const int C = K, CW = C / CHAIN_NW, NT = CW / 32, KS = C / 16, FR = NT * KS;
        const int64_t nq = (int64_t)CHAIN_NW * FR * 64;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
            const int lane = (int)(q & 63), fi = (int)((q >> 6) % FR), w = (int)((q >> 6) / FR);
            const int rho = lane & 31, kg = lane >> 5, cc = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
            const int ks = fi / NT, t = fi % NT;
            const int r = w * CW + 32 * t + cc, c0 = 16 * ks + 8 * kg;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = in[(int64_t)(c0 + e) * C + r] * (rs ? rs[c0 + e] : 1.f);
            u32x4 o;
            o.x = bf_pack(v[0], v[1]);
            o.y = bf_pack(v[2], v[3]);
            o.z = bf_pack(v[4], v[5]);
            o.w = bf_pack(v[6], v[7]);
            *reinterpret_cast<u32x4*>(out + q * 8) = o;
        }
        return;
    }
    if (jobs.transpose[j] == 0 && K % 8 == 0) {   // the big one (per-image scaled weights, nimg * N * K elements): 8 per thread, 16-byte stores
        const int64_t nk = (int64_t)N * K;
        for (int64_t i8 = (int64_t)blockIdx.x * 256 + threadIdx.x; i8 * 8 < total; i8 += (int64_t)gridDim.x * 256) {
            const int64_t i = i8 * 8;
            const int64_t img = i / nk, e = i - img * nk;
            const int k = (int)(e % K);
            f8 v = f8_ld(in + e);
            if (ks) v = f8_mul(v, f8_ld(ks + img * K + k));
            u32x4 o;
            o.x = bf_pack(v.lo.x, v.lo.y);
            o.y = bf_pack(v.lo.z, v.lo.w);
            o.z = bf_pack(v.hi.x, v.hi.y);
            o.w = bf_pack(v.hi.z, v.hi.w);
            *reinterpret_cast<u32x4*>(out + i) = o;
        }
        return;
    }
    if (jobs.transpose[j] == 1) {   // out[k][n] = in[n][k] * rs[n]: 32 x 32 tiles through LDS, reads coalesced along k, writes along n
        __shared__ float tile[32][33];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
        const int tn = (N + 31) / 32, tk = (K + 31) / 32;
        for (int t = blockIdx.x; t < tn * tk; t += gridDim.x) {
            const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + ty + 8 * r, k = k0 + tx;
                tile[ty + 8 * r][tx] = (n < N && k < K) ? in[(int64_t)n * K + k] * (rs ? rs[n] : 1.f) : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + ty + 8 * r, n = n0 + tx;
                if (k < K && n < N) out[(int64_t)k * N + n] = (bf16_t)(bf_pack(tile[tx][ty + 8 * r], 0.f) & 0xffffu);
            }
            __syncthreads();
        }
        return;
    }
    if (jobs.transpose[j] == 2) {   // dense 3x3: out[oc][tap * Ci + ic] = in[(oc * Ci + ic) * 9 + tap].  A tile = one oc x 64 ic: 576
        __shared__ float buf2[576];   // contiguous floats in, nine runs of 64 bf16 out (the element-per-thread form reads with stride 9)
        const int Ci = K / 9, nch = (Ci + 63) / 64;
        for (int t = blockIdx.x; t < N * nch; t += gridDim.x) {
            const int oc = t / nch, ic0 = (t % nch) * 64;
            const int nic = Ci - ic0 < 64 ? Ci - ic0 : 64;
            const float* src = in + ((int64_t)oc * Ci + ic0) * 9;
            for (int q = threadIdx.x; q < nic * 9; q += 256) buf2[q] = src[q];
            __syncthreads();
            for (int q = threadIdx.x; q < 576; q += 256) {
                const int tap = q >> 6, i = q & 63;
                if (i < nic) out[(int64_t)oc * K + tap * Ci + ic0 + i] = (bf16_t)(bf_pack(buf2[i * 9 + tap], 0.f) & 0xffffu);
            }
            __syncthreads();
        }
        return;
    }
    if (jobs.transpose[j] == 3) {   // its transposed conv: out[ic][tap * Co + oc] = in[(oc * Ci + ic) * 9 + 8 - tap].  A tile = 8 ic x 32 oc:
        __shared__ float buf3[32 * 73];   // 32 runs of 72 contiguous floats in, 72 runs of 32 bf16 out
        const int Ci = K / 9, Co = N, nic = (Ci + 7) / 8, noc = (Co + 31) / 32;
        for (int t = blockIdx.x; t < nic * noc; t += gridDim.x) {
            const int ic0 = (t / noc) * 8, oc0 = (t % noc) * 32;
            for (int q = threadIdx.x; q < 32 * 72; q += 256) {
                const int ol = q / 72, r = q - ol * 72;
                const int oc = oc0 + ol, ic = ic0 + r / 9;
                buf3[ol * 73 + r] = (oc < Co && ic < Ci) ? in[((int64_t)oc * Ci + ic0) * 9 + r] : 0.f;
            }
            __syncthreads();
            for (int q = threadIdx.x; q < 8 * 9 * 32; q += 256) {
                const int ol = q & 31, tap = (q >> 5) % 9, il = q / 288;
                if (oc0 + ol < Co && ic0 + il < Ci)
                    out[(int64_t)(ic0 + il) * (9 * Co) + tap * Co + oc0 + ol] = (bf16_t)(bf_pack(buf3[ol * 73 + il * 9 + 8 - tap], 0.f) & 0xffffu);
            }
            __syncthreads();
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i indexes the OUTPUT
        const int64_t img = i / ((int64_t)N * K);
        const int64_t e = i % ((int64_t)N * K);
        float v;
        if (jobs.transpose[j] == 1) {   // out[k][n] = in[n][k] * rs[n]
            const int k = (int)(e / N), n = (int)(e % N);
            v = in[(int64_t)n * K + k] * (rs ? rs[n] : 1.f);
        } else if (jobs.transpose[j] == 2) {   // dense 3x3, N = Co, K = 9 Ci: out[oc][tap * Ci + ic] = in[(oc * Ci + ic) * 9 + tap]
            const int Ci = K / 9;
            const int oc = (int)(e / K), r = (int)(e % K);
            const int tap = r / Ci, ic = r % Ci;
            v = in[((int64_t)oc * Ci + ic) * 9 + tap];
        } else if (jobs.transpose[j] == 3) {   // its transposed conv: out[ic][tap * Co + oc] = in[(oc * Ci + ic) * 9 + 8 - tap]
            const int Ci = K / 9, Co = N;
            const int ic = (int)(e / (9 * Co)), r = (int)(e % (9 * Co));
            const int tap = r / Co, oc = r % Co;
            v = in[((int64_t)oc * Ci + ic) * 9 + (8 - tap)];
        } else if (jobs.transpose[j] == 4) {   // 2x2 stride-2 conv (misc.hip WP_DOWN): out[oc][ij * C + ic] = in[oc][ic][ij], K = 4 C
            const int Cc = K / 4;
            const int oc = (int)(e / K), r = (int)(e % K);
            const int ij = r / Cc, ic = r % Cc;
            v = in[((int64_t)oc * Cc + ic) * 4 + ij];
        } else if (jobs.transpose[j] == 5) {   // WP_DOWN_T: out[ij * C + ic][oc]
            const int Cc = K / 4;
            const int r = (int)(e / N), oc = (int)(e % N);
            const int ij = r / Cc, ic = r % Cc;
            v = in[((int64_t)oc * Cc + ic) * 4 + ij];
        } else if (jobs.transpose[j] == 6) {   // WP_UP (1x1 conv + PixelShuffle(2)): out[ij * G + kk][ic] = in[4 kk + ij][ic], G = N / 4
            const int G = N / 4;
            const int r = (int)(e / K), ic = (int)(e % K);
            const int ij = r / G, kk = r % G;
            v = in[(int64_t)(4 * kk + ij) * K + ic];
        } else if (jobs.transpose[j] == 7) {   // WP_UP_T: out[ic][ij * G + kk] = in[4 kk + ij][ic]
            const int G = N / 4;
            const int ic = (int)(e / N), r = (int)(e % N);
            const int ij = r / G, kk = r % G;
            v = in[(int64_t)(4 * kk + ij) * K + ic];
        } else {                   // out[img][n][k] = in[n][k] * kscale[img][k]
            const int k = (int)(e % K);
            v = in[e] * (ks ? ks[img * K + k] : 1.f);
        }
        out[i] = (bf16_t)(bf_pack(v, 0.f) & 0xffffu);
    }
}

// out[m][k] = x[m][k] * s[m / P][k]
__global__ __launch_bounds__(256) void scale_rows_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ simg, bf16_t* __restrict__ out,
                                                              int64_t M, int C, int P) {
    const int nq = C / 8;
    const int64_t total = M * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / nq;
        const int q = (int)(i % nq);
        const u32x4 w = *reinterpret_cast<const u32x4*>(x + m * C + 8 * q);
        const f8 sc = f8_ld(simg + (m / P) * C + 8 * q);
        u32x4 o;
        o.x = bf_pack(bf_lo(w.x) * sc.lo.x, bf_hi(w.x) * sc.lo.y);
        o.y = bf_pack(bf_lo(w.y) * sc.lo.z, bf_hi(w.y) * sc.lo.w);
        o.z = bf_pack(bf_lo(w.z) * sc.hi.x, bf_hi(w.z) * sc.hi.y);
        o.w = bf_pack(bf_lo(w.w) * sc.hi.z, bf_hi(w.w) * sc.hi.w);
        *reinterpret_cast<u32x4*>(out + m * C + 8 * q) = o;
    }
}

// part[b][j][k] = sum over the j-th pixel slice of image b of dts * t2   (layout of misc.hip's sca_ds_part_kernel)
__global__ __launch_bounds__(256) void sca_ds_part_bf16_kernel(const bf16_t* __restrict__ dts, const bf16_t* __restrict__ t2, float* __restrict__ part,
                                                               int C, int P, int nslices) {
    __shared__ float red[256 * 8];
    const int b = blockIdx.z, j = blockIdx.y;
    const int nq = C / 8;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    const int tid = threadIdx.x, ql = tid % qb, pl = tid / qb;
    const int q = blockIdx.x * qb + ql;
    const bool qok = q < nq;
    const int per = (P + nslices - 1) / nslices;
    const int pbeg = j * per;
    int pend = pbeg + per;
    if (pend > P) pend = P;
    f8 acc = f8_zero();
    if (qok)
        for (int px = pbeg + pl; px < pend; px += pb) {
            const int64_t o = ((int64_t)b * P + px) * C + 8 * q;
            const u32x4 a = *reinterpret_cast<const u32x4*>(dts + o), c = *reinterpret_cast<const u32x4*>(t2 + o);
            acc.lo = f4_fma(make_float4(bf_lo(a.x), bf_hi(a.x), bf_lo(a.y), bf_hi(a.y)), make_float4(bf_lo(c.x), bf_hi(c.x), bf_lo(c.y), bf_hi(c.y)), acc.lo);
            acc.hi = f4_fma(make_float4(bf_lo(a.z), bf_hi(a.z), bf_lo(a.w), bf_hi(a.w)), make_float4(bf_lo(c.z), bf_hi(c.z), bf_lo(c.w), bf_hi(c.w)), acc.hi);
        }
    *reinterpret_cast<float4*>(&red[tid * 8]) = acc.lo;
    *reinterpret_cast<float4*>(&red[tid * 8 + 4]) = acc.hi;
    __syncthreads();
    if (pl == 0 && qok) {
        float4 s0 = *reinterpret_cast<const float4*>(&red[ql * 8]), s1 = *reinterpret_cast<const float4*>(&red[ql * 8 + 4]);
        for (int i = 1; i < pb; ++i) {
            s0 = f4_add(s0, *reinterpret_cast<const float4*>(&red[(i * qb + ql) * 8]));
            s1 = f4_add(s1, *reinterpret_cast<const float4*>(&red[(i * qb + ql) * 8 + 4]));
        }
        float* dst = part + ((int64_t)b * nslices + j) * C + 8 * q;
        stg4(dst, s0);
        stg4(dst + 4, s1);
    }
}

int grid_for(int64_t items) {
    int64_t g = cdiv64(items, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

int ln_bwd_bf16_num_blocks(int64_t M, int C) {
    const int G = lnb_group(C), gpb = 256 / G;
    int64_t nb = cdiv64(M, gpb);
    if (nb > 1024) nb = 1024;   // column partials: [nblk][2][C] fp32 read back once by the reducer
    if (nb < 1) nb = 1;
    return (int)nb;
}

int launch_ln_fwd_bf16(const bf16_t* x, const float* w, const float* b, bf16_t* y, float* mu, float* rstd, int64_t M, int C, float eps,
                       hipStream_t s) {
    trace_tag("ln_fwd_bf16");
    DCPT_CHECK_ARG(C % 8 == 0 && C <= 1024, "ln_fwd_bf16: C=%d must be a multiple of 8, at most 1024", C);
    const int G = lnb_group(C), gpb = 256 / G;
    int64_t nb = cdiv64(M, gpb);
    if (nb > 8192) nb = 8192;
    if (C / 8 <= G) ln_fwd_bf16_kernel<1><<<dim3((unsigned)nb), dim3(256), 0, s>>>(x, w, b, y, mu, rstd, M, C, eps, G, nullptr, 0);
    else ln_fwd_bf16_kernel<2><<<dim3((unsigned)nb), dim3(256), 0, s>>>(x, w, b, y, mu, rstd, M, C, eps, G, nullptr, 0);
    DCPT_CHECK_LAUNCH("ln_fwd_bf16");
    return DCPT_OK;
}

int launch_ln_act_fwd_bf16(const bf16_t* x, const float* w, const float* b, const bf16_t* res, int relu, bf16_t* y, float* mu, float* rstd,
                           int64_t M, int C, float eps, hipStream_t s) {
    trace_tag("ln_act_fwd_bf16");
    DCPT_CHECK_ARG(C % 8 == 0 && C <= 1024, "ln_act_fwd_bf16: C=%d must be a multiple of 8, at most 1024", C);
    const int G = lnb_group(C), gpb = 256 / G;
    int64_t nb = cdiv64(M, gpb);
    if (nb > 8192) nb = 8192;
    if (C / 8 <= G) ln_fwd_bf16_kernel<1><<<dim3((unsigned)nb), dim3(256), 0, s>>>(x, w, b, y, mu, rstd, M, C, eps, G, res, relu);
    else ln_fwd_bf16_kernel<2><<<dim3((unsigned)nb), dim3(256), 0, s>>>(x, w, b, y, mu, rstd, M, C, eps, G, res, relu);
    DCPT_CHECK_LAUNCH("ln_act_fwd_bf16");
    return DCPT_OK;
}

int launch_ln_bwd_bf16(const bf16_t* gy, const bf16_t* x, const float* mu, const float* rstd, const float* w, const bf16_t* dres, bf16_t* dx,
                       float* part, int nblk, int64_t M, int C, hipStream_t s) {
    trace_tag("ln_bwd_bf16");
    DCPT_CHECK_ARG(C % 8 == 0 && C <= 1024, "ln_bwd_bf16: C=%d must be a multiple of 8, at most 1024", C);
    const int G = lnb_group(C), gpb = 256 / G;
    const int64_t iters = cdiv64(M, (int64_t)nblk * gpb);
    if (C / 8 <= G) ln_bwd_bf16_kernel<1><<<dim3(nblk), dim3(256), 0, s>>>(gy, x, mu, rstd, w, dres, dx, part, M, C, G, iters, nullptr, nullptr);
    else ln_bwd_bf16_kernel<2><<<dim3(nblk), dim3(256), 0, s>>>(gy, x, mu, rstd, w, dres, dx, part, M, C, G, iters, nullptr, nullptr);
    DCPT_CHECK_LAUNCH("ln_bwd_bf16");
    return DCPT_OK;
}

int launch_ln_act_bwd_bf16(const bf16_t* gy, const bf16_t* x, const float* mu, const float* rstd, const float* w, const bf16_t* ymask,
                           bf16_t* gmasked, bf16_t* dx, float* part, int nblk, int64_t M, int C, hipStream_t s) {
    trace_tag("ln_act_bwd_bf16");
    DCPT_CHECK_ARG(C % 8 == 0 && C <= 1024, "ln_act_bwd_bf16: C=%d must be a multiple of 8, at most 1024", C);
    const int G = lnb_group(C), gpb = 256 / G;
    const int64_t iters = cdiv64(M, (int64_t)nblk * gpb);
    if (C / 8 <= G) ln_bwd_bf16_kernel<1><<<dim3(nblk), dim3(256), 0, s>>>(gy, x, mu, rstd, w, nullptr, dx, part, M, C, G, iters, ymask, gmasked);
    else ln_bwd_bf16_kernel<2><<<dim3(nblk), dim3(256), 0, s>>>(gy, x, mu, rstd, w, nullptr, dx, part, M, C, G, iters, ymask, gmasked);
    DCPT_CHECK_LAUNCH("ln_act_bwd_bf16");
    return DCPT_OK;
}

int launch_cast_f32_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t s) {
    DCPT_CHECK_ARG(n % 8 == 0, "cast: element count must be a multiple of 8");
    cast_f2b_kernel<<<dim3(grid_for(n / 8)), dim3(256), 0, s>>>(x, y, n / 8);
    DCPT_CHECK_LAUNCH("cast_f32_bf16");
    return DCPT_OK;
}
int launch_cast_bf16_f32(const bf16_t* x, float* y, int64_t n, hipStream_t s) {
    DCPT_CHECK_ARG(n % 8 == 0, "cast: element count must be a multiple of 8");
    cast_b2f_kernel<<<dim3(grid_for(n / 8)), dim3(256), 0, s>>>(x, y, n / 8);
    DCPT_CHECK_LAUNCH("cast_bf16_f32");
    return DCPT_OK;
}

template <int MAXJ>
static int launch_wpack_bf16_t(const WpackBJobsT<MAXJ>& jobs, hipStream_t s, int grid_cap = 256) {
    DCPT_CHECK_ARG(jobs.n >= 1 && jobs.n <= MAXJ, "wpack_bf16: bad job count");
    int64_t mx = 0;
    for (int j = 0; j < jobs.n; ++j) {
        const int64_t t = (int64_t)(jobs.nimg[j] > 0 ? jobs.nimg[j] : 1) * jobs.N[j] * jobs.K[j];
        if (t > mx) mx = t;
    }
    int g = grid_for(mx);
    if (g > 4096) g = 4096;   // (a tile of the gathered 3 x 3 packs is ~1 us of latency: the largest has 16 384 of them)
    if (MAXJ > WPACKB_MAX_JOBS && g > grid_cap) g = grid_cap;
    wpack_bf16_kernel<MAXJ><<<dim3(g, jobs.n), dim3(256), 0, s>>>(jobs);
    DCPT_CHECK_LAUNCH("wpack_bf16");
    return DCPT_OK;
}
int launch_wpack_bf16(const WpackBJobs& jobs, hipStream_t s) { return launch_wpack_bf16_t(jobs, s); }
int launch_wpack_bf16(const WpackBJobsL& jobs, hipStream_t s, int grid_cap) { return launch_wpack_bf16_t(jobs, s, grid_cap); }

int launch_scale_rows_bf16(const bf16_t* x, const float* simg, bf16_t* out, int64_t M, int C, int P, hipStream_t s) {
    DCPT_CHECK_ARG(C % 8 == 0, "scale_rows_bf16: C=%d", C);
    scale_rows_bf16_kernel<<<dim3(grid_for(M * (C / 8))), dim3(256), 0, s>>>(x, simg, out, M, C, P);
    DCPT_CHECK_LAUNCH("scale_rows_bf16");
    return DCPT_OK;
}

int launch_sca_ds_part_bf16(const bf16_t* dts, const bf16_t* t2, float* ds_part, int B, int C, int P, int nslices, hipStream_t s) {
    DCPT_CHECK_ARG(C % 8 == 0 && B <= 65535, "sca_ds_bf16: C=%d", C);
    const int nq = C / 8;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    sca_ds_part_bf16_kernel<<<dim3(cdiv(nq, qb), nslices, B), dim3(256), 0, s>>>(dts, t2, ds_part, C, P, nslices);
    DCPT_CHECK_LAUNCH("sca_ds_part_bf16");
    return DCPT_OK;
}
