// Per-pixel channel LayerNorm on NHWC rows (reference basicsr/archs/nafnet_arch.py:25-64).
//
// A group of G lanes (G = power of two >= C/4, at most 64) owns one pixel row; lane `lig` holds the
// float4 quads lig, lig+G, ...  Row reductions are XOR butterflies over the G lanes (no LDS).
// Statistics are the reference's two-pass biased mean/variance; 1/sqrt(var+eps) is stored as rstd.
#include "kernels.h"

namespace {

__host__ __device__ inline int ln_group(int C) {
    const int q = C / 4;
    int g = 1;
    while (g < q && g < 64) g <<= 1;
    return g;
}

// NQ = float4 quads per lane (C <= 16 * G * NQ / 4): the row is loaded ONCE into registers and reused by the mean pass, the
// centred-variance pass and the normalisation.
template <int NQ>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ mu,
                                                       float* __restrict__ rstd, int64_t M, int C, float eps, int G,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ y, const float* __restrict__ res, int relu) {
    const int tid = threadIdx.x;
    const int gpb = 256 / G;
    const int lig = tid % G;
    const int nq = C / 4;
    for (int64_t rb = blockIdx.x; rb * gpb < M; rb += gridDim.x) {
    const int64_t row = rb * gpb + tid / G;
    const bool valid = row < M;
    const float* xr = x + (valid ? row : 0) * (int64_t)C;
    float4 v[NQ];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = lig + i * G;
        v[i] = (valid && q < nq) ? ldg4(xr + 4 * q) : f4_zero();
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) sum += f4_sum(v[i]);
    sum = group_sum(sum, G);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (lig + i * G < nq) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    sq = group_sum(sq, G);
    const float var = sq / (float)C;
    const float rs = 1.0f / sqrtf(var + eps);
    if (valid && lig == 0) {
        mu[row] = mean;
        rstd[row] = rs;
    }
    if (y != nullptr && valid) {
        float* yr = y + row * (int64_t)C;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lig + i * G;
            if (q >= nq) continue;
            const float4 ww = ldg4(w + 4 * q);
            float4 o;
            if (b != nullptr) {
                const float4 bb = ldg4(b + 4 * q);
                o.x = fmaf((v[i].x - mean) * rs, ww.x, bb.x);
                o.y = fmaf((v[i].y - mean) * rs, ww.y, bb.y);
                o.z = fmaf((v[i].z - mean) * rs, ww.z, bb.z);
                o.w = fmaf((v[i].w - mean) * rs, ww.w, bb.w);
            } else {  // Restormer BiasFree_LayerNorm (restormer_arch.py:36-40): x / sqrt(var + eps) * w, numerator not centred
                o = make_float4(v[i].x * rs * ww.x, v[i].y * rs * ww.y, v[i].z * rs * ww.z, v[i].w * rs * ww.w);
            }
            if (res) o = f4_add(o, ldg4(res + row * (int64_t)C + 4 * q));
            if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            stg4(yr + 4 * q, o);
        }
    }
    }
}

struct LnBwdP {
    const float* gy;
    const float* x;
    const float* mu;
    const float* rstd;
    const float* w;
    const float* dres;
    const float* ymask;  // optional: zero the incoming gradient where this (post-ReLU) tensor is <= 0
    float* gmasked;      // optional: the masked incoming gradient is also written here (residual branch)
    float* dx;
    float* part;  // [nblk][3][C]
    int64_t M;
    int C, G;
    int64_t iters;
    int biasfree;  // Restormer BiasFree_LayerNorm: y = x * rstd * w (variance about the mean, numerator not centred)
};

template <int NQ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdP p) {
    __shared__ float red[1024 * NQ];  // (256/G) row groups x C columns, C <= 4*G*NQ
    const int tid = threadIdx.x, G = p.G, gpb = 256 / G, gid = tid / G, lig = tid % G;
    const int nq = p.C / 4;
    float4 w[NQ], aw[NQ], ab[NQ], ad[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = lig + j * G;
        w[j] = (q < nq) ? ldg4(p.w + 4 * q) : f4_zero();
        aw[j] = f4_zero();
        ab[j] = f4_zero();
        ad[j] = f4_zero();
    }
    const float invC = 1.0f / (float)p.C;
    for (int64_t it = 0; it < p.iters; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * gpb + gid;
        const bool valid = row < p.M;
        const int64_t ro = (valid ? row : 0) * (int64_t)p.C;
        const float mean = valid ? p.mu[row] : 0.f, rs = valid ? p.rstd[row] : 0.f;
        float4 g[NQ], xh[NQ], dr[NQ];
        float s1 = 0.f, s2 = 0.f;
        // all loads of the row are issued together (the residual gradient too: it used to be fetched after the row
        // reductions, a second full memory round trip per row)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = lig + j * G;
            dr[j] = (p.dres && valid && q < nq) ? ldg4(p.dres + ro + 4 * q) : f4_zero();
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = lig + j * G;
            if (valid && q < nq) {
                g[j] = ldg4(p.gy + ro + 4 * q);
                if (p.ymask) {
                    const float4 ym = ldg4(p.ymask + ro + 4 * q);
                    g[j] = make_float4(ym.x > 0.f ? g[j].x : 0.f, ym.y > 0.f ? g[j].y : 0.f, ym.z > 0.f ? g[j].z : 0.f,
                                       ym.w > 0.f ? g[j].w : 0.f);
                }
                if (p.gmasked) stg4(p.gmasked + ro + 4 * q, g[j]);
                const float4 xv = ldg4(p.x + ro + 4 * q);
                xh[j] = make_float4((xv.x - mean) * rs, (xv.y - mean) * rs, (xv.z - mean) * rs, (xv.w - mean) * rs);
            } else {
                g[j] = f4_zero();
                xh[j] = f4_zero();
            }
            const float4 gw = f4_mul(g[j], w[j]);
            if (p.biasfree) {
                // s2 = mean_c(gw * x * rstd);  x*rstd = xhat + mu*rstd
                const float mr = mean * rs;
                const float4 xr = make_float4(xh[j].x + mr, xh[j].y + mr, xh[j].z + mr, xh[j].w + mr);
                s2 += (valid && q < nq) ? f4_sum(f4_mul(gw, xr)) : 0.f;
            } else {
                s1 += f4_sum(gw);
                s2 += f4_sum(f4_mul(gw, xh[j]));
            }
        }
        s1 = group_sum(s1, G) * invC;  // mean_c(g)            (0 for BiasFree)
        s2 = group_sum(s2, G) * invC;  // mean_c(g * xhat)     (BiasFree: mean_c(g * x * rstd))
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = lig + j * G;
            if (valid && q < nq) {
                const float4 gw = f4_mul(g[j], w[j]);
                float4 d;
                d.x = rs * (gw.x - xh[j].x * s2 - s1);
                d.y = rs * (gw.y - xh[j].y * s2 - s1);
                d.z = rs * (gw.z - xh[j].z * s2 - s1);
                d.w = rs * (gw.w - xh[j].w * s2 - s1);
                d = f4_add(d, dr[j]);
                stg4(p.dx + ro + 4 * q, d);
                if (p.biasfree) {
                    const float mr = mean * rs;
                    aw[j] = f4_fma(g[j], make_float4(xh[j].x + mr, xh[j].y + mr, xh[j].z + mr, xh[j].w + mr), aw[j]);
                } else {
                    aw[j] = f4_fma(g[j], xh[j], aw[j]);
                }
                ab[j] = f4_add(ab[j], g[j]);
                ad[j] = f4_add(ad[j], d);
            }
        }
    }
    // deterministic block reduction of the three column accumulators
    float* part = p.part + (int64_t)blockIdx.x * 3 * p.C;
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = lig + j * G;
            if (q < nq) {
                const float4 v = (which == 0) ? aw[j] : (which == 1) ? ab[j] : ad[j];
                *reinterpret_cast<float4*>(&red[gid * p.C + 4 * q]) = v;
            }
        }
        __syncthreads();
        for (int c = tid; c < p.C; c += 256) {
            float s = 0.f;
            for (int gI = 0; gI < gpb; ++gI) s += red[gI * p.C + c];
            part[which * p.C + c] = s;
        }
    }
}

// out_j[c] = sum_r part[r][j][c]; block = 16 columns x 16 row groups (fixed order: deterministic)
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, int R, int nj, int C,
                                                             float* out0, float* out1, float* out2) {
    __shared__ float red[16][16];
    const int j = blockIdx.y;
    float* out = (j == 0) ? out0 : (j == 1) ? out1 : out2;
    if (out == nullptr) return;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int r = rg; r < R; r += 16) s += part[((int64_t)r * nj + j) * C + c];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i][cl];
        out[c] = t;
    }
}

}  // namespace

static int launch_ln_rows(const float* x, float* mu, float* rstd, int64_t M, int C, float eps, const float* w, const float* b,
                          float* y, const float* res, int relu, hipStream_t s) {
    DCPT_CHECK_ARG(C <= 2048, "LayerNorm: C=%d > 2048", C);
    const int G = ln_group(C), gpb = 256 / G;
    const int nqpl = cdiv(C / 4, G);
    const dim3 grid((unsigned)cdiv64(M, gpb));
    if (nqpl <= 1) ln_stats_kernel<1><<<grid, dim3(256), 0, s>>>(x, mu, rstd, M, C, eps, G, w, b, y, res, relu);
    else if (nqpl <= 2) ln_stats_kernel<2><<<grid, dim3(256), 0, s>>>(x, mu, rstd, M, C, eps, G, w, b, y, res, relu);
    else if (nqpl <= 4) ln_stats_kernel<4><<<grid, dim3(256), 0, s>>>(x, mu, rstd, M, C, eps, G, w, b, y, res, relu);
    else ln_stats_kernel<8><<<grid, dim3(256), 0, s>>>(x, mu, rstd, M, C, eps, G, w, b, y, res, relu);
    DCPT_CHECK_LAUNCH("ln_rows");
    return DCPT_OK;
}

int launch_ln_stats(const float* x, float* mu, float* rstd, int64_t M, int C, float eps, hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && C > 0 && M > 0, "ln_stats: C=%d must be a positive multiple of 4", C);
    DCPT_TRY(launch_ln_rows(x, mu, rstd, M, C, eps, nullptr, nullptr, nullptr, nullptr, 0, s));
    return DCPT_OK;
}

int launch_ln_act_fwd(const float* x, const float* w, const float* b, const float* res, int relu, float* y, float* mu,
                      float* rstd, int64_t M, int C, float eps, hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && C > 0 && M > 0, "ln_fwd: C=%d must be a positive multiple of 4", C);
    DCPT_TRY(launch_ln_rows(x, mu, rstd, M, C, eps, w, b, y, res, relu, s));
    return DCPT_OK;
}

int launch_ln_fwd(const float* x, const float* w, const float* b, float* y, float* mu, float* rstd, int64_t M, int C,
                  float eps, hipStream_t s) {
    return launch_ln_act_fwd(x, w, b, nullptr, 0, y, mu, rstd, M, C, eps, s);
}

int ln_bwd_num_blocks(int64_t M, int C) {
    const int G = ln_group(C), gpb = 256 / G;
    int64_t nb = cdiv64(M, gpb);
    if (nb > 1024) nb = 1024;  // HBM-bound: keep >= 4 blocks per CU in flight
    return (int)nb;
}

int launch_ln_bwd(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                  float* dx, float* part, int nblk, int64_t M, int C, hipStream_t s) {
    return launch_ln_act_bwd(gy, x, mu, rstd, w, dres, nullptr, nullptr, dx, part, nblk, M, C, s);
}

int launch_ln_act_bwd(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                      const float* ymask, float* gmasked, float* dx, float* part, int nblk, int64_t M, int C, hipStream_t s) {
    return launch_ln_bwd_ex(gy, x, mu, rstd, w, dres, ymask, gmasked, 0, dx, part, nblk, M, C, s);
}

int launch_ln_bwd_ex(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                     const float* ymask, float* gmasked, int biasfree, float* dx, float* part, int nblk, int64_t M, int C,
                     hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && C > 0 && C <= 2048, "ln_bwd: C=%d must be a multiple of 4, <= 2048", C);
    LnBwdP p;
    p.biasfree = biasfree;
    p.ymask = ymask; p.gmasked = gmasked;
    p.gy = gy; p.x = x; p.mu = mu; p.rstd = rstd; p.w = w; p.dres = dres; p.dx = dx; p.part = part;
    p.M = M; p.C = C; p.G = ln_group(C);
    const int gpb = 256 / p.G;
    p.iters = cdiv64(M, (int64_t)nblk * gpb);
    const int nqpl = cdiv(C / 4, p.G);
    if (nqpl <= 1) ln_bwd_kernel<1><<<dim3(nblk), dim3(256), 0, s>>>(p);
    else if (nqpl <= 2) ln_bwd_kernel<2><<<dim3(nblk), dim3(256), 0, s>>>(p);
    else if (nqpl <= 4) ln_bwd_kernel<4><<<dim3(nblk), dim3(256), 0, s>>>(p);
    else ln_bwd_kernel<8><<<dim3(nblk), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ln_bwd");
    return DCPT_OK;
}

int launch_colpart_reduce(const float* part, int R, int nj, int C, float* out0, float* out1, float* out2, hipStream_t s) {
    DCPT_CHECK_ARG(nj >= 1 && nj <= 3, "colpart_reduce: nj=%d", nj);
    colpart_reduce_kernel<<<dim3(cdiv(C, 16), nj), dim3(256), 0, s>>>(part, R, nj, C, out0, out1, out2);
    DCPT_CHECK_LAUNCH("colpart_reduce");
    return DCPT_OK;
}
