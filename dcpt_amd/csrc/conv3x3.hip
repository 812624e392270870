// Network-edge 3x3 convolutions between the 3-channel NCHW image and the NHWC feature space
// (reference basicsr/archs/nafnet_arch.py:202-219 intro/ending, :252,:271-272 use).  One side has
// <= 4 channels, so these are bandwidth-bound direct convolutions on the VALU (K = 27 is far too
// small for MFMA).  The image side stays NCHW-contiguous: the layout change to/from NHWC is fused
// into these kernels, so no separate permute pass ever touches the big tensors.
#include "kernels.h"

namespace {

constexpr int MAXCS = 4;

// y[p][c] = sum_{s,tap} x[p+off(tap)][s] * W(c,s,tap) (+bias[c]);  thread = (pixel, channel quad)
__global__ __launch_bounds__(256) void conv3x3_s2b_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                          int W, int Cs, int Cb, int wmode) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cs*9][Cb]
    const int tid = threadIdx.x;
    for (int i = tid; i < Cs * 9 * Cb; i += 256) {
        const int st = i / Cb, c = i % Cb;
        const int s = st / 9, tap = st % 9;
        wl[i] = (wmode == 0) ? w[((int64_t)c * Cs + s) * 9 + tap] : w[((int64_t)s * Cb + c) * 9 + (8 - tap)];
    }
    __syncthreads();
    const int nq = Cb / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + tid;
    const int64_t npix = (int64_t)B * H * W;
    if (idx >= npix * nq) return;
    const int q = (int)(idx % nq);
    const int64_t pix = idx / nq;
    const int xw = (int)(pix % W);
    const int64_t t = pix / W;
    const int yh = (int)(t % H);
    const int64_t b = t / H;
    float4 acc = bias ? ldg4(bias + 4 * q) : f4_zero();
    for (int s = 0; s < Cs; ++s) {
        const float* xs = x + ((b * Cs + s) * H) * (int64_t)W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = yh - 1 + ky;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xw - 1 + kx;
                if (xx < 0 || xx >= W) continue;
                const float v = xs[(int64_t)yy * W + xx];
                const float4 wv = *reinterpret_cast<const float4*>(&wl[(s * 9 + ky * 3 + kx) * Cb + 4 * q]);
                acc = f4_fma(make_float4(v, v, v, v), wv, acc);
            }
        }
    }
    stg4(y + pix * Cb + 4 * q, acc);
}

// y[p][s] = sum_{c,tap} x[p+off(tap)][c] * W(s,c,tap) (+bias[s]) (+res[p][s]);  G lanes per pixel
template <int CS>
__global__ __launch_bounds__(256) void conv3x3_b2s_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ y, int B, int H, int W, int Cb, int wmode, int G,
                                                          int64_t iters) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [CS*9][Cb]
    const int tid = threadIdx.x;
    for (int i = tid; i < CS * 9 * Cb; i += 256) {
        const int st = i / Cb, c = i % Cb;
        const int s = st / 9, tap = st % 9;
        wl[i] = (wmode == 0) ? w[((int64_t)s * Cb + c) * 9 + tap] : w[((int64_t)c * CS + s) * 9 + (8 - tap)];
    }
    __syncthreads();
    const int lig = tid % G, gid = tid / G, gpb = 256 / G;
    const int nq = Cb / 4;
    const int64_t npix = (int64_t)B * H * W;
    float out_acc[CS];
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t pix = (it * gridDim.x + blockIdx.x) * gpb + gid;
        const bool valid = pix < npix;
        const int64_t pp = valid ? pix : 0;
        const int xw = (int)(pp % W);
        const int64_t t = pp / W;
        const int yh = (int)(t % H);
        const int64_t b = t / H;
#pragma unroll
        for (int s = 0; s < CS; ++s) out_acc[s] = 0.f;
        for (int q = lig; q < nq; q += G) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = yh - 1 + ky;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = xw - 1 + kx;
                    if (!valid || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const float4 v = ldg4(x + ((b * H + yy) * (int64_t)W + xx) * Cb + 4 * q);
                    const int tap = ky * 3 + kx;
#pragma unroll
                    for (int s = 0; s < CS; ++s) {
                        const float4 wv = *reinterpret_cast<const float4*>(&wl[(s * 9 + tap) * Cb + 4 * q]);
                        out_acc[s] += f4_sum(f4_mul(v, wv));
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < CS; ++s) out_acc[s] = group_sum(out_acc[s], G);
        if (valid && lig < CS) {
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < CS; ++s)
                if (s == lig) v = out_acc[s];
            const int64_t o = ((b * CS + lig) * H + yh) * (int64_t)W + xw;
            if (bias) v += bias[lig];
            if (res) v += res[o];
            y[o] = v;
        }
    }
}

// Weight gradient partials: acc[s*9+tap] (float4 over 4 big channels) += big[p][4q..] * small[p+off][s]
// thread = (pixel-slot, channel quad); block partials [nblk][Cs*9+1][Cb] (last row: column sum of big)
template <int CS>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ big, const float* __restrict__ small,
                                                            float* __restrict__ part, int B, int H, int W, int Cb,
                                                            int64_t iters) {
    __shared__ float4 red[256];
    const int nq = Cb / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    const int tid = threadIdx.x, ql = tid % qb, pl = tid / qb;
    const int q = blockIdx.y * qb + ql;
    const bool qok = q < nq;
    const int64_t npix = (int64_t)B * H * W;
    float4 acc[CS * 9 + 1];
#pragma unroll
    for (int i = 0; i < CS * 9 + 1; ++i) acc[i] = f4_zero();
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t pix = (it * gridDim.x + blockIdx.x) * pb + pl;
        if (!qok || pix >= npix) continue;
        const int xw = (int)(pix % W);
        const int64_t t = pix / W;
        const int yh = (int)(t % H);
        const int64_t b = t / H;
        const float4 g = ldg4(big + pix * Cb + 4 * q);
        acc[CS * 9] = f4_add(acc[CS * 9], g);
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const float* xs = small + ((b * CS + s) * H) * (int64_t)W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = yh - 1 + ky;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = xw - 1 + kx;
                    float v = 0.f;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = xs[(int64_t)yy * W + xx];
                    acc[s * 9 + ky * 3 + kx] = f4_fma(g, make_float4(v, v, v, v), acc[s * 9 + ky * 3 + kx]);
                }
            }
        }
    }
    float* pp = part + (int64_t)blockIdx.x * (CS * 9 + 1) * Cb;
#pragma unroll
    for (int i = 0; i < CS * 9 + 1; ++i) {
        __syncthreads();
        red[tid] = acc[i];
        __syncthreads();
        if (pl == 0 && qok) {
            float4 s = red[ql];
            for (int j = 1; j < pb; ++j) s = f4_add(s, red[j * qb + ql]);
            stg4(pp + (int64_t)i * Cb + 4 * q, s);
        }
    }
}

// G[c][s][tap] = sum_r part[r][s*9+tap][c]; row Cs*9 is the column sum of big
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ part, int R, int Cs, int Cb,
                                                                   float* __restrict__ dW, float* __restrict__ bsum,
                                                                   int omode) {
    __shared__ float red[4][64];
    const int j = blockIdx.y;  // 0 .. Cs*9
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int nj = Cs * 9 + 1;
    float sacc = 0.f;
    if (c < Cb)
        for (int r = rg; r < R; r += 4) sacc += part[((int64_t)r * nj + j) * Cb + c];
    red[rg][cl] = sacc;
    __syncthreads();
    if (rg == 0 && c < Cb) {
        const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (j == Cs * 9) {
            if (bsum) bsum[c] = v;
        } else {
            const int s = j / 9, tap = j % 9;
            if (omode == 0) dW[((int64_t)c * Cs + s) * 9 + tap] = v;
            else dW[((int64_t)s * Cb + c) * 9 + (8 - tap)] = v;
        }
    }
}

// stage 1: part[c][j] = sum over the j-th slice of (b, hw);  stage 2 (gridDim.y == 1 launch): out[c] = sum_j part[c][j]
__global__ __launch_bounds__(256) void nchw_channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                               int C, int HW, int nsl, int stage) {
    __shared__ float red[256];
    const int c = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    float s = 0.f;
    if (stage == 0) {
        const int64_t tot = (int64_t)B * HW;
        const int64_t per = (tot + nsl - 1) / nsl;
        const int64_t beg = j * per;
        int64_t end = beg + per;
        if (end > tot) end = tot;
        for (int64_t i = beg + tid; i < end; i += 256) {
            const int64_t b = i / HW, p = i - b * HW;
            s += x[(b * C + c) * HW + p];
        }
    } else {
        for (int i = tid; i < nsl; i += 256) s += x[(int64_t)c * nsl + i];
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) out[(stage == 0) ? (int64_t)c * nsl + j : c] = red[0];
}

// tiled transposes between [B][C][HW] and [B][HW][C]
__global__ __launch_bounds__(256) void layout_transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R,
                                                               int Cc) {
    // per batch: in [R][Cc] -> out [Cc][R]
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* xi = x + (int64_t)b * R * Cc;
    float* yo = y + (int64_t)b * R * Cc;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? xi[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) yo[(int64_t)c * R + r] = tile[tx][i];
    }
}

inline int pow2_group(int nq) {
    int g = 1;
    while (g < nq && g < 64) g <<= 1;
    return g;
}

}  // namespace

int launch_conv3x3_s2b(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cs, int Cb,
                       int wmode, hipStream_t s) {
    DCPT_CHECK_ARG(Cb % 4 == 0 && Cs >= 1 && Cs * 9 * Cb * 4 <= 65536, "conv3x3_s2b: Cs=%d Cb=%d unsupported", Cs, Cb);
    const int64_t n = (int64_t)B * H * W * (Cb / 4);
    conv3x3_s2b_kernel<<<dim3((unsigned)cdiv64(n, 256)), dim3(256), Cs * 9 * Cb * sizeof(float), s>>>(x, w, bias, y, B, H, W,
                                                                                                       Cs, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_s2b");
    return DCPT_OK;
}

int launch_conv3x3_b2s(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W,
                       int Cs, int Cb, int wmode, hipStream_t s) {
    DCPT_CHECK_ARG(Cb % 4 == 0 && Cs >= 1 && Cs <= MAXCS && Cs * 9 * Cb * 4 <= 65536, "conv3x3_b2s: Cs=%d (<=4) Cb=%d (%%4)", Cs, Cb);
    const int G = pow2_group(Cb / 4) < MAXCS ? MAXCS : pow2_group(Cb / 4);
    const int gpb = 256 / G;
    const int64_t npix = (int64_t)B * H * W;
    int64_t nblk = cdiv64(npix, gpb);
    if (nblk > 8192) nblk = 8192;
    const int64_t iters = cdiv64(npix, nblk * gpb);
#define GO(CS) conv3x3_b2s_kernel<CS><<<dim3((unsigned)nblk), dim3(256), CS * 9 * Cb * sizeof(float), s>>>(x, w, bias, res, y, B, H, W, Cb, wmode, G, iters)
    if (Cs == 1) GO(1);
    else if (Cs == 2) GO(2);
    else if (Cs == 3) GO(3);
    else GO(4);
#undef GO
    DCPT_CHECK_LAUNCH("conv3x3_b2s");
    return DCPT_OK;
}

int conv3x3_wgrad_num_blocks(int B, int H, int W, int Cb) {
    const int nq = Cb / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    int64_t nblk = cdiv64((int64_t)B * H * W, (int64_t)pb * 64);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

int launch_conv3x3_wgrad(const float* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H,
                         int W, int Cs, int Cb, int omode, hipStream_t s) {
    DCPT_CHECK_ARG(Cb % 4 == 0 && Cs >= 1 && Cs <= MAXCS, "conv3x3_wgrad: Cs=%d (<=4) Cb=%d (%%4)", Cs, Cb);
    const int nq = Cb / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    const int64_t iters = cdiv64((int64_t)B * H * W, (int64_t)nblk * pb);
#define GO(CS) conv3x3_wgrad_kernel<CS><<<dim3(nblk, cdiv(nq, qb)), dim3(256), 0, s>>>(big, small, part, B, H, W, Cb, iters)
    if (Cs == 1) GO(1);
    else if (Cs == 2) GO(2);
    else if (Cs == 3) GO(3);
    else GO(4);
#undef GO
    DCPT_CHECK_LAUNCH("conv3x3_wgrad");
    conv3x3_wgrad_reduce_kernel<<<dim3(cdiv(Cb, 64), Cs * 9 + 1), dim3(256), 0, s>>>(part, nblk, Cs, Cb, dW, bsum, omode);
    DCPT_CHECK_LAUNCH("conv3x3_wgrad_reduce");
    return DCPT_OK;
}

int launch_nchw_channel_sum(const float* x, float* part, float* out, int B, int C, int HW, hipStream_t s) {
    nchw_channel_sum_kernel<<<dim3(C, 128), dim3(256), 0, s>>>(x, part, B, C, HW, 128, 0);
    DCPT_CHECK_LAUNCH("nchw_channel_sum");
    nchw_channel_sum_kernel<<<dim3(C, 1), dim3(256), 0, s>>>(part, out, B, C, HW, 128, 1);
    DCPT_CHECK_LAUNCH("nchw_channel_sum2");
    return DCPT_OK;
}

int launch_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, hipStream_t s) {
    // in [C][HW] -> out [HW][C]
    DCPT_CHECK_ARG(B <= 65535, "nchw_to_nhwc: B too large");
    layout_transpose_kernel<<<dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, s>>>(x, y, C, HW);
    DCPT_CHECK_LAUNCH("nchw_to_nhwc");
    return DCPT_OK;
}

int launch_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, hipStream_t s) {
    DCPT_CHECK_ARG(B <= 65535, "nhwc_to_nchw: B too large");
    layout_transpose_kernel<<<dim3(cdiv(C, 32), cdiv(HW, 32), B), dim3(256), 0, s>>>(x, y, HW, C);
    DCPT_CHECK_LAUNCH("nhwc_to_nchw");
    return DCPT_OK;
}
