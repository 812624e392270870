// Network-edge 3x3 convolutions between the 3-channel NCHW image and the NHWC feature space
// (reference basicsr/archs/nafnet_arch.py:202-219 intro/ending, :252,:271-272 use).  One side has
// <= 4 channels, so these are bandwidth-bound direct convolutions on the VALU (K = 27 is far too
// small for MFMA).  The image side stays NCHW-contiguous: the layout change to/from NHWC is fused
// into these kernels, so no separate permute pass ever touches the big tensors.
#include "bf16.h"
#include "bf16_ops.h"
#include "bufops.h"
#include "kernels.h"

namespace {

constexpr int MAXCS = 4;

// the big (NHWC feature) side is fp32 or bf16 storage (ST = float / bf16_t); offsets are in bytes
template <typename ST>
__device__ __forceinline__ float4 big_ld4(rsrc_t r, uint32_t off) {
    if constexpr (sizeof(ST) == 4) return buf_ld4(r, off);
    else return bbuf_ld4(r, off);
}
template <typename ST>
__device__ __forceinline__ void big_st4(rsrc_t r, uint32_t off, float4 v) {
    if constexpr (sizeof(ST) == 4) buf_st4(r, off, v);
    else bbuf_st4(r, off, v);
}
constexpr int WGRAD_BLOCKS = 768;   // weight-gradient kernel: one resident round (3 blocks/CU), few partial rows to reduce

// All three kernels share one shape: a thread owns one channel quad of the big side and one pixel column, keeps its
// CS*9 weight (or gradient) float4s in registers and streams down a strip of rows, carrying the three output rows that
// an input row feeds as running accumulators -- so every input element is loaded once per thread, the small-side scalars
// are 3*CS loads per row, and there is no LDS traffic in the loop (the previous version re-read every weight from LDS per
// output and ran at < 1 TB/s).  Loads/stores are range-checked buffer accesses relative to image b: padding and strip
// halos are sentinel offsets, not branches.
struct EdgeMap {
    int QB, PB;     // quads x pixel columns per block (QB * PB = 256)
    int nqc, nwc;   // quad chunks, column chunks
    int strips, RS; // row strips per image, rows per strip
};
inline __host__ __device__ EdgeMap edge_map(int B, int H, int W, int Cb, int target = 2048) {
    EdgeMap m;
    const int nq = Cb / 4;
    int qb = MAXCS;   // at least MAXCS lanes per pixel: lane s of a pixel's group writes small channel s in b2s
    while (qb < nq && qb < 64) qb <<= 1;
    m.QB = qb;
    m.PB = 256 / qb;
    m.nqc = (nq + qb - 1) / qb;
    m.nwc = (W + m.PB - 1) / m.PB;
    // about `target` blocks, strips of at least 8 rows
    int64_t st = target / ((int64_t)B * m.nqc * m.nwc);
    if (st > H / 8) st = H / 8;
    if (st < 1) st = 1;
    m.RS = (int)((H + st - 1) / st);
    m.strips = (H + m.RS - 1) / m.RS;
    return m;
}

struct EdgeBlk {
    int q, x, b, y0, y1;
    bool qok, ok;
};
__device__ __forceinline__ EdgeBlk edge_block(const EdgeMap& m, int H, int W, int Cb) {
    EdgeBlk k;
    const int tid = threadIdx.x;
    const int ql = tid % m.QB, pl = tid / m.QB;
    const int qc = blockIdx.x % m.nqc, wc = blockIdx.x / m.nqc;
    k.q = qc * m.QB + ql;
    k.x = wc * m.PB + pl;
    k.b = blockIdx.z;
    k.y0 = blockIdx.y * m.RS;
    k.y1 = (k.y0 + m.RS < H) ? k.y0 + m.RS : H;
    k.qok = k.q < Cb / 4;
    k.ok = k.qok && k.x < W;
    return k;
}

// small -> big:  y[p][c] = sum_{s,tap} x[p+off(tap)][s] * W(c,s,tap) (+bias[c])      x NCHW [B][CS][H][W], y NHWC
// wmode 0: W = w[c][s][tap] (intro forward);  wmode 1: W = w[s][c][8-tap] (ending dgrad)
template <int CS, typename ST = float>
__global__ __launch_bounds__(256) void conv3x3_s2b_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, ST* __restrict__ y, int B, int H,
                                                          int W, int Cb, int wmode) {
    constexpr uint32_t ES = sizeof(ST);
    const EdgeMap m = edge_map(B, H, W, Cb);
    const EdgeBlk k = edge_block(m, H, W, Cb);
    float4 wv[CS * 9];
#pragma unroll
    for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4 v = f4_zero();
            if (k.qok) {
                float* vp = &v.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * k.q + j;
                    vp[j] = (wmode == 0) ? w[((int64_t)c * CS + s) * 9 + t] : w[((int64_t)s * Cb + c) * 9 + (8 - t)];
                }
            }
            wv[s * 9 + t] = v;
        }
    const float4 bv = (bias && k.qok) ? ldg4(bias + 4 * k.q) : f4_zero();
    const rsrc_t rs_x = make_rsrc(x + (int64_t)k.b * CS * H * W);
    const rsrc_t rs_y = make_rsrc(y + ((int64_t)k.b * H + k.y0) * W * Cb);   // big side: window at the strip's first row
    const uint32_t cl = (k.ok && k.x > 0) ? 0u : COL_SENT, cc = k.ok ? 0u : COL_SENT, cr = (k.ok && k.x + 1 < W) ? 0u : COL_SENT;
    float4 a0 = f4_zero(), a1 = f4_zero();
    for (int r = k.y0 - 1; r <= k.y1; ++r) {
        float4 a2 = f4_zero();
        const bool rin = r >= 0 && r < H;
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const uint32_t o = rin ? (uint32_t)((s * H + r) * W + k.x) * 4u : ROW_SENT;
            const float vl = buf_ld1(rs_x, (o - 4u) | cl), vc = buf_ld1(rs_x, o | cc), vr = buf_ld1(rs_x, (o + 4u) | cr);
            const float4 l4 = make_float4(vl, vl, vl, vl), c4 = make_float4(vc, vc, vc, vc), r4 = make_float4(vr, vr, vr, vr);
            a0 = f4_fma(wv[s * 9 + 6], l4, f4_fma(wv[s * 9 + 7], c4, f4_fma(wv[s * 9 + 8], r4, a0)));   // ky = 2 -> row r-1
            a1 = f4_fma(wv[s * 9 + 3], l4, f4_fma(wv[s * 9 + 4], c4, f4_fma(wv[s * 9 + 5], r4, a1)));   // ky = 1 -> row r
            a2 = f4_fma(wv[s * 9 + 0], l4, f4_fma(wv[s * 9 + 1], c4, f4_fma(wv[s * 9 + 2], r4, a2)));   // ky = 0 -> row r+1
        }
        const int yo = r - 1;
        big_st4<ST>(rs_y, (k.ok && yo >= k.y0) ? ((uint32_t)((yo - k.y0) * W + k.x) * (uint32_t)Cb + 4u * (uint32_t)k.q) * ES : ROW_SENT, f4_add(a0, bv));
        a0 = a1;
        a1 = a2;
    }
}

// big -> small:  y[p][s] = sum_{c,tap} x[p+off(tap)][c] * W(s,c,tap) (+bias[s]) (+res[p][s])     x NHWC, y/res NCHW [B][CS][H][W]
// wmode 0: W = w[s][c][tap] (ending forward);  wmode 1: W = w[c][s][8-tap] (intro dgrad).  One wave holds all quads of a pixel.
template <int CS, typename ST = float>
__global__ __launch_bounds__(256) void conv3x3_b2s_kernel(const ST* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ y, int B, int H, int W, int Cb, int wmode) {
    const EdgeMap m = edge_map(B, H, W, Cb);
    const EdgeBlk k = edge_block(m, H, W, Cb);
    float4 wv[CS * 9];
#pragma unroll
    for (int s = 0; s < CS; ++s)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4 v = f4_zero();
            if (k.qok) {
                float* vp = &v.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * k.q + j;
                    vp[j] = (wmode == 0) ? w[((int64_t)s * Cb + c) * 9 + t] : w[((int64_t)c * CS + s) * 9 + (8 - t)];
                }
            }
            wv[s * 9 + t] = v;
        }
    const int rb = k.y0 > 0 ? k.y0 - 1 : 0;
    const rsrc_t rs_x = make_rsrc(x + ((int64_t)k.b * H + rb) * W * Cb);   // big side: window one row above the strip
    const rsrc_t rs_y = make_rsrc(y + (int64_t)k.b * CS * H * W);
    const rsrc_t rs_r = make_rsrc(res ? res + (int64_t)k.b * CS * H * W : y);
    const uint32_t cl = (k.ok && k.x > 0) ? 0u : COL_SENT, cc = k.ok ? 0u : COL_SENT, cr = (k.ok && k.x + 1 < W) ? 0u : COL_SENT;
    constexpr uint32_t ES = sizeof(ST);
    const uint32_t st = ES * (uint32_t)Cb;
    const int ql = threadIdx.x % m.QB;
    float a0[CS], a1[CS];
#pragma unroll
    for (int s = 0; s < CS; ++s) a0[s] = a1[s] = 0.f;
    for (int r = k.y0 - 1; r <= k.y1; ++r) {
        const uint32_t o = (r >= 0 && r < H) ? ((uint32_t)((r - rb) * W + k.x) * (uint32_t)Cb + 4u * (uint32_t)k.q) * ES : ROW_SENT;
        // (this loop is where the packed-fp32 operand-select problem was found: the whole library is built without packed fp32,
        // dcpt_amd/build.py NO_PACKED_FP32 and DESIGN.md 4h)
        const float4 xl = big_ld4<ST>(rs_x, (o - st) | cl), xc = big_ld4<ST>(rs_x, o | cc), xr = big_ld4<ST>(rs_x, (o + st) | cr);
        float a2[CS];
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            a0[s] += f4_sum(f4_fma(wv[s * 9 + 6], xl, f4_fma(wv[s * 9 + 7], xc, f4_mul(wv[s * 9 + 8], xr))));
            a1[s] += f4_sum(f4_fma(wv[s * 9 + 3], xl, f4_fma(wv[s * 9 + 4], xc, f4_mul(wv[s * 9 + 5], xr))));
            a2[s] = f4_sum(f4_fma(wv[s * 9 + 0], xl, f4_fma(wv[s * 9 + 1], xc, f4_mul(wv[s * 9 + 2], xr))));
        }
        const int yo = r - 1;
        float mine = 0.f;
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const float t = group_sum(a0[s], m.QB);   // over the channel quads of this pixel (consecutive lanes)
            if (ql == s) mine = t;
        }
        if (ql < CS) {
            const uint32_t oo = (k.x < W && yo >= k.y0) ? (uint32_t)((ql * H + yo) * W + k.x) * 4u : ROW_SENT;
            if (bias) mine += bias[ql];
            if (res) mine += buf_ld1(rs_r, oo);
            buf_st1(rs_y, oo, mine);
        }
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            a0[s] = a1[s];
            a1[s] = a2[s];
        }
    }
}

// Weight gradient partials: acc[s*9+tap] (float4 over 4 big channels) += big[p][4q..] * small[p+off][s]
// block partials [nblk][CS*9+1][Cb] (last row: column sum of big), nblk = conv3x3_wgrad_num_blocks()
template <int CS, typename ST = float>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const ST* __restrict__ big, const float* __restrict__ small,
                                                            float* __restrict__ part, int B, int H, int W, int Cb) {
    __shared__ float4 red[256];
    const EdgeMap m = edge_map(B, H, W, Cb, WGRAD_BLOCKS);
    const EdgeBlk k = edge_block(m, H, W, Cb);
    const int tid = threadIdx.x, ql = tid % m.QB, pl = tid / m.QB;
    float4 acc[CS * 9 + 1];
#pragma unroll
    for (int i = 0; i < CS * 9 + 1; ++i) acc[i] = f4_zero();
    const rsrc_t rs_g = make_rsrc(big + ((int64_t)k.b * H + k.y0) * W * Cb);   // big side: window at the strip's first row
    const rsrc_t rs_s = make_rsrc(small + (int64_t)k.b * CS * H * W);
    const uint32_t cl = (k.ok && k.x > 0) ? 0u : COL_SENT, cc = k.ok ? 0u : COL_SENT, cr = (k.ok && k.x + 1 < W) ? 0u : COL_SENT;
    // small-side window: rows r-1, r, r+1 x columns x-1, x, x+1 per small channel, shifted down one row per step
    float sm[CS][3][3];
    auto load_row = [&](int r, int slot) {
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const uint32_t o = (r >= 0 && r < H) ? (uint32_t)((s * H + r) * W + k.x) * 4u : ROW_SENT;
            sm[s][slot][0] = buf_ld1(rs_s, (o - 4u) | cl);
            sm[s][slot][1] = buf_ld1(rs_s, o | cc);
            sm[s][slot][2] = buf_ld1(rs_s, (o + 4u) | cr);
        }
    };
    load_row(k.y0 - 1, 0);
    load_row(k.y0, 1);
    for (int r = k.y0; r < k.y1; ++r) {
        load_row(r + 1, 2);
        const float4 g = big_ld4<ST>(rs_g, k.ok ? ((uint32_t)((r - k.y0) * W + k.x) * (uint32_t)Cb + 4u * (uint32_t)k.q) * (uint32_t)sizeof(ST) : ROW_SENT);
        acc[CS * 9] = f4_add(acc[CS * 9], g);
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = sm[s][ky][kx];
                    acc[s * 9 + ky * 3 + kx] = f4_fma(g, make_float4(v, v, v, v), acc[s * 9 + ky * 3 + kx]);
                }
#pragma unroll
        for (int s = 0; s < CS; ++s)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                sm[s][0][kx] = sm[s][1][kx];
                sm[s][1][kx] = sm[s][2][kx];
            }
    }
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * m.nwc + blockIdx.x / m.nqc;
    float* pp = part + blk * (CS * 9 + 1) * Cb;
#pragma unroll
    for (int i = 0; i < CS * 9 + 1; ++i) {
        __syncthreads();
        red[tid] = acc[i];
        __syncthreads();
        if (pl == 0 && k.qok) {
            float4 s = red[ql];
            for (int j = 1; j < m.PB; ++j) s = f4_add(s, red[j * m.QB + ql]);
            stg4(pp + (int64_t)i * Cb + 4 * k.q, s);
        }
    }
}

// G[c][s][tap] = sum_r part[r][s*9+tap][c]; row Cs*9 is the column sum of big.  block = 16 columns x 16 row groups (fixed order)
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ part, int R, int Cs, int Cb,
                                                                   float* __restrict__ dW, float* __restrict__ bsum,
                                                                   int omode) {
    __shared__ float red[16][16];
    const int j = blockIdx.y;  // 0 .. Cs*9
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int nj = Cs * 9 + 1;
    float sacc = 0.f;
    if (c < Cb) {
#pragma unroll 8
        for (int r = rg; r < R; r += 16) sacc += part[((int64_t)r * nj + j) * Cb + c];
    }
    red[rg][cl] = sacc;
    __syncthreads();
    if (rg == 0 && c < Cb) {
        float v = red[0][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += red[i][cl];
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

}  // namespace

#define EDGE_CHECK(name, TARGET)                                                                                                     \
    DCPT_CHECK_ARG(Cb % 4 == 0 && Cs >= 1 && Cs <= MAXCS && B <= 65535 && (double)H * W * Cs * 4.0 < 1.0e9 &&                   \
                       ((double)edge_map(B, H, W, Cb, TARGET).RS + 3.0) * W * Cb * 4.0 < 1.0e9,                                       \
                   name ": Cs=%d (1..4), Cb=%d (%%4), image %dx%d", Cs, Cb, H, W)
#define EDGE_GO(KERNEL, ...)                                                     \
    do {                                                                         \
        const EdgeMap m = edge_map(B, H, W, Cb);                                 \
        const dim3 grid(m.nqc * m.nwc, m.strips, B);                             \
        if (Cs == 1) KERNEL<1><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);          \
        else if (Cs == 2) KERNEL<2><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);     \
        else if (Cs == 3) KERNEL<3><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);     \
        else KERNEL<4><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);                  \
    } while (0)

#define EDGE_GO_B(KERNEL, ...)                                                            \
    do {                                                                                  \
        const EdgeMap m = edge_map(B, H, W, Cb);                                          \
        const dim3 grid(m.nqc * m.nwc, m.strips, B);                                      \
        if (Cs == 1) KERNEL<1, bf16_t><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);           \
        else if (Cs == 2) KERNEL<2, bf16_t><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);      \
        else if (Cs == 3) KERNEL<3, bf16_t><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);      \
        else KERNEL<4, bf16_t><<<grid, dim3(256), 0, s>>>(__VA_ARGS__);                   \
    } while (0)

// the same three kernels with the feature side in bf16 storage (image side, weights, partial sums fp32)
int launch_conv3x3_s2b_bf16(const float* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int Cs, int Cb, int wmode,
                            hipStream_t s) {
    EDGE_CHECK("conv3x3_s2b_bf16", 2048);
    EDGE_GO_B(conv3x3_s2b_kernel, x, w, bias, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_s2b_bf16");
    return DCPT_OK;
}
int launch_conv3x3_b2s_bf16(const bf16_t* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cs, int Cb,
                            int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_b2s_bf16", 2048);
    DCPT_CHECK_ARG(Cb <= 256, "conv3x3_b2s_bf16: Cb=%d > 256 (one wave must hold all channel quads of a pixel)", Cb);
    EDGE_GO_B(conv3x3_b2s_kernel, x, w, bias, res, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_b2s_bf16");
    return DCPT_OK;
}

int launch_conv3x3_s2b(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cs, int Cb,
                       int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_s2b", 2048);
    EDGE_GO(conv3x3_s2b_kernel, x, w, bias, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_s2b");
    return DCPT_OK;
}

int launch_conv3x3_b2s(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W,
                       int Cs, int Cb, int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_b2s", 2048);
    DCPT_CHECK_ARG(Cb <= 256, "conv3x3_b2s: Cb=%d > 256 (one wave must hold all channel quads of a pixel)", Cb);
    EDGE_GO(conv3x3_b2s_kernel, x, w, bias, res, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_b2s");
    return DCPT_OK;
}

int conv3x3_wgrad_num_blocks(int B, int H, int W, int Cb) {
    const EdgeMap m = edge_map(B, H, W, Cb, WGRAD_BLOCKS);
    return B * m.strips * m.nwc;
}

int launch_conv3x3_wgrad_bf16(const bf16_t* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H, int W, int Cs,
                              int Cb, int omode, hipStream_t s) {
    EDGE_CHECK("conv3x3_wgrad_bf16", WGRAD_BLOCKS);
    DCPT_CHECK_ARG(nblk == conv3x3_wgrad_num_blocks(B, H, W, Cb), "conv3x3_wgrad_bf16: nblk mismatch");
    {
        const EdgeMap m = edge_map(B, H, W, Cb, WGRAD_BLOCKS);
        const dim3 grid(m.nqc * m.nwc, m.strips, B);
        if (Cs == 1) conv3x3_wgrad_kernel<1, bf16_t><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else if (Cs == 2) conv3x3_wgrad_kernel<2, bf16_t><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else if (Cs == 3) conv3x3_wgrad_kernel<3, bf16_t><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else conv3x3_wgrad_kernel<4, bf16_t><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
    }
    DCPT_CHECK_LAUNCH("conv3x3_wgrad_bf16");
    conv3x3_wgrad_reduce_kernel<<<dim3(cdiv(Cb, 16), Cs * 9 + 1), dim3(256), 0, s>>>(part, nblk, Cs, Cb, dW, bsum, omode);
    DCPT_CHECK_LAUNCH("conv3x3_wgrad_reduce");
    return DCPT_OK;
}

int launch_conv3x3_wgrad(const float* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H,
                         int W, int Cs, int Cb, int omode, hipStream_t s) {
    EDGE_CHECK("conv3x3_wgrad", WGRAD_BLOCKS);
    DCPT_CHECK_ARG(nblk == conv3x3_wgrad_num_blocks(B, H, W, Cb), "conv3x3_wgrad: nblk mismatch");
    {
        const EdgeMap m = edge_map(B, H, W, Cb, WGRAD_BLOCKS);
        const dim3 grid(m.nqc * m.nwc, m.strips, B);
        if (Cs == 1) conv3x3_wgrad_kernel<1><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else if (Cs == 2) conv3x3_wgrad_kernel<2><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else if (Cs == 3) conv3x3_wgrad_kernel<3><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
        else conv3x3_wgrad_kernel<4><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, Cb);
    }
    DCPT_CHECK_LAUNCH("conv3x3_wgrad");
    conv3x3_wgrad_reduce_kernel<<<dim3(cdiv(Cb, 16), Cs * 9 + 1), dim3(256), 0, s>>>(part, nblk, Cs, Cb, dW, bsum, omode);
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
