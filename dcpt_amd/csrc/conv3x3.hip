// Network-edge 3x3 convolutions between the 3-channel NCHW image and the NHWC feature space
// (reference basicsr/archs/nafnet_arch.py:202-219 intro/ending, :252,:271-272 use).  One side has
// <= 4 channels.  The image side stays NCHW-contiguous: the layout change to/from NHWC is fused
// into these kernels, so no separate permute pass ever touches the big tensors.
// Two forms of each of the three kernels: direct convolutions on the VALU (any Cs <= 4, Cb % 4 == 0 -- first half of the file; 27 FMAs
// per feature element make them VALU-bound at 2.5 - 7 x their traffic time) and, for the shapes of every NAFNet configuration (Cs <= 3,
// Cb = 32 / 64), exact-fp32 MFMA kernels (second half, round 5: K = 27 is small for a GEMM tile, not for v_mfma_f32_32x32x2_f32).
#include "bf16.h"
#include "prof.h"
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


// ================================================================================================================================
// MFMA forms of the three edge kernels (round 5).  The VALU kernels above spend 27 FMAs per feature element (K = Cs * 9 taps) and, with packed
// fp32 off (build.py), run at 2.5 - 4 x their HBM time: 237 / 409 / 256 us (s2b / b2s / wgrad, B = 32, 256 x 256, Cb = 64, fp32) against
// ~107 us of traffic each.  The same sums as exact-fp32 matrix products on v_mfma_f32_32x32x2_f32 (lane l: A[i = l & 31][k = l >> 5],
// B[k = l >> 5][j = l & 31], D[i = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][j = l & 31]) take 7.5 - 9.7 GF = 48 - 62 us of the matrix pipe:
//   s2b    y[p][c]  = sum_k S[p][k] Wk[k][c],  S[p][k = s * 9 + tap] = x[s][p + off(tap)]  (A = S: one image scalar per lane and step, the
//          nine-fold reuse is the L1's; B = the 14 x Cb/32 weight fragments, resident in registers).  A wave owns 32 consecutive pixels of
//          a row; a lane's channels are c = NT (l & 31) + nt, so its NT accumulators of a pixel are adjacent in memory (one store each).
//   wgrad  D[c][j]  = sum_p big[p][c] S[p][j],  S[p][Cs * 9] = 1 (the column sum of big = the bias gradient rides as column 27).
//          A = big read as 128-byte runs (32 channels of a pixel, the lane halves take neighbouring pixels), B = S as above.
//   b2s    Z[q][j = s * 9 + tap] = sum_c x[q][c] W[s][c][tap] at the UNSHIFTED pixel q (a plain GEMM over the channels: A = 16-byte pieces
//          of a pixel's row, the contraction order permuted to match), then y[p][s] = sum_tap Z[p + off(tap)][(s, tap)] gathered from a
//          wave-private three-row ring of Z in LDS.  A wave owns a band of 32 Z columns = 30 output columns (the edge columns are
//          recomputed by the neighbouring band: no block-level synchronisation at all) and streams down a strip of rows.
// Taken for Cs <= 3 and Cb = 32 / 64 (every NAFNet / head configuration of the repo); anything else keeps the VALU kernels.  DCPT_EDGE_MFMA=0
// switches back (A/B and parity reference).  All three are run-to-run deterministic (fixed assignment, fixed summation order).
constexpr int EM_NW = 4;   // waves per block

template <typename ST>
__device__ __forceinline__ float big_ld1(rsrc_t r, uint32_t off) {
    if constexpr (sizeof(ST) == 4) return buf_ld1(r, off);
    else return __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0) << 16);
}
#define EM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ void em_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int CS, int NT, typename ST>
__global__ __launch_bounds__(256) void conv3x3_s2b_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               ST* __restrict__ y, int B, int H, int W, int wmode) {
    constexpr int KS = (CS * 9 + 1) / 2, ES = sizeof(ST), Cb = 32 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, kh = lane >> 5;
    float wf[KS][NT], bv[NT];
    int koff[KS];
    uint32_t kbit[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int k = 2 * kk + kh;
        const bool kv = k < CS * 9;
        const int s = kv ? k / 9 : 0, t = kv ? k % 9 : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = i * NT + nt;
            wf[kk][nt] = kv ? ((wmode == 0) ? w[((int64_t)c * CS + s) * 9 + t] : w[((int64_t)s * Cb + c) * 9 + (8 - t)]) : 0.f;
        }
        koff[kk] = ((s * H + (t / 3 - 1)) * W + (t % 3 - 1)) * 4;
        kbit[kk] = kv ? (1u << t) : 0u;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = bias ? bias[i * NT + nt] : 0.f;
    const int tpr = (W + 31) / 32;
    const int tiles = B * H * tpr;
    // tile t = (image b, row yy, 32-pixel column tile tx), walked with a constant stride: the position is carried, not divided out per tile
    struct Pos {
        int b, yy, tx;
    };
    const int tstep = (int)gridDim.x * EM_NW;
    const int s_tx = tstep % tpr, s_rows = tstep / tpr, s_yy = s_rows % H, s_b = s_rows / H;
    auto advance = [&](Pos& p) {
        p.tx += s_tx;
        const int c1 = p.tx >= tpr ? 1 : 0;
        p.tx -= c1 * tpr;
        p.yy += s_yy + c1;
        const int c2 = p.yy >= H ? 1 : 0;
        p.yy -= c2 * H;
        p.b += s_b + c2;
    };
    // the image scalars of a tile (zeros for padding taps and for tiles past the end); the next tile's are loaded before this tile's MFMAs
    auto load_a = [&](const Pos& p, float (&a)[KS]) {
        const bool tv = p.b < B;
        const rsrc_t rs_x = make_rsrc(x + (int64_t)(tv ? p.b : 0) * CS * H * W);
        const int xp = p.tx * 32 + i;
        const uint32_t colm = (tv && xp < W) ? ((xp > 0 ? 1u : 0u) | 2u | (xp + 1 < W ? 4u : 0u)) : 0u;
        const uint32_t m9 = (p.yy > 0 ? colm : 0u) | (colm << 3) | (p.yy + 1 < H ? colm << 6 : 0u);
        const int base = (p.yy * W + xp) * 4;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[kk] = buf_ld1(rs_x, (m9 & kbit[kk]) ? (uint32_t)(base + koff[kk]) : ROW_SENT);
    };
    Pos cur, nxt;
    {
        const int t0 = (int)blockIdx.x * EM_NW + wave;
        const int by = t0 / tpr;
        cur.tx = t0 - by * tpr;
        cur.b = by / H;
        cur.yy = by - cur.b * H;
    }
    // Schedule, pinned by sched_barriers: prefetch of the next tile | 28 MFMAs | hand-over an -> a | 16 stores.  Left alone the scheduler sinks
    // the prefetch to its first use.  The hand-over sits BEFORE the stores: vmcnt counts loads and stores together and stores may retire out
    // of order, so a wait for the prefetch issued after the stores is a wait for the stores' acknowledgements too (vmcnt(13) .. (0)).
    float a[KS], an[KS];
    load_a(cur, an);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) asm volatile("v_mov_b32 %0, %1" : "=v"(a[kk]) : "v"(an[kk]));   // (the loop is entered with nothing in flight)
    for (; cur.b < B; cur = nxt) {
        nxt = cur;
        advance(nxt);
        load_a(nxt, an);
        __builtin_amdgcn_sched_barrier(0);
        const int x0 = cur.tx * 32, by = cur.b * H + cur.yy;
        const rsrc_t rs_y = make_rsrc(y + (int64_t)by * W * Cb, (uint32_t)(W * Cb * ES));
        floatx16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = bv[nt];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = EM_MFMA(a[kk], wf[kk][nt], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("v_mov_b32 %0, %1" : "=v"(a[kk]) : "v"(an[kk]));   // (real copies: a and an are two live ranges)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = x0 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const uint32_t off = px < W ? (uint32_t)((px * Cb + i * NT) * ES) : ROW_SENT;
            if constexpr (NT == 2) {
                if constexpr (ES == 4) {
                    floatx2 v;
                    v.x = acc[0][r];
                    v.y = acc[1][r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs_y, off, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(bf_pack(acc[0][r], acc[1][r]), rs_y, off, 0, 0);
                }
            } else {
                if constexpr (ES == 4) buf_st1(rs_y, off, acc[0][r]);
                else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(bf_pack(acc[0][r], 0.f) & 0xffffu), rs_y, off, 0, 0);
            }
        }
    }
}

// part[blk][CS * 9 + 1][Cb] as conv3x3_wgrad_kernel writes it (conv3x3_wgrad_reduce_kernel finishes); a wave takes `rpw` whole image rows as ONE
// stream of 16-pixel batches (8 MFMA steps x MT tiles), three batches in rotation: the loads of batch n + 2 are issued before the MFMAs of
// batch n (a batch's MFMAs are 1024 cycles, the stream comes from HBM: one batch ahead was slower than none).  Row m of tile mt is channel
// MT m + mt: a lane's MT values of a pixel are adjacent in memory -- one load.
template <int CS, int MT, typename ST>
__global__ __launch_bounds__(256) void conv3x3_wgrad_mfma_kernel(const ST* __restrict__ big, const float* __restrict__ small, float* __restrict__ part,
                                                                 int B, int H, int W, int rpw) {
    constexpr int NJ = CS * 9 + 1, ES = sizeof(ST), U = 8, Cb = 32 * MT;
    static_assert(NJ <= 32, "the S columns and the ones column must fit one MFMA tile");
    __shared__ float red[EM_NW][MT * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, kh = lane >> 5;
    const bool jtap = j < CS * 9, jone = j == CS * 9;
    const int s = jtap ? j / 9 : 0, tap = jtap ? j % 9 : 0, dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int rows = B * H;
    const int r0 = ((int)blockIdx.x * EM_NW + wave) * rpw;
    const int rend = (r0 + rpw < rows) ? r0 + rpw : rows;
    floatx16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    struct BPos {
        int rr, b, yy, x0;   // image row b * H + yy, first pixel of the batch
    };
    auto next = [&](BPos p) {
        p.x0 += 2 * U;
        if (p.x0 >= W) {
            p.x0 = 0;
            ++p.rr;
            if (++p.yy == H) {
                p.yy = 0;
                ++p.b;
            }
        }
        return p;
    };
    auto load_batch = [&](const BPos& p, float (&sv)[U], float (&av)[U][MT]) {   // (rows past the wave's range: every offset out of range, zeros)
        const bool rv = p.rr < rend;
        const rsrc_t rs_g = make_rsrc(big + (int64_t)(rv ? p.rr : 0) * W * Cb, (uint32_t)(W * Cb * ES));
        const rsrc_t rs_s = make_rsrc(small + (int64_t)(rv ? p.b : 0) * CS * H * W);
        const int ys = p.yy + dy;
        const bool rok = rv && jtap && ys >= 0 && ys < H;
        const int srow = ((s * H + ys) * W + dx) * 4;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int xp = p.x0 + 2 * u + kh, cs = xp + dx;
            const bool pin = rv && xp < W;
            const float v = buf_ld1(rs_s, (rok && pin && cs >= 0 && cs < W) ? (uint32_t)(srow + xp * 4) : ROW_SENT);
            sv[u] = v + ((jone && pin) ? 1.f : 0.f);   // (the ones column: its lanes load nothing -- an add, not a select: no branch around the load)
            const uint32_t off = pin ? (uint32_t)((xp * Cb + j * MT) * ES) : ROW_SENT;
            if constexpr (MT == 2 && ES == 4) {
                const floatx2 f = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs_g, off, 0, 0));
                av[u][0] = f.x;
                av[u][1] = f.y;
            } else if constexpr (MT == 2) {
                const uint32_t wv = __builtin_amdgcn_raw_buffer_load_b32(rs_g, off, 0, 0);
                av[u][0] = bf_lo(wv);
                av[u][1] = bf_hi(wv);
            } else {
                av[u][0] = big_ld1<ST>(rs_g, off);
            }
        }
    };
    auto mfma_batch = [&](const float (&sv)[U], const float (&av)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = EM_MFMA(av[u][mt], sv[u], acc[mt]);
    };
    if (r0 < rows) {
        const int nb = (rend - r0) * ((W + 2 * U - 1) / (2 * U));
        BPos p;
        p.rr = r0;
        p.b = r0 / H;
        p.yy = r0 - p.b * H;
        p.x0 = 0;
        float svA[U], avA[U][MT], svB[U], avB[U][MT], svC[U], avC[U][MT];
        load_batch(p, svA, avA);
        p = next(p);
        load_batch(p, svB, avB);
        for (int k = 0; k < nb; k += 3) {   // (pinned: the scheduler otherwise sinks every prefetch to its first use)
            p = next(p);
            load_batch(p, svC, avC);
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(svA, avA);
            __builtin_amdgcn_sched_barrier(0);
            p = next(p);
            load_batch(p, svA, avA);
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(svB, avB);
            __builtin_amdgcn_sched_barrier(0);
            p = next(p);
            load_batch(p, svB, avB);
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(svC, avC);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(mt * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    float* pp = part + (int64_t)blockIdx.x * NJ * Cb;
    for (int e = threadIdx.x; e < MT * 16 * 64; e += 256) {
        const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];   // fixed order
        const int l = e & 63, r = (e >> 6) & 15, mt = e >> 10;
        const int jj = l & 31, c = MT * (8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) + mt;
        if (jj < NJ) pp[jj * Cb + c] = v;
    }
}

struct B2sGeom {
    int bands, RS, strips;
};
inline __host__ __device__ B2sGeom b2s_geom(int B, int H, int W) {
    B2sGeom g;
    g.bands = (W + 29) / 30;
    int64_t rs = ((int64_t)B * H * g.bands + 4095) / 4096;   // about 4096 wave units (4 per SIMD), strips of 8 .. H rows (2 halo rows each)
    if (rs < 8) rs = 8;
    if (rs > H) rs = H;
    g.RS = (int)rs;
    g.strips = (H + g.RS - 1) / g.RS;
    return g;
}

template <int CS, int CB, typename ST>
__global__ __launch_bounds__(256) void conv3x3_b2s_mfma_kernel(const ST* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ res, float* __restrict__ y, int B, int H, int W, int wmode) {
    constexpr int ES = sizeof(ST), KS = CB / 2, CPL = 16 / ES, NG = CB / (2 * CPL), NJ = CS * 9, ZS = 28, NO = 30 * CS, NOI = (NO + 63) / 64;
    static_assert(NJ <= ZS && KS == NG * CPL, "b2s geometry");
    __shared__ float zr[EM_NW][3][32 * ZS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, kh = lane >> 5;
    const B2sGeom g = b2s_geom(B, H, W);
    const int unit = (int)blockIdx.x * EM_NW + wave;   // -> (image, strip, band), bands fastest
    if (unit >= B * g.strips * g.bands) return;         // (no block-level synchronisation below)
    const int band = unit % g.bands, bs = unit / g.bands;
    const int strip = bs % g.strips, b = bs / g.strips;
    const int y0 = strip * g.RS, y1 = (y0 + g.RS < H) ? y0 + g.RS : H;
    const int x0 = band * 30 - 1;   // image column of Z column 0; outputs are the columns x0 + 1 .. x0 + 30
    // B operand: Wz[c][j], step kk of half kh contracts channel c = (kk / CPL) * 2 CPL + kh * CPL + kk % CPL (the 16-byte pieces of the loads)
    float wf[KS];
    {
        const bool jv = i < NJ;
        const int s = jv ? i / 9 : 0, t = jv ? i % 9 : 0;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = (kk / CPL) * 2 * CPL + kh * CPL + kk % CPL;
            wf[kk] = jv ? ((wmode == 0) ? w[((int64_t)s * CB + c) * 9 + t] : w[((int64_t)c * CS + s) * 9 + (8 - t)]) : 0.f;
        }
    }
    const rsrc_t rs_x = make_rsrc(x + (int64_t)b * H * W * CB);
    const rsrc_t rs_y = make_rsrc(y + (int64_t)b * CS * H * W);
    const rsrc_t rs_r = make_rsrc(res ? res + (int64_t)b * CS * H * W : y);
    const int xp = x0 + i;
    const bool cok = xp >= 0 && xp < W;
    float* const zw = &zr[wave][0][0];
    auto load_row = [&](int r, u32x4 (&v)[NG]) {   // the 16-byte pieces of Z row r's pixels (zeros where the pixel does not exist or r > y1)
        const bool ok = cok && r >= 0 && r < H && r <= y1;
        const uint32_t off = ok ? (uint32_t)(((r * W + xp) * CB + kh * CPL) * ES) : ROW_SENT;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) v[gq] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + (uint32_t)(gq * 2 * CPL * ES), 0, 0);
    };
    u32x4 v[NG], vn[NG];
    load_row(y0 - 1, vn);
    int it = 0;
    for (int r = y0 - 1; r <= y1; ++r, ++it) {   // hand-over | the next row's loads | this row's MFMAs, Z row, gather (pinned, see the s2b kernel)
        const int slot = it % 3;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                                                     : "=&v"(v[gq].x), "=&v"(v[gq].y), "=&v"(v[gq].z), "=&v"(v[gq].w)
                                                     : "v"(vn[gq].x), "v"(vn[gq].y), "v"(vn[gq].z), "v"(vn[gq].w));
        load_row(r + 1, vn);
        // this iteration's outputs (row yo = r - 1): offsets and the residual, requested before the MFMAs
        const int yo = r - 1;
        const bool yok = yo >= y0 && yo < y1;
        uint32_t ooff[NOI];
        float resv[NOI];
#pragma unroll
        for (int q = 0; q < NOI; ++q) {
            const int o = q * 64 + lane;
            const int px = o / CS, s = o - px * CS;
            const int xo = x0 + 1 + px;
            ooff[q] = (yok && o < NO && xo < W) ? (uint32_t)(((s * H + yo) * W + xo) * 4) : ROW_SENT;
            resv[q] = res ? buf_ld1(rs_r, ooff[q]) : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        floatx16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            float av[CPL];
            if constexpr (ES == 4) {
                const floatx4 f = __builtin_bit_cast(floatx4, v[gq]);
                av[0] = f.x; av[1] = f.y; av[2] = f.z; av[3] = f.w;
            } else {
                av[0] = bf_lo(v[gq].x); av[1] = bf_hi(v[gq].x); av[2] = bf_lo(v[gq].y); av[3] = bf_hi(v[gq].y);
                av[4] = bf_lo(v[gq].z); av[5] = bf_hi(v[gq].z); av[6] = bf_lo(v[gq].w); av[7] = bf_hi(v[gq].w);
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) acc = EM_MFMA(av[e], wf[gq * CPL + e], acc);
        }
        if (i < ZS) {
            float* const zs = zw + slot * 32 * ZS;
#pragma unroll
            for (int q = 0; q < 16; ++q) zs[(8 * (q >> 2) + 4 * kh + (q & 3)) * ZS + i] = acc[q];
        }
        em_lds_fence();
        // ---- output row yo = r - 1 from the Z rows yo - 1, yo, yo + 1 (slots it - 2, it - 1, it) ----
        if (yok) {
#pragma unroll
            for (int q = 0; q < NOI; ++q) {
                const int o = q * 64 + lane;
                const int px = o / CS, s = o - px * CS;
                if (ooff[q] != ROW_SENT) {
                    float sum = bias ? bias[s] : 0.f;
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
                        const float* const zs = zw + ((it + 1 + dyi) % 3) * 32 * ZS + px * ZS + s * 9 + dyi * 3;   // slot of it - 2 + dyi
#pragma unroll
                        for (int dxi = 0; dxi < 3; ++dxi) sum += zs[dxi * ZS + dxi];
                    }
                    buf_st1(rs_y, ooff[q], sum + resv[q]);
                }
            }
        }
        em_lds_fence();
    }
}

bool edge_mfma_ok(int Cs, int Cb) {
    static const int on = dcpt_tuning("DCPT_EDGE_MFMA", 1);
    return on != 0 && Cs >= 1 && Cs <= 3 && (Cb == 32 || Cb == 64);
}
int wgrad_mfma_rpw(int B, int H) { return cdiv(B * H, 4096) > 0 ? cdiv(B * H, 4096) : 1; }   // rows per wave: about 4096 waves

template <typename ST>
int launch_s2b_mfma(const float* x, const float* w, const float* bias, ST* y, int B, int H, int W, int Cs, int Cb, int wmode, hipStream_t s) {
    const int64_t tiles = (int64_t)B * H * cdiv(W, 32);
    DCPT_CHECK_ARG(tiles < (1 << 30) && (double)H * W * Cs * 4.0 < 1.0e9, "conv3x3_s2b: image %dx%d x %d too large", H, W, B);
    const int64_t slots = (Cb == 64 && Cs == 3) ? 768 : 1024;   // resident blocks (3 / 4 per CU by registers): one round, every wave walks its tiles
    const dim3 grid((unsigned)(cdiv64(tiles, EM_NW) < slots ? cdiv64(tiles, EM_NW) : slots));
#define S2B(CS_, NT_) conv3x3_s2b_mfma_kernel<CS_, NT_, ST><<<grid, dim3(256), 0, s>>>(x, w, bias, y, B, H, W, wmode)
    if (Cb == 64) {
        if (Cs == 1) S2B(1, 2);
        else if (Cs == 2) S2B(2, 2);
        else S2B(3, 2);
    } else {
        if (Cs == 1) S2B(1, 1);
        else if (Cs == 2) S2B(2, 1);
        else S2B(3, 1);
    }
#undef S2B
    return DCPT_OK;
}

template <typename ST>
int launch_b2s_mfma(const ST* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cs, int Cb, int wmode,
                    hipStream_t s) {
    const B2sGeom g = b2s_geom(B, H, W);
    const int64_t units = (int64_t)B * g.strips * g.bands;
    DCPT_CHECK_ARG(units < (1 << 30) && (double)H * W * Cb * 4.0 < 1.0e9 && (double)H * W * Cs * 4.0 < 1.0e9, "conv3x3_b2s: image %dx%d x %d too large", H, W, B);
    const dim3 grid((unsigned)cdiv64(units, EM_NW));
#define B2S(CS_, CB_) conv3x3_b2s_mfma_kernel<CS_, CB_, ST><<<grid, dim3(256), 0, s>>>(x, w, bias, res, y, B, H, W, wmode)
    if (Cb == 64) {
        if (Cs == 1) B2S(1, 64);
        else if (Cs == 2) B2S(2, 64);
        else B2S(3, 64);
    } else {
        if (Cs == 1) B2S(1, 32);
        else if (Cs == 2) B2S(2, 32);
        else B2S(3, 32);
    }
#undef B2S
    return DCPT_OK;
}

template <typename ST>
int launch_wgrad_mfma(const ST* big, const float* small, float* part, int nblk, int B, int H, int W, int Cs, int Cb, hipStream_t s) {
    const int rpw = wgrad_mfma_rpw(B, H);
    DCPT_CHECK_ARG((int64_t)B * H < (1 << 30) && (double)W * Cb * 4.0 < 1.0e9 && (double)H * W * Cs * 4.0 < 1.0e9, "conv3x3_wgrad: image %dx%d x %d too large", H, W, B);
    const dim3 grid((unsigned)nblk);
#define WG(CS_, MT_) conv3x3_wgrad_mfma_kernel<CS_, MT_, ST><<<grid, dim3(256), 0, s>>>(big, small, part, B, H, W, rpw)
    if (Cb == 64) {
        if (Cs == 1) WG(1, 2);
        else if (Cs == 2) WG(2, 2);
        else WG(3, 2);
    } else {
        if (Cs == 1) WG(1, 1);
        else if (Cs == 2) WG(2, 1);
        else WG(3, 1);
    }
#undef WG
    return DCPT_OK;
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
    trace_tag(edge_mfma_ok(Cs, Cb) ? "edge.s2b_mfma" : "edge.s2b_valu");
    if (edge_mfma_ok(Cs, Cb)) DCPT_TRY(launch_s2b_mfma<bf16_t>(x, w, bias, y, B, H, W, Cs, Cb, wmode, s));
    else EDGE_GO_B(conv3x3_s2b_kernel, x, w, bias, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_s2b_bf16");
    return DCPT_OK;
}
int launch_conv3x3_b2s_bf16(const bf16_t* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cs, int Cb,
                            int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_b2s_bf16", 2048);
    DCPT_CHECK_ARG(Cb <= 256, "conv3x3_b2s_bf16: Cb=%d > 256 (one wave must hold all channel quads of a pixel)", Cb);
    trace_tag(edge_mfma_ok(Cs, Cb) ? "edge.b2s_mfma" : "edge.b2s_valu");
    if (edge_mfma_ok(Cs, Cb)) DCPT_TRY(launch_b2s_mfma<bf16_t>(x, w, bias, res, y, B, H, W, Cs, Cb, wmode, s));
    else EDGE_GO_B(conv3x3_b2s_kernel, x, w, bias, res, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_b2s_bf16");
    return DCPT_OK;
}

int launch_conv3x3_s2b(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cs, int Cb,
                       int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_s2b", 2048);
    trace_tag(edge_mfma_ok(Cs, Cb) ? "edge.s2b_mfma" : "edge.s2b_valu");
    if (edge_mfma_ok(Cs, Cb)) DCPT_TRY(launch_s2b_mfma<float>(x, w, bias, y, B, H, W, Cs, Cb, wmode, s));
    else EDGE_GO(conv3x3_s2b_kernel, x, w, bias, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_s2b");
    return DCPT_OK;
}

int launch_conv3x3_b2s(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W,
                       int Cs, int Cb, int wmode, hipStream_t s) {
    EDGE_CHECK("conv3x3_b2s", 2048);
    DCPT_CHECK_ARG(Cb <= 256, "conv3x3_b2s: Cb=%d > 256 (one wave must hold all channel quads of a pixel)", Cb);
    trace_tag(edge_mfma_ok(Cs, Cb) ? "edge.b2s_mfma" : "edge.b2s_valu");
    if (edge_mfma_ok(Cs, Cb)) DCPT_TRY(launch_b2s_mfma<float>(x, w, bias, res, y, B, H, W, Cs, Cb, wmode, s));
    else EDGE_GO(conv3x3_b2s_kernel, x, w, bias, res, y, B, H, W, Cb, wmode);
    DCPT_CHECK_LAUNCH("conv3x3_b2s");
    return DCPT_OK;
}

static int wgrad_valu_blocks(int B, int H, int W, int Cb) {
    const EdgeMap m = edge_map(B, H, W, Cb, WGRAD_BLOCKS);
    return B * m.strips * m.nwc;
}
static int wgrad_mfma_blocks(int B, int H) { return cdiv(B * H, EM_NW * wgrad_mfma_rpw(B, H)); }
// capacity of the partial buffer in blocks (the caller sizes its workspace before Cs is known to this function: the larger of the two forms)
int conv3x3_wgrad_num_blocks(int B, int H, int W, int Cb) {
    const int a = wgrad_valu_blocks(B, H, W, Cb), b = wgrad_mfma_blocks(B, H);
    return a > b ? a : b;
}

int launch_conv3x3_wgrad_bf16(const bf16_t* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H, int W, int Cs,
                              int Cb, int omode, hipStream_t s) {
    EDGE_CHECK("conv3x3_wgrad_bf16", WGRAD_BLOCKS);
    DCPT_CHECK_ARG(nblk == conv3x3_wgrad_num_blocks(B, H, W, Cb), "conv3x3_wgrad_bf16: nblk mismatch");
    if (edge_mfma_ok(Cs, Cb)) {
        nblk = wgrad_mfma_blocks(B, H);
        trace_tag("edge.wgrad_mfma");
        DCPT_TRY(launch_wgrad_mfma<bf16_t>(big, small, part, nblk, B, H, W, Cs, Cb, s));
    } else {
        trace_tag("edge.wgrad_valu");
        nblk = wgrad_valu_blocks(B, H, W, Cb);
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
    if (edge_mfma_ok(Cs, Cb)) {
        nblk = wgrad_mfma_blocks(B, H);
        trace_tag("edge.wgrad_mfma");
        DCPT_TRY(launch_wgrad_mfma<float>(big, small, part, nblk, B, H, W, Cs, Cb, s));
    } else {
        trace_tag("edge.wgrad_valu");
        nblk = wgrad_valu_blocks(B, H, W, Cb);
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
