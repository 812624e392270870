// The narrow level's forward 1 x 1 chains in fp32 storage, C = 64 (reference basicsr/archs/nafnet_arch.py:165-186):
//   FFN   out = y + (conv5(SimpleGate(conv4(LayerNorm2(y)) + b4)) + b5) * gamma      (the block's second half)
//   HEAD  t1 = conv1(LayerNorm1(inp)) + b1                                           (the first two links of the first half)
// each as ONE pass over the input.  Same wave-local scheme as the bf16 kernels of ffn_bf16.hip -- a wave owns groups of 32 pixels, its
// products are computed transposed (D[n][m] = sum_k W[n][k] x[m][k]: a lane holds one pixel and runs of four consecutive channels), the
// weights (48 KB) live in registers -- with what the exact fp32 matrix instruction changes:
//   * v_mfma_f32_32x32x2_f32 takes ONE value per lane and step, k = 2 s + (lane >> 5).  The contraction order is free, so step s uses
//     k = s + 32 (lane >> 5): lane (pixel m, half h) then needs exactly the contiguous half row x[m][32 h .. 32 h + 31] -- eight 16-byte LDS
//     reads -- and the LayerNorm, whose row a lane pair (m, m + 32) already holds that way, leaves its OUTPUT in the registers that are the
//     MFMA's B operand: LayerNorm(y) never touches LDS or HBM.
//   * 64 cycles per instruction: 192 MFMAs = 12 288 cycles per group and wave against ~15 000 cycles of HBM time for its 32 KB at four
//     waves per CU -- the kernel sits on the fp32-MFMA / HBM corner, the chain of separate GEMMs it replaces on neither.
// The gate goes through the y slot in place (the lane that read a 16-byte residual piece writes the gate piece) to be re-read as half rows;
// v (t1) and out are stored straight from the accumulators, 16 bytes per lane (the lane pair of a pixel writes 32 contiguous bytes, the
// sixteen stores of a group complete its rows: L2 merges them into whole lines) -- no staging buffer, which is what pays for the ring.
// vmcnt counts stores as well, and they may retire before older loads: a counted wait then can only wait LONGER than needed, never too
// short (loads retire in order and the count bounds them from above).
// Nothing else is written: the backward pass takes LayerNorm2(y) and the gate from the weight-gradient GEMM's operand loaders (A_LN on y,
// A_SG on v) and the statistics from mu / rstd.
// Results: fp32 MFMA products are exact and the k order is fixed, so a pixel's output depends on its row only (batch-consistency and
// run-to-run reproducibility are bit-exact); against the chain of GEMM kernels only the summation order differs.
#include "bufops.h"
#include "ffn_f32.h"

namespace {

#ifdef F32_ABL_NOMFMA   // ablation builds only (tools/build_variant.sh)
#define FFN_MFMA(a, b, c) (c)
#else
#define FFN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

constexpr int FW = 4;   // waves per block
// per-wave LDS: a ring of four y slots of 32 rows x 256 B (three groups = 24 KB in flight per wave: with one wave per SIMD the HBM latency
// is covered by bytes in flight, not by occupancy)
constexpr int S_SLOT = 8192, S_RING = 4, S_WAVE = S_RING * S_SLOT;
constexpr int S_TAB = FW * S_WAVE;   // fp32 tables: b4[128] | b5[64] | gamma[64] | lnw[64] | lnb[64]
constexpr int T_B4 = 0, T_B5 = 128, T_GM = 192, T_LW = 256, T_LB = 320, T_N = 384;

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int C, int HEAD>
__global__ __launch_bounds__(256) void ffn_fwd_f32_kernel(const FfnFwdF p) {
    static_assert(C == 64, "ffn_fwd_f32: C = 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[S_TAB + T_N * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tab = reinterpret_cast<float*>(smem + S_TAB);
    for (int i = tid; i < T_N; i += 256)
        tab[i] = i < T_B5 ? (p.b4 ? p.b4[i] : 0.f)
                          : i < T_LW ? (HEAD ? 0.f : (i < T_GM ? (p.b5 ? p.b5[i - T_B5] : 0.f) : (p.gamma ? p.gamma[i - T_GM] : 1.f)))
                                     : i < T_LB ? p.lnw[i - T_LW] : p.lnb[i - T_LB];
    const int fr = lane & 31, kh = lane >> 5;
    // A operands: row n = 32 j + (lane & 31) of the weight, k = s + 32 kh for step s
    float W4f[4][32], W5f[2][32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 w = ldg4(p.W4 + (32 * j + fr) * C + 32 * kh + 4 * q);
            W4f[j][4 * q + 0] = w.x; W4f[j][4 * q + 1] = w.y; W4f[j][4 * q + 2] = w.z; W4f[j][4 * q + 3] = w.w;
        }
    if constexpr (!HEAD) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 w = ldg4(p.W5 + (32 * j + fr) * C + 32 * kh + 4 * q);
                W5f[j][4 * q + 0] = w.x; W5f[j][4 * q + 1] = w.y; W5f[j][4 * q + 2] = w.z; W5f[j][4 * q + 3] = w.w;
            }
    }
    __syncthreads();

    unsigned char* const wb = smem + wave * S_WAVE;
    const uint32_t wb_lds = lds_addr(reinterpret_cast<const float*>(wb));
    const int64_t ng = (p.M + 31) / 32;
    const int64_t wg = (int64_t)blockIdx.x * FW + wave, TW = (int64_t)gridDim.x * FW;
    // a group = 32 rows x 256 B = eight DMAs of 4 rows: lane -> row 4 pc + (lane >> 4), LDS chunk lane & 15 = global chunk ^ (row & 15)
    auto issue = [&](int64_t gi, int s) {
        const i32x4 rs = make_rsrc_dma(p.y + (gi < ng ? gi : 0) * (32 * C));
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int row = 4 * pc + (lane >> 4);
            const bool ok = gi < ng && gi * 32 + row < p.M;
            dma16(rs, wb_lds + (uint32_t)(s * S_SLOT + pc * 1024), ok ? (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) * 16)) : ROW_SENT, 0);
        }
    };
    issue(wg, 0);
    issue(wg + TW, 1);
    issue(wg + 2 * TW, 2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    int s = 0;
    for (int64_t gi = wg; gi < ng; gi += TW) {
        unsigned char* const ys = wb + s * S_SLOT;
        issue(gi + 3 * TW, (s + 3) & 3);   // (the slot of the previous group: its last LDS access was fenced at the end of that iteration)
        const int64_t r0 = gi * 32;
        const bool rowok = r0 + fr < p.M;
        // ---- LayerNorm of pixel fr: this lane holds the half row 32 kh .., the other half is in lane ^ 32; its output IS the B operand ----
        float xf[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 w = *reinterpret_cast<const float4*>(ys + fr * 256 + (((8 * kh + c) ^ (fr & 15)) * 16));
            xf[4 * c + 0] = w.x; xf[4 * c + 1] = w.y; xf[4 * c + 2] = w.z; xf[4 * c + 3] = w.w;
        }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) sum += xf[e];
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            xf[e] -= mean;
            sq += xf[e] * xf[e];
        }
        sq += __shfl_xor(sq, 32);
        const float rs = 1.0f / sqrtf(sq * (1.0f / C) + p.eps);
        if (p.mu && kh == 0 && rowok) {
            p.mu[r0 + fr] = mean;
            p.rstd[r0 + fr] = rs;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 lw = *reinterpret_cast<const float4*>(tab + T_LW + 32 * kh + 4 * c), lb = *reinterpret_cast<const float4*>(tab + T_LB + 32 * kh + 4 * c);
            xf[4 * c + 0] = xf[4 * c + 0] * rs * lw.x + lb.x;
            xf[4 * c + 1] = xf[4 * c + 1] * rs * lw.y + lb.y;
            xf[4 * c + 2] = xf[4 * c + 2] * rs * lw.z + lb.z;
            xf[4 * c + 3] = xf[4 * c + 3] * rs * lw.w + lb.w;
        }
        // residual pieces of this lane's output channels 32 j + 8 g + 4 kh + i (16-byte chunk 8 j + 2 g + kh of the row)
        float4 yres[8];
        if constexpr (!HEAD) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) yres[4 * j + g] = *reinterpret_cast<const float4*>(ys + fr * 256 + (((8 * j + 2 * g + kh) ^ (fr & 15)) * 16));
        }
        // ---- v^T = W4 LN(y)^T + b4 ----
        floatx16 acc1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(tab + T_B4 + 32 * j + 8 * g + 4 * kh);
                acc1[j][4 * g + 0] = b.x; acc1[j][4 * g + 1] = b.y; acc1[j][4 * g + 2] = b.z; acc1[j][4 * g + 3] = b.w;
            }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[j] = FFN_MFMA(W4f[j][ks], xf[ks], acc1[j]);
        if (p.v) {
            const rsrc_t rsV = make_rsrc(p.v + r0 * (2 * C));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    buf_st4(rsV, rowok ? (uint32_t)(fr * 512 + (8 * j + 2 * g + kh) * 16) : ROW_SENT,
                            make_float4(acc1[j][4 * g + 0], acc1[j][4 * g + 1], acc1[j][4 * g + 2], acc1[j][4 * g + 3]));
        }
        if constexpr (!HEAD) {
            lds_fence();   // (the residual pieces are in registers: the slot now carries the gate)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(ys + fr * 256 + (((8 * j + 2 * g + kh) ^ (fr & 15)) * 16)) =
                        make_float4(acc1[j][4 * g + 0] * acc1[j + 2][4 * g + 0], acc1[j][4 * g + 1] * acc1[j + 2][4 * g + 1],
                                    acc1[j][4 * g + 2] * acc1[j + 2][4 * g + 2], acc1[j][4 * g + 3] * acc1[j + 2][4 * g + 3]);
            lds_fence();
            // ---- out^T = y^T + (W5 gate^T + b5) * gamma ----
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(ys + fr * 256 + (((8 * kh + c) ^ (fr & 15)) * 16));
                xf[4 * c + 0] = w.x; xf[4 * c + 1] = w.y; xf[4 * c + 2] = w.z; xf[4 * c + 3] = w.w;
            }
            floatx16 acc2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(tab + T_B5 + 32 * j + 8 * g + 4 * kh);
                    acc2[j][4 * g + 0] = b.x; acc2[j][4 * g + 1] = b.y; acc2[j][4 * g + 2] = b.z; acc2[j][4 * g + 3] = b.w;
                }
#pragma unroll
            for (int ks = 0; ks < 32; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j] = FFN_MFMA(W5f[j][ks], xf[ks], acc2[j]);
            const rsrc_t rsO = make_rsrc(p.out + r0 * C);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 gm = *reinterpret_cast<const float4*>(tab + T_GM + 32 * j + 8 * g + 4 * kh);
                    const float4 yv = yres[4 * j + g];
                    buf_st4(rsO, rowok ? (uint32_t)(fr * 256 + (8 * j + 2 * g + kh) * 16) : ROW_SENT,
                            make_float4(yv.x + acc2[j][4 * g + 0] * gm.x, yv.y + acc2[j][4 * g + 1] * gm.y, yv.z + acc2[j][4 * g + 2] * gm.z,
                                        yv.w + acc2[j][4 * g + 3] * gm.w));
                }
        }
        lds_fence();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the next group has landed (two more groups may be in flight)
        s = (s + 1) & 3;
    }
    dma_wait_all();
}


// ---- backward data path of the second half:  dout -> dg = (dout * gamma) W5 -> dv = SimpleGate'(dg; v) -> dxn2 = dv W4 -> dy = dout + LN2'(dxn2; y) ----
// A group's inputs (dout, v, y) are 1 KB per pixel in fp32: with 32 pixels a two-slot ring would be 64 KB per wave.  The 16 x 16 x 4
// instruction (same FLOP rate, 32 cycles) makes the group 16 pixels: four lanes per pixel (q = lane >> 4), step s contracts
// k = s + (K / 4) q, so lane (pixel m, quarter q) feeds the contiguous quarter row x[m][(K / 4) q ...] and holds, per 16 x 16 output tile
// t, the four consecutive channels 16 t + 4 q + r -- one 16-byte piece.  dv overwrites v and dy overwrites dout IN PLACE (the lane that
// reads a piece writes it); both leave as full rows.  LayerNorm statistics are recomputed from y; the LayerNorm's weight / bias gradients
// accumulate per lane and channel and are reduced once ([wave][2][C] partials).  Weights: 192 registers (wT5 64, wT4 128).
constexpr int G_DO = 0, G_V = 4096, G_Y = 12288, G_SLOT = 16384, G_WAVE = 2 * G_SLOT;
constexpr int G_TAB = FW * G_WAVE;   // lnw[64]
typedef float floatx4v __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(256) void ffn_bwd_f32_kernel(const FfnBwdF p) {
    static_assert(C == 64, "ffn_bwd_f32: C = 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[G_TAB + 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tab = reinterpret_cast<float*>(smem + G_TAB);
    if (tid < 64) tab[tid] = p.lnw[tid];
    const int fm = lane & 15, q = lane >> 4;
    // A operands: wT5[k][n] (row k = 16 t + fm of dg, n = s + 16 q), wT4[c][j] (row c = 16 t + fm of dxn2, j = s + 32 q)
    float W5f[4][16], W4f[4][32];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 w = ldg4(p.wT5 + (16 * t + fm) * C + 16 * q + 4 * e);
            W5f[t][4 * e + 0] = w.x; W5f[t][4 * e + 1] = w.y; W5f[t][4 * e + 2] = w.z; W5f[t][4 * e + 3] = w.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 w = ldg4(p.wT4 + (16 * t + fm) * (2 * C) + 32 * q + 4 * e);
            W4f[t][4 * e + 0] = w.x; W4f[t][4 * e + 1] = w.y; W4f[t][4 * e + 2] = w.z; W4f[t][4 * e + 3] = w.w;
        }
    }
    __syncthreads();

    unsigned char* const wb = smem + wave * G_WAVE;
    const uint32_t wb_lds = lds_addr(reinterpret_cast<const float*>(wb));
    const int64_t ng = (p.M + 15) / 16;
    const int64_t wg = (int64_t)blockIdx.x * FW + wave, TW = (int64_t)gridDim.x * FW;
    auto issue = [&](int64_t gi, int s) {
        const int64_t gg = gi < ng ? gi : 0;
        const i32x4 rsD = make_rsrc_dma(p.dout + gg * (16 * C)), rsY = make_rsrc_dma(p.y + gg * (16 * C)), rsV = make_rsrc_dma(p.v + gg * (32 * C));
        const uint32_t base = wb_lds + (uint32_t)(s * G_SLOT);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {   // 256-byte rows: 4 rows per DMA
            const int row = 4 * pc + (lane >> 4);
            const bool ok = gi < ng && gi * 16 + row < p.M;
            const uint32_t vo = ok ? (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) * 16)) : ROW_SENT;
            dma16(rsD, base + (uint32_t)(G_DO + pc * 1024), vo, 0);
            dma16(rsY, base + (uint32_t)(G_Y + pc * 1024), vo, 0);
        }
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {   // 512-byte rows: 2 rows per DMA
            const int row = 2 * pc + (lane >> 5);
            const bool ok = gi < ng && gi * 16 + row < p.M;
            dma16(rsV, base + (uint32_t)(G_V + pc * 1024), ok ? (uint32_t)(row * 512 + (((lane & 31) ^ (row & 15)) * 16)) : ROW_SENT, 0);
        }
    };
    issue(wg, 0);
    issue(wg + TW, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    float aw[4][4], ab[4][4];   // LayerNorm weight / bias gradient partials of this lane's channels 16 t + 4 q + r
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) aw[t][r] = ab[t][r] = 0.f;

    int s = 0;
    for (int64_t gi = wg; gi < ng; gi += TW) {
        unsigned char* const sl = wb + s * G_SLOT;
        const int sw = fm;   // swizzle key of this lane's pixel row (row & 15 = fm)
        // ---- dg^T = wT5 dout^T ----
        float xb[32];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 w = *reinterpret_cast<const float4*>(sl + G_DO + fm * 256 + (((4 * q + e) ^ sw) * 16));
            xb[4 * e + 0] = w.x; xb[4 * e + 1] = w.y; xb[4 * e + 2] = w.z; xb[4 * e + 3] = w.w;
        }
        floatx4v dg[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dg[t] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 16; ++st)
#pragma unroll
            for (int t = 0; t < 4; ++t) dg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W5f[t][st], xb[st], dg[t], 0, 0, 0);
        // ---- SimpleGate backward in place: dv1 = dg * v2, dv2 = dg * v1 (pieces 4 t + q and 16 + 4 t + q of the v row) ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4* const a1 = reinterpret_cast<float4*>(sl + G_V + fm * 512 + (((4 * t + q) ^ sw) * 16));
            float4* const a2 = reinterpret_cast<float4*>(sl + G_V + fm * 512 + (((16 + 4 * t + q) ^ sw) * 16));
            const float4 v1 = *a1, v2 = *a2;
            const float4 d = make_float4(dg[t][0], dg[t][1], dg[t][2], dg[t][3]);
            *a1 = f4_mul(d, v2);
            *a2 = f4_mul(d, v1);
        }
        lds_fence();
        // ---- dxn2^T = wT4 dv^T ----
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 w = *reinterpret_cast<const float4*>(sl + G_V + fm * 512 + (((8 * q + e) ^ sw) * 16));
            xb[4 * e + 0] = w.x; xb[4 * e + 1] = w.y; xb[4 * e + 2] = w.z; xb[4 * e + 3] = w.w;
        }
        floatx4v dx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 32; ++st)
#pragma unroll
            for (int t = 0; t < 4; ++t) dx[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W4f[t][st], xb[st], dx[t], 0, 0, 0);
        // ---- LayerNorm2 backward for pixel fm: this lane holds channels 16 t + 4 q + r, the rest of the row is in the three lanes ^ 16, ^ 32 ----
        float xh[4][4];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 yv = *reinterpret_cast<const float4*>(sl + G_Y + fm * 256 + (((4 * t + q) ^ sw) * 16));
            xh[t][0] = yv.x; xh[t][1] = yv.y; xh[t][2] = yv.z; xh[t][3] = yv.w;
            sum += (yv.x + yv.y) + (yv.z + yv.w);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[t][r] -= mean;
                sq += xh[t][r] * xh[t][r];
            }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rs = 1.0f / sqrtf(sq * (1.0f / C) + p.eps);
        const bool rowok = gi * 16 + fm < p.M;
        float s1 = 0.f, s2 = 0.f;
        float gwv[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 lw = *reinterpret_cast<const float4*>(tab + 16 * t + 4 * q);
            const float lws[4] = {lw.x, lw.y, lw.z, lw.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[t][r] *= rs;
                const float gx = dx[t][r];
                aw[t][r] += rowok ? gx * xh[t][r] : 0.f;
                ab[t][r] += rowok ? gx : 0.f;
                const float gw = gx * lws[r];
                gwv[t][r] = gw;
                s1 += gw;
                s2 += gw * xh[t][r];
            }
        }
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        s1 *= (1.0f / C);
        s2 *= (1.0f / C);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4* const a = reinterpret_cast<float4*>(sl + G_DO + fm * 256 + (((4 * t + q) ^ sw) * 16));
            const float4 dres = *a;
            *a = make_float4(rs * (gwv[t][0] - xh[t][0] * s2 - s1) + dres.x, rs * (gwv[t][1] - xh[t][1] * s2 - s1) + dres.y,
                             rs * (gwv[t][2] - xh[t][2] * s2 - s1) + dres.z, rs * (gwv[t][3] - xh[t][3] * s2 - s1) + dres.w);
        }
        lds_fence();
        // ---- outputs back into registers as full rows, slot re-staged, counted wait, stores ----
        u32x4 ody[4], odv[8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 4 + (lane >> 4), ch = lane & 15;
            ody[it] = *reinterpret_cast<const u32x4*>(sl + G_DO + row * 256 + ((ch ^ (row & 15)) * 16));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 2 + (lane >> 5), ch = lane & 31;
            odv[it] = *reinterpret_cast<const u32x4*>(sl + G_V + row * 512 + ((ch ^ (row & 15)) * 16));
        }
        lds_fence();
        issue(gi + 2 * TW, s);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        const int64_t r0 = gi * 16;
        const rsrc_t rsO = make_rsrc(p.dy + r0 * C), rsV = make_rsrc(p.dv + r0 * (2 * C));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 4 + (lane >> 4), ch = lane & 15;
            __builtin_amdgcn_raw_buffer_store_b128(ody[it], rsO, r0 + row < p.M ? (uint32_t)(row * 256 + ch * 16) : ROW_SENT, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 2 + (lane >> 5), ch = lane & 31;
            __builtin_amdgcn_raw_buffer_store_b128(odv[it], rsV, r0 + row < p.M ? (uint32_t)(row * 512 + ch * 16) : ROW_SENT, 0, 0);
        }
        s ^= 1;
    }
    dma_wait_all();
    // column sums over this wave's pixels: the 16 pixel lanes of each quarter (fixed butterfly order: deterministic)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = aw[t][r], b = ab[t][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (fm == 0) {
                const int c = 16 * t + 4 * q + r;
                p.lnpart[(wg * 2 + 0) * C + c] = a;
                p.lnpart[(wg * 2 + 1) * C + c] = b;
            }
        }
}

}  // namespace

bool ffn_fwd_f32_ok(int C) { return C == 64; }

static int launch(const FfnFwdF& p, int head, hipStream_t s) {
    int64_t blocks = cdiv64(cdiv64(p.M, 32), FW);
    if (blocks > 256) blocks = 256;   // one block per CU, persistent over its groups
    if (head) ffn_fwd_f32_kernel<64, 1><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
    else ffn_fwd_f32_kernel<64, 0><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ffn_fwd_f32");
    return DCPT_OK;
}

int launch_ffn_fwd_f32(const FfnFwdF& p, int C, hipStream_t s) {
    DCPT_CHECK_ARG(ffn_fwd_f32_ok(C), "ffn_fwd_f32: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.y && p.out && p.W4 && p.W5 && p.lnw && p.lnb && p.M > 0 && (p.mu == nullptr) == (p.rstd == nullptr), "ffn_fwd_f32: null argument");
    return launch(p, 0, s);
}

int launch_ln_conv_f32(const FfnFwdF& p, int C, hipStream_t s) {
    DCPT_CHECK_ARG(ffn_fwd_f32_ok(C), "ln_conv_f32: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.y && p.v && p.W4 && p.lnw && p.lnb && p.M > 0 && (p.mu == nullptr) == (p.rstd == nullptr), "ln_conv_f32: null argument");
    return launch(p, 1, s);
}

int ffn_bwd_f32_waves(int64_t M) {
    int64_t blocks = cdiv64(cdiv64(M, 16), FW);
    if (blocks > 256) blocks = 256;
    return (int)blocks * FW;
}

int launch_ffn_bwd_f32(const FfnBwdF& p, int C, hipStream_t s) {
    DCPT_CHECK_ARG(ffn_fwd_f32_ok(C), "ffn_bwd_f32: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.dout && p.v && p.y && p.wT5 && p.wT4 && p.lnw && p.dv && p.dy && p.lnpart && p.M > 0, "ffn_bwd_f32: null argument");
    ffn_bwd_f32_kernel<64><<<dim3((unsigned)(ffn_bwd_f32_waves(p.M) / FW)), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ffn_bwd_f32");
    return DCPT_OK;
}
