// Fused second half of a NAFBlock ("FFN": LayerNorm2 -> conv4 -> SimpleGate -> conv5 -> residual; reference
// basicsr/archs/nafnet_arch.py:180-186) for the NARROW levels in bf16 storage, C = 64.
//
// At C = 64 a row of the feature map is 128 bytes and the block's two GEMMs do 24 KFLOP per pixel: the chain of separate kernels
// (ln_fwd, conv4 + gate epilogue, conv5 + residual epilogue) is pure HBM traffic -- 9 passes over an [M][C] tensor where the data
// flow needs 4 (read y; write v twice as wide, write out).  This kernel does the chain per group of 32 pixels without leaving the CU:
//
//   * everything is WAVE-LOCAL.  A wave owns groups of 32 pixels (one MFMA tile of pixels): its lanes normalise those rows, the
//     normalised rows are the B operand of its own MFMAs, the gate product and the residual are applied by the lanes that hold the
//     accumulators.  No block barrier in the loop, no cross-wave traffic; a block is four such waves sharing only the parameter tables.
//   * the products are computed TRANSPOSED, D[n][m] = sum_k W[n][k] x[m][k] (A operand = weight fragment, B operand = pixel fragment):
//     a lane then holds ONE pixel and 16 channels of it per 32 x 32 tile in runs of four consecutive channels -- bias, gate, gamma and
//     the residual are elementwise per lane, and four channels pack into one 8-byte LDS write.  The weights (24 KB in bf16) live in
//     REGISTERS for the whole kernel (96 VGPRs: one wave per SIMD has 512).
//   * y comes in by LDS-DMA through a three-slot ring per wave (two groups in flight), 16-byte chunks XOR-swizzled by row so that the
//     row-wise LayerNorm reads, the fragment reads and the 8-byte epilogue accesses are conflict-free or 2-way; v and out leave through
//     small swizzled staging buffers as full 16-byte, row-contiguous stores.
//   * vmcnt counts loads and stores together on gfx9 and the two kinds retire out of order with respect to each other, so the ring's
//     counted wait is placed BEFORE the group's stores are issued (a whole group's work after the previous group's stores): the count it
//     waits for is then an upper bound of the loads in flight whatever the stores do.
//
// Rounding points are the unfused path's (nafblock_bf16.hip): LN2(y) rounded to bf16 (it is an MFMA operand), v = conv4 + bias rounded
// once on store, SimpleGate(v) = product of the UNROUNDED halves rounded once (operand of conv5), out rounded once.
#include "bf16_ops.h"
#include "prof.h"
#include "ffn_bf16.h"

namespace {

constexpr int FW = 4;   // waves per block
// per-wave LDS bytes: y ring 3 x 4 KB | LN2(y) 4 KB | gate 4 KB | v staging 8 KB | out staging 4 KB
constexpr int F_RING = 0, F_XN = 12288, F_GT = 16384, F_VST = 20480, F_OST = 28672, F_WAVE = 32768;
constexpr int F_TAB = FW * F_WAVE;   // fp32 tables: b4[128] | b5[64] | gamma[64] | lnw[64] | lnb[64]
constexpr int T_B4 = 0, T_B5 = 128, T_GM = 192, T_LW = 256, T_LB = 320, T_N = 384;

typedef __attribute__((address_space(3))) unsigned char* lds_p;

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// HEAD = 1: only the first two links, t1 = conv1(LayerNorm1(inp)) + b1 (the block's FIRST half up to the depthwise conv: `y` is the block
// input, `v` receives t1, `xn2` LN1(inp) for conv1's weight gradient; W5 / b5 / gamma / out / g are not touched)
template <int C, int HEAD>
__global__ __launch_bounds__(256) void ffn_fwd_bf16_kernel(const FfnFwdB p) {
    static_assert(C == 64, "ffn_fwd_bf16: C = 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[F_TAB + T_N * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tab = reinterpret_cast<float*>(smem + F_TAB);
    for (int i = tid; i < T_N; i += 256)
        tab[i] = i < T_B5 ? p.b4[i] : i < T_LW ? (HEAD ? 0.f : (i < T_GM ? p.b5[i - T_B5] : p.gamma[i - T_GM])) : i < T_LB ? p.lnw[i - T_LW] : p.lnb[i - T_LB];
    // weight fragments (A operands of the transposed products): rows n = 32 j + (lane & 31), k = 16 ks + 8 (lane >> 5) .. + 7
    const int fr = lane & 31, kh = lane >> 5;
    bf16x8 W4f[4][4], W5f[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) W4f[j][ks] = *reinterpret_cast<const bf16x8*>(p.W4 + (32 * j + fr) * C + 16 * ks + 8 * kh);
    if constexpr (!HEAD) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) W5f[j][ks] = *reinterpret_cast<const bf16x8*>(p.W5 + (32 * j + fr) * C + 16 * ks + 8 * kh);
    }
    __syncthreads();

    unsigned char* const wb = smem + wave * F_WAVE;
    const uint32_t wb_lds = lds_addr(reinterpret_cast<const float*>(wb));
    const int64_t ng = (p.M + 31) / 32;
    const int64_t wg = (int64_t)blockIdx.x * FW + wave, TW = (int64_t)gridDim.x * FW;

    // one group = 32 rows x 128 B = four DMAs of 8 rows; lane -> row 8 pc + (lane >> 3), LDS chunk lane & 7 = global chunk ^ (row & 7)
    const uint32_t dma_voff = (uint32_t)(lane >> 3) * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    auto issue = [&](int64_t gi, int s) {
        const i32x4 rs = make_rsrc_dma(p.y + (gi < ng ? gi : 0) * (32 * C));
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const bool ok = gi < ng && gi * 32 + 8 * pc + (lane >> 3) < p.M;
            dma16(rs, wb_lds + (uint32_t)(F_RING + s * 4096 + pc * 1024), ok ? dma_voff + (uint32_t)pc * 1024u : ROW_SENT, 0);
        }
    };
    issue(wg, 0);
    issue(wg + TW, 1);
    issue(wg + 2 * TW, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");

    const int lr = lane >> 1, hf = lane & 1;   // LayerNorm: row lr of the group, channels 32 hf .. + 31 (four 16-byte chunks)
    int s = 0;
    for (int64_t gi = wg; gi < ng; gi += TW) {
        unsigned char* const ys = wb + F_RING + s * 4096;
        // ---- LayerNorm2 of the group's rows (two-pass statistics in fp32 over the bf16 inputs, as ln_fwd_bf16) ----
        float x[32];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(ys + lr * 128 + (((4 * hf + c) ^ (lr & 7)) * 16));
            x[8 * c + 0] = bf_lo(w.x); x[8 * c + 1] = bf_hi(w.x); x[8 * c + 2] = bf_lo(w.y); x[8 * c + 3] = bf_hi(w.y);
            x[8 * c + 4] = bf_lo(w.z); x[8 * c + 5] = bf_hi(w.z); x[8 * c + 6] = bf_lo(w.w); x[8 * c + 7] = bf_hi(w.w);
        }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) sum += x[e];
        sum += __shfl_xor(sum, 1);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            x[e] -= mean;
            sq += x[e] * x[e];
        }
        sq += __shfl_xor(sq, 1);
        const float rs = 1.0f / sqrtf(sq * (1.0f / C) + p.eps);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 w0 = *reinterpret_cast<const float4*>(tab + T_LW + 32 * hf + 8 * c), w1 = *reinterpret_cast<const float4*>(tab + T_LW + 32 * hf + 8 * c + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(tab + T_LB + 32 * hf + 8 * c), b1 = *reinterpret_cast<const float4*>(tab + T_LB + 32 * hf + 8 * c + 4);
            u32x4 o;
            o.x = bf_pack(x[8 * c + 0] * rs * w0.x + b0.x, x[8 * c + 1] * rs * w0.y + b0.y);
            o.y = bf_pack(x[8 * c + 2] * rs * w0.z + b0.z, x[8 * c + 3] * rs * w0.w + b0.w);
            o.z = bf_pack(x[8 * c + 4] * rs * w1.x + b1.x, x[8 * c + 5] * rs * w1.y + b1.y);
            o.w = bf_pack(x[8 * c + 6] * rs * w1.z + b1.z, x[8 * c + 7] * rs * w1.w + b1.w);
            *reinterpret_cast<u32x4*>(wb + F_XN + lr * 128 + (((4 * hf + c) ^ (lr & 7)) * 16)) = o;
        }
        lds_fence();
        // ---- v^T = W4 LN2(y)^T + b4: lane = pixel fr, registers = channels 32 j + 8 g + 4 kh + i ----
        bf16x8 xf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(wb + F_XN + fr * 128 + (((2 * ks + kh) ^ (fr & 7)) * 16));
        floatx16 acc1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(tab + T_B4 + 32 * j + 8 * g + 4 * kh);
                acc1[j][4 * g + 0] = b.x; acc1[j][4 * g + 1] = b.y; acc1[j][4 * g + 2] = b.z; acc1[j][4 * g + 3] = b.w;
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W4f[j][ks], xf[ks], acc1[j], 0, 0, 0);
        // v staging (16 chunks of 16 B per row, chunk ^ (row & 15)) and the gate = product of the unrounded halves (8 chunks, ^ (row & 7))
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w.x = bf_pack(acc1[j][4 * g + 0], acc1[j][4 * g + 1]);
                w.y = bf_pack(acc1[j][4 * g + 2], acc1[j][4 * g + 3]);
                *reinterpret_cast<u32x2*>(wb + F_VST + fr * 256 + (((4 * j + g) ^ (fr & 15)) * 16) + kh * 8) = w;
            }
        if constexpr (!HEAD) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w.x = bf_pack(acc1[j][4 * g + 0] * acc1[j + 2][4 * g + 0], acc1[j][4 * g + 1] * acc1[j + 2][4 * g + 1]);
                w.y = bf_pack(acc1[j][4 * g + 2] * acc1[j + 2][4 * g + 2], acc1[j][4 * g + 3] * acc1[j + 2][4 * g + 3]);
                *reinterpret_cast<u32x2*>(wb + F_GT + fr * 128 + (((4 * j + g) ^ (fr & 7)) * 16) + kh * 8) = w;
            }
        lds_fence();
        // ---- out^T = y^T + (W5 gate^T + b5) * gamma ----
        bf16x8 gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) gf[ks] = *reinterpret_cast<const bf16x8*>(wb + F_GT + fr * 128 + (((2 * ks + kh) ^ (fr & 7)) * 16));
        floatx16 acc2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(tab + T_B5 + 32 * j + 8 * g + 4 * kh);
                acc2[j][4 * g + 0] = b.x; acc2[j][4 * g + 1] = b.y; acc2[j][4 * g + 2] = b.z; acc2[j][4 * g + 3] = b.w;
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W5f[j][ks], gf[ks], acc2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gm = *reinterpret_cast<const float4*>(tab + T_GM + 32 * j + 8 * g + 4 * kh);
                const uint32_t off = (uint32_t)(fr * 128 + (((4 * j + g) ^ (fr & 7)) * 16) + kh * 8);
                const float4 yv = bf4_unpack(*reinterpret_cast<const u32x2*>(ys + off));
                float4 o;
                o.x = yv.x + acc2[j][4 * g + 0] * gm.x;
                o.y = yv.y + acc2[j][4 * g + 1] * gm.y;
                o.z = yv.z + acc2[j][4 * g + 2] * gm.z;
                o.w = yv.w + acc2[j][4 * g + 3] * gm.w;
                *reinterpret_cast<u32x2*>(wb + F_OST + off) = bf4_pack(o);
            }
        }
        lds_fence();
        // ---- ring: the slot of this group is free; wait for the next group BEFORE this group's stores are issued ----
        issue(gi + 3 * TW, s);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        // ---- stores: full 16-byte chunks, row-contiguous ----
        const int64_t r0 = gi * 32;
        if constexpr (!HEAD) {
            const rsrc_t rsO = make_rsrc(p.out + r0 * C);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const u32x4 w = *reinterpret_cast<const u32x4*>(wb + F_OST + row * 128 + ((ch ^ (row & 7)) * 16));
                __builtin_amdgcn_raw_buffer_store_b128(w, rsO, r0 + row < p.M ? (uint32_t)(row * 128 + ch * 16) : ROW_SENT, 0, 0);
            }
        }
        if (p.v) {
            const rsrc_t rsV = make_rsrc(p.v + r0 * (2 * C));
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4), ch = lane & 15;
                const u32x4 w = *reinterpret_cast<const u32x4*>(wb + F_VST + row * 256 + ((ch ^ (row & 15)) * 16));
                __builtin_amdgcn_raw_buffer_store_b128(w, rsV, r0 + row < p.M ? (uint32_t)(row * 256 + ch * 16) : ROW_SENT, 0, 0);
            }
        }
        if (p.xn2) {   // (kept for a caller whose backward still reads LN2(y) and the gate: two more passes)
            const rsrc_t rsX = make_rsrc(p.xn2 + r0 * C), rsG = make_rsrc((HEAD ? p.xn2 : p.g) + r0 * C);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const uint32_t lo = (uint32_t)(row * 128 + ((ch ^ (row & 7)) * 16));
                const uint32_t go = r0 + row < p.M ? (uint32_t)(row * 128 + ch * 16) : ROW_SENT;
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(wb + F_XN + lo), rsX, go, 0, 0);
                if constexpr (!HEAD) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(wb + F_GT + lo), rsG, go, 0, 0);
            }
        }
        if (p.mu && hf == 0 && r0 + lr < p.M) {
            p.mu[r0 + lr] = mean;
            p.rstd[r0 + lr] = rs;
        }
        lds_fence();
        s = s == 2 ? 0 : s + 1;
    }
    dma_wait_all();
}


// ---- backward data path of the same half:  dout -> dg = (dout * gamma) W5 -> dv = SimpleGate'(dg; v) -> dxn2 = dv W4 -> dy = dout + LN2'(dxn2; y) ----
// Same wave-local scheme, products transposed (lane = pixel).  Inputs of a group (dout 4 KB, v 8 KB, y 4 KB) come by LDS-DMA into a
// two-slot ring; dv overwrites v and dy overwrites dout IN PLACE (the lane that reads an 8-byte piece writes the piece), so the slot is also
// the staging buffer of the two outputs: they are read back as full rows into registers, the slot is re-staged, and the stores go out
// after the ring's counted wait (see the note on vmcnt above).  dv is written for the conv4 weight-gradient GEMM, which stays the
// transposing-read TN kernel, as does conv5's (operands dout and the saved gate).  The LayerNorm statistics are recomputed from y (the
// forward pass's two-pass formula on the same bf16 values: identical), its weight / bias gradients accumulate per lane and channel over
// all the wave's groups and are reduced once at the end (partials [wave][2][C] for colpart_reduce).
constexpr int B_DO = 0, B_V = 4096, B_Y = 12288, B_SLOT = 16384, B_WAVE = 2 * B_SLOT;
constexpr int B_TAB = FW * B_WAVE;   // fp32 table: lnw[64]

// TAIL = 1: only the last two links, dx = dres + LayerNorm'(dz W; x) -- the end of the block's FIRST half (dz = dt1 [M][2C] arrives where v
// does, W = conv1^T where wT4 does, x = the block input, dres = dy; nothing is gated and no dv is written)
template <int C, int TAIL>
__global__ __launch_bounds__(256) void ffn_bwd_bf16_kernel(const FfnBwdB p) {
    static_assert(C == 64, "ffn_bwd_bf16: C = 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_TAB + 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tab = reinterpret_cast<float*>(smem + B_TAB);
    if (tid < 64) tab[tid] = p.lnw[tid];
    const int fr = lane & 31, kh = lane >> 5;
    // A operands: wT5[k][n] = W5[n][k] gamma[n] (rows k of dg, contraction over n), wT4[c][j] = W4[j][c] (rows c of dxn2, contraction over j)
    bf16x8 W5f[2][4], W4f[2][8];
    if constexpr (!TAIL) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) W5f[j][ks] = *reinterpret_cast<const bf16x8*>(p.wT5 + (32 * j + fr) * C + 16 * ks + 8 * kh);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) W4f[j][ks] = *reinterpret_cast<const bf16x8*>(p.wT4 + (32 * j + fr) * (2 * C) + 16 * ks + 8 * kh);
    __syncthreads();

    unsigned char* const wb = smem + wave * B_WAVE;
    const uint32_t wb_lds = lds_addr(reinterpret_cast<const float*>(wb));
    const int64_t ng = (p.M + 31) / 32;
    const int64_t wg = (int64_t)blockIdx.x * FW + wave, TW = (int64_t)gridDim.x * FW;
    // rows of 128 B (dout, y): 8 rows per DMA, chunk ^ (row & 7); rows of 256 B (v): 4 rows per DMA, 16 chunks, chunk ^ (row & 15)
    const uint32_t voff128 = (uint32_t)(lane >> 3) * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    auto issue = [&](int64_t gi, int s) {
        const int64_t gg = gi < ng ? gi : 0;
        const i32x4 rsD = make_rsrc_dma(p.dout + gg * (32 * C)), rsY = make_rsrc_dma(p.y + gg * (32 * C)), rsV = make_rsrc_dma(p.v + gg * (64 * C));
        const uint32_t base = wb_lds + (uint32_t)(s * B_SLOT);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const bool ok = gi < ng && gi * 32 + 8 * pc + (lane >> 3) < p.M;
            const uint32_t vo = ok ? voff128 + (uint32_t)pc * 1024u : ROW_SENT;
            dma16(rsD, base + (uint32_t)(B_DO + pc * 1024), vo, 0);
            dma16(rsY, base + (uint32_t)(B_Y + pc * 1024), vo, 0);
        }
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int row = 4 * pc + (lane >> 4);
            const bool ok = gi < ng && gi * 32 + row < p.M;
            dma16(rsV, base + (uint32_t)(B_V + pc * 1024), ok ? (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) * 16)) : ROW_SENT, 0);
        }
    };
    issue(wg, 0);
    issue(wg + TW, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    float aw[2][16], ab[2][16];   // LayerNorm weight / bias gradient partials of this lane's channels 32 t + 8 g + 4 kh + i
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) aw[t][r] = ab[t][r] = 0.f;

    int s = 0;
    for (int64_t gi = wg; gi < ng; gi += TW) {
        unsigned char* const sl = wb + s * B_SLOT;
        if constexpr (!TAIL) {
        // ---- dg^T = wT5 dout^T ----
        bf16x8 df[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) df[ks] = *reinterpret_cast<const bf16x8*>(sl + B_DO + fr * 128 + (((2 * ks + kh) ^ (fr & 7)) * 16));
        floatx16 dg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dg[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) dg[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W5f[j][ks], df[ks], dg[j], 0, 0, 0);
        // ---- SimpleGate backward in place: dv1 = dg * v2, dv2 = dg * v1 ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* const a1 = sl + B_V + fr * 256 + (((4 * j + g) ^ (fr & 15)) * 16) + kh * 8;
                unsigned char* const a2 = sl + B_V + fr * 256 + (((8 + 4 * j + g) ^ (fr & 15)) * 16) + kh * 8;
                const float4 v1 = bf4_unpack(*reinterpret_cast<const u32x2*>(a1)), v2 = bf4_unpack(*reinterpret_cast<const u32x2*>(a2));
                const float4 d = make_float4(dg[j][4 * g + 0], dg[j][4 * g + 1], dg[j][4 * g + 2], dg[j][4 * g + 3]);
                *reinterpret_cast<u32x2*>(a1) = bf4_pack(f4_mul(d, v2));
                *reinterpret_cast<u32x2*>(a2) = bf4_pack(f4_mul(d, v1));
            }
        lds_fence();
        }
        // ---- dxn2^T = wT4 dv^T ----
        floatx16 dx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sl + B_V + fr * 256 + (((2 * ks + kh) ^ (fr & 15)) * 16));
#pragma unroll
            for (int j = 0; j < 2; ++j) dx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W4f[j][ks], vf, dx[j], 0, 0, 0);
        }
        // ---- LayerNorm2 backward for pixel fr: this lane holds channels 32 t + 8 g + 4 kh + i, the other half of the row is in lane ^ 32 ----
        float xh[2][16];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 yv = bf4_unpack(*reinterpret_cast<const u32x2*>(sl + B_Y + fr * 128 + (((4 * t + g) ^ (fr & 7)) * 16) + kh * 8));
                xh[t][4 * g + 0] = yv.x; xh[t][4 * g + 1] = yv.y; xh[t][4 * g + 2] = yv.z; xh[t][4 * g + 3] = yv.w;
                sum += (yv.x + yv.y) + (yv.z + yv.w);
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xh[t][r] -= mean;
                sq += xh[t][r] * xh[t][r];
            }
        sq += __shfl_xor(sq, 32);
        const float rs = 1.0f / sqrtf(sq * (1.0f / C) + p.eps);
        float s1 = 0.f, s2 = 0.f;
        const bool rowok = gi * 32 + fr < p.M;   // (rows past M: dout = v = 0 -> dx = 0, but xhat is not: keep them out of the column sums)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 lw = *reinterpret_cast<const float4*>(tab + 32 * t + 8 * g + 4 * kh);
                const float lws[4] = {lw.x, lw.y, lw.z, lw.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    xh[t][r] *= rs;
                    const float gx = dx[t][r];
                    if (rowok) {
                        aw[t][r] += gx * xh[t][r];
                        ab[t][r] += gx;
                    }
                    const float gw = gx * lws[i];
                    dx[t][r] = gw;
                    s1 += gw;
                    s2 += gw * xh[t][r];
                }
            }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        s1 *= (1.0f / C);
        s2 *= (1.0f / C);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* const a = sl + B_DO + fr * 128 + (((4 * t + g) ^ (fr & 7)) * 16) + kh * 8;
                const float4 dres = bf4_unpack(*reinterpret_cast<const u32x2*>(a));
                float4 o;
                o.x = rs * (dx[t][4 * g + 0] - xh[t][4 * g + 0] * s2 - s1) + dres.x;
                o.y = rs * (dx[t][4 * g + 1] - xh[t][4 * g + 1] * s2 - s1) + dres.y;
                o.z = rs * (dx[t][4 * g + 2] - xh[t][4 * g + 2] * s2 - s1) + dres.z;
                o.w = rs * (dx[t][4 * g + 3] - xh[t][4 * g + 3] * s2 - s1) + dres.w;
                *reinterpret_cast<u32x2*>(a) = bf4_pack(o);
            }
        lds_fence();
        // ---- outputs back into registers as full rows, slot re-staged, counted wait, stores ----
        u32x4 ody[4], odv[8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            ody[it] = *reinterpret_cast<const u32x4*>(sl + B_DO + row * 128 + ((ch ^ (row & 7)) * 16));
        }
        if (!TAIL && p.dv) {   // (no dv for a caller whose conv4 weight gradient comes from ffn_wgrad_bf16)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4), ch = lane & 15;
                odv[it] = *reinterpret_cast<const u32x4*>(sl + B_V + row * 256 + ((ch ^ (row & 15)) * 16));
            }
        }
        lds_fence();
        issue(gi + 2 * TW, s);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        const int64_t r0 = gi * 32;
        const rsrc_t rsO = make_rsrc(p.dy + r0 * C), rsV = make_rsrc(((TAIL || !p.dv) ? p.dy : p.dv) + r0 * (TAIL ? C : 2 * C));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            __builtin_amdgcn_raw_buffer_store_b128(ody[it], rsO, r0 + row < p.M ? (uint32_t)(row * 128 + ch * 16) : ROW_SENT, 0, 0);
        }
        if (!TAIL && p.dv) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4), ch = lane & 15;
                __builtin_amdgcn_raw_buffer_store_b128(odv[it], rsV, r0 + row < p.M ? (uint32_t)(row * 256 + ch * 16) : ROW_SENT, 0, 0);
            }
        }
        s ^= 1;
    }
    dma_wait_all();
    // column sums over this wave's pixels: the 32 pixel lanes of each half (fixed butterfly order: deterministic)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = aw[t][r], b = ab[t][r];
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (fr == 0) {
                const int c = 32 * t + 8 * (r >> 2) + 4 * kh + (r & 3);
                p.lnpart[(wg * 2 + 0) * C + c] = a;
                p.lnpart[(wg * 2 + 1) * C + c] = b;
            }
        }
}


// ---- weight gradients of the same half, accumulated where the operands are recomputed ---------------------------------------------------
// conv5:  G5[n][k] = sum_m dout[m][n] g[m][k]   (g = SimpleGate(v));   conv4:  G4[j][k] = sum_m dv[m][j] LN2(y)[m][k]
// Both contract over PIXELS.  With the TN GEMM kernel they cost 5 tensor passes of reads plus the 5 that wrote g, LN2(y) and dv first; here
// a wave recomputes the three operands for its 32 pixels from (dout, v, y) -- the SimpleGate backward and the LayerNorm exactly as the
// data-path kernel does them, bf16-rounded as its MFMA operands are -- leaves them in LDS in place of their inputs, and feeds them to the
// matrix pipe through the transposing LDS read (ds_read_b64_tr_b16: 8 consecutive pixels of one column per lane).  The 64 x 64 and
// 128 x 64 products stay in the wave's accumulators (192 registers) over ALL its groups; the column sums (bias gradients) fall out of
// the A fragments (8 pixels of a column per lane).  One fp32 slab per wave for the fixed-order reducer.  No stores in the loop: the ring's
// vmcnt counts are exact.
constexpr int W_G = 2 * B_SLOT, W_WAVE = 2 * B_SLOT + 4096;   // per wave: two (dout | v | y) slots + the gate buffer
constexpr int W_TAB = FW * W_WAVE;                             // fp32 tables: lnw[64] | lnb[64]

typedef __attribute__((address_space(3))) bf16x4* lds_tr_p;
// 8 consecutive pixels mrow .. mrow + 7 of column `col` of a row-major bf16 tile (row pitch PITCH bytes, 16-byte chunk c of row r at chunk
// position c ^ (r & SW)): lane t of a 16-lane group supplies the address of row mrow + (t >> 2), columns col16 + 4 (t & 3) .. + 3 and
// receives column col16 + t
template <int PITCH, int SW>
__device__ __forceinline__ bf16x8 tr_cols(const unsigned char* tile, int mrow, int col16, int t) {
    const int r0 = mrow + (t >> 2), r1 = r0 + 4;
    const int c = col16 + 4 * (t & 3);
    const int chunk = c >> 3, within = (c & 7) * 2;
    const unsigned char* a0 = tile + r0 * PITCH + ((chunk ^ (r0 & SW)) << 4) + within;
    const unsigned char* a1 = tile + r1 * PITCH + ((chunk ^ (r1 & SW)) << 4) + within;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(const_cast<unsigned char*>(a0)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(const_cast<unsigned char*>(a1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float sum8(bf16x8 v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    return ((bf_lo(w.x) + bf_hi(w.x)) + (bf_lo(w.y) + bf_hi(w.y))) + ((bf_lo(w.z) + bf_hi(w.z)) + (bf_lo(w.w) + bf_hi(w.w)));
}

template <int C>
__global__ __launch_bounds__(256) void ffn_wgrad_bf16_kernel(const FfnWgB p) {
    static_assert(C == 64, "ffn_wgrad_bf16: C = 64");
    __shared__ __attribute__((aligned(16))) unsigned char smem[W_TAB + 128 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tab = reinterpret_cast<float*>(smem + W_TAB);
    if (tid < 128) tab[tid] = tid < 64 ? p.lnw[tid] : p.lnb[tid - 64];
    const int fr = lane & 31, kh = lane >> 5;
    bf16x8 W5f[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) W5f[j][ks] = *reinterpret_cast<const bf16x8*>(p.wT5 + (32 * j + fr) * C + 16 * ks + 8 * kh);
    __syncthreads();

    unsigned char* const wb = smem + wave * W_WAVE;
    const uint32_t wb_lds = lds_addr(reinterpret_cast<const float*>(wb));
    const int64_t ng = (p.M + 31) / 32;
    const int64_t wg = (int64_t)blockIdx.x * FW + wave, TW = (int64_t)gridDim.x * FW;
    const uint32_t voff128 = (uint32_t)(lane >> 3) * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    auto issue = [&](int64_t gi, int s) {
        const int64_t gg = gi < ng ? gi : 0;
        const i32x4 rsD = make_rsrc_dma(p.dout + gg * (32 * C)), rsY = make_rsrc_dma(p.y + gg * (32 * C)), rsV = make_rsrc_dma(p.v + gg * (64 * C));
        const uint32_t base = wb_lds + (uint32_t)(s * B_SLOT);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const bool ok = gi < ng && gi * 32 + 8 * pc + (lane >> 3) < p.M;
            const uint32_t vo = ok ? voff128 + (uint32_t)pc * 1024u : ROW_SENT;
            dma16(rsD, base + (uint32_t)(B_DO + pc * 1024), vo, 0);
            dma16(rsY, base + (uint32_t)(B_Y + pc * 1024), vo, 0);
        }
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int row = 4 * pc + (lane >> 4);
            const bool ok = gi < ng && gi * 32 + row < p.M;
            dma16(rsV, base + (uint32_t)(B_V + pc * 1024), ok ? (uint32_t)(row * 256 + (((lane & 15) ^ (row & 15)) * 16)) : ROW_SENT, 0);
        }
    };
    issue(wg, 0);
    issue(wg + TW, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");

    floatx16 G5[2][2], G4[4][2];   // [n / j tile][k tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) G5[a][b][r] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) G4[a][b][r] = 0.f;
    float cs5[2] = {0.f, 0.f}, cs4[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of dout / dv at column 32 a + (lane & 31), this lane's pixel half

    const int t16 = lane & 15, c16 = 16 * ((lane >> 4) & 1);
    int s = 0;
    for (int64_t gi = wg; gi < ng; gi += TW) {
        unsigned char* const sl = wb + s * B_SLOT;
        unsigned char* const gb = wb + W_G;
        const bool rowok = gi * 32 + fr < p.M;
        // ---- LN2(y) in place (rows past M: zeros, so that they add nothing to G4) ----
        {
            float xh[2][16];
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 yv = bf4_unpack(*reinterpret_cast<const u32x2*>(sl + B_Y + fr * 128 + (((4 * t + g) ^ (fr & 7)) * 16) + kh * 8));
                    xh[t][4 * g + 0] = yv.x; xh[t][4 * g + 1] = yv.y; xh[t][4 * g + 2] = yv.z; xh[t][4 * g + 3] = yv.w;
                    sum += (yv.x + yv.y) + (yv.z + yv.w);
                }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xh[t][r] -= mean;
                    sq += xh[t][r] * xh[t][r];
                }
            sq += __shfl_xor(sq, 32);
            const float rs = rowok ? 1.0f / sqrtf(sq * (1.0f / C) + p.eps) : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 lw = *reinterpret_cast<const float4*>(tab + 32 * t + 8 * g + 4 * kh);
                    const float4 lb = *reinterpret_cast<const float4*>(tab + 64 + 32 * t + 8 * g + 4 * kh);
                    float4 o;
                    o.x = rowok ? xh[t][4 * g + 0] * rs * lw.x + lb.x : 0.f;
                    o.y = rowok ? xh[t][4 * g + 1] * rs * lw.y + lb.y : 0.f;
                    o.z = rowok ? xh[t][4 * g + 2] * rs * lw.z + lb.z : 0.f;
                    o.w = rowok ? xh[t][4 * g + 3] * rs * lw.w + lb.w : 0.f;
                    *reinterpret_cast<u32x2*>(sl + B_Y + fr * 128 + (((4 * t + g) ^ (fr & 7)) * 16) + kh * 8) = bf4_pack(o);
                }
        }
        // ---- dg^T = wT5 dout^T, SimpleGate backward in place, the gate ----
        {
            bf16x8 df[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) df[ks] = *reinterpret_cast<const bf16x8*>(sl + B_DO + fr * 128 + (((2 * ks + kh) ^ (fr & 7)) * 16));
            floatx16 dg[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dg[j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) dg[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W5f[j][ks], df[ks], dg[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned char* const a1 = sl + B_V + fr * 256 + (((4 * j + g) ^ (fr & 15)) * 16) + kh * 8;
                    unsigned char* const a2 = sl + B_V + fr * 256 + (((8 + 4 * j + g) ^ (fr & 15)) * 16) + kh * 8;
                    const float4 v1 = bf4_unpack(*reinterpret_cast<const u32x2*>(a1)), v2 = bf4_unpack(*reinterpret_cast<const u32x2*>(a2));
                    const float4 d = make_float4(dg[j][4 * g + 0], dg[j][4 * g + 1], dg[j][4 * g + 2], dg[j][4 * g + 3]);
                    *reinterpret_cast<u32x2*>(a1) = bf4_pack(f4_mul(d, v2));
                    *reinterpret_cast<u32x2*>(a2) = bf4_pack(f4_mul(d, v1));
                    *reinterpret_cast<u32x2*>(gb + fr * 128 + (((4 * j + g) ^ (fr & 7)) * 16) + kh * 8) = bf4_pack(f4_mul(v1, v2));
                }
        }
        lds_fence();
        // ---- the two pixel contractions: A = 8 consecutive pixels of a dout / dv column, B = of a gate / LN2(y) column ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int mrow = 16 * ks + 8 * kh;
            bf16x8 Yg[2], Yx[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                Yg[b] = tr_cols<128, 7>(gb, mrow, 32 * b + c16, t16);
                Yx[b] = tr_cols<128, 7>(sl + B_Y, mrow, 32 * b + c16, t16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16x8 X = tr_cols<128, 7>(sl + B_DO, mrow, 32 * a + c16, t16);
                cs5[a] += sum8(X);
#pragma unroll
                for (int b = 0; b < 2; ++b) G5[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Yg[b], G5[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x8 X = tr_cols<256, 15>(sl + B_V, mrow, 32 * a + c16, t16);
                cs4[a] += sum8(X);
#pragma unroll
                for (int b = 0; b < 2; ++b) G4[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Yx[b], G4[a][b], 0, 0, 0);
            }
        }
        lds_fence();
        issue(gi + 2 * TW, s);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        s ^= 1;
    }
    dma_wait_all();
    // ---- this wave's slabs: G[tile row i = (r & 3) + 8 (r >> 2) + 4 kh][tile column lane & 31] ----
    float* const o5 = p.g5 + wg * (int64_t)(C * C);
    float* const o4 = p.g4 + wg * (int64_t)(2 * C * C);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o5[(32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * C + 32 * b + fr] = G5[a][b][r];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o4[(32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh) * C + 32 * b + fr] = G4[a][b][r];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float t = cs5[a] + __shfl_xor(cs5[a], 32);
        if (kh == 0) p.cs5[wg * C + 32 * a + fr] = t;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float t = cs4[a] + __shfl_xor(cs4[a], 32);
        if (kh == 0) p.cs4[wg * (2 * C) + 32 * a + fr] = t;
    }
}

}  // namespace

bool ffn_fwd_bf16_ok(int C) { return C == 64; }

int launch_ffn_fwd_bf16(const FfnFwdB& p, int C, hipStream_t s) {
    trace_tag("ffn64.fwd");
    DCPT_CHECK_ARG(ffn_fwd_bf16_ok(C), "ffn_fwd_bf16: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.y && p.out && p.W4 && p.W5 && p.b4 && p.b5 && p.gamma && p.lnw && p.lnb && p.M > 0, "ffn_fwd_bf16: null argument");
    DCPT_CHECK_ARG((p.xn2 == nullptr) == (p.g == nullptr) && (p.mu == nullptr) == (p.rstd == nullptr), "ffn_fwd_bf16: xn2 / g and mu / rstd come in pairs");
    const int64_t ng = cdiv64(p.M, 32);
    int64_t blocks = cdiv64(ng, FW);
    if (blocks > 256) blocks = 256;   // one block per CU, persistent over its groups
    ffn_fwd_bf16_kernel<64, 0><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ffn_fwd_bf16");
    return DCPT_OK;
}

// t1 = conv1(LayerNorm1(inp)) + b1 in one pass: p.y = inp, p.W4 / p.b4 = conv1, p.v = t1 [M][2C], p.xn2 = LN1(inp) (kept for conv1's weight
// gradient), p.mu / p.rstd optional
int launch_ln_conv_bf16(const FfnFwdB& p, int C, hipStream_t s) {
    trace_tag("ffn64.ln_conv");
    DCPT_CHECK_ARG(ffn_fwd_bf16_ok(C), "ln_conv_bf16: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.y && p.v && p.xn2 && p.W4 && p.b4 && p.lnw && p.lnb && p.M > 0, "ln_conv_bf16: null argument");
    int64_t blocks = cdiv64(cdiv64(p.M, 32), FW);
    if (blocks > 256) blocks = 256;
    ffn_fwd_bf16_kernel<64, 1><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ln_conv_bf16");
    return DCPT_OK;
}

int ffn_bwd_bf16_waves(int64_t M) {
    int64_t blocks = cdiv64(cdiv64(M, 32), FW);
    if (blocks > 256) blocks = 256;
    return (int)blocks * FW;
}

int launch_ffn_bwd_bf16(const FfnBwdB& p, int C, hipStream_t s) {
    trace_tag("ffn64.bwd");
    DCPT_CHECK_ARG(ffn_fwd_bf16_ok(C), "ffn_bwd_bf16: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.dout && p.v && p.y && p.wT5 && p.wT4 && p.lnw && p.dy && p.lnpart && p.M > 0, "ffn_bwd_bf16: null argument");
    ffn_bwd_bf16_kernel<64, 0><<<dim3((unsigned)(ffn_bwd_bf16_waves(p.M) / FW)), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ffn_bwd_bf16");
    return DCPT_OK;
}

// dx = dres + LayerNorm'(dz W; x) in one pass: p.v = dz [M][2C], p.wT4 = W^T [C][2C], p.y = x, p.dout = dres, p.dy = dx, p.lnpart as above
int launch_conv_ln_bwd_tail_bf16(const FfnBwdB& p, int C, hipStream_t s) {
    trace_tag("ffn64.bwd_tail");
    DCPT_CHECK_ARG(ffn_fwd_bf16_ok(C), "conv_ln_bwd_tail_bf16: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.dout && p.v && p.y && p.wT4 && p.lnw && p.dy && p.lnpart && p.M > 0, "conv_ln_bwd_tail_bf16: null argument");
    ffn_bwd_bf16_kernel<64, 1><<<dim3((unsigned)(ffn_bwd_bf16_waves(p.M) / FW)), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("conv_ln_bwd_tail_bf16");
    return DCPT_OK;
}

int launch_ffn_wgrad_bf16(const FfnWgB& p, int C, hipStream_t s) {
    trace_tag("ffn64.wgrad");
    DCPT_CHECK_ARG(ffn_fwd_bf16_ok(C), "ffn_wgrad_bf16: C=%d not supported (64)", C);
    DCPT_CHECK_ARG(p.dout && p.v && p.y && p.wT5 && p.lnw && p.lnb && p.g5 && p.g4 && p.cs5 && p.cs4 && p.M > 0, "ffn_wgrad_bf16: null argument");
    ffn_wgrad_bf16_kernel<64><<<dim3((unsigned)(ffn_bwd_bf16_waves(p.M) / FW)), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("ffn_wgrad_bf16");
    return DCPT_OK;
}
