// fp32-class NT GEMM on the bf16 matrix pipe by SPLIT OPERANDS ("bf16x3"):  C[m][n] = sum_k A[m][k] * Bw[n][k]  with fp32 A, Bw, C.
//
// An fp32 value is the exact sum of three bfloat16 pieces obtained by truncation, x = x0 + x1 + x2 (x0 = the top 8 significand bits,
// x1 = the top 8 of the exact remainder x - x0, x2 = the exact rest: 24 bits in all).  A product a*b is then nine piece products, each of
// them EXACT in fp32 (8 x 8 significand bits); the kernel keeps the six with i + j <= 2,
//     a*b ~= a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0,
// the dropped ones (a1 b2, a2 b1, a2 b2) are <= 2^-24 |a b| each, i.e. below the rounding error an fp32 FMA makes on the full product.
// Sums are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The matrix pipe runs bf16 at 16x the fp32-MFMA rate, so six bf16 MFMAs
// replace eight fp32 MFMAs of a k = 16 block at 3/8 of their issue time -- an fp32-CLASS GEMM (not bit-identical to the fp32-MFMA path:
// the summation order and the dropped 2^-24 terms differ; tests/test_gpu_x3.py measures both paths against fp64).  This is a SEPARATELY
// REPORTED mode (dcpt_set_gemm_x3): the fp32-MFMA kernels stay the product default and the headline.
//
// Data path.  WEIGHTS are rewritten ("split") once per launch into a tiled image of 12-KB pieces -- piece (row block of 128, k-tile of 16) =
// 3 planes x [128 rows][16 k] bf16, the two 16-byte halves of a row swapped on rows with bit 3 set -- which is byte for byte what the GEMM
// keeps in LDS, so a B half-tile is ONE contiguous 24-KB LDS-DMA read.  ACTIVATIONS are NOT rewritten (a separate split pass over every A
// operand cost 12 ms per training step: 4 B read + 6 B written per element): the A half-tiles go HBM -> LDS as fp32 by LDS-DMA in the fp32
// kernel's own swizzled [128 rows][32 floats] image, a wave reads its fragment (8 consecutive k of a row: two ds_read_b128) and splits it
// IN REGISTERS (~44 VALU operations per fragment, issued while the partner wave of the SIMD runs its MFMAs).  The GEMM is the 256 x 256
// kernel of gemm_bf16_256.hip (two alternating wave groups, half-tile ring of two k-tiles = the CU's whole 160 KB, counted vmcnt,
// interleaved wave outputs) with a k-tile of 32 walked as two 16-k sub-steps: eight phases of 12 MFMAs; the epilogue is the fp32 kernel's
// own (gemm_nt_epi.h).
#include "gemm_nt_epi.h"
#include "bf16.h"
#include "prof.h"

namespace {

constexpr int PIECE = 12288;       // bytes of a piece: 3 planes x 128 rows x 32 B
constexpr int PLANE = 4096;
constexpr int STG3 = 4 * PIECE;    // one k-tile in LDS: A-lo | A-hi | B-lo | B-hi
constexpr int NSTG = 3;

// ---- split: fp32 [R][K] (row stride ld) -> tiled 3-plane image [ceil(R / 128)][K / 16][3][128][16] bf16 ------------------------------
// A block converts a [128 rows][64 k] tile: coalesced 256-B row reads, split in registers, pieces assembled in LDS, linear 16-B writes.
// grid.z = batch: problem z reads the SAME X (weights) with scale row z of kscale and writes image z (the per-image SCA-scaled conv3 weights).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ X, int64_t R, int K, int ld, unsigned char* __restrict__ out,
                                                     const float* __restrict__ kscale, int64_t img_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[4 * PIECE];
    const int tid = threadIdx.x;
    const int64_t rb = blockIdx.y;
    const int kq = blockIdx.x;                 // group of four k-tiles
    const int KT = K / 16;
    const int nkt = (KT - 4 * kq) < 4 ? (KT - 4 * kq) : 4;
    // thread -> (row, 16-B column chunk): 16 chunks of 4 floats per row of 64 k; 16 rows per pass, 8 passes
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int r = ps * 16 + (tid >> 4), c4 = tid & 15;
        const int64_t row = rb * 128 + r;
        const int k = 64 * kq + 4 * c4;
        float4 v = f4_zero();
        if (row < R && k < K) {
            v = ldg4(X + row * (int64_t)ld + k);
            if (kscale) v = f4_mul(v, ldg4(kscale + (int64_t)blockIdx.z * K + k));
        }
        const float xs[4] = {v.x, v.y, v.z, v.w};
        uint32_t pc[3][2];   // plane -> two packed bf16 pairs
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t b0 = __builtin_bit_cast(uint32_t, xs[e]) & 0xffff0000u;
            const float r1 = xs[e] - __builtin_bit_cast(float, b0);
            const uint32_t b1 = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, b1);
            const uint32_t b2 = __builtin_bit_cast(uint32_t, r2) & 0xffff0000u;   // (r2 has <= 8 significant bits: exact)
            const uint32_t bs[3] = {b0, b1, b2};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (e & 1) pc[p][e >> 1] |= bs[p];
                else pc[p][e >> 1] = bs[p] >> 16;
            }
        }
        // element k -> k-tile kt = c4 / 4, position 4 (c4 % 4) .. +3 inside the tile: half kh = (c4 % 4) / 2, 8-byte piece (c4 % 2)
        const int kt = c4 >> 2, kh = (c4 >> 1) & 1, q = c4 & 1;
        unsigned char* dst = sm + kt * PIECE + r * 32 + ((kh ^ ((r >> 3) & 1)) * 16) + q * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            u32x2 w;
            w.x = pc[p][0];
            w.y = pc[p][1];
            *reinterpret_cast<u32x2*>(dst + p * PLANE) = w;
        }
    }
    __syncthreads();
    unsigned char* o = out + (int64_t)blockIdx.z * img_bytes + ((rb * KT + 4 * kq) * (int64_t)PIECE);
    for (int i = tid; i < nkt * (PIECE / 16); i += 256) *reinterpret_cast<u32x4*>(o + (int64_t)i * 16) = *reinterpret_cast<const u32x4*>(sm + i * 16);
}

typedef __attribute__((address_space(3))) const volatile bf16x8* lds_frag_p;
typedef __attribute__((address_space(3))) const volatile floatx4* lds_f4_p;

struct GemmX3 {
    GemmNT g;                   // the fp32 problem: A is read as it is (fp32, LDS-DMA), Bw through its split image
    const unsigned char* B3;    // tiled split image of Bw [N / 128][K / 16] pieces (per batch problem: + b * sB3 bytes)
    int64_t sB3;
};

// 8 fp32 values -> their three bf16 pieces (truncating splits, exact: x = hi + mid + lo), packed as MFMA operands
__device__ __forceinline__ void split8(const floatx4 lo4, const floatx4 hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    u32x4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // the upper halves of a pair of words, packed {even element: low half, odd element: high half}, are one v_perm_b32
        const uint32_t a0 = __builtin_bit_cast(uint32_t, x[2 * i]), a1 = __builtin_bit_cast(uint32_t, x[2 * i + 1]);
        w0[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        const float r0 = x[2 * i] - __builtin_bit_cast(float, a0 & 0xffff0000u);
        const float r1 = x[2 * i + 1] - __builtin_bit_cast(float, a1 & 0xffff0000u);
        const uint32_t b0 = __builtin_bit_cast(uint32_t, r0), b1 = __builtin_bit_cast(uint32_t, r1);
        w1[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xffff0000u);   // <= 8 significant bits left: its low half is zero
        const float s1 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
        w2[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s0), 0x07060302u);
    }
    p0 = __builtin_bit_cast(bf16x8, w0);
    p1 = __builtin_bit_cast(bf16x8, w1);
    p2 = __builtin_bit_cast(bf16x8, w2);
}

// one pair of fp32 values -> one dword of each piece operand (the same arithmetic as split8)
__device__ __forceinline__ void split_pair(const float x0, const float x1, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    const uint32_t a0 = __builtin_bit_cast(uint32_t, x0), a1 = __builtin_bit_cast(uint32_t, x1);
    w0 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    const float r0 = x0 - __builtin_bit_cast(float, a0 & 0xffff0000u);
    const float r1 = x1 - __builtin_bit_cast(float, a1 & 0xffff0000u);
    const uint32_t b0 = __builtin_bit_cast(uint32_t, r0), b1 = __builtin_bit_cast(uint32_t, r1);
    w1 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xffff0000u);
    const float s1 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
    w2 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s0), 0x07060302u);
}

// ---- epilogue straight from the accumulators (no LDS parking) ------------------------------------------------------------------------
// A 32 x 32 accumulator tile holds, on lane l, column l & 31 and the rows (r & 3) + 8 (r >> 2) + 4 (l >> 5): four registers r = 4 g .. 4 g + 3
// are four consecutive ROWS of one column.  A 4 x 4 transpose over the four lanes of a quad (two DPP butterfly steps, 16 VALU operations
// per four registers) turns them into four consecutive COLUMNS of one row -- a float4 at (row 8 g + 4 h + q, columns 4 cq .., q = l & 3,
// cq = (l & 31) >> 2, h = l >> 5) -- which is what the row epilogues of the fp32 kernel compute on: 16-byte loads and stores, every
// instruction covering eight rows x 128 contiguous bytes (whole cache lines).  The tile never visits LDS and no barrier separates the
// loop from the first store.  NOT the default (-DX3_EPI_REGS selects it; parity-green, tools/x3_check.sh): measured equal to the LDS-parked
// row epilogue on every level-3 launch (plain 169.9 vs 167.1 us, the step 97.8 vs 96.8 ms) -- the ~40 us a launch spends after its
// k-loop are not the parking (9 us when ablated alone) but the drain of 67-134 MB of output through an L2 whose clock the loop has just
// pulled down to 1.35 GHz (profiles/r3/power_and_clock.txt: the same epilogue costs 9 us when the loop is compiled out).
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float4 quad_transpose(float r0, float r1, float r2, float r3, bool b0, bool b1) {
    // butterfly over lane bit 0 (quad_perm [1,0,3,2]) then bit 1 ([2,3,0,1]): new[i][q] = (q_b == i_b) ? old[i][q] : old[i ^ b][q ^ b]
    const float s0 = dpp_quad<0xB1>(r1), s1 = dpp_quad<0xB1>(r0), s2 = dpp_quad<0xB1>(r3), s3 = dpp_quad<0xB1>(r2);
    const float n0 = b0 ? s0 : r0, n1 = b0 ? r1 : s1, n2 = b0 ? s2 : r2, n3 = b0 ? r3 : s3;
    const float t0 = dpp_quad<0x4E>(n2), t2 = dpp_quad<0x4E>(n0), t1 = dpp_quad<0x4E>(n3), t3 = dpp_quad<0x4E>(n1);
    return make_float4(b1 ? t0 : n0, b1 ? t1 : n1, b1 ? n2 : t2, b1 ? n3 : t3);
}

// one wave's 32 rows x 256 columns (acc[j]: columns 32 j ..): rows mw .. mw + 31 of the problem, columns n0 + 32 j (plain) resp. the two
// gate halves n0h + 32 j and Ch + n0h + 32 (j - 4) (E_BIASGATE).  red: 1 KiB of LDS per wave for the column sums of E_DOTCOL.
template <int EK>
__device__ __forceinline__ void epilogue_regs(const GemmNT& p, floatx16 (&acc)[8], int64_t mw, int n0, int lane, float* __restrict__ red) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const int q = lane & 3, cq = (lane & 31) >> 2, h = lane >> 5;
    const int ldres = p.ldres ? p.ldres : p.ldc;
    const int64_t mb = mw < p.M ? mw : 0;
    const rsrc_t rsC = make_rsrc(p.C + mb * (int64_t)p.ldc);
    uint32_t rowC[4], rowR[4], rowX[4];   // byte offsets of this lane's four rows (g = 0..3) in C / res / aux, ROW_SENT past M
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rl = 8 * g + 4 * h + q;
        const bool ok = mw + rl < p.M;
        rowC[g] = ok ? (uint32_t)rl * (uint32_t)p.ldc * 4u : ROW_SENT;
        rowR[g] = ok ? (uint32_t)rl * (uint32_t)ldres * 4u : ROW_SENT;
        rowX[g] = ok ? (uint32_t)rl * (uint32_t)p.N * 8u : ROW_SENT;
    }
    if constexpr (EK == E_BIASGATE) {
        const int Ch = p.N / 2;
        const rsrc_t rsG = make_rsrc(p.gate + mb * (int64_t)Ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 32 * j + 4 * cq;
            float4 bA = f4_zero(), bB = f4_zero();
            if (p.bias) {
                bA = ldg4(p.bias + n);
                bB = ldg4(p.bias + Ch + n);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v1 = f4_add(quad_transpose(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], b0, b1), bA);
                const float4 v2 = f4_add(quad_transpose(acc[j + 4][4 * g], acc[j + 4][4 * g + 1], acc[j + 4][4 * g + 2], acc[j + 4][4 * g + 3], b0, b1), bB);
                buf_st4(rsC, rowC[g] + 4u * (uint32_t)n, v1);
                buf_st4(rsC, rowC[g] + 4u * (uint32_t)(Ch + n), v2);
                const bool ok = rowC[g] != ROW_SENT;
                buf_st4(rsG, ok ? ((uint32_t)(8 * g + 4 * h + q) * (uint32_t)Ch + (uint32_t)n) * 4u : ROW_SENT, f4_mul(v1, v2));
            }
        }
    } else {
        rsrc_t rsR = rsC, rsX = rsC;
        if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_MUL || EK == E_DOTCOL) rsR = make_rsrc(p.res + mb * (int64_t)ldres);
        if constexpr (EK == E_SGBWD) rsX = make_rsrc(p.aux + mb * (2 * (int64_t)p.N));
        float4 dot[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + 32 * j + 4 * cq;
            const uint32_t nb = 4u * (uint32_t)n;
            float4 bias = f4_zero(), cs = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (EK == E_BIAS || EK == E_RESID || EK == E_MUL) {
                if (p.bias) bias = ldg4(p.bias + n);
            }
            if constexpr (EK == E_RESID || EK == E_ADDSCALED) {
                if (p.cscale) cs = ldg4(p.cscale + n);
            }
            float4 pre1[4], pre2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_MUL || EK == E_DOTCOL) pre1[g] = buf_ld4(rsR, rowR[g] == ROW_SENT ? ROW_SENT : rowR[g] + nb);
                if constexpr (EK == E_SGBWD) {
                    const uint32_t xo = rowX[g] == ROW_SENT ? ROW_SENT : rowX[g] + nb;
                    pre1[g] = buf_ld4(rsX, xo);
                    pre2[g] = buf_ld4(rsX, xo == ROW_SENT ? ROW_SENT : xo + 4u * (uint32_t)p.N);
                }
            }
            dot[j] = f4_zero();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = quad_transpose(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], b0, b1);
                const uint32_t o = rowC[g] == ROW_SENT ? ROW_SENT : rowC[g] + nb;
                if constexpr (EK == E_PLAIN) {
                    buf_st4(rsC, o, v);
                } else if constexpr (EK == E_BIAS) {
                    buf_st4(rsC, o, f4_add(v, bias));
                } else if constexpr (EK == E_RESID) {
                    buf_st4(rsC, o, f4_fma(f4_add(v, bias), cs, pre1[g]));
                } else if constexpr (EK == E_ADDSCALED) {
                    buf_st4(rsC, o, f4_fma(cs, pre1[g], v));
                } else if constexpr (EK == E_MUL) {
                    buf_st4(rsC, o, f4_mul(f4_add(v, bias), pre1[g]));
                } else if constexpr (EK == E_SGBWD) {
                    buf_st4(rsC, o, f4_mul(v, pre2[g]));
                    buf_st4(rsC, o == ROW_SENT ? ROW_SENT : o + 4u * (uint32_t)p.N, f4_mul(v, pre1[g]));
                } else {   // E_DOTCOL
                    buf_st4(rsC, o, v);
                    dot[j] = f4_fma(v, pre1[g], dot[j]);   // rows past M loaded 0
                }
            }
        }
        if constexpr (EK == E_DOTCOL) {
            // column sums over the wave's 32 rows: the lanes q = 0..3 of a quad and the two halves h hold different rows of the same
            // columns (fixed butterfly order); lane (q = 0, h = 0) of column group cq leaves the wave's partial in LDS
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 d = dot[j];
#pragma unroll
                for (int m = 1; m <= 2; m <<= 1) {
                    d.x += __shfl_xor(d.x, m); d.y += __shfl_xor(d.y, m); d.z += __shfl_xor(d.z, m); d.w += __shfl_xor(d.w, m);
                }
                d.x += __shfl_xor(d.x, 32); d.y += __shfl_xor(d.y, 32); d.z += __shfl_xor(d.z, 32); d.w += __shfl_xor(d.w, 32);
                if (q == 0 && h == 0) *reinterpret_cast<float4*>(red + 32 * j + 4 * cq) = d;
            }
        }
    }
}

// LDS of one k-tile of 32: A-lo | A-hi as fp32 [128 rows][32 floats] (the fp32 kernel's swizzled image, 16 KB each), B-lo | B-hi as two
// consecutive pieces each (24 KB each)
constexpr int AHT = 16384, BHT = 2 * PIECE, STGF = 2 * AHT + 2 * BHT;   // 80 KB: two stages are the whole 160 KB of a CU

template <int EK>
__global__ __launch_bounds__(512) void gemm_nt_x3_kernel(const GemmX3 pin) {
    constexpr bool GATE = (EK == E_BIASGATE);
    GemmNT p = pin.g;
    const unsigned char* B3 = pin.B3;
    if (gridDim.y > 1) {
        const int b1 = blockIdx.y / p.nb2, b2 = blockIdx.y % p.nb2;
        p.A += b1 * p.sA1 + b2 * p.sA2;
        p.C += b1 * p.sC1 + b2 * p.sC2;
        if (p.res) p.res += b1 * p.sR1 + b2 * p.sR2;
        if (p.cscale) p.cscale += b1 * p.sS1 + b2 * p.sS2;
        B3 += (int64_t)blockIdx.y * pin.sB3;
    }
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STGF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
#ifdef X3_ABL_CLOCK
    const uint64_t clk0_ = __builtin_readcyclecounter(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    // wave -> 32 rows x 256 columns of the tile (waves 0..3 in A-lo, 4..7 in A-hi; eight 32 x 32 tiles each): the in-register split of an A
    // fragment -- what this kernel's load phases are made of -- is done by exactly ONE wave per tile (two with 64 x 128 wave tiles: 193 vs
    // 215 us with four, 128 x 64); the B fragments are plain 16-byte records, reading each in eight waves costs LDS cycles that are free
    const int ah = wave >> 2, ar = (wave & 3) * 32;
    const int Ch = p.N / 2;
    const int tilesN = p.N / 256;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(lin / tilesN) * 256;
    const int n0 = (lin % tilesN) * (GATE ? 128 : 256);
    const int KT = p.K / 16;
    const int rbB[2] = {n0 / 128, (GATE ? Ch + n0 : n0 + 128) / 128};
    const i32x4 rsA = make_rsrc_dma(p.A + (m0 < p.M ? m0 : 0) * (int64_t)p.lda);
    const i32x4 rsB0 = make_rsrc_dma(B3 + (int64_t)rbB[0] * KT * PIECE);
    const i32x4 rsB1 = make_rsrc_dma(B3 + (int64_t)rbB[1] * KT * PIECE);
    // A half-tile = 16 DMAs of 8 rows (wave w: rows 8 w.. and 64 + 8 w..), lane -> row (lane >> 3), LDS slot lane & 7 = k-quad ^ ((row >> 1) & 7);
    // B half-tile = 24 linear KiB (wave w: KiB w, 8 + w, 16 + w)
    uint32_t voffA[2][2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int row = 8 * wave + (lane >> 3) + 64 * ps;
        const uint32_t ch = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 128 + row;
            voffA[h][ps] = (m0 + r < p.M) ? (uint32_t)r * (uint32_t)p.lda * 4u + ch : ROW_SENT;
        }
    }
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_base = lds_addr(reinterpret_cast<const float*>(smem));
    // X: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi; s: stage; t: k-tile of 32
    auto stage = [&](int X, int s, int t, uint32_t dead) {
        if (X < 2) {
            const uint32_t dst = lds_base + (uint32_t)(s * STGF + X * AHT) + (uint32_t)wave * 1024u;
            dma16(rsA, dst, voffA[X][0] | dead, (uint32_t)t * 128u);
            dma16(rsA, dst + 8192u, voffA[X][1] | dead, (uint32_t)t * 128u);
        } else {
            const uint32_t dst = lds_base + (uint32_t)(s * STGF + 2 * AHT + (X - 2) * BHT) + (uint32_t)wave * 1024u;
            const uint32_t src = ((uint32_t)t * BHT + (uint32_t)wave * 1024u + lane_off) | dead;
            const i32x4 rs = X == 2 ? rsB0 : rsB1;
            dma16(rs, dst, src, 0);
            dma16(rs, dst + 8192u, src + 8192u, 0);
            dma16(rs, dst + 16384u, src + 16384u, 0);
        }
    };

    // A fragment (fp32): row (lane & 31) of this wave's 32 rows; sub-step u (16 k), half fh = lane >> 5: the 8 consecutive k are k-quads
    // 4 u + 2 fh and + 1
    const int fr = lane & 31, fh = lane >> 5, fi = (fr >> 1) & 7;
    const unsigned char* abase = smem + ah * AHT + (ar + fr) * 128;
    int aslot[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) aslot[u][q] = ((4 * u + 2 * fh + q) ^ fi) * 16;
    // B fragments (split image): row fr of a 32-row tile, 16-byte half fh of the 32-byte row, swapped on rows with bit 3
    const unsigned char* bbase = smem + 2 * AHT + fr * 32 + ((fh ^ ((fr >> 3) & 1)) * 16);
    // one opaque base register per stage: every fragment read is then base + a 16-bit immediate (left to itself the compiler materialises
    // ~50 address registers for the offsets beyond 64 KB, which this kernel's register budget does not have)
    const uint32_t b0_ = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const void*)bbase);
    const uint32_t a0_ = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const void*)abase);
    uint32_t bst[2] = {b0_, b0_ + (uint32_t)STGF}, ast[2] = {a0_, a0_ + (uint32_t)STGF};
    asm volatile("" : "+v"(bst[0]), "+v"(bst[1]), "+v"(ast[0]), "+v"(ast[1]));

    floatx16 acc[8];   // n-tile j: columns 32 j.. of the tile (j < 4: B-lo, else B-hi)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // the wave's A fragment [sub-step parity][piece] -- the fragment of sub-step u + 1 is built (read one phase, split in quarters over the
    // next phases) while the MFMAs of sub-step u run --; the two B tiles of the current phase [tile][piece]
    bf16x8 fa[2][3], fb[2][3];
    uint32_t fw[3][4];  // the fragment under construction [piece][pair of k]
    floatx4 ra0, ra1;   // its eight fp32 values

    const int nkt = p.K / 32;   // (K % 64 == 0: k-tiles of 32 in pairs)
    // prologue: k-tile 0 completely, the A halves of k-tile 1
    stage(0, 0, 0, 0);
    stage(1, 0, 0, 0);
    stage(2, 0, 0, 0);
    stage(3, 0, 0, 0);
    stage(0, 1, 1, 0);
    stage(1, 1, 1, 0);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // A(0), B-lo(0) landed (this wave's part; the barrier publishes all)
    __builtin_amdgcn_s_barrier();
    ra0 = *(lds_f4_p)(uintptr_t)(ast[0] + aslot[0][0]);
    ra1 = *(lds_f4_p)(uintptr_t)(ast[0] + aslot[0][1]);
    split8(ra0, ra1, fa[0][0], fa[0][1], fa[0][2]);
    if (grp == 1) __builtin_amdgcn_s_barrier();

// raw fragment of sub-step U of stage S (two ds_read_b128: the 8 consecutive k of this lane's row)
#define DCPT_RD_A(S, U)                                                                                                            \
    ra0 = *(lds_f4_p)(uintptr_t)(ast[S] + aslot[U][0]);                                                                           \
    ra1 = *(lds_f4_p)(uintptr_t)(ast[S] + aslot[U][1]);
// split pair I (k = 2 I, 2 I + 1) of the raw fragment
#ifdef X3_ABL_NOSPLIT
#define DCPT_SPLIT_Q(I) fw[0][I] = fw[1][I] = fw[2][I] = __builtin_bit_cast(uint32_t, (I) < 2 ? ra0[2 * ((I)&1)] : ra1[2 * ((I)&1)]);
#else
#define DCPT_SPLIT_Q(I) split_pair((I) < 2 ? ra0[2 * ((I)&1)] : ra1[2 * ((I)&1)], (I) < 2 ? ra0[2 * ((I)&1) + 1] : ra1[2 * ((I)&1) + 1], fw[0][I], fw[1][I], fw[2][I]);
#endif
#define DCPT_PUT_A(PAR)                                                                                                            \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                                            \
        u32x4 w_;                                                                                                                 \
        w_.x = fw[pl][0]; w_.y = fw[pl][1]; w_.z = fw[pl][2]; w_.w = fw[pl][3];                                                     \
        fa[PAR][pl] = __builtin_bit_cast(bf16x8, w_);                                                                             \
    }
// the two B tiles of phase Q (columns 64 Q..): half Q / 2, rows 64 (Q % 2).. of that half-tile's piece of sub-step U
#define DCPT_LD_B(S, U, Q)                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                 \
        fb[j][pl] = *(lds_frag_p)(uintptr_t)(bst[S] + ((Q) >> 1) * BHT + (U)*PIECE + pl * PLANE + (((Q)&1) * 2 + j) * 1024);
#ifdef X3_ABL_NOMFMA
#define DCPT_MM(U, J, PA, PB) asm volatile("" ::"v"(fa[U][PA]), "v"(fb[(J)&1][PB]))
#else
#define DCPT_MM(U, J, PA, PB) acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[U][PA], fb[(J)&1][PB], acc[J], 0, 0, 0)
#endif
#define DCPT_MFMA(U, Q)                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                                \
    DCPT_MM(U, 2 * (Q), 2, 0); DCPT_MM(U, 2 * (Q) + 1, 2, 0);                                                                       \
    DCPT_MM(U, 2 * (Q), 0, 2); DCPT_MM(U, 2 * (Q) + 1, 0, 2);                                                                       \
    DCPT_MM(U, 2 * (Q), 1, 1); DCPT_MM(U, 2 * (Q) + 1, 1, 1);                                                                       \
    DCPT_MM(U, 2 * (Q), 1, 0); DCPT_MM(U, 2 * (Q) + 1, 1, 0);                                                                       \
    DCPT_MM(U, 2 * (Q), 0, 1); DCPT_MM(U, 2 * (Q) + 1, 0, 1);                                                                       \
    DCPT_MM(U, 2 * (Q), 0, 0); DCPT_MM(U, 2 * (Q) + 1, 0, 0);                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    __builtin_amdgcn_s_barrier();

// compile-time ablations (tools/build_variant.sh ... -DX3_ABL_*): never defined in the product build
#ifdef X3_ABL_NODMA
#define X3_STAGE(...)
#else
#define X3_STAGE(...) __VA_ARGS__
#endif
#ifdef X3_ABL_NOLDB
#define X3_LDB(...)
    DCPT_LD_B(0, 0, 0)
#else
#define X3_LDB(...) __VA_ARGS__
#endif
#ifdef X3_ABL_NOLDA
#define X3_LDA(...)
#else
#define X3_LDA(...) __VA_ARGS__
#endif
    // Eight phases g = 4 u + Q per k-tile of 32 (stage s = t & 1): sub-step u = 0, 1 x column quarters Q = 0..3, 12 MFMAs each.  The load
    // half of a phase is kept EVEN (the other wave of the SIMD runs 12 MFMAs = 384 cycles meanwhile; a phase with the whole 44-operation
    // split of a fragment in it took ~2x that and cost a quarter of the loop): six B fragment reads, one quarter of the NEXT sub-step's A
    // split (raw fragment read at Q = 0, pairs 0 / 1 / 2+3 split at Q = 1 / 2 / 3), at most three DMAs.
    //   reads     A(t) sub-step 1 raw: g0;  A(t+1) sub-step 0 raw: g4 (other stage);  B-lo(t): g0 1 4 5;  B-hi(t): g2 3 6 7
    //   staging   (>= two phases after the last read of the slot, LDS-DMA rule)  g0 B-lo(t+1), g1 B-hi(t+1) -> other stage;
    //             g2 A-lo(t+2), g3 A-hi(t+2) -> this stage
    //   waits     (read >= one phase after the wait)  g1 vmcnt(10): B-hi(t);  g3 vmcnt(10): A(t+1);  g7 vmcnt(7): B-lo(t+1)
#ifdef X3_ABL_NOLOOP
    for (int t = 0; t < 0; t += 2) {
#else
    for (int t = 0; t < nkt; t += 2) {
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kt = t + s;
            const uint32_t dead1 = (kt + 1 < nkt) ? 0u : ROW_SENT, dead2 = (kt + 2 < nkt) ? 0u : ROW_SENT;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // column quarter 0
                X3_LDB(DCPT_LD_B(s, u, 0))
                if (u == 0) {
                    X3_LDA(DCPT_RD_A(s, 1))
                    X3_STAGE(stage(2, s ^ 1, kt + 1, dead1));
                } else {
                    X3_LDA(DCPT_RD_A(s ^ 1, 0))
                }
                __builtin_amdgcn_s_barrier();
                DCPT_MFMA(u, 0)
                // quarter 1
                X3_LDB(DCPT_LD_B(s, u, 1))
                DCPT_SPLIT_Q(0)
                if (u == 0) {
                    X3_STAGE(stage(3, s ^ 1, kt + 1, dead1));
                    X3_STAGE(asm volatile("s_waitcnt vmcnt(10)" ::: "memory"));
                }
                __builtin_amdgcn_s_barrier();
                DCPT_MFMA(u, 1)
                // quarter 2
                X3_LDB(DCPT_LD_B(s, u, 2))
                DCPT_SPLIT_Q(1)
                if (u == 0) X3_STAGE(stage(0, s, kt + 2, dead2));
                __builtin_amdgcn_s_barrier();
                DCPT_MFMA(u, 2)
                // quarter 3
                X3_LDB(DCPT_LD_B(s, u, 3))
                DCPT_SPLIT_Q(2)
                DCPT_SPLIT_Q(3)
                DCPT_PUT_A(u ^ 1)
                if (u == 0) {
                    X3_STAGE(stage(1, s, kt + 2, dead2));
                    X3_STAGE(asm volatile("s_waitcnt vmcnt(10)" ::: "memory"));
                } else {
                    X3_STAGE(asm volatile("s_waitcnt vmcnt(7)" ::: "memory"));
                }
                __builtin_amdgcn_s_barrier();
                DCPT_MFMA(u, 3)
            }
        }
    }
#undef DCPT_RD_A
#undef DCPT_SPLIT_Q
#undef DCPT_PUT_A
#undef DCPT_LD_B
#undef DCPT_MM
#undef DCPT_MFMA
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
    dma_wait_all();
    __syncthreads();
#ifdef X3_ABL_CLOCK
    const uint64_t clk1_ = __builtin_readcyclecounter(), rt1_ = __builtin_amdgcn_s_memrealtime();
#endif

#ifdef X3_ABL_NOEPI
    {
        float keep = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[j][r];
        if (keep == 12345.678f) p.C[0] = keep;
        return;
    }
#endif
#ifndef X3_EPI_REGS
    // epilogue: the 128 x 128 sub-blocks (a, b) of the tile through LDS -> the fp32 kernel's row epilogues.  Rows of half a belong to waves
    // 4 a .. 4 a + 3 (32 rows each); every wave parks its own accumulators of the current half and both groups meet at the barriers.
    float* const Cs0 = reinterpret_cast<float*>(smem);
    if constexpr (GATE) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* const Cs = Cs0;   // [128][256]: columns 0..127 first gate half, 128..255 second
            if (ah == a) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int nl = j * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = ar + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[ml * 256 + nl] = acc[j][r];
                    }
                }
            }
            __syncthreads();
            epilogue_gate<128, 256, 512>(p, Cs, m0 + a * 128, n0, tid);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = q >> 1, b = q & 1;
            float* const Cs = Cs0 + (q & 1) * (128 * 128);
#ifndef X3_ABL_NOPARK
            if (ah == a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = j * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = ar + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[ml * 128 + nl] = acc[4 * b + j][r];
                    }
                }
            }
#else
            if (acc[4 * b][0] == 12345.678f) Cs[tid] = acc[4 * b + 1][1];
#endif
            __syncthreads();
#ifndef X3_ABL_NOROWS
            epilogue_rows<EK, 128, 128, 512>(p, Cs, m0 + a * 128, n0 + b * 128, tid);
#else
            if (Cs[tid] == 12345.678f) p.C[tid] = Cs[tid + 1];
#endif
            if constexpr (EK == E_DOTCOL) __syncthreads();
        }
    }
#else
    // epilogue straight from the accumulators (epilogue_regs): every wave stores its own 32 rows, both wave groups at once
    {
        float* const red = reinterpret_cast<float*>(smem) + wave * 256;
        epilogue_regs<EK>(p, acc, m0 + ah * 128 + ar, n0, lane, red);
        if constexpr (EK == E_DOTCOL) {
            // colpart[row tile of 128][n]: the four waves of a half (32 rows each) in fixed order
            __syncthreads();
            const int a = tid >> 8, c = tid & 255;
            const float* r4 = reinterpret_cast<const float*>(smem) + (4 * a) * 256 + c;
            const float t = ((r4[0] + r4[256]) + r4[512]) + r4[768];
            if (m0 + a * 128 < p.M && n0 + c < p.N) p.colpart[((m0 + a * 128) / 128) * (int64_t)p.N + n0 + c] = t;
        }
    }
#endif
#ifdef X3_ABL_CLOCK
    __syncthreads();
    if (tid == 0 && lin == 0) {   // main-loop shader cycles and 100-MHz ticks of block 0 into C[0][0..1]
        p.C[0] = (float)(clk1_ - clk0_);
        p.C[1] = (float)(rt1_ - rt0_);
    }
#endif
}

// ---- weight-gradient (TN) GEMM in the same arithmetic:  G[n][k] = sum_m X[m][n] * Y[m][k]  over a chunk of pixels -> fp32 slab ------------
// Both operands are fp32 activations.  Their [16 m][128 columns] half-tiles go HBM -> LDS by LDS-DMA as they are (row-major, 512-B runs, a
// ring of two pixel steps).  An MFMA operand is 8 consecutive PIXELS of one column: a CONVERT pass gathers them (eight ds_read_b32, stride-1
// across the lanes: conflict-free without a swizzle), splits them in registers and writes the three pieces as lane-linear 16-byte records
// into a fragment-major plane buffer -- each of the 16 fragments of a pixel step ONCE per block (wave w converts X fragment w and Y fragment
// w, one step ahead), where the first version split every fragment in each of the 2-4 waves that use it (264 VALU operations per wave and
// pixel step next to 48 MFMAs; now 88).  The MFMA phases then read their fragments as plain ds_read_b128 from the planes.  256 x 256 output
// tile, waves as 4 (n) x 2 (k) interleaved over the halves, four phases of 12 MFMAs per 16 pixels, two alternating wave groups as in the NT
// kernel.  LDS: 64 KB fp32 ring + 2 x 48 KB planes = the CU's 160 KB.  Partial column sums of X (bias gradients) in the fp32 kernel's
// [split * tiles_k + tile_k][N] layout (this kernel's tile_k covers two of that kernel's: the odd row is written as zeros).
constexpr int THT = 8192;            // TN half-tile: 16 rows x 128 floats
constexpr int TSTG = 4 * THT;        // X-lo | X-hi | Y-lo | Y-hi
constexpr int TPL = 48 * 1024;       // plane buffer of one pixel step: 16 fragments x 3 pieces x 1 KiB
constexpr int TPL0 = 2 * TSTG;       // planes start behind the fp32 ring

typedef __attribute__((address_space(3))) const volatile float* lds_f1_p;
typedef __attribute__((address_space(3))) bf16x8* lds_wr_p;

__device__ __forceinline__ void split8s(const float x[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    floatx4 a, b;
    a.x = x[0]; a.y = x[1]; a.z = x[2]; a.w = x[3];
    b.x = x[4]; b.y = x[5]; b.z = x[6]; b.w = x[7];
    split8(a, b, p0, p1, p2);
}

__global__ __launch_bounds__(512) void gemm_tn_x3_kernel(const GemmTN p, int fp32_tiles_k) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TSTG + 2 * TPL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave & 3, wn = wave >> 2;      // rows 32 wm.. of each X half (n), columns 64 wn.. of each Y half (k)
    const int tilesK = p.K / 256, tilesN = p.N / 256;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / (tilesN * tilesK);
    const int tile = lin % (tilesN * tilesK);
    const int tile_n = tile / tilesK, tile_k = tile % tilesK;
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int64_t mbeg = (int64_t)split * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const int64_t mw = mbeg < p.M ? mbeg : 0;
    const i32x4 rsX = make_rsrc_dma(p.X + mw * (int64_t)p.ldx + n0);
    const i32x4 rsY = make_rsrc_dma(p.Y + mw * (int64_t)p.ldy + k0);
    // a half-tile = 8 DMAs of two 512-byte row runs: wave w stages rows 2 w and 2 w + 1
    const int drow = 2 * wave + (lane >> 5);
    const uint32_t dcol = (uint32_t)(lane & 31) * 16u;
    const uint32_t voffX[2] = {(uint32_t)drow * (uint32_t)p.ldx * 4u + dcol, (uint32_t)drow * (uint32_t)p.ldx * 4u + 512u + dcol};
    const uint32_t voffY[2] = {(uint32_t)drow * (uint32_t)p.ldy * 4u + dcol, (uint32_t)drow * (uint32_t)p.ldy * 4u + 512u + dcol};
    const uint32_t lds_base = lds_addr(reinterpret_cast<const float*>(smem)) + (uint32_t)wave * 1024u;
    const int64_t nmt = (mend - mbeg + 15) / 16;
    // Hx: 0 X-lo, 1 X-hi, 2 Y-lo, 3 Y-hi; fp32 stage st (0 / 1); pixel step t (past the chunk: zeros)
    auto stage = [&](int Hx, int st, int64_t t) {
        const int64_t row = mbeg + t * 16 + drow;
        const bool ok = t < nmt && row < mend;
        const uint32_t dst = lds_base + (uint32_t)(st * TSTG + Hx * THT);
        if (Hx < 2) dma16(rsX, dst, ok ? voffX[Hx] : ROW_SENT, (uint32_t)(t * 16) * (uint32_t)p.ldx * 4u);
        else dma16(rsY, dst, ok ? voffY[Hx - 2] : ROW_SENT, (uint32_t)(t * 16) * (uint32_t)p.ldy * 4u);
    };

    const int fr = lane & 31, fh = lane >> 5;
    // convert source: this wave's X fragment = columns 32 wave.. of the 256-column X tile, its Y fragment likewise
    const unsigned char* cvx = smem + (wave >> 2) * THT + (8 * fh) * 512 + ((wave & 3) * 32 + fr) * 4;
    const unsigned char* cvy = cvx + 2 * THT;
    // plane records: fragment f (X: n-tile 0..7, Y: 8 + k-tile), piece pl at ((f * 3 + pl) * 1024) + lane * 16
    unsigned char* const pl_wr = smem + TPL0 + lane * 16;
    const unsigned char* const pl_rd = smem + TPL0 + lane * 16;
    const int fx_lo = wm, fx_hi = 4 + wm, fy_lo = 8 + 2 * wn, fy_hi = 12 + 2 * wn;

    floatx16 acc[2][2][2];   // [X half][Y half][k-tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;
    bf16x8 fa[3], fb[2][2][3];
    float cs_lo = 0.f, cs_hi = 0.f;
    const bool do_cs = p.colsum != nullptr && tid < 128;

// gather the fragment (X or Y of this wave) of fp32 stage ST: 8 consecutive pixels of this lane's column
#define DCPT_GATHER(SRC, ST) _Pragma("unroll") for (int e = 0; e < 8; ++e) gr[e] = *(lds_f1_p)((SRC) + (ST)*TSTG + e * 512);
// split the gathered fragment and write its three pieces into plane buffer PB as fragment F
#define DCPT_SPLITW(PB, F)                                                                                                         \
    {                                                                                                                             \
        bf16x8 q0_, q1_, q2_;                                                                                                     \
        split8s(gr, q0_, q1_, q2_);                                                                                               \
        *(lds_wr_p)(pl_wr + (PB)*TPL + ((F)*3 + 0) * 1024) = q0_;                                                                 \
        *(lds_wr_p)(pl_wr + (PB)*TPL + ((F)*3 + 1) * 1024) = q1_;                                                                 \
        *(lds_wr_p)(pl_wr + (PB)*TPL + ((F)*3 + 2) * 1024) = q2_;                                                                 \
    }
#define DCPT_COLSUM(ST)                                                                                                            \
    {                                                                                                                             \
        float s0_ = 0.f, s1_ = 0.f;                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                          \
            s0_ += *(lds_f1_p)(smem + (ST)*TSTG + r * 512 + tid * 4);                                                             \
            s1_ += *(lds_f1_p)(smem + (ST)*TSTG + THT + r * 512 + tid * 4);                                                       \
        }                                                                                                                         \
        cs_lo += s0_;                                                                                                             \
        cs_hi += s1_;                                                                                                             \
    }
    float gr[8];   // the gathered fragment between its read phase and its split phase

    // prologue: raw step 0 -> stage 0, raw step 1 -> stage 1; step 0 converted by everybody; then X(2) -> stage 0 (the loop's order of
    // outstanding DMAs before its first phase is ..., Y(t+1), X(t+2))
    stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
    stage(0, 1, 1); stage(1, 1, 1); stage(2, 1, 1); stage(3, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    DCPT_GATHER(cvx, 0)
    DCPT_SPLITW(0, wave)
    DCPT_GATHER(cvy, 0)
    DCPT_SPLITW(0, 8 + wave)
    if (do_cs && tile_k == 0) DCPT_COLSUM(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    stage(0, 0, 2); stage(1, 0, 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // X(1) landed (Y(1), X(2) may be in flight)
    __syncthreads();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind from here on

#define DCPT_RD_X(PB, F) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) fa[pl] = *(lds_frag_p)(pl_rd + (PB)*TPL + ((F)*3 + pl) * 1024);
#define DCPT_RD_Y(PB, H, F)                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                 \
        fb[H][j][pl] = *(lds_frag_p)(pl_rd + (PB)*TPL + (((F) + j) * 3 + pl) * 1024);
#ifdef X3_ABL_NOMFMA
#define DCPT_MM(ACC, I, BH, PA, PB_) asm volatile("" ::"v"(fa[PA]), "v"(fb[BH][I][PB_]))
#else
#define DCPT_MM(ACC, I, BH, PA, PB_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA], fb[BH][I][PB_], ACC, 0, 0, 0)
#endif
#define DCPT_MFMA(AH, BH)                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                                \
    DCPT_MM(acc[AH][BH][0], 0, BH, 2, 0); DCPT_MM(acc[AH][BH][1], 1, BH, 2, 0);                                                     \
    DCPT_MM(acc[AH][BH][0], 0, BH, 0, 2); DCPT_MM(acc[AH][BH][1], 1, BH, 0, 2);                                                     \
    DCPT_MM(acc[AH][BH][0], 0, BH, 1, 1); DCPT_MM(acc[AH][BH][1], 1, BH, 1, 1);                                                     \
    DCPT_MM(acc[AH][BH][0], 0, BH, 1, 0); DCPT_MM(acc[AH][BH][1], 1, BH, 1, 0);                                                     \
    DCPT_MM(acc[AH][BH][0], 0, BH, 0, 1); DCPT_MM(acc[AH][BH][1], 1, BH, 0, 1);                                                     \
    DCPT_MM(acc[AH][BH][0], 0, BH, 0, 0); DCPT_MM(acc[AH][BH][1], 1, BH, 0, 0);                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    __builtin_amdgcn_s_barrier();

#ifdef X3_ABL_NORD
#define X3_RD(...)
    DCPT_RD_X(0, fx_lo)
    DCPT_RD_Y(0, 0, fy_lo)
    DCPT_RD_Y(0, 1, fy_hi)
#else
#define X3_RD(...) __VA_ARGS__
#endif
#ifdef X3_ABL_NOCONV
#define X3_CONV(...)
#else
#define X3_CONV(...) __VA_ARGS__
#endif
    // Step t: plane buffer u = t & 1 holds its fragments; raw stage u ^ 1 holds raw step t + 1, converted during this step into plane
    // buffer u ^ 1.  The load half of every phase is kept short of the 384 cycles of the other group's 12 MFMAs: no phase both reads and
    // uses the same data (the gather of a fragment is issued a phase before its 44-operation split), and nothing waits for LDS writes
    // before ph3.
    //   ph0  read X-lo, Y-lo;  gather X of raw(t+1);  column sums of raw(t+1);  DMA Y(t+2) -> stage u     MFMA (lo,lo)
    //   ph1  read Y-hi;  split + write X;  vmcnt(4): Y(t+1) landed                                       MFMA (lo,hi)
    //   ph2  read X-hi;  gather Y of raw(t+1);  DMA X(t+3) -> stage u ^ 1                                MFMA (hi,hi)
    //   ph3  split + write Y;  vmcnt(4): X(t+2) landed;  lgkmcnt(0): plane t + 1 complete                MFMA (hi,lo)
    // LDS-DMA rules: the X halves of a raw stage are last read at ph0 (gather, column sums) and re-staged at ph2, the Y halves read at
    // ph2 and re-staged at ph0 of the next step (>= two phases); every half is read >= one phase after the wait that retired it (X: ph3 ->
    // ph0, Y: ph1 -> ph2); five phases of flight for every DMA.  A plane buffer is rewritten from ph1 on, three phases after its last
    // read (X-hi, ph2 of the step before).
    for (int64_t t = 0; t < nmt; t += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // u = parity of the step
            const int64_t tt = t + u;
            // ph0
            X3_RD(DCPT_RD_X(u, fx_lo))
            X3_RD(DCPT_RD_Y(u, 0, fy_lo))
            X3_CONV(DCPT_GATHER(cvx, u ^ 1))
            if (do_cs && tt + 1 < nmt && (int)((tt + 1) % tilesK) == tile_k) DCPT_COLSUM(u ^ 1)
            X3_STAGE(stage(2, u, tt + 2);)
            X3_STAGE(stage(3, u, tt + 2);)
            __builtin_amdgcn_s_barrier();
            DCPT_MFMA(0, 0)
            // ph1
            X3_RD(DCPT_RD_Y(u, 1, fy_hi))
            X3_CONV(DCPT_SPLITW(u ^ 1, wave))
            X3_STAGE(asm volatile("s_waitcnt vmcnt(4)" ::: "memory");)
            __builtin_amdgcn_s_barrier();
            DCPT_MFMA(0, 1)
            // ph2
            X3_RD(DCPT_RD_X(u, fx_hi))
            X3_CONV(DCPT_GATHER(cvy, u ^ 1))
            X3_STAGE(stage(0, u ^ 1, tt + 3);)
            X3_STAGE(stage(1, u ^ 1, tt + 3);)
            __builtin_amdgcn_s_barrier();
            DCPT_MFMA(1, 1)
            // ph3
            X3_CONV(DCPT_SPLITW(u ^ 1, 8 + wave))
            X3_STAGE(asm volatile("s_waitcnt vmcnt(4)" ::: "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            DCPT_MFMA(1, 0)
        }
    }
#undef DCPT_GATHER
#undef DCPT_SPLITW
#undef DCPT_COLSUM
#undef DCPT_RD_X
#undef DCPT_RD_Y
#undef DCPT_MM
#undef DCPT_MFMA
    if (grp == 0) __builtin_amdgcn_s_barrier();
    dma_wait_all();
    __syncthreads();

#ifdef X3_ABL_NOEPI
    {
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[a][b][i][r];
        if (keep == 12345.678f) p.slab[0] = keep;
        return;
    }
#endif
    // slab tile [256 n][256 k] of this split: the four 128 x 128 quadrants through LDS (two alternating 64 KB buffers) into the fp32
    // kernel's plain row epilogue -- 16-byte stores, 512-byte runs (4-byte stores straight from the accumulators cost ~6x per byte)
    GemmNT sp{};
    sp.C = p.slab + (int64_t)split * p.N * p.K;
    sp.M = p.N;
    sp.N = p.K;
    sp.ldc = p.K;
    float* const Cs0 = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int a = q >> 1, b = q & 1;
        float* const Cs = Cs0 + (q & 1) * (128 * 128);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kl = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[nl * 128 + kl] = acc[a][b][j][r];
            }
        }
        __syncthreads();
        epilogue_rows<E_PLAIN, 128, 128, 512>(sp, Cs, (int64_t)n0 + a * 128, k0 + b * 128, tid);
    }
    if (p.colsum != nullptr && tid < 128) {
        // the fp32 kernel's layout has fp32_tiles_k (= K / 128) partial rows per split: this tile_k owns rows 2 tile_k (the sums) and + 1 (zeros)
        float* c0 = p.colsum + ((int64_t)split * fp32_tiles_k + 2 * tile_k) * p.N + n0;
        c0[tid] = cs_lo;
        c0[128 + tid] = cs_hi;
        if (2 * tile_k + 1 < fp32_tiles_k) {
            c0[p.N + tid] = 0.f;
            c0[p.N + 128 + tid] = 0.f;
        }
    }
}

// process-wide state of the opt-in mode: a caller-provided scratch for the split images (dcpt_set_gemm_x3)
unsigned char* g_x3_scratch = nullptr;
size_t g_x3_bytes = 0;
int g_x3_min_tiles = 192;
long long g_x3_scratch_misses = 0;   // eligible launches that fell back to the fp32 kernels because the scratch was too small

size_t split_bytes(int64_t R, int K) { return (size_t)cdiv64(R, 128) * (size_t)(K / 16) * PIECE; }

}  // namespace

// nimg > 1: the same matrix with scale row z of kscale ([nimg][K]) into image z (images img_bytes apart)
int launch_split3(const float* X, int64_t R, int K, int ld, void* out, const float* kscale, int nimg, hipStream_t s) {
    DCPT_CHECK_ARG(X && out && R > 0 && K > 0 && K % 16 == 0 && ld % 4 == 0, "split3: bad argument (K=%d must be a multiple of 16)", K);
    DCPT_CHECK_ARG(cdiv64(R, 128) < 65536 && nimg >= 1 && nimg < 65536, "split3: too many row blocks / images");
    split3_kernel<<<dim3((unsigned)cdiv(K / 16, 4), (unsigned)cdiv64(R, 128), (unsigned)nimg), dim3(256), 0, s>>>(X, R, K, ld, (unsigned char*)out,
                                                                                                                  kscale, (int64_t)split_bytes(R, K));
    DCPT_CHECK_LAUNCH("split3");
    return DCPT_OK;
}

bool gemm_x3_enabled() { return g_x3_scratch != nullptr; }

// eligibility: plain (or per-image scaled) fp32 A operand, row-wise epilogue, full 256-column tiles, k-tiles of 32 in pairs, enough tiles
bool gemm_nt_x3_ok(const GemmNT& p, int aload, int epi) {
    if (!g_x3_scratch || !(aload == A_PLAIN || aload == A_SCALE)) return false;
    if (!(epi == E_PLAIN || epi == E_BIAS || epi == E_RESID || epi == E_SGBWD || epi == E_BIASGATE || epi == E_DOTCOL || epi == E_ADDSCALED ||
          epi == E_MUL))
        return false;
    if (epi == E_SGBWD && p.rowpart) return false;
    if (p.K % 64 != 0 || p.N % 256 != 0 || p.lda % 4 != 0) return false;
    int64_t nb = (int64_t)(p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1);
    int64_t M = p.M;
    if (aload == A_SCALE) {   // the per-image scale moves into per-image weights: one GEMM problem per image
        if (nb != 1 || p.P <= 0 || p.M % p.P != 0 || p.M / p.P >= 65536) return false;
        nb = p.M / p.P;
        M = p.P;
    }
    const int64_t tiles = cdiv64(M, 256) * (p.N / 256) * nb;
    if (tiles < g_x3_min_tiles) return false;
    if (split_bytes(p.N, p.K) * (size_t)nb > g_x3_bytes) {
        ++g_x3_scratch_misses;
        return false;
    }
    return true;
}

int launch_gemm_nt_x3(const GemmNT& pin, int aload, int epi, hipStream_t s) {
    GemmX3 x{};
    x.g = pin;
    GemmNT& p = x.g;
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;
    const size_t bbytes = split_bytes(p.N, p.K);
    x.B3 = g_x3_scratch;
    x.sB3 = (int64_t)bbytes;
    if (aload == A_SCALE) {
        const int nimg = (int)(p.M / p.P);
        const int ldres = p.ldres ? p.ldres : p.ldc;
        DCPT_TRY(launch_split3(p.Bw, p.N, p.K, p.K, g_x3_scratch, p.simg, nimg, s));
        p.M = p.P;
        p.nb1 = nimg;
        p.nb2 = 1;
        p.sA1 = (int64_t)p.P * p.lda;
        p.sC1 = (int64_t)p.P * p.ldc;
        p.sR1 = (int64_t)p.P * ldres;
        p.sS1 = 0;
        p.sA2 = p.sC2 = p.sR2 = p.sS2 = 0;
    } else {
        const int nb = p.nb1 * p.nb2;
        for (int b = 0; b < nb; ++b) {   // (already batched problems bring their own weights: one split each)
            const float* Bb = p.Bw + (b / p.nb2) * p.sB1 + (b % p.nb2) * p.sB2;
            DCPT_TRY(launch_split3(Bb, p.N, p.K, p.K, g_x3_scratch + bbytes * b, nullptr, 1, s));
        }
    }
    const dim3 grid((unsigned)(cdiv64(p.M, 256) * (p.N / 256)), (unsigned)(p.nb1 * p.nb2));
    switch (epi) {
        case E_PLAIN: gemm_nt_x3_kernel<E_PLAIN><<<grid, dim3(512), 0, s>>>(x); break;
        case E_BIAS: gemm_nt_x3_kernel<E_BIAS><<<grid, dim3(512), 0, s>>>(x); break;
        case E_RESID: gemm_nt_x3_kernel<E_RESID><<<grid, dim3(512), 0, s>>>(x); break;
        case E_SGBWD: gemm_nt_x3_kernel<E_SGBWD><<<grid, dim3(512), 0, s>>>(x); break;
        case E_BIASGATE: gemm_nt_x3_kernel<E_BIASGATE><<<grid, dim3(512), 0, s>>>(x); break;
        case E_DOTCOL: gemm_nt_x3_kernel<E_DOTCOL><<<grid, dim3(512), 0, s>>>(x); break;
        case E_ADDSCALED: gemm_nt_x3_kernel<E_ADDSCALED><<<grid, dim3(512), 0, s>>>(x); break;
        case E_MUL: gemm_nt_x3_kernel<E_MUL><<<grid, dim3(512), 0, s>>>(x); break;
        default: dcpt_set_error("gemm_nt_x3: epilogue %d not supported", epi); return DCPT_ERR_ARG;
    }
    DCPT_CHECK_LAUNCH("gemm_nt_x3");
    return DCPT_OK;
}

// shapes the TN kernel takes: full 256 x 256 output tiles, plain operands, dense enough rows
bool gemm_tn_x3_shape_ok(int N, int K) { return g_x3_scratch != nullptr && N % 256 == 0 && K % 256 == 0; }

// split plan for such a shape: one block per CU (tiles x splits ~ 256), chunks of >= 256 pixels, multiples of 32
bool gemm_tn_x3_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split) {
    if (!gemm_tn_x3_shape_ok(N, K)) return false;
    const int64_t tiles = (int64_t)(N / 256) * (K / 256);
    int64_t want = 256 / tiles;
    const int64_t max_by_rows = cdiv64(M, 256);
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    if (tiles * want < g_x3_min_tiles && g_x3_min_tiles > 1) return false;   // too little work for one-block-per-CU tiles
    const int64_t rps = cdiv64(cdiv64(M, want), 32) * 32;
    *rows_per_split = rps;
    *splits = (int)cdiv64(M, rps);
    return true;
}

bool gemm_tn_x3_ok(const GemmTN& p, int xload, int yload) {
    if (!gemm_tn_x3_shape_ok(p.N, p.K) || xload != A_PLAIN || yload != A_PLAIN) return false;
    if ((p.nb1 > 1) || (p.nb2 > 1) || p.ldx % 4 != 0 || p.ldy % 4 != 0 || p.rows_per_split % 16 != 0) return false;
    const int64_t blocks = (int64_t)(p.N / 256) * (p.K / 256) * p.splits;
    return blocks >= (g_x3_min_tiles > 1 ? 128 : 1);
}

int launch_gemm_tn_x3(const GemmTN& p, int fp32_tiles_k, hipStream_t s) {
    const unsigned blocks = (unsigned)((p.N / 256) * (p.K / 256) * p.splits);
    gemm_tn_x3_kernel<<<dim3(blocks), dim3(512), 0, s>>>(p, fp32_tiles_k);
    DCPT_CHECK_LAUNCH("gemm_tn_x3");
    return DCPT_OK;
}

extern "C" long long dcpt_gemm_x3_scratch_misses(void) { return g_x3_scratch_misses; }

extern "C" int dcpt_set_gemm_x3(void* scratch, size_t bytes, int min_tiles) {
    g_x3_scratch = (unsigned char*)scratch;
    g_x3_bytes = scratch ? bytes : 0;
    g_x3_min_tiles = min_tiles > 0 ? min_tiles : 192;
    return DCPT_OK;
}
