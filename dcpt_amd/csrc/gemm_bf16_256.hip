// bf16 NT GEMM with 256 x 256 macro-tiles for the wide levels of the encoder (C[m][n] = sum_k A[m][k] * Bw[n][k], M = pixels,
// N, K = 256 ... 2048 channels): the 1x1 convolutions of a level-2/3/4 NAFBlock, forward and data-gradient.
//
// Why a second kernel.  In the 128 x 128 kernel (gemm_bf16.hip) a k-tile of 64 costs a CU 512 cycles of MFMA per block, 512 cycles
// of L2 -> LDS fill (32 KB at the 64 B/clk vector-memory rate) and 256 cycles of LDS operand reads, and the three do not overlap
// (profiles/r2/ntb_ablation.txt): 421-750 TF/s.  A 256 x 256 tile halves the fill bytes per flop (64 KB for 2048 MFMA cycles) and the
// LDS read bytes per flop, and the schedule below keeps the matrix pipe busy while the fill and the reads happen:
//
//   * 8 waves = two GROUPS of four; a SIMD hosts one wave of each group.  A k-tile is four PHASES (one 64 x 32 x 64 quadrant product
//     per wave and phase: 8 x v_mfma_f32_32x32x16_bf16 = 256 cycles); a phase is  [LDS fragment reads + 2 LDS-DMA issues] barrier
//     [8 MFMAs] barrier.  Group 1 runs one barrier behind group 0, so on every SIMD one wave issues MFMAs while its partner reads
//     fragments and feeds the DMA queue -- the pipe alternates between the two waves instead of idling through a common load phase.
//   * LDS = 2 k-tiles x {A-lo, A-hi, B-lo, B-hi} half-tiles of [128 rows][128 B] (128 KB), XOR-swizzled like the 128 x 128 kernel's
//     tiles (one conflict-free ds_read_b128 = one MFMA operand; the image is what an LDS-DMA writes).  A wave's 128 x 64 output is
//     INTERLEAVED over the halves (64 rows of A-lo + 64 of A-hi, 32 columns of B-lo + 32 of B-hi), so half-tiles die one after
//     the other inside a k-tile and each phase re-stages ONE of them, two k-tiles ahead.
//   * The DMAs are never drained in the loop: every phase ends its load part with a counted  s_waitcnt vmcnt(8)  -- four half-tiles
//     (4 phases, ~2000 cycles) stay in flight across the barriers.  Ordering rules (MI355X_MICROARCH.md, LDS-DMA): a half-tile is
//     read at least one phase after the wait that retired it, and re-staged at least two phases after its last read.
//
// Phase p of k-tile t (stage s = t & 1), quadrant = (A half, B half):
//     p0: read A-lo, B-lo   stage B-hi(t+1)   MFMA (lo,lo)        p2: read A-hi    stage A-lo(t+2)   MFMA (hi,hi)
//     p1: read B-hi         stage A-hi(t+1)   MFMA (lo,hi)        p3: --           stage B-lo(t+2)   MFMA (hi,lo)
// (both B sub-fragments stay in registers for the whole k-tile.)  Epilogue: the four 128 x 128 quadrants go through LDS one after the
// other (two alternating 64 KB buffers) into the same epilogue8<> as the 128-row kernel -- bias, residual, SimpleGate backward,
// bias+gate (the two B halves are the two gate halves), column dots, LayerNorm backward.
#include "bf16.h"
#include "gemm_bf16_epi.h"

namespace {

constexpr int HT = 16384;       // bytes of a half-tile
constexpr int STG = 4 * HT;     // one k-tile: A-lo | A-hi | B-lo | B-hi

typedef __attribute__((address_space(3))) const volatile bf16x8* lds_frag_p;

// TALL: a 512 x 128 tile for N == 128 (the classifier head's stage-0 convs: 2.1 M pixels x 128 channels) -- four A half-tiles of 128 rows against
// ONE B half-tile per k-tile (80 KB a stage, 160 KB for the two), the same 8-wave two-group schedule: phase i of a k-tile is the quadrant product
// (A_i, B), the B fragments stay in registers for the whole k-tile.  Staging, as early as the two-phase rule allows (a half-tile is re-staged
// two phases after its last read):  p0: A2(t+1)   p1: A3(t+1)   p2: A0(t+2) + B(t+2)   p3: A1(t+2)  -- every piece is issued six phases before
// its read, so the counted waits leave 12 (p0, p1, p3) / 14 (p2) DMAs of a wave in flight.
template <int EK, int AM = 0, bool TALL = false>   // AM 1: implicit 3 x 3 (conv3): A rows are pixels, a k-tile of 64 channels lies inside one tap (gC % 64 == 0)
__global__ __launch_bounds__(512) void gemm_nt_bf16_256_kernel(const GemmNTB pin) {
    constexpr bool GATE = (EK == EB_BIASGATE);
    constexpr bool CONV = AM == 1;
    static_assert(!(TALL && GATE), "the tall tile has one B half: no gate epilogue");
    constexpr int NA = TALL ? 4 : 2;            // A half-tiles of a k-tile
    constexpr int STGK = (NA + (TALL ? 1 : 2)) * HT;   // bytes of a stage
    constexpr int BOFF = NA * HT;               // the B half-tile(s) behind the A half-tiles
    GemmNTB p = pin;
    int lin, batch;
    xcd_remap_batched(lin, batch);
    if (gridDim.y > 1) {
        const int64_t b = batch;
        p.A += b * p.sA;
        p.Bw += b * p.sB;
        p.C += b * p.sC;
        if (p.res) p.res += b * p.sR;
    }
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STGK];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;              // 0: leads, 1: one barrier behind
    const int wm = wave >> 2, wn = wave & 3;
    const int Ch = p.N / 2;
    const int tilesN = GATE ? Ch / 128 : TALL ? p.N / 128 : p.N / 256;
    const int64_t m0 = (int64_t)(lin / tilesN) * (TALL ? 512 : 256);
    const int n0 = (lin % tilesN) * ((GATE || TALL) ? 128 : 256);
    const int nlo = n0, nhi = GATE ? Ch + n0 : n0 + 128;   // first weight row of the two B halves

    // conv3: the window starts one image row + one pixel before the tile's first pixel (clipped at the tensor start)
    int64_t apix0 = (m0 < p.M ? m0 : 0);
    if constexpr (CONV) {
        apix0 -= p.gW + 1;
        if (apix0 < 0) apix0 = 0;
    }
    const i32x4 rsA = make_rsrc_dma(p.A + apix0 * (int64_t)(CONV ? p.gC : p.lda));
    const i32x4 rsB = make_rsrc_dma(p.Bw);
    // staging map of a half-tile (16 DMAs of 8 rows): wave w issues rows 8 w .. 8 w + 7 and 64 + 8 w .. ; lane -> row (lane >> 3),
    // LDS slot lane & 7 = logical 16-byte chunk ^ ((row >> 1) & 7)
    uint32_t voffA[NA][2], voffB[2][2];
    uint32_t tapok[NA][2];   // conv3: bit t set = tap t of this thread's row lies inside the image
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int row = 8 * wave + (lane >> 3) + 64 * ps;
        const uint32_t ch = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            const int r = h * 128 + row;
            if constexpr (CONV) {
                const int64_t m = m0 + r;
                const bool ok = m < p.M;
                const int64_t mm = ok ? m : 0;
                const int pw = (int)(mm % p.gW), ph = (int)((mm / p.gW) % p.gH);
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hh = ph + t / 3 - 1, ww = pw + t % 3 - 1;
                    if (ok && hh >= 0 && hh < p.gH && ww >= 0 && ww < p.gW) bits |= 1u << t;
                }
                tapok[h][ps] = bits;
                voffA[h][ps] = (uint32_t)((mm - apix0) * p.gC) * 2u + ch;
            } else {
                voffA[h][ps] = (m0 + r < p.M) ? (uint32_t)r * (uint32_t)p.lda * 2u + ch : ROW_SENT;
            }
            if (h < 2) voffB[h][ps] = (uint32_t)((h ? nhi : nlo) + row) * (uint32_t)p.K * 2u + ch;
        }
    }
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(smem));
    const uint32_t lds_w = lds0 + (uint32_t)wave * 1024u;
    // X: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi   (TALL: 0..3 A0..A3, 4 B)
    auto stage = [&](int X, int s, int kt, uint32_t dead) {
        const uint32_t dst = lds_w + (uint32_t)(s * STGK + X * HT);
        const uint32_t soff = (uint32_t)kt * 128u;
#ifdef DCPT_ABL_NOA   // ablation builds: the DMA is issued but range-checked away (zero fill, no memory access)
        if (X < 2) dead = ROW_SENT;
#endif
#ifdef DCPT_ABL_NOB
        if (X >= 2) dead = ROW_SENT;
#endif
        if (X < NA) {
            if constexpr (CONV) {
                // k-tile kt = 64 channels of ONE tap: the same shift for every row, per-row validity from the tap masks.  The k-tiles run
                // CHANNEL CHUNK-major, tap-minor (chunk kt / 9, tap kt % 9): the nine shifted reads of a chunk's rows follow each other and
                // hit the XCD's L2; tap-major, a block came back to its rows only after a whole pass over the channels and fetched the input
                // 7 x over the fabric (1131 MB for a 157-MB tensor at the head's 256 x 256 stage, profiles/r5/conv3_tap_order/)
                const uint32_t chunk = ((uint32_t)kt * 7282u) >> 16, tap = (uint32_t)kt - 9u * chunk;   // kt / 9, kt % 9   (kt < 2048)
                const int ch0 = (int)chunk * 64;
                const int ky = (int)((tap * 11u) >> 5), kx = (int)tap - 3 * ky;
                const uint32_t shift = (uint32_t)((((ky - 1) * p.gW + (kx - 1)) * p.gC + ch0) * 2);
                dma16(rsA, dst, (((tapok[X][0] >> tap) & 1u) ? voffA[X][0] + shift : ROW_SENT) | dead, 0);
                dma16(rsA, dst + 8192u, (((tapok[X][1] >> tap) & 1u) ? voffA[X][1] + shift : ROW_SENT) | dead, 0);
            } else {
                dma16(rsA, dst, voffA[X][0] | dead, soff);
                dma16(rsA, dst + 8192u, voffA[X][1] | dead, soff);
            }
        } else {
            uint32_t soffB = soff;
            if constexpr (CONV) {   // the weights' k index of k-tile kt: tap * gC + 64 * chunk
                const uint32_t chunk = ((uint32_t)kt * 7282u) >> 16, tap = (uint32_t)kt - 9u * chunk;
                soffB = (tap * (uint32_t)p.gC + 64u * chunk) * 2u;
            }
            dma16(rsB, dst, voffB[X - NA][0] | dead, soffB);
            dma16(rsB, dst + 8192u, voffB[X - NA][1] | dead, soffB);
        }
    };

    // fragment addresses: row (lane & 31) of the wave's sub-tile, 16-byte slot ((2 j + fh) ^ fi) of k-step j
    const int fi = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    const unsigned char* abase = smem + (wm * 64 + (lane & 31)) * 128;
    const unsigned char* bbase = smem + BOFF + (wn * 32 + (lane & 31)) * 128;
    int slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = ((2 * j + fh) ^ fi) * 16;

    constexpr int NBH = TALL ? 1 : 2;
    floatx16 acc[NA][NBH][2];   // [A half][B half][m-tile]
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NBH; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;
    bf16x8 fa[2][4], fb[NBH][4];   // A sub-tile in use [m-tile][k-step]; B sub-tiles [half][k-step]

    const int nkt = p.K / 64;   // even (K % 128 == 0)
    if constexpr (TALL) {
        // prologue in the steady order of k-tiles -2 and -1: A0(0) B(0) A1(0) | A2(0) A3(0) A0(1) B(1) A1(1); the loop goes on with A2(1), A3(1), ...
        stage(0, 0, 0, 0);
        stage(4, 0, 0, 0);
        stage(1, 0, 0, 0);
        stage(2, 0, 0, 0);
        stage(3, 0, 0, 0);
        stage(0, 1, 1, 0);
        stage(4, 1, 1, 0);
        stage(1, 1, 1, 0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // A0(0), B(0) landed (this wave's part; the barrier publishes all)
    } else {
    // prologue: A-lo(0) B-lo(0) B-hi(0) A-hi(0) A-lo(1) B-lo(1); the loop stages B-hi(1), A-hi(1), A-lo(2), ... in that rhythm
    stage(0, 0, 0, 0);
    stage(2, 0, 0, 0);
    stage(3, 0, 0, 0);
    stage(1, 0, 0, 0);
    stage(0, 1, 1, 0);
    stage(2, 1, 1, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A-lo(0), B-lo(0) landed (this wave's part; the barrier publishes all)
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind from here on

#ifdef DCPT_ABL_NOMFMA   // ablation build: operands stay live, no matrix work
#define DCPT_MFMA_OP(ACC, FA, FB) asm volatile("" ::"v"(FA), "v"(FB))
#else
#define DCPT_MFMA_OP(ACC, FA, FB) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA, FB, ACC, 0, 0, 0)
#endif
// (fragments are read in the order the MFMAs consume them: k-step by k-step)
#define DCPT_LOAD_A(S, H)                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)                           \
        fa[i][j] = *(lds_frag_p)(abase + (S)*STGK + (H)*HT + i * 4096 + slot[j]);
#define DCPT_LOAD_B(S, H)                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[H][j] = *(lds_frag_p)(bbase + (S)*STGK + (H)*HT + slot[j]);
#define DCPT_LOAD_AB(S)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                       \
        fa[0][j] = *(lds_frag_p)(abase + (S)*STGK + slot[j]);                                                             \
        fb[0][j] = *(lds_frag_p)(bbase + (S)*STGK + slot[j]);                                                             \
        fa[1][j] = *(lds_frag_p)(abase + (S)*STGK + 4096 + slot[j]);                                                      \
    }
#define DCPT_MFMA(AH, BH)                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)                           \
        DCPT_MFMA_OP(acc[AH][BH][i], fa[i][j], fb[BH][j]);                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();
#define DCPT_PUBLISH()                                                                                                    \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                      \
    __builtin_amdgcn_s_barrier();
#define DCPT_PUBLISH_N(N)                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                                                                 \
    __builtin_amdgcn_s_barrier();

    if constexpr (TALL) {
        for (int t = 0; t < nkt; t += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kt = t + s;
                const uint32_t dead1 = (kt + 1 < nkt) ? 0u : ROW_SENT, dead2 = (kt + 2 < nkt) ? 0u : ROW_SENT;
                // phase 0: (A0, B)
                DCPT_LOAD_AB(s)
                stage(2, s ^ 1, kt + 1, dead1);
                DCPT_PUBLISH_N(12)
                DCPT_MFMA(0, 0)
                // phase 1: (A1, B)
                DCPT_LOAD_A(s, 1)
                stage(3, s ^ 1, kt + 1, dead1);
                DCPT_PUBLISH_N(12)
                DCPT_MFMA(1, 0)
                // phase 2: (A2, B)
                DCPT_LOAD_A(s, 2)
                stage(0, s, kt + 2, dead2);
                stage(4, s, kt + 2, dead2);
                DCPT_PUBLISH_N(14)
                DCPT_MFMA(2, 0)
                // phase 3: (A3, B)
                DCPT_LOAD_A(s, 3)
                stage(1, s, kt + 2, dead2);
                DCPT_PUBLISH_N(12)
                DCPT_MFMA(3, 0)
            }
        }
    } else
    for (int t = 0; t < nkt; t += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kt = t + s;
            const uint32_t dead1 = (kt + 1 < nkt) ? 0u : ROW_SENT, dead2 = (kt + 2 < nkt) ? 0u : ROW_SENT;
            // phase 0: (A-lo, B-lo)
            DCPT_LOAD_AB(s)
            stage(3, s ^ 1, kt + 1, dead1);
            DCPT_PUBLISH()
            DCPT_MFMA(0, 0)
            // phase 1: (A-lo, B-hi)
            DCPT_LOAD_B(s, 1)
            stage(1, s ^ 1, kt + 1, dead1);
            DCPT_PUBLISH()
            DCPT_MFMA(0, 1)
            // phase 2: (A-hi, B-hi)
            DCPT_LOAD_A(s, 1)
            stage(0, s, kt + 2, dead2);
            DCPT_PUBLISH()
            DCPT_MFMA(1, 1)
            // phase 3: (A-hi, B-lo)
            stage(2, s, kt + 2, dead2);
            DCPT_PUBLISH()
            DCPT_MFMA(1, 0)
        }
    }
#undef DCPT_LOAD_A
#undef DCPT_LOAD_B
#undef DCPT_LOAD_AB
#undef DCPT_MFMA
#undef DCPT_MFMA_OP
#undef DCPT_PUBLISH
#undef DCPT_PUBLISH_N
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
    dma_wait_all();                               // the zero-filling DMAs past the last k-tile, before LDS is reused
    __syncthreads();

    // epilogue: quadrant (a, b) of every wave = the 128 x 128 sub-block (a, b) of the tile
    float* const Cs0 = reinterpret_cast<float*>(smem);
#ifdef DCPT_ABL_NOEPI   // ablation build: accumulators stay live, nothing is parked or written
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(acc[a][b][i]));
    return;
#endif
    if constexpr (TALL) {
        // four 128 x 128 quadrants (A0..A3 against the one B half): whole rows of a 128-column output, so the LayerNorm epilogues run as they are
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float* const Cs = Cs0 + (a & 1) * (128 * 128);   // alternate two 64 KB buffers: one barrier per quadrant
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = wn * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Cs[ml * 128 + nl] = acc[a][0][i][r];
                }
            }
            __syncthreads();
            epilogue8<EK, 128, 128, 512>(p, Cs, m0 + a * 128, n0, tid);
            if constexpr (EK == EB_DOTCOL || EK == EB_LNBWD2 || EK == EB_LNBWDM) __syncthreads();   // (their column sums reuse the buffer)
        }
    } else if constexpr (EK == EB_LNFWD || EK == EB_LNBWDM) {
        // a LayerNorm epilogue needs whole rows: the tile is parked as two [128][256] halves (N == 256: one column tile per row block)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* const Cs = Cs0;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int nl = b * 128 + wn * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[ml * 256 + nl] = acc[a][b][i][r];
                    }
                }
            __syncthreads();
            epilogue8<EK, 128, 256, 512>(p, Cs, m0 + a * 128, n0, tid);
            __syncthreads();
        }
    } else if constexpr (GATE) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* const Cs = Cs0;   // [128][256]: columns 0..127 first gate half, 128..255 second
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int nl = b * 128 + wn * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[ml * 256 + nl] = acc[a][b][i][r];
                    }
                }
            __syncthreads();
            epilogue8<EK, 128, 256, 512>(p, Cs, m0 + a * 128, n0, tid);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = q >> 1, b = q & 1;
            float* const Cs = Cs0 + (q & 1) * (128 * 128);   // alternate two 64 KB buffers: one barrier per quadrant
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = wn * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Cs[ml * 128 + nl] = acc[a][b][i][r];
                }
            }
            __syncthreads();
            epilogue8<EK, 128, 128, 512>(p, Cs, m0 + a * 128, n0 + b * 128, tid);
            if constexpr (EK == EB_DOTCOL || EK == EB_LNBWD2) __syncthreads();   // (their column sums reuse the buffer)
        }
    }
}

}  // namespace

// Eligibility of a launch for the 256 x 256 kernel: plain operands, full 256-column tiles (128 per gate half), k-tiles in pairs, and
// enough tiles to fill most of the 256 CUs (one block per CU).
bool gemm_nt_bf16_256_ok(const GemmNTB& p, int epi, int min_tiles) {
    if (p.gather2 || epi == EB_SCATTER || epi == EB_SCATTER_ADD || epi == EB_LNBWD2) return false;
    const bool lnepi = epi == EB_LNFWD || epi == EB_LNBWDM;
    if (lnepi && p.N != 256) return false;
    if (p.conv3 && ((epi != EB_PLAIN && !lnepi) || p.gC % 64 != 0 || p.K != 9 * p.gC || p.K / 64 >= 2048 || (p.nb > 1))) return false;
    if (p.K % 128 != 0 || p.K < 128) return false;
    if (p.N % 256 != 0) return false;   // (gate: 128 columns of each half)
    const int nb = p.nb > 0 ? p.nb : 1;
    const int64_t tiles = cdiv64(p.M, 256) * (p.N / 256) * nb;
    return tiles >= min_tiles;
}

// The 512 x 128 tile (TALL): N == 128 exactly, plain or implicit-3x3 A, k-tiles in pairs, the epilogues the classifier head launches at that
// width, and enough tiles for one block per CU.
bool gemm_nt_bf16_tall_ok(const GemmNTB& p, int epi, int min_tiles) {
    if (p.N != 128 || p.gather2 || p.K % 128 != 0 || p.K < 128 || (p.nb > 1)) return false;
    // (the masked LayerNorm-backward epilogue stays on the 128-row kernel: on this tile it measured +0.2 ms per launch at the head's stage 0 --
    // four quadrant epilogues with their column sums in a row and no second block on the CU to run under them; profiles/r6/head_tall_tile/)
    if (!(epi == EB_PLAIN || epi == EB_RESID || epi == EB_LNFWD)) return false;
    if (p.conv3 && (epi == EB_RESID || p.gC % 64 != 0 || p.K != 9 * p.gC || p.K / 64 >= 2048)) return false;
    return cdiv64(p.M, 512) >= min_tiles;
}

int launch_gemm_nt_bf16_tall(const GemmNTB& p, int epi, hipStream_t s) {
    const dim3 grid((unsigned)cdiv64(p.M, 512));
    if (p.conv3) {
        if (epi == EB_LNFWD) gemm_nt_bf16_256_kernel<EB_LNFWD, 1, true><<<grid, dim3(512), 0, s>>>(p);
        else gemm_nt_bf16_256_kernel<EB_PLAIN, 1, true><<<grid, dim3(512), 0, s>>>(p);
    } else {
        if (epi == EB_LNFWD) gemm_nt_bf16_256_kernel<EB_LNFWD, 0, true><<<grid, dim3(512), 0, s>>>(p);
        else if (epi == EB_RESID) gemm_nt_bf16_256_kernel<EB_RESID, 0, true><<<grid, dim3(512), 0, s>>>(p);
        else gemm_nt_bf16_256_kernel<EB_PLAIN, 0, true><<<grid, dim3(512), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_nt_bf16 tall");
    return DCPT_OK;
}

int launch_gemm_nt_bf16_256(const GemmNTB& p, int epi, hipStream_t s) {
    const unsigned nb = (unsigned)(p.nb > 0 ? p.nb : 1);
    const dim3 grid((unsigned)(cdiv64(p.M, 256) * (p.N / 256)), nb);
    if (p.conv3) {
        if (epi == EB_LNFWD) gemm_nt_bf16_256_kernel<EB_LNFWD, 1><<<grid, dim3(512), 0, s>>>(p);
        else if (epi == EB_LNBWDM) gemm_nt_bf16_256_kernel<EB_LNBWDM, 1><<<grid, dim3(512), 0, s>>>(p);
        else gemm_nt_bf16_256_kernel<EB_PLAIN, 1><<<grid, dim3(512), 0, s>>>(p);
        DCPT_CHECK_LAUNCH("gemm_nt_bf16_256 conv3");
        return DCPT_OK;
    }
    switch (epi) {
        case EB_PLAIN: gemm_nt_bf16_256_kernel<EB_PLAIN><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_BIAS: gemm_nt_bf16_256_kernel<EB_BIAS><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_RESID: gemm_nt_bf16_256_kernel<EB_RESID><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_SGBWD: gemm_nt_bf16_256_kernel<EB_SGBWD><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_BIASGATE: gemm_nt_bf16_256_kernel<EB_BIASGATE><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_DOTCOL: gemm_nt_bf16_256_kernel<EB_DOTCOL><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_LNFWD: gemm_nt_bf16_256_kernel<EB_LNFWD><<<grid, dim3(512), 0, s>>>(p); break;
        case EB_LNBWDM: gemm_nt_bf16_256_kernel<EB_LNBWDM><<<grid, dim3(512), 0, s>>>(p); break;
        default: dcpt_set_error("gemm_nt_bf16_256: epilogue %d not supported", epi); return DCPT_ERR_ARG;
    }
    DCPT_CHECK_LAUNCH("gemm_nt_bf16_256");
    return DCPT_OK;
}
