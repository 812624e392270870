// Grouped bf16 weight-gradient GEMM with 256 x 256 output tiles:  G_p[n][k] = sum_m X_p[m][n] * Y_p[m][k]  for up to TNG_MAX problems
// in ONE launch (the four weight gradients of a wide NAFBlock -- conv5, conv4, conv3, conv1 -- once dt1 exists; a dense 3 x 3 of the
// classifier head with its gathered-tap Y operand; reference basicsr/archs/nafnet_arch.py:165-186, degrad_classify_arch.py:132-243), and
// the FINISHER that turns the partial sums into parameter gradients together with every other small parameter-gradient reduction of the
// block -- one launch instead of ten.
//
// Why grouped, why 256.  A CU holds one 256 x 256 fp32 tile in its accumulators, the chip 256 of them; a single 1024 x 512 weight
// gradient has 8 such tiles, so all CUs working on it means 32 partial sums per output element (64 MB of fp32 slabs written and read
// back for 96 MB of operands).  The 128 x 128 kernel (gemm_bf16.hip) has the same slab bytes and twice the operand traffic from L2
// (every X element K/128 times, every Y element N/128 times): 614-688 TF/s + a 9.6-us reducer per problem.  The four problems of a block
// in one launch are 24 tiles at C = 512, i.e. 10 partial sums per element (86 MB of slabs per block backward instead of 256 MB), half the
// L2 -> LDS fill per flop, 1 + 1 launches instead of 4 + 8 (847 TF/s for the grouped launch at level 3).
//
// Kernel = the schedule of the NT kernel (gemm_bf16_256.hip: 8 waves in two groups one barrier apart, a SIMD alternates between one
// wave issuing 8 MFMAs and its partner reading fragments + feeding the LDS-DMA queue; half-tile ring, counted vmcnt, DMAs never
// drained in the loop) with the operand side of the 128-wide TN kernel: a "k-tile" is 64 PIXELS, its half-tiles are row-major
// [64 pixels][128 columns] bf16 images (X-lo, X-hi, Y-lo, Y-hi; 16 KB each, 16-byte chunk c of pixel row r at position
// c ^ 4 (r & 3)), and an MFMA operand -- 8 consecutive pixels of one column -- comes out of LDS through ds_read_b64_tr_b16, two per
// operand, conflict-free (a half-wave reads 4 rows x 64 contiguous bytes, the swizzle puts the rows into different bank quarters).
// Phases of k-tile t (stage s = t & 1), as in the NT kernel:
//     p0: read X-lo, Y-lo   stage Y-hi(t+1)   MFMA (lo,lo)        p2: read X-hi    stage X-lo(t+2)   MFMA (hi,hi)
//     p1: read Y-hi         stage X-hi(t+1)   MFMA (lo,hi)        p3: --           stage Y-lo(t+2)   MFMA (hi,lo)
// Bias gradients (column sums of X) ride along as v_dot2c_f32_bf16 on the X fragments a wave holds anyway: the four waves that share
// an X strip take one 32-column tile each, the strip's lower 128 columns in the blocks of k tile column 0 and the upper 128 in those of
// k tile column 1 (16 VALU instructions per k-tile in the load slot of a phase that reads few fragments, the same load on every sibling tile).
//
// Slabs are stored in the ACCUMULATORS' OWN LAYOUT (a block's 256 KB = 256 chunks of [64 lanes][4 floats]: fully coalesced 16-byte
// stores straight from the registers, no LDS transpose); only the finisher knows the map back to (n, k): a lane's float4 is 4
// consecutive rows of one column, a chunk pair (lane < 32, lane >= 32) is an 8-row x 32-column patch -- the finisher's block owns 8
// full rows, so gain / bias gradients (row dots with W, nafblock.hip's gain algebra) need no second pass.  Fixed split order,
// no atomics: bit-reproducible.
#include "bf16.h"
#include "kernels.h"
#include "prof.h"

#ifdef TN256_TIMELINE
// Diagnostic build only (tools/tn256_timeline.py): per-block wall-clock stamps (100 MHz) at the start, after the prologue, at each quarter
// of the first segment's k-loop and at the end, plus the XCC id and the block's (problem, pixel range, tile)
__device__ unsigned long long g_tl_tn256[1024][10];
extern "C" int dcpt_timeline_read_tn256(unsigned long long* host, int nblk) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl_tn256), sizeof(unsigned long long) * 10 * (size_t)nblk);
}
#define TL256(i) if (threadIdx.x == 0 && blockIdx.x < 1024) g_tl_tn256[blockIdx.x][i] = wall_clock64();
#else
#define TL256(i)
#endif

namespace {

constexpr int HT = 16384;       // bytes of a half-tile
constexpr int STG = 4 * HT;     // one k-tile: X-lo | X-hi | Y-lo | Y-hi
constexpr int TILE_F = 65536;   // floats of one block's partial tile
constexpr int FIN_THREADS = 512;   // threads of a finisher block

typedef __attribute__((address_space(3))) bf16x4* lds_tr_p;
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* a) {   // rows r .. r + 3 and r + 4 .. r + 7 of the lane's column
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(const_cast<unsigned char*>(a)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_p)(const_cast<unsigned char*>(a + 4 * 256)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float4 f4_bcast(float s) { return make_float4(s, s, s, s); }
__device__ __forceinline__ void dot_ones(bf16x8 f, float (&c)[2]) {   // c[0] + c[1] += the 8 values (two independent chains: the kernel has no
    bf16x2v one;                                                       // register to spare for four)
    one.x = (__bf16)1.0f;
    one.y = (__bf16)1.0f;
    c[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), one, c[0], false);
    c[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), one, c[1], false);
    c[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), one, c[0], false);
    c[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), one, c[1], false);
}

template <bool YCONV>
__global__ __launch_bounds__(512) void gemm_tn_bf16_256_kernel(const GemmTNG g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;              // 0: leads, 1: one barrier behind
    const int wm = wave >> 2, wn = wave & 3;
    TnProb p = g.p[0];   // (wave-uniform selects on kernel arguments: no dynamically indexed copy of the table)
    int split, tile, tiles, pidx = 0;
    if (g.xcd_slots > 0) {   // XCD-aligned order (GemmTNG): slot b >> 3 of XCD b & 7
        const int sh = 6 * (int)(blockIdx.x & 7);
        int slot = (int)(blockIdx.x >> 3), unit = -1, tin = 0, usz = 1;
#pragma unroll
        for (int i = 0; i < TNG_MAX; ++i) {
            if (i < g.n && unit < 0) {
                const int u0 = (int)((g.xq[i] >> sh) & 63ull), u1 = (int)((g.xq[i] >> (sh + 6)) & 63ull);
                const int tl = (g.p[i].N >> 8) * g.p[i].tiles_k, us = tl < 4 ? tl : 4;
                const int cnt = (u1 - u0) * us;
                if (slot < cnt) {
                    p = g.p[i];
                    pidx = i;
                    usz = us;
                    unit = u0 + slot / us;
                    tin = slot % us;
                } else {
                    slot -= cnt;
                }
            }
        }
        if (unit < 0) return;   // an idle slot of this XCD
        tiles = (p.N >> 8) * p.tiles_k;
        const int ups = tiles / usz;
        split = unit / ups;
        tile = (unit - split * ups) * usz + tin;
    } else {
        const int lin = xcd_remap(blockIdx.x, gridDim.x);
#pragma unroll
        for (int i = 1; i < TNG_MAX; ++i)
            if (i < g.n && lin >= g.p[i].blk0) {
                p = g.p[i];
                pidx = i;
            }
        const int rel = lin - p.blk0;
        tiles = (p.N >> 8) * p.tiles_k;
        split = rel / tiles;   // (consecutive blocks = the tiles of one pixel range: one XCD's L2 serves them)
        tile = rel - split * tiles;
    }
#ifdef DCPT_TUNING   // diagnostic builds: only the blocks of ONE problem of the group run (per-problem fabric traffic, tools/exp_r5f.sh)
    if (g.only >= 0 && pidx != g.only) return;
#endif
    TL256(0)
#ifdef TN256_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_tl_tn256[blockIdx.x][8] = ((unsigned long long)(xcc & 15u) << 48) | ((unsigned long long)pidx << 32) | ((unsigned long long)split << 16) | (unsigned)tile;
        g_tl_tn256[blockIdx.x][9] = 1;
    }
#endif
    const int tile_n = tile / p.tiles_k, tile_k = tile - tile_n * p.tiles_k;
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int64_t mbeg = (int64_t)split * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const int brows = mbeg < mend ? (int)(mend - mbeg) : 0;   // rows of this block
    // segments: the block's rows in pieces of seg_rows (an image), each with its own partial-sum slot; 0 = the whole range is one piece
    const int seg_rows = p.seg_rows > 0 ? p.seg_rows : (int)p.rows_per_split;
    const int nseg = (brows + seg_rows - 1) / seg_rows;
    const int slot0 = p.seg_rows > 0 ? (int)(mbeg / seg_rows) : split;
    const int64_t mw = mbeg < p.M ? mbeg : 0;

    const i32x4 rsX = make_rsrc_dma(p.X + mw * (int64_t)p.ldx + n0);
    // gathered Y (weight gradient of a dense 3 x 3): the window starts one image row + one pixel before the block's first pixel (clipped at
    // the tensor start); a 128-column half of the tile lies inside ONE tap (gC % 128 == 0)
    int64_t ypix0 = mw;
    if constexpr (YCONV) {
        ypix0 -= p.gW + 1;
        if (ypix0 < 0) ypix0 = 0;
    }
    const i32x4 rsY = make_rsrc_dma(YCONV ? p.Y + ypix0 * (int64_t)p.gC : p.Y + mw * (int64_t)p.ldy + k0);
    // staging map of a half-tile (two passes of 32 pixel rows): wave w issues rows 4 w .. 4 w + 3 of each pass; lane -> row (lane >> 4),
    // chunk POSITION lane & 15, which holds source chunk (lane & 15) ^ 4 (row & 3)
    int srow[2];
    uint32_t voffX[2][2], voffY[2][2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int row = 32 * ps + 4 * wave + (lane >> 4);
        const int c = (lane & 15) ^ (4 * (row & 3));
        srow[ps] = row;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            voffX[h][ps] = ((uint32_t)row * (uint32_t)p.ldx + (uint32_t)(128 * h + 8 * c)) * 2u;
            if constexpr (YCONV) {   // (tap, channel) of the half's first column; offset relative to the row's own pixel (may be negative: wraps)
                const int kk = k0 + 128 * h, tap = kk / p.gC, ch = kk - tap * p.gC;
                const int ky = tap / 3, kx = tap - 3 * ky;
                voffY[h][ps] = (uint32_t)(((int)(mw - ypix0) + row) * p.gC + ch + 8 * c + ((ky - 1) * p.gW + (kx - 1)) * p.gC) * 2u;
            } else {
                voffY[h][ps] = ((uint32_t)row * (uint32_t)p.ldy + (uint32_t)(128 * h + 8 * c)) * 2u;
            }
        }
    }
    // gathered Y: pixel coordinates of this thread's two rows in the Y tile staged last (advanced by 64 pixels per tile), and the two halves' taps
    int yh[2] = {0, 0}, yw[2] = {0, 0};
    int ydy[2] = {0, 0}, ydx[2] = {0, 0};
    if constexpr (YCONV) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = (k0 + 128 * h) / p.gC;
            ydy[h] = tap / 3 - 1;
            ydx[h] = tap - 3 * (tap / 3) - 1;
        }
    }
    const int adv_h = YCONV ? 64 / p.gW : 0, adv_w = YCONV ? 64 % p.gW : 0;
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(smem));
    const uint32_t lds_w = lds0 + (uint32_t)wave * 1024u;

    // fragment addresses: lane t = lane & 15 of a 16-lane group passes row (t >> 2) (+ 8 fh for the upper 8 pixels of a 16-pixel step),
    // the 8 bytes at columns 4 (t & 3) .. of its 16-column group; the read returns column t of the 4 x 16 block
    const int t16 = lane & 15, g16 = (lane >> 4) & 1, fh = lane >> 5;
    const int swz = 4 * ((t16 >> 2) & 3);
    const int rowoff = ((t16 >> 2) + 8 * fh) * 256;
    const unsigned char* abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = wm * 64 + i * 32 + 16 * g16 + 4 * (t16 & 3);
        abase[i] = smem + rowoff + (((c >> 3) ^ swz) << 4) + (c & 7) * 2;
    }
    const unsigned char* bbase;
    {
        const int c = wn * 32 + 16 * g16 + 4 * (t16 & 3);
        bbase = smem + 2 * HT + rowoff + (((c >> 3) ^ swz) << 4) + (c & 7) * 2;
    }
    // wave (wm, wn) sums X columns  (wn >> 1) * 128 + wm * 64 + (wn & 1) * 32 + 0..31 -- the two 128-column halves in DIFFERENT tiles of the
    // strip where it has two or more (k tile columns 0 and 1): the sums are 16 VALU instructions per k-tile, and with all of them in the first
    // tile column (and in the waves' load phase) those blocks ran 18 % longer than their siblings (95 us against 80 at level 3,
    // tools/tn256_timeline.py) -- which stretches the launch AND takes the siblings out of step, so that the late ones find the shared operand
    // columns evicted from the XCD's L2 (459 MB over the fabric for 335 MB of operands)
    float cs[2] = {0.f, 0.f};
    const bool do_cs = !YCONV && (p.colsum != nullptr) && (p.tiles_k == 1 ? tile_k == 0 : tile_k == (wn >> 1));   // (gathered Y: never with column sums)

#ifdef TN_ABL_NOMFMA   // ablation build: operands stay live, no matrix work
#define TN_MFMA_OP(ACC, FA, FB) asm volatile("" ::"v"(FA), "v"(FB))
#else
#define TN_MFMA_OP(ACC, FA, FB) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA, FB, ACC, 0, 0, 0)
#endif
#define TN_LOAD_A(S, H)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)                           \
        fa[i][j] = tr_read8(abase[i] + (S)*STG + (H)*HT + j * 4096);
#define TN_LOAD_B(S, H)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[H][j] = tr_read8(bbase + (S)*STG + (H)*HT + j * 4096);
#define TN_LOAD_AB(S)                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                       \
        fa[0][j] = tr_read8(abase[0] + (S)*STG + j * 4096);                                                               \
        fb[0][j] = tr_read8(bbase + (S)*STG + j * 4096);                                                                  \
        fa[1][j] = tr_read8(abase[1] + (S)*STG + j * 4096);                                                               \
    }
#define TN_MFMA(AH, BH)                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i)                           \
        TN_MFMA_OP(acc[AH][BH][i], fa[i][j], fb[BH][j]);                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();
// column sums of X half H from the fragments in fa: issued in the load slot of the phase that has the FEWEST fragment reads while fa
// still holds that half (phase 1 for the lower half, phase 3 for the upper one; phases 0 and 2 read 12 and 8 fragments, 1 and 3 read 4 and 0)
#define TN_COLSUM(H)                                                                                                      \
    if (do_cs && (wn >> 1) == (H)) {                                                                                      \
        if (wn & 1) {                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) dot_ones(fa[1][j], cs);                                         \
        } else {                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) dot_ones(fa[0][j], cs);                                         \
        }                                                                                                                 \
    }
#define TN_PUBLISH()                                                                                                      \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                      \
    __builtin_amdgcn_s_barrier();

    for (int seg = 0; seg < nseg; ++seg) {
        const int row0 = seg * seg_rows;
        const int nrows = brows - row0 < seg_rows ? brows - row0 : seg_rows;
        const int nkt = (nrows + 63) >> 6;
        // Q: 0 X-lo, 1 X-hi, 2 Y-lo, 3 Y-hi of pixel tile kt into stage s; rows past the segment (and whole tiles past its last one) are
        // range-checked away: the DMA zero-fills
        auto stage = [&](int Q, int s, int kt) {
            const uint32_t dst = lds_w + (uint32_t)(s * STG + Q * HT);
#ifdef TN_ABL_NOLOAD   // ablation builds (tools/build_variant.sh): every DMA is range-checked away (zero fill, no memory traffic)
            const int left = 0;
#else
            const int left = nrows - kt * 64;
#endif
            if (Q < 2) {
                const uint32_t soff = (uint32_t)(row0 + kt * 64) * 2u * (uint32_t)p.ldx;
                dma16(rsX, dst, srow[0] < left ? voffX[Q][0] : ROW_SENT, soff);
                dma16(rsX, dst + 8192u, srow[1] < left ? voffX[Q][1] : ROW_SENT, soff);
            } else if constexpr (YCONV) {
                // the Y stages come in the order lo(k), hi(k), lo(k + 1), hi(k + 1), ...: a lo stage moves the pixel coordinates to its tile
                if (Q == 2) {
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        if (kt == 0) {
                            const int64_t m = mw + row0 + srow[ps];
                            yw[ps] = (int)(m % p.gW);
                            yh[ps] = (int)((m / p.gW) % p.gH);
                        } else {
                            int w2 = yw[ps] + adv_w, h2 = yh[ps] + adv_h;
                            if (w2 >= p.gW) {
                                w2 -= p.gW;
                                ++h2;
                            }
                            while (h2 >= p.gH) h2 -= p.gH;
                            yw[ps] = w2;
                            yh[ps] = h2;
                        }
                    }
                }
                const int dy = ydy[Q - 2], dx = ydx[Q - 2];
                // (the tile's advance is added to the per-lane offset, not passed as the scalar offset: a tap above / left of the pixel makes
                // the lane's own part negative, and the range check looks at the lane's part alone)
                const uint32_t adv = (uint32_t)(row0 + kt * 64) * 2u * (uint32_t)p.gC;
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int hh = yh[ps] + dy, ww = yw[ps] + dx;
                    const bool ok = srow[ps] < left && hh >= 0 && hh < p.gH && ww >= 0 && ww < p.gW;
                    dma16(rsY, dst + 8192u * ps, ok ? voffY[Q - 2][ps] + adv : ROW_SENT, 0);
                }
            } else {
                const uint32_t soff = (uint32_t)(row0 + kt * 64) * 2u * (uint32_t)p.ldy;
                dma16(rsY, dst, srow[0] < left ? voffY[Q - 2][0] : ROW_SENT, soff);
                dma16(rsY, dst + 8192u, srow[1] < left ? voffY[Q - 2][1] : ROW_SENT, soff);
            }
        };
        floatx16 acc[2][2][2];   // [X half][Y half][n-tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;
        bf16x8 fa[2][4], fb[2][4];   // X strip in use [n-tile][pixel step]; Y strips [half][pixel step]

        const int nkt2 = (nkt + 1) & ~1;   // the loop runs k-tiles in pairs (compile-time stage index); an odd tail multiplies zeros
        stage(0, 0, 0);
        stage(2, 0, 0);
        stage(3, 0, 0);
        stage(1, 0, 0);
        stage(0, 1, 1);
        stage(2, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X-lo(0), Y-lo(0) landed (this wave's part; the barrier publishes all)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind from here on
        if (seg == 0) { TL256(1) }

        for (int t = 0; t < nkt2; t += 2) {
#ifdef TN256_TIMELINE
            if (seg == 0 && nkt2 >= 8) {
                const int qt = (nkt2 / 4) & ~1;
                if (t == qt) { TL256(2) }
                if (t == 2 * qt) { TL256(3) }
                if (t == 3 * qt) { TL256(4) }
            }
#endif
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kt = t + s;
                // phase 0: (X-lo, Y-lo)
                TN_LOAD_AB(s)
                stage(3, s ^ 1, kt + 1);
                TN_PUBLISH()
                TN_MFMA(0, 0)
                // phase 1: (X-lo, Y-hi)
                TN_LOAD_B(s, 1)
                stage(1, s ^ 1, kt + 1);
                TN_COLSUM(0)
                TN_PUBLISH()
                TN_MFMA(0, 1)
                // phase 2: (X-hi, Y-hi)
                TN_LOAD_A(s, 1)
                stage(0, s, kt + 2);
                TN_PUBLISH()
                TN_MFMA(1, 1)
                // phase 3: (X-hi, Y-lo)
                stage(2, s, kt + 2);
                TN_COLSUM(1)
                TN_PUBLISH()
                TN_MFMA(1, 0)
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier: every wave is past its last fragment read
        dma_wait_all();                               // (the zero-filling DMAs past the last k-tile, before the next segment restages)
        if (seg == 0) { TL256(5) }

#ifdef TN_ABL_NOEPI   // ablation build: accumulators stay live, nothing is written
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(acc[a][b][i]));
#else
        // the partial tile in the accumulators' own layout: chunk (wave, a, b, i, q) = [64 lanes][4 floats]
        float* const out = p.slab + ((int64_t)(slot0 + seg) * tiles + tile) * TILE_F + wave * (32 * 256) + lane * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gidx = ((a * 2 + b) * 2 + i) * 4 + q;
                        const floatx16 v = acc[a][b][i];
                        *reinterpret_cast<float4*>(out + gidx * 256) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    }
#endif
    }
#undef TN_LOAD_A
#undef TN_LOAD_B
#undef TN_LOAD_AB
#undef TN_COLSUM
#undef TN_MFMA
#undef TN_MFMA_OP
#undef TN_PUBLISH
    TL256(6)
    if (do_cs) {   // lanes l and l + 32 hold the two pixel halves of column l
        const float own = cs[0] + cs[1];
        const float tot = own + __shfl_xor(own, 32);
        if (lane < 32) p.colsum[(int64_t)split * p.N + n0 + (wn >> 1) * 128 + wm * 64 + (wn & 1) * 32 + lane] = tot;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Finisher: block ranges -> jobs.
//   slab job: rows 8 rg .. 8 rg + 7 of one problem.  Thread (cgb = tid >> 6, lane): column groups cg = cgb, cgb + 16, ... of 32 columns;
//             lane & 31 = column, lane >> 5 = row half; its float4 = rows 4 h + 0..3.
//   cols job: out_j[c] = sum_r part[r][j][c]   (LayerNorm weight / bias gradients; the depthwise taps with their own output map)
//   sca job : dWsca[n][k] = sum_b ds[b][n] pooled[b][k],  dbsca[n] = sum_b ds[b][n]
__device__ __forceinline__ void fin_slab(const FinSlab& j, int rg, float* red /* [16][8] + [8] */) {
    const int tid = threadIdx.x, lane = tid & 63, cgb = tid >> 6;   // 512 threads: 8 column groups at a time (two blocks per CU: the slab jobs
                                                                     // of a wide NAFBlock -- 384 blocks -- are resident in one round)
    const int h = lane >> 5, kl = lane & 31;
    const int tile_n = rg >> 5, r32 = rg & 31;
    const int a = r32 >> 4, wm = (r32 >> 3) & 1, i = (r32 >> 2) & 1, q = r32 & 3;
    const int tiles = (j.N >> 8) * j.tiles_k;
    const int64_t sstride = (int64_t)tiles * TILE_F;
    const int nbase = 8 * rg + 4 * h;
    float rsc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rsc[e] = j.rowscale ? j.rowscale[nbase + e] : 1.f;
    float4 dot = f4_zero();
    for (int cg = cgb; cg < (j.K >> 5); cg += FIN_THREADS / 64) {
        const int tile_k = cg >> 3, b = (cg >> 2) & 1, wn = cg & 3;
        const int chunk = (wm * 4 + wn) * 32 + ((a * 2 + b) * 2 + i) * 4 + q;
        const float* base = j.slab + (int64_t)(tile_n * j.tiles_k + tile_k) * TILE_F + chunk * 256 + lane * 4;
        const int k = cg * 32 + kl;
        float4 gs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gs[u] = f4_zero();
        int s = 0;
        if (j.kscale == nullptr) {
            for (; s + 7 < j.splits; s += 8) {   // eight independent sums keep eight loads in flight (fixed order: deterministic)
#pragma unroll
                for (int u = 0; u < 8; ++u) gs[u] = f4_add(gs[u], ldg4(base + (int64_t)(s + u) * sstride));
            }
            // the tail (fewer than eight partial sums left) as eight loads at once as well: the ones past the end re-read the last partial sum
            // with weight 0 (a load that depends on the loop counter of a runtime loop waits for the one before it: with 10 partial sums
            // per element that was three round trips per column group instead of two)
            if (s < j.splits) {
                const int rem = j.splits - s;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 v = ldg4(base + (int64_t)(s + (u < rem ? u : rem - 1)) * sstride);
                    gs[u] = f4_fma(v, f4_bcast(u < rem ? 1.f : 0.f), gs[u]);
                }
            }
        } else {   // every partial sum lies inside one image: it is weighted by that image's per-column scale (SCA)
            const float* ks = j.kscale + k;
            for (; s + 7 < j.splits; s += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    gs[u] = f4_fma(ldg4(base + (int64_t)(s + u) * sstride), f4_bcast(ks[(int64_t)((s + u) / j.ks_div) * j.K]), gs[u]);
            }
            if (s < j.splits) {
                const int rem = j.splits - s;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int su = s + (u < rem ? u : rem - 1);
                    const float4 v = ldg4(base + (int64_t)su * sstride);
                    gs[u] = f4_fma(v, f4_bcast(u < rem ? ks[(int64_t)(su / j.ks_div) * j.K] : 0.f), gs[u]);
                }
            }
        }
        const float4 G = f4_add(f4_add(f4_add(gs[0], gs[1]), f4_add(gs[2], gs[3])), f4_add(f4_add(gs[4], gs[5]), f4_add(gs[6], gs[7])));
        const float ge[4] = {G.x, G.y, G.z, G.w};
        float de[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t ko = k;   // position of column k inside a row of dW
        if (j.conv3) {     // packed k = tap * Ci + ic  ->  dW[n][ic][tap]
            const int Ci = j.K / 9, tap = k / Ci;
            ko = (int64_t)(k - tap * Ci) * 9 + tap;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t o = (int64_t)(nbase + e) * j.K + ko;
            j.dW[o] = rsc[e] * ge[e];
            if (j.dgain) de[e] = j.W[o] * ge[e];
        }
        dot = f4_add(dot, make_float4(de[0], de[1], de[2], de[3]));
    }
    if (j.dgain == nullptr && j.dbias == nullptr) return;   // (uniform for the block)
    // row dots: 32 lanes of a half wave, then the waves; column sums of X: 32 threads per row over the blocks' partial sums
    dot.x = group_sum(dot.x, 32);
    dot.y = group_sum(dot.y, 32);
    dot.z = group_sum(dot.z, 32);
    dot.w = group_sum(dot.w, 32);
    if (kl == 0) {
        red[cgb * 8 + 4 * h + 0] = dot.x;
        red[cgb * 8 + 4 * h + 1] = dot.y;
        red[cgb * 8 + 4 * h + 2] = dot.z;
        red[cgb * 8 + 4 * h + 3] = dot.w;
    }
    if (tid < 256) {
        const int row = tid >> 5, jj = tid & 31;
        float c = 0.f;
        if (j.colsum)
            for (int s = jj; s < j.cs_rows; s += 32) c += j.colsum[(int64_t)s * j.N + 8 * rg + row];
        c = group_sum(c, 32);
        if (jj == 0) red[FIN_THREADS / 8 + row] = c;
    }
    __syncthreads();
    if (tid < 8) {
        const int n = 8 * rg + tid;
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < FIN_THREADS / 64; ++w) d += red[w * 8 + tid];
        const float c = red[FIN_THREADS / 8 + tid];
        if (j.dgain) j.dgain[n] = d + (j.wbias ? j.wbias[n] * c : 0.f);
        if (j.dbias) j.dbias[n] = (j.rowscale ? j.rowscale[n] : 1.f) * c;
    }
}

__device__ __forceinline__ void fin_cols(const FinCols& j, int blk, float* red /* [FIN_THREADS / 16][16] */) {
    const int nbx = (j.C + 15) >> 4;
    const int jj = blk / nbx, bx = blk - jj * nbx;
    const int cl = threadIdx.x & 15, rgp = threadIdx.x >> 4;
    const int c = bx * 16 + cl;
    float s = 0.f;
    if (c < j.C) {
#pragma unroll 4
        for (int r = rgp; r < j.R; r += FIN_THREADS / 16) s += j.part[((int64_t)r * j.nj + jj) * j.C + c];
    }
    red[rgp * 16 + cl] = s;
    __syncthreads();
    if (rgp == 0 && c < j.C) {
        float t = red[cl];
#pragma unroll
        for (int i = 1; i < FIN_THREADS / 16; ++i) t += red[i * 16 + cl];
        if (j.mode == 0) {
            float* o = jj == 0 ? j.out0 : j.out1;
            if (o) o[c] = t;
        } else {   // depthwise taps: rows 0..8 -> dw2[c][tap], row 9 -> db2[c]
            if (jj < 9) j.out0[c * 9 + jj] = t;
            else j.out1[c] = t;
        }
    }
}

__device__ __forceinline__ void fin_sca(const FinSca& j, int blk) {
    const int nb0 = (int)(((int64_t)j.C * j.C + FIN_THREADS - 1) / FIN_THREADS);
    if (blk < nb0) {
        const int64_t i = (int64_t)blk * FIN_THREADS + threadIdx.x;
        if (i >= (int64_t)j.C * j.C) return;
        const int n = (int)(i / j.C), k = (int)(i % j.C);
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < j.B; ++b) s = fmaf(j.ds[(int64_t)b * j.C + n], j.pooled[(int64_t)b * j.C + k], s);
        j.dW[i] = s;
    } else {
        const int i = (blk - nb0) * FIN_THREADS + threadIdx.x;
        if (i >= j.C) return;
        float s = 0.f;
        for (int b = 0; b < j.B; ++b) s += j.ds[(int64_t)b * j.C + i];
        j.db[i] = s;
    }
}

__global__ __launch_bounds__(FIN_THREADS) void wgrad_finish_kernel(const FinJobs jobs) {
    __shared__ float red[FIN_THREADS];
    const int blk = blockIdx.x;
    if (blk < jobs.slab_end) {
        FinSlab q = jobs.slab[0];
        int b0 = 0;
#pragma unroll
        for (int i = 1; i < TNG_MAX; ++i)
            if (i < jobs.nslab && blk >= jobs.slab_blk0[i]) {
                q = jobs.slab[i];
                b0 = jobs.slab_blk0[i];
            }
        fin_slab(q, blk - b0, red);
    } else if (blk < jobs.cols_end) {
        FinCols q = jobs.cols[0];
        int b0 = jobs.cols_blk0[0];
#pragma unroll
        for (int i = 1; i < FIN_MAX_COLS; ++i)
            if (i < jobs.ncols && blk >= jobs.cols_blk0[i]) {
                q = jobs.cols[i];
                b0 = jobs.cols_blk0[i];
            }
        fin_cols(q, blk - b0, red);
    } else {
        fin_sca(jobs.sca, blk - jobs.cols_end);
    }
}

}  // namespace

bool gemm_tn_bf16_256_ok(int N, int K) {
    static const int on = dcpt_tuning("DCPT_TN256", 1);
    return on && N >= 256 && K >= 256 && N % 256 == 0 && K % 256 == 0 && N <= 4096 && K <= 16384;
}

// Pixel ranges of a grouped launch: one block per CU in a single round.  Every tile of every problem gets about
// target_blocks / (tiles of the group) blocks, i.e. the same number of pixels per block everywhere (the launch ends when its longest block
// does), in multiples of 64 pixels and of at least 256.  A problem with img_P[i] > 0 needs its partial sums per image (the finisher weights
// them with that image's scale): its blocks take a divisor of the image (large images) or a whole number of images, one slot each (small
// ones).  Returns false -- and plans nothing -- if such a problem cannot be cut that way (image not a multiple of 64 pixels, or so small
// that the per-image slots would cost more traffic than a scaled copy of the operand: the caller makes that copy instead).
// (a block's 256 KB of partial sums are worth 512 pixels of its two operand tiles: with fewer than ~1000 pixels per block a launch moves
// more slab bytes than operand bytes -- small batches fill fewer CUs instead; DCPT at 128 x 128: 48.7 -> ... us per grouped launch)
constexpr int64_t TN_MIN_ROWS_DEFAULT = 1024;

// The XCD-aligned block order of GemmTNG: units (<= 4 consecutive tiles of one pixel range) dealt to the 8 XCDs in contiguous runs of
// about the same number of blocks; a run boundary that would fall between the two units of an 8-tile pixel range moves to the range's end
// (or start) while the XCD keeps at most 32 blocks -- one per CU.  Left off (xcd_slots = 0) where a problem's tile count is no multiple of its unit or a
// problem has more than 63 units.
static void plan_xcd(GemmTNG& g) {
    static const int on = dcpt_tuning("DCPT_TN_XCD", 1);
    g.xcd_slots = 0;
    g.only = dcpt_tuning("DCPT_TN_ONLY", -1);
    for (int i = 0; i < TNG_MAX; ++i) g.xq[i] = 0;
    if (!on) return;
    int usz[TNG_MAX], ups[TNG_MAX], units[TNG_MAX], off[TNG_MAX + 1], total_blocks = 0;
    off[0] = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = (g.p[i].N / 256) * g.p[i].tiles_k;
        usz[i] = tl < 4 ? tl : 4;
        if (tl % usz[i] != 0) return;
        ups[i] = tl / usz[i];
        units[i] = g.p[i].splits * ups[i];
        if (units[i] > 63) return;
        off[i + 1] = off[i] + units[i];
        total_blocks += units[i] * usz[i];
    }
    const int U = off[g.n];
    auto prob_of = [&](int u) {
        int i = 0;
        while (i + 1 < g.n && u >= off[i + 1]) ++i;
        return i;
    };
    int cut[9];
    cut[0] = 0;
    int cum = 0, u = 0, slots = 0;
    for (int x = 0; x < 8; ++x) {
        const int target = (int)(((int64_t)total_blocks * (x + 1) + 4) / 8);
        int run = 0;
        while (u < U) {
            const int i = prob_of(u);
            if (x < 7 && cum + usz[i] / 2 >= target && run > 0) break;   // (the last XCD takes what is left)
            cum += usz[i];
            run += usz[i];
            ++u;
        }
        if (x < 7 && u < U) {   // boundary inside a pixel range's units?  move it to the range's end, else to its start
            const int i = prob_of(u), r = (u - off[i]) % ups[i];
            if (r != 0 && ups[i] == 2) {   // (larger tile sets share enough inside their units)
                const int fwd = ups[i] - r;
                if (run + fwd * usz[i] <= 32) {
                    u += fwd;
                    cum += fwd * usz[i];
                    run += fwd * usz[i];
                } else if (run - r * usz[i] > 0) {
                    u -= r;
                    cum -= r * usz[i];
                    run -= r * usz[i];
                }
            }
        }
        cut[x + 1] = u;
        if (run > slots) slots = run;
    }
    for (int i = 0; i < g.n; ++i) {
        unsigned long long q = 0;
        for (int x = 0; x <= 8; ++x) {
            int v = cut[x] - off[i];
            v = v < 0 ? 0 : (v > units[i] ? units[i] : v);
            q |= (unsigned long long)v << (6 * x);
        }
        g.xq[i] = q;
    }
    g.xcd_slots = slots;
}

bool gemm_tn_bf16_256_plan(GemmTNG& g, const int* img_P, int target_blocks) {
    static const int64_t TN_MIN_ROWS = dcpt_tuning("DCPT_TN_MIN_ROWS", (int)TN_MIN_ROWS_DEFAULT);
    static const int force_blocks = dcpt_tuning("DCPT_TN_BLOCKS", 0);   // experiments: blocks per launch (default: one per CU)
    if (force_blocks > 0) target_blocks = force_blocks;
    int tiles = 0;
    for (int i = 0; i < g.n; ++i) {
        g.p[i].tiles_k = g.p[i].K / 256;
        tiles += (g.p[i].N / 256) * g.p[i].tiles_k;
        const int P = img_P ? img_P[i] : 0;
        if (P > 0 && (P % 64 != 0 || P < 512 || g.p[i].M % P != 0)) return false;
    }
    const int s0 = tiles > 0 && target_blocks / tiles > 1 ? target_blocks / tiles : 1;
    // first the problems that are tied to image boundaries, then the free ones share the blocks that are left
    int used = 0, tiles_free = 0;
    for (int i = 0; i < g.n; ++i) {
        TnProb& p = g.p[i];
        const int P = img_P ? img_P[i] : 0;
        const int tl = (p.N / 256) * p.tiles_k;
        p.seg_rows = 0;
        if (P <= 0) {
            tiles_free += tl;
            continue;
        }
        int64_t rows = cdiv64(cdiv64(p.M, s0), 64) * 64;
        if (rows >= P) {   // whole images per block, a slot per image
            const int64_t ipb = (rows + P / 2) / P;
            rows = ipb * P;
            p.seg_rows = P;
        } else {           // a divisor of the image per block (the largest one not much above the target)
            int64_t best = 64;
            for (int64_t d = 64; d <= P; d += 64)
                if (P % d == 0 && d <= rows + rows / 4) best = d;
            rows = best;
        }
        p.rows_per_split = rows;
        p.splits = (int)cdiv64(p.M, rows);
        p.slots = p.seg_rows > 0 ? (int)(p.M / p.seg_rows) : p.splits;
        used += p.splits * tl;
    }
    int sf = tiles_free > 0 ? (target_blocks - used) / tiles_free : 1;
    if (sf < 1) sf = 1;
    int blk = 0;
    for (int i = 0; i < g.n; ++i) {
        TnProb& p = g.p[i];
        if (!(img_P && img_P[i] > 0)) {
            int64_t rows = cdiv64(cdiv64(p.M, sf), 64) * 64;
            if (rows < TN_MIN_ROWS) rows = TN_MIN_ROWS;
            p.rows_per_split = rows;
            p.splits = (int)cdiv64(p.M, rows);
            p.slots = p.splits;
        }
        p.blk0 = blk;
        blk += p.splits * (p.N / 256) * p.tiles_k;
    }
    plan_xcd(g);
    return true;
}

size_t gemm_tn_bf16_256_slab_floats(const TnProb& p) { return (size_t)p.slots * (p.N / 256) * (p.K / 256) * TILE_F; }
size_t gemm_tn_bf16_256_colsum_floats(const TnProb& p) { return (size_t)p.splits * p.N; }

int launch_gemm_tn_bf16_256(const GemmTNG& g, hipStream_t s) {
    trace_tag(g.p[0].yconv ? "tn_bf16.256_grouped_conv3" : "tn_bf16.256_grouped");
    DCPT_CHECK_ARG(g.n >= 1 && g.n <= TNG_MAX, "gemm_tn_bf16_256: %d problems", g.n);
    int blocks = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < g.n; ++i) {
        const TnProb& p = g.p[i];
        DCPT_CHECK_ARG(p.X && p.Y && p.slab && p.M > 0 && gemm_tn_bf16_256_ok(p.N, p.K) && p.tiles_k == p.K / 256,
                       "gemm_tn_bf16_256: problem %d: null operand or N=%d / K=%d not multiples of 256", i, p.N, p.K);
        DCPT_CHECK_ARG(p.ldx % 8 == 0 && p.ldx >= p.N && (p.yconv || (p.ldy % 8 == 0 && p.ldy >= p.K)), "gemm_tn_bf16_256: row strides must be multiples of 8");
        DCPT_CHECK_ARG((p.yconv != 0) == (g.p[0].yconv != 0), "gemm_tn_bf16_256: gathered and plain Y operands cannot share a launch");
        if (p.yconv)
            DCPT_CHECK_ARG(p.gC % 128 == 0 && p.K == 9 * p.gC && p.seg_rows == 0 && p.colsum == nullptr && p.gW >= 1 && p.gH >= 1 &&
                               (double)(p.rows_per_split + 192 + 2 * p.gW) * p.gC * 2.0 < 1.0e9,
                           "gemm_tn_bf16_256: gathered Y needs K == 9 * gC, gC %% 128 == 0, no column sums, no image segments");
        DCPT_CHECK_ARG(p.splits >= 1 && p.rows_per_split >= 64 && p.rows_per_split % 64 == 0 && (int64_t)p.splits * p.rows_per_split >= p.M &&
                           p.blk0 == blocks && (p.seg_rows == 0 ? p.slots == p.splits
                                                                : (p.seg_rows % 64 == 0 && p.rows_per_split % p.seg_rows == 0 && p.M % p.seg_rows == 0 &&
                                                                   (int64_t)p.slots * p.seg_rows == p.M)),
                       "gemm_tn_bf16_256: bad split plan for problem %d", i);
        DCPT_CHECK_ARG((double)(p.rows_per_split + 192) * (double)(p.ldx > p.ldy ? p.ldx : p.ldy) * 2.0 < 1.0e9,
                       "gemm_tn_bf16_256: pixel range too large for 32-bit window offsets");
        blocks += p.splits * (p.N / 256) * p.tiles_k;
        flops += 2.0 * (double)p.M * p.N * p.K;
        bytes += ((double)p.M * p.N + (double)p.M * p.K) * 2.0 + (double)gemm_tn_bf16_256_slab_floats(p) * 4.0;
    }
    ProfScope prof(s, PROF_TN + 257, g.p[0].M, g.p[0].N, g.p[0].K, flops, bytes);
    const unsigned grid = g.xcd_slots > 0 ? 8u * (unsigned)g.xcd_slots : (unsigned)blocks;
    if (g.p[0].yconv) gemm_tn_bf16_256_kernel<true><<<dim3(grid), dim3(512), 0, s>>>(g);
    else gemm_tn_bf16_256_kernel<false><<<dim3(grid), dim3(512), 0, s>>>(g);
    DCPT_CHECK_LAUNCH("gemm_tn_bf16_256");
    return DCPT_OK;
}

int launch_wgrad_finish(FinJobs& j, hipStream_t s) {
    trace_tag("wgrad_finish");
    DCPT_CHECK_ARG(j.nslab >= 0 && j.nslab <= TNG_MAX && j.ncols >= 0 && j.ncols <= FIN_MAX_COLS, "wgrad_finish: bad job counts");
    int blk = 0;
    for (int i = 0; i < j.nslab; ++i) {
        const FinSlab& q = j.slab[i];
        DCPT_CHECK_ARG(q.slab && q.dW && q.N % 256 == 0 && q.K % 256 == 0 && q.tiles_k == q.K / 256 && q.splits >= 1,
                       "wgrad_finish: slab job %d: null pointer or bad shape", i);
        DCPT_CHECK_ARG(!(q.dgain || q.dbias) || q.colsum, "wgrad_finish: gain / bias gradients need column sums");
        DCPT_CHECK_ARG(!q.dgain || q.W, "wgrad_finish: the gain gradient needs the weights");
        DCPT_CHECK_ARG(!q.kscale || q.ks_div >= 1, "wgrad_finish: ks_div");
        DCPT_CHECK_ARG(!q.colsum || q.cs_rows >= 1, "wgrad_finish: cs_rows");
        j.slab_blk0[i] = blk;
        blk += q.N / 8;
    }
    j.slab_end = blk;
    for (int i = 0; i < j.ncols; ++i) {
        const FinCols& q = j.cols[i];
        DCPT_CHECK_ARG(q.part && q.R >= 1 && q.C >= 1 && ((q.mode == 0 && q.nj >= 1 && q.nj <= 2) || (q.mode == 1 && q.nj == 10 && q.out0 && q.out1)),
                       "wgrad_finish: column job %d", i);
        j.cols_blk0[i] = blk;
        blk += cdiv(q.C, 16) * q.nj;
    }
    j.cols_end = blk;
    if (j.sca.ds) {
        DCPT_CHECK_ARG(j.sca.pooled && j.sca.dW && j.sca.db && j.sca.B >= 1 && j.sca.C >= 1, "wgrad_finish: sca job");
        blk += (int)cdiv64((int64_t)j.sca.C * j.sca.C, FIN_THREADS) + cdiv(j.sca.C, FIN_THREADS);
    }
    if (blk == 0) return DCPT_OK;
    wgrad_finish_kernel<<<dim3((unsigned)blk), dim3(FIN_THREADS), 0, s>>>(j);
    DCPT_CHECK_LAUNCH("wgrad_finish");
    return DCPT_OK;
}
