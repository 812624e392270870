// TN GEMM on fp32 MFMA (weight gradients):  G[n][k] = sum_m X'(m,n) * Y'(m,k).
//
// Both operands are NHWC activations/gradients: the reduction index m (pixels) is the slow axis,
// so MFMA operand fragments are plain ds_read_b32 of consecutive floats (lane i reads column i of
// row mm + lane>>5): no transposition anywhere, and the LDS tiles are plain row-major [32][BN] images --
// which is what an LDS-DMA writes, so operands without a fused transform (dO always; plain / gathered /
// implicit-conv activations) go global -> LDS without touching VGPRs (buffer_load_dwordx4 ... lds).
// The pixel range is split into `splits` chunks (grid.y); every chunk writes its own [N][K] slab and a
// deterministic reduce kernel (misc.hip) sums the slabs -- no float atomics.  Blocks of the first k-tile
// column also emit the column sums of X' (bias gradients) from the LDS tile they already hold.
#include "gemm_operand.h"
#include "prof.h"

#ifdef DCPT_TIMELINE
// Diagnostic build only (tools/timeline_tn.py): per-block phase stamps (100 MHz wall clock) + shader cycles around the loop
__device__ unsigned long long g_tl_tn[1 << 15][8];
extern "C" int dcpt_timeline_read_tn(unsigned long long* host, int nblk) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl_tn), sizeof(unsigned long long) * 8 * (size_t)nblk);
}
#define TLT(i) if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < (1 << 15)) { g_tl_tn[blockIdx.x][i] = wall_clock64(); g_tl_tn[blockIdx.x][4 + i] = clock64(); }
#else
#define TLT(i)
#endif

namespace {

constexpr int BR = 32;  // reduction rows per LDS tile

template <int KIND>
constexpr bool is_dma() { return KIND == A_PLAIN || KIND == A_GATHER || KIND == A_CONV3; }

template <int BN, int BKo, int WN, int WK, int XK, int YK>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTN pin) {
    static_assert(WN * WK == 4, "4 waves");
    static_assert(is_dma<XK>(), "X (the output gradient) is never transformed");
    GemmTN p = pin;
    TLT(0)
    int lin, batch;
    xcd_remap_batched(lin, batch);
    if (gridDim.y > 1) {
        const int b1 = batch / p.nb2, b2 = batch % p.nb2;
        p.X += b1 * p.sX1 + b2 * p.sX2;
        p.Y += b1 * p.sY1 + b2 * p.sY2;
        p.slab += (int64_t)batch * p.splits * p.N * p.K;
    }
    constexpr int TN = BN / (WN * 32), TK = BKo / (WK * 32);
    // staging: pass i of the 256 threads covers tile floats [1024 i, 1024 i + 1024) of the row-major [BR][width] tile
    constexpr int X_IT = BR * BN / 1024, Y_IT = BR * BKo / 1024;
    constexpr bool Y_DMA = is_dma<YK>();
    static_assert(1024 % BKo == 0 || (Y_DMA && YK == A_PLAIN), "96-wide Y tiles: plain operand only");
    static_assert(1024 % BN == 0 || XK == A_PLAIN, "96-wide X tiles: plain operand only");
    __shared__ __attribute__((aligned(16))) float Xs[2][BR * BN];
    __shared__ __attribute__((aligned(16))) float Ys[2][BR * BKo];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    // 1-D grid, XCD-aware: the tiles of one pixel chunk (split) get consecutive logical ids, i.e. run
    // on one XCD at about the same time, so its X/Y row panels are fetched from HBM once and re-read
    // from that XCD's L2 by the other tiles.
    const int tilesK = (p.K + BKo - 1) / BKo;
    const int tilesN = (p.N + BN - 1) / BN;
    const int split = lin / (tilesN * tilesK);
    const int tile = lin % (tilesN * tilesK);
    const int tile_n = tile / tilesK, tile_k = tile % tilesK;
    const int n0 = tile_n * BN, k0 = tile_k * BKo;
    const int64_t mbeg = (int64_t)split * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;

    Operand ox, oy;
    ox.ptr = p.X; ox.M = mend; ox.ncols = p.N; ox.ld = p.ldx;
    ox.mu = nullptr; ox.rstd = nullptr; ox.lnw = nullptr; ox.lnb = nullptr; ox.simg = nullptr; ox.P = 1;
    ox.gH = p.gH; ox.gW = p.gW; ox.gC = p.gC;
    oy.ptr = p.Y; oy.M = mend; oy.ncols = p.K; oy.ld = p.ldy;
    oy.mu = p.mu; oy.rstd = p.rstd; oy.lnw = p.lnw; oy.lnb = p.lnb; oy.simg = p.simg; oy.P = p.P;
    oy.gH = p.gH; oy.gW = p.gW; oy.gC = p.gC;
    open_window<XK>(ox, mbeg < p.M ? mbeg : 0);
    open_window<YK>(oy, mbeg < p.M ? mbeg : 0);

    int xrow[X_IT], xc[X_IT], yrow[Y_IT], ycol[Y_IT];
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        xrow[i] = (1024 * i + 4 * tid) / BN;
        xc[i] = (1024 * i + 4 * tid) % BN;
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
        yrow[i] = (1024 * i + 4 * tid) / BKo;
        ycol[i] = (1024 * i + 4 * tid) % BKo;
    }
    const int yc = ycol[0];   // the same in every pass when 1024 % BKo == 0 (the transformed-Y loaders)
    RawVec ry[Y_IT];
    RowCtx rcy[Y_IT];
    // A_LN / A_LNBF on Y: this thread always handles the same 4 columns, so weight/bias are loaded once
    float4 kw = f4_zero(), kb = f4_zero();
    if constexpr (YK == A_LN || YK == A_LNBF) {
        if (k0 + yc < p.K) {
            kw = ldg4(p.lnw + k0 + yc);
            if constexpr (YK == A_LN) kb = ldg4(p.lnb + k0 + yc);
        }
    }
    // LDS-DMA: a wave's 64 lanes write 1 KiB of consecutive tile floats
    const uint32_t lds_x = lds_addr(&Xs[0][0]) + wave * 1024, lds_y = lds_addr(&Ys[0][0]) + wave * 1024;
    // Plain rows (the common case): the byte offset of this thread's quad inside the chunk window is fixed, the pixel tile
    // advances through the scalar offset of the DMA instruction, and only rows past the chunk end need the sentinel --
    // a few VALU per tile instead of 64-bit row arithmetic per piece.
    uint32_t xfix[X_IT], yfix[Y_IT];
#pragma unroll
    for (int i = 0; i < X_IT; ++i) xfix[i] = (n0 + xc[i] < p.N) ? ((uint32_t)xrow[i] * (uint32_t)p.ldx + (uint32_t)(n0 + xc[i])) * 4u : COL_SENT;
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) yfix[i] = (k0 + ycol[i] < p.K) ? ((uint32_t)yrow[i] * (uint32_t)p.ldy + (uint32_t)(k0 + ycol[i])) * 4u : COL_SENT;
    auto gload = [&](int64_t mt, int buf) {
        const int left = (int)(mend - mt);                    // rows of the chunk from this tile on (wave-uniform)
        const uint32_t step = (uint32_t)(mt - mbeg);          // tile's first row inside the chunk
        if constexpr (XK == A_PLAIN) {
#pragma unroll
            for (int i = 0; i < X_IT; ++i)
                dma16(ox.rsd, lds_x + (buf * BR * BN + i * 1024) * 4, (xrow[i] < left) ? xfix[i] : ROW_SENT, step * (uint32_t)p.ldx * 4u);
        } else {
#pragma unroll
            for (int i = 0; i < X_IT; ++i) {
                RowCtx rc;
                make_row<XK>(ox, mt + xrow[i], rc);
                dma16(ox.rsd, lds_x + (buf * BR * BN + i * 1024) * 4, elem_voff<XK>(ox, rc, n0 + xc[i]), 0);
            }
        }
        if constexpr (YK == A_PLAIN) {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i)
                dma16(oy.rsd, lds_y + (buf * BR * BKo + i * 1024) * 4, (yrow[i] < left) ? yfix[i] : ROW_SENT, step * (uint32_t)p.ldy * 4u);
        } else {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i) {
                make_row<YK>(oy, mt + yrow[i], rcy[i]);
                if constexpr (Y_DMA) {
                    dma16(oy.rsd, lds_y + (buf * BR * BKo + i * 1024) * 4, elem_voff<YK>(oy, rcy[i], k0 + yc), 0);
                } else {
                    load_raw<YK>(oy, rcy[i], k0 + yc, ry[i]);
                    if constexpr (YK == A_LN || YK == A_LNBF) {
                        ry[i].b = kw;
                        ry[i].c = kb;
                    }
                }
            }
        }
    };
    auto lstore = [&](int buf) {
        if constexpr (!Y_DMA) {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i)
                *reinterpret_cast<float4*>(&Ys[buf][1024 * i + 4 * tid]) = finish<YK>(rcy[i], ry[i]);
        }
    };

    floatx16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs = 0.f;
    // column sums of X' (bias gradients): the tilesK blocks that share an (n-tile, split) take turns,
    // one pixel tile each, so no block carries the whole extra cost; their partial rows are summed by
    // the slab reducer.  colsum layout: [split * tilesK + tile_k][N].
    const bool do_cs = (p.colsum != nullptr) && (tid < BN);

    const int64_t nmt = (mend - mbeg + BR - 1) / BR;
    if (nmt > 0) {
        gload(mbeg, 0);
        lstore(0);
    }
    dma_wait_all();
    __syncthreads();
    TLT(1)
    const int x_off = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
    const int y_off = (lane >> 5) * BKo + wk * TK * 32 + (lane & 31);
    for (int64_t t = 0; t < nmt; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < nmt) gload(mbeg + (t + 1) * BR, buf ^ 1);
        const float* xs = &Xs[buf][x_off];
        const float* ys = &Ys[buf][y_off];
        // software-pipelined fragments: the ds_reads of step s+2 are issued before the MFMAs of step s (a ring of three
        // fragment sets), so an LDS round trip has two MFMA groups (512 cycles) to complete
        float a[3][TN], b[3][TK];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
#pragma unroll
            for (int i = 0; i < TN; ++i) a[pf][i] = xs[2 * pf * BN + i * 32];
#pragma unroll
            for (int j = 0; j < TK; ++j) b[pf][j] = ys[2 * pf * BKo + j * 32];
        }
#pragma unroll
        for (int st = 0; st < BR / 2; ++st) {
            const int cur = st % 3, nxt = (st + 2) % 3;
            if (st + 2 < BR / 2) {
#pragma unroll
                for (int i = 0; i < TN; ++i) a[nxt][i] = xs[(2 * st + 4) * BN + i * 32];
#pragma unroll
                for (int j = 0; j < TK; ++j) b[nxt][j] = ys[(2 * st + 4) * BKo + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the later steps' ds_reads ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_cs && (int)(t % tilesK) == tile_k) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < BR; ++r) s += Xs[buf][r * BN + tid];
            cs += s;
        }
        if (t + 1 < nmt) lstore(buf ^ 1);
        dma_wait_all();   // this wave's LDS-DMA pieces of the next tile have landed; the barrier publishes them
        __syncthreads();
    }

    TLT(2)
    // slab stores through a buffer window at the tile's first element; rows past N / columns past K are dropped
    const rsrc_t rsS = make_rsrc(p.slab + (int64_t)split * p.N * p.K + (int64_t)n0 * p.K + k0);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int kl = (wk * TK + j) * 32 + (lane & 31);
            const bool kok = k0 + kl < p.K;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (wn * TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const uint32_t o = (kok && n0 + nl < p.N) ? ((uint32_t)nl * (uint32_t)p.K + (uint32_t)kl) * 4u : ROW_SENT;
                buf_st1(rsS, o, acc[i][j][r]);
            }
        }
    if (do_cs && n0 + tid < p.N) p.colsum[((int64_t)split * tilesK + tile_k) * p.N + n0 + tid] = cs;
    TLT(3)
}

}  // namespace

// Output-tile shape for an N x K gradient: 64 for narrow sides, otherwise whichever of 128 x 128, 128 x 96, 96 x 128 pads the
// least (Restormer's 96 / 192 / 288 / 576-wide layers); the 96-wide tiles take plain operands and no column sums.
static int tn_narrow() {   // experiment knob: 128 x 64 output tiles everywhere (three resident blocks per CU); measured +-0 on the training step
    static const int v = dcpt_tuning("DCPT_TN_NARROW", 0);
    return v;
}

static void tn_tile_shape(int N, int K, bool plain, int* bn, int* bk) {
    *bn = (N <= 64) ? 64 : 128;
    *bk = (K <= 64 || tn_narrow()) ? 64 : 128;
    if (tn_narrow()) return;
    static const int use96 = dcpt_tuning("DCPT_TN_96", 1);
    if (N <= 64 || K <= 64 || !plain || !use96) return;
    const int64_t a128 = (int64_t)cdiv(N, 128) * 128 * cdiv(K, 128) * 128;
    const int64_t a_k96 = (int64_t)cdiv(N, 128) * 128 * cdiv(K, 96) * 96;
    const int64_t a_n96 = (int64_t)cdiv(N, 96) * 96 * cdiv(K, 128) * 128;
    if (a_k96 < a128 && a_k96 <= a_n96) *bk = 96;
    else if (a_n96 < a128) *bn = 96;
}

namespace {

template <int XK, int YK>
int launch_cfg(const GemmTN& p, hipStream_t s) {
    const unsigned nbatch = (unsigned)(p.nb1 * p.nb2);
    constexpr bool PLAIN = (XK == A_PLAIN && YK == A_PLAIN);
    int bn, bk;
    tn_tile_shape(p.N, p.K, PLAIN && p.colsum == nullptr, &bn, &bk);   // colsum layouts assume gemm_tn_tiles_k()
    if constexpr (PLAIN) {
        if (bk == 96) {
            const int tiles = cdiv(p.N, 128) * cdiv(p.K, 96);
            gemm_tn_kernel<128, 96, 4, 1, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
            DCPT_CHECK_LAUNCH("gemm_tn");
            return DCPT_OK;
        }
        if (bn == 96) {
            const int tiles = cdiv(p.N, 96) * cdiv(p.K, 128);
            gemm_tn_kernel<96, 128, 1, 4, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
            DCPT_CHECK_LAUNCH("gemm_tn");
            return DCPT_OK;
        }
    }
    if (bn == 64 && bk == 64) {
        const int tiles = cdiv(p.N, 64) * cdiv(p.K, 64);
        gemm_tn_kernel<64, 64, 2, 2, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
    } else if (bk == 64) {
        const int tiles = cdiv(p.N, 128) * cdiv(p.K, 64);
        gemm_tn_kernel<128, 64, 4, 1, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
    } else if (bn == 64) {
        const int tiles = cdiv(p.N, 64) * cdiv(p.K, 128);
        gemm_tn_kernel<64, 128, 1, 4, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
    } else {
        const int tiles = cdiv(p.N, 128) * cdiv(p.K, 128);
        gemm_tn_kernel<128, 128, 2, 2, XK, YK><<<dim3(tiles * p.splits, nbatch), dim3(256), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_tn");
    return DCPT_OK;
}

}  // namespace

int gemm_tn_tiles_k(int N, int K) {   // launches with column sums never use the 96-wide tiles
    (void)N;
    const int bk = (K <= 64 || tn_narrow()) ? 64 : 128;
    return cdiv(K, bk);
}

void gemm_tn_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split) {
    if (gemm_tn_x3_plan(M, N, K, splits, rows_per_split)) return;   // (opt-in bf16x3 mode: one 256 x 256-tile block per CU)
    int bn, bk;
    tn_tile_shape(N, K, true, &bn, &bk);   // (a transformed-operand launch of the same shape uses at most as many tiles)
    const int64_t tiles = (int64_t)cdiv(N, bn) * cdiv(K, bk);
    // 2 blocks are co-resident per CU (LDS), 256 CUs = 512 block slots.  Cost model in units of one reduction row of a block:
    // rounds(s) * (rows per split + ~96 rows of prologue/epilogue) for the GEMM, plus the slab round trip (written here, read
    // by the reducer) at ~2e-5 row units per slab element.  The common shapes land on one full round (tiles * s = 512); the
    // model also handles shapes with more tiles than slots (a dense 3x3 at 1024 channels has 576: one split would run
    // 1.125 rounds, 8 splits run 9.0) and keeps the split count low where the slab traffic would dominate.
    const int64_t max_by_rows = cdiv64(M, 256);  // at least 256 rows per split
    const int64_t slots = (bn == 64 || bk == 64) ? 768 : 512;
    int64_t smax = 4 * slots / tiles;
    if (smax < 16) smax = 16;
    if (smax > max_by_rows) smax = max_by_rows;
    if (smax > 65535) smax = 65535;
    if (smax < 1) smax = 1;
    static const int plan_model = dcpt_tuning("DCPT_TN_PLAN", 1);
    int64_t want = 1;
    if (plan_model) {
        double best = 1e300;
        for (int64_t sp = 1; sp <= smax; ++sp) {
            const int64_t r = cdiv64(cdiv64(M, sp), 32) * 32;
            const int64_t nsp = cdiv64(M, r);
            const double rounds = (double)cdiv64(tiles * nsp, slots);
            const double cost = rounds * (double)(r + 96) + 2.0e-5 * (double)nsp * N * K;
            if (cost < best * 0.999) {
                best = cost;
                want = sp;
            }
        }
    } else {
        want = slots / tiles;
        if (want > max_by_rows) want = max_by_rows;
        if (want < 1) want = 1;
        if (want > 65535) want = 65535;
    }
    int64_t rps = cdiv64(cdiv64(M, want), 32) * 32;
    *rows_per_split = rps;
    *splits = (int)cdiv64(M, rps);
}

// Plan whose splits never straddle an image of P pixels (P % rows_per_split == 0); false if there is none close to the
// occupancy target (the caller then keeps the scaled operand loader).
bool gemm_tn_plan_images(int64_t M, int N, int K, int P, int* splits, int64_t* rows_per_split) {
    int sp;
    int64_t rps;
    gemm_tn_plan(M, N, K, &sp, &rps);
    if (P % 32 != 0 || M % P != 0) return false;
    int64_t best = 0;
    for (int64_t d = 32; d <= P; d += 32)
        if (P % d == 0 && d <= rps + rps / 2) best = d;   // largest image-aligned chunk not much above the target
    if (best == 0 || best * 2 < rps) return false;         // would need far more (smaller) splits than the target
    *rows_per_split = best;
    *splits = (int)(M / best);
    return *splits <= 65535;
}

int launch_gemm_tn(const GemmTN& pin, int xload, int yload, hipStream_t s) {
    GemmTN p = pin;
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;
    DCPT_CHECK_ARG(p.nb1 * p.nb2 == 1 || p.colsum == nullptr, "gemm_tn: column sums are not supported for batched problems");
    DCPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_tn: empty problem");
    DCPT_CHECK_ARG(p.N % 4 == 0 && p.K % 4 == 0, "gemm_tn: N=%d, K=%d must be multiples of 4", p.N, p.K);
    DCPT_CHECK_ARG(p.splits >= 1 && p.splits <= 65535 && p.rows_per_split % 32 == 0, "gemm_tn: bad split plan");
    // 32-bit offsets inside a block's buffer windows (gemm_operand.h): one pixel chunk of either operand, one slab tile
    DCPT_CHECK_ARG((double)(p.rows_per_split + 64) * (double)(p.ldx > p.ldy ? p.ldx : p.ldy) * 4.0 < 1.0e9 && (double)p.N * p.K < 2.0e8,
                   "gemm_tn: pixel chunk or slab too large for 32-bit window offsets");
    const double bytes = (double)p.M * p.N + (double)p.M * p.K * (yload == A_SG ? 2 : 1) + (double)p.splits * p.N * p.K;
    ProfScope prof(s, PROF_TN + xload * 8 + yload, p.M, p.N, p.K, 2.0 * (double)p.M * p.N * p.K * p.nb1 * p.nb2,
                   bytes * 4.0 * p.nb1 * p.nb2);
    if (gemm_tn_x3_ok(p, xload, yload)) return launch_gemm_tn_x3(p, gemm_tn_tiles_k(p.N, p.K), s);   // (opt-in mode; off by default)
#define CASE(XK, YK) \
    if (xload == XK && yload == YK) return launch_cfg<XK, YK>(p, s);
    CASE(A_PLAIN, A_PLAIN)
    CASE(A_PLAIN, A_LN)
    CASE(A_PLAIN, A_SCALE)
    CASE(A_PLAIN, A_SG)
    CASE(A_PLAIN, A_GATHER)
    CASE(A_GATHER, A_PLAIN)
    CASE(A_PLAIN, A_CONV3)
    CASE(A_PLAIN, A_LNBF)
#undef CASE
    dcpt_set_error("gemm_tn: unsupported loader combination %d/%d", xload, yload);
    return DCPT_ERR_ARG;
}
