// bf16-storage MFMA GEMMs (fp32 accumulate on v_mfma_f32_32x32x16_bf16) for the bf16 path of the NAFBlock.
//
// NT:  C[m][n] = sum_k A[m][k] * Bw[n][k].  Same skeleton as the fp32 kernel (gemm_nt.hip): 128 x BN x (128 bytes of k) tiles,
// 4 waves, LDS tiles of [rows][128 B] whose eight 16-byte chunks are XOR-swizzled (chunk q of row r at slot q ^ ((r >> 1) & 7)):
// one ds_read_b128 per (row, 16-k group) is exactly one MFMA operand (8 bf16 per lane, k = 16 j + 8 (lane >> 5) + 0..7) and is
// conflict-free, and the image is what an LDS-DMA writes, so BOTH operands go HBM -> LDS without touching VGPRs.  Every operand of
// this path is plain (the SCA scale of conv3 is folded into per-image weights by the caller), so there is no register-staged
// loader at all.  A k-tile is 64 elements: 4 MFMAs of 32 cycles per 32x32 tile against 16 MFMAs of 64 cycles in fp32 for the same
// LDS bytes, i.e. the kernel is bound by LDS reads / HBM rather than by the matrix pipe -- as is the whole bf16 block.
// Epilogues run on the fp32 accumulators parked in LDS; a thread owns 8 consecutive columns (16-byte bf16 accesses).
//
// TN:  G[n][k] = sum_m X[m][n] * Y[m][k]  (weight gradients, fp32 slabs).  The contraction index m is the slow axis of both
// operands, and a bf16 MFMA operand needs 8 CONSECUTIVE m per lane: the row-major [64 m][width] LDS tiles (again plain DMA images)
// are read with ds_read_b64_tr_b16, the gfx950 transposing LDS read (a 16-lane group reads a [4 m][16 col] block, lane t supplies
// the address of row t / 4, columns 4 (t % 4).., and receives column t -- verified by tools/ubench/tr_probe.hip), two reads per
// operand.  16-byte chunk c of tile row r sits at chunk position c ^ swz(r) (swz = 4 (r & 3) for 256-byte rows, 4 ((r >> 1) & 1)
// for 128-byte rows): the 32 lanes of a half-wave then hit 32 distinct 8-byte slots of a bank row.
#include "bf16.h"
#include "gemm_bf16_epi.h"
#include "prof.h"

namespace {

constexpr int KT = 64;          // k elements per NT tile (128 bytes per row)

template <int BM, int BN, int WM, int WN, int EK, int AM = 0>   // AM: 0 plain A, 1 implicit 3x3 (conv3), 2 gathered 2x2 cells (gather2)
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const GemmNTB pin) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr bool CONV = AM == 1, GATH = AM == 2;
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
    constexpr bool GATE = (EK == EB_BIASGATE);
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
#ifndef NTB_STAGES
#define NTB_STAGES 2
#endif
    constexpr int ST = NTB_STAGES;   // k-tiles resident in LDS: the one in use + ST - 1 in flight
    constexpr int SM_BYTES = ST * (A_BYTES + B_BYTES);
    static_assert(SM_BYTES >= BM * BN * 4, "epilogue staging does not fit");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    unsigned char* const As0 = smem;                  // [ST][A_BYTES]
    unsigned char* const Bs0 = smem + ST * A_BYTES;   // [ST][B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int Ch = p.N / 2;
    const int tilesN = GATE ? (Ch + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    const int64_t m0 = (int64_t)(lin / tilesN) * BM;
    const int n0 = (lin % tilesN) * (GATE ? BN / 2 : BN);

    // conv3: the window starts one image row + one pixel before the tile's first pixel (clipped at the tensor start)
    int64_t apix0 = (m0 < p.M ? m0 : 0);
    if constexpr (CONV) {
        apix0 -= p.gW + 1;
        if (apix0 < 0) apix0 = 0;
    }
    if constexpr (GATH) apix0 = fine_elem(p, apix0);   // (elements, not pixels)
    const i32x4 rsA = make_rsrc_dma(p.A + (GATH ? apix0 : apix0 * (int64_t)(CONV ? p.gC : p.lda)));
    const i32x4 rsB = make_rsrc_dma(p.Bw + (GATE ? 0 : (int64_t)n0 * p.K));
    // staging map: thread -> row (tid >> 3) + 32 * pass, LDS slot tid & 7 = logical 16-byte chunk slot ^ ((row >> 1) & 7)
    const int lrow = tid >> 3;
    const int lk = 8 * ((tid & 7) ^ ((tid >> 4) & 7));   // first k element of this thread's chunk inside a k-tile
    uint32_t aoff[A_IT], boff[B_IT];
    int ah[A_IT], aw[A_IT];   // conv3: pixel coordinates of this thread's rows
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = lrow + 32 * i;
        if constexpr (CONV) {
            const int64_t m = m0 + r;
            const bool ok = m < p.M;
            const int64_t mm = ok ? m : 0;
            aw[i] = (int)(mm % p.gW);
            ah[i] = ok ? (int)((mm / p.gW) % p.gH) : -100000;   // invalid row: every tap out of bounds
            aoff[i] = (uint32_t)((mm - apix0) * p.gC) * 2u;
        } else if constexpr (GATH) {
            aoff[i] = (m0 + r < p.M) ? (uint32_t)((fine_elem(p, m0 + r) - apix0) * 2) : ROW_SENT;
        } else {
            aoff[i] = (m0 + r < p.M) ? ((uint32_t)r * (uint32_t)p.lda + (uint32_t)lk) * 2u : ROW_SENT;
        }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int nl = lrow + 32 * i;
        if constexpr (GATE) {
            const int hl = nl % (BN / 2);
            const int n = (nl < BN / 2 ? 0 : Ch) + n0 + hl;
            boff[i] = (n0 + hl < Ch) ? ((uint32_t)n * (uint32_t)p.K + (uint32_t)lk) * 2u : ROW_SENT;
        } else {
            boff[i] = (n0 + nl < p.N) ? ((uint32_t)nl * (uint32_t)p.K + (uint32_t)lk) * 2u : ROW_SENT;
        }
    }
    const uint32_t lds_a = lds_addr(reinterpret_cast<const float*>(As0)) + wave * 1024;
    const uint32_t lds_b = lds_addr(reinterpret_cast<const float*>(Bs0)) + wave * 1024;
    auto gload = [&](int kt, int buf) {
        const uint32_t ksent = (kt * KT + lk < p.K) ? 0u : COL_SENT;   // ragged K: only the last k-tile (K % 8 == 0)
#pragma unroll
        for (int i = 0; i < B_IT; ++i) dma16(rsB, lds_b + buf * B_BYTES + i * 4096, boff[i] + ksent, (uint32_t)kt * 128u);
        if constexpr (CONV) {
            const int k = kt * KT + lk;
            const int tap = k / p.gC, ch = k - tap * p.gC;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int shift = ((ky - 1) * p.gW + (kx - 1)) * p.gC + ch;   // elements, relative to the row's own pixel
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int hh = ah[i] + ky - 1, ww = aw[i] + kx - 1;
                const bool ok = k < p.K && hh >= 0 && hh < p.gH && ww >= 0 && ww < p.gW;
                dma16(rsA, lds_a + buf * A_BYTES + i * 4096, ok ? aoff[i] + (uint32_t)(shift * 2) : COL_SENT, 0);
            }
        } else if constexpr (GATH) {
            const int k = kt * KT + lk;
            const int ij = k / p.gC, ch = k - ij * p.gC;
            const uint32_t shift = (k < p.K) ? (uint32_t)((((ij >> 1) * (2 * p.gW) + (ij & 1)) * p.gC + ch) * 2) : COL_SENT;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) dma16(rsA, lds_a + buf * A_BYTES + i * 4096, aoff[i] + shift, 0);
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) dma16(rsA, lds_a + buf * A_BYTES + i * 4096, aoff[i] + ksent, (uint32_t)kt * 128u);
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (p.K + KT - 1) / KT;
    // prologue: ST - 1 k-tiles in flight (tiles past the end are issued too -- their offsets are out of range, they land as zeros --
    // so that the counted waits below see a fixed number of DMAs per k-tile)
#pragma unroll
    for (int q = 0; q < ST - 1; ++q) gload(q, q);
    if constexpr (ST == 2) dma_wait_all();
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (A_IT + B_IT)) : "memory");
    __syncthreads();
    const int fi = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    const int a_row = (wm * TM * 32 + (lane & 31)) * 128;
    const int b_row = (wn * TN * 32 + (lane & 31)) * 128;
    int slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = ((2 * j + fh) ^ fi) * 16;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt % ST;
        if constexpr (ST == 2) {
            if (kt + 1 < nkt) gload(kt + 1, buf ^ 1);
        } else {
            gload(kt + ST - 1, (kt + ST - 1) % ST);   // refills the buffer read in iteration kt - 1 (published free by the last barrier)
        }
        const unsigned char* as = As0 + buf * A_BYTES + a_row;
        const unsigned char* bs = Bs0 + buf * B_BYTES + b_row;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(as + i * 4096 + slot[j]);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) bf[jn] = *reinterpret_cast<const bf16x8*>(bs + jn * 4096 + slot[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[jn], acc[i][jn], 0, 0, 0);
        }
        if constexpr (ST == 2) dma_wait_all();
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * (A_IT + B_IT)) : "memory");   // k-tile kt + 1 has landed
        __syncthreads();
    }
    if constexpr (ST > 2) {
        dma_wait_all();   // the zero tiles past the end, before the staging area is reused
        __syncthreads();
    }

    float* const Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = (wm * TM + i) * 32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = rb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[ml * BN + nl] = acc[i][j][r];
            }
        }
    }
    __syncthreads();
    epilogue8<EK, BM, BN>(p, Cs, m0, n0, tid);
}

template <int EK>
int launch_nt(const GemmNTB& p, hipStream_t s) {
    constexpr bool GATE = (EK == EB_BIASGATE);
    const unsigned nb = (unsigned)(p.nb > 0 ? p.nb : 1);
    const int ncols = GATE ? p.N / 2 : p.N;
    if constexpr (EK == EB_PLAIN || EK == EB_LNFWD || EK == EB_LNBWDM) {
        if (p.conv3) {
            if (ncols <= 64) gemm_nt_bf16_kernel<128, 64, 4, 1, EK, 1><<<dim3((unsigned)(cdiv64(p.M, 128) * cdiv(ncols, 64)), nb), dim3(256), 0, s>>>(p);
            else gemm_nt_bf16_kernel<128, 128, 2, 2, EK, 1><<<dim3((unsigned)(cdiv64(p.M, 128) * cdiv(ncols, 128)), nb), dim3(256), 0, s>>>(p);
            DCPT_CHECK_LAUNCH("gemm_nt_bf16 conv3");
            return DCPT_OK;
        }
    }
    if constexpr (EK == EB_PLAIN || EK == EB_BIAS) {
        if (p.gather2) {
            if (ncols <= 64) gemm_nt_bf16_kernel<128, 64, 4, 1, EK, 2><<<dim3((unsigned)(cdiv64(p.M, 128) * cdiv(ncols, 64)), nb), dim3(256), 0, s>>>(p);
            else gemm_nt_bf16_kernel<128, 128, 2, 2, EK, 2><<<dim3((unsigned)(cdiv64(p.M, 128) * cdiv(ncols, 128)), nb), dim3(256), 0, s>>>(p);
            DCPT_CHECK_LAUNCH("gemm_nt_bf16 gather2");
            return DCPT_OK;
        }
    }
    if (ncols <= (GATE ? 32 : 64)) {
        const int64_t tiles = cdiv64(p.M, 128) * cdiv(ncols, GATE ? 32 : 64);
        gemm_nt_bf16_kernel<128, 64, 4, 1, EK><<<dim3((unsigned)tiles, nb), dim3(256), 0, s>>>(p);
    } else {
        const int64_t tiles = cdiv64(p.M, 128) * cdiv(ncols, GATE ? 64 : 128);
        gemm_nt_bf16_kernel<128, 128, 2, 2, EK><<<dim3((unsigned)tiles, nb), dim3(256), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_nt_bf16");
    return DCPT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
constexpr int BRB = 64;   // reduction rows (pixels) per TN tile

template <int W>
__device__ __forceinline__ int tn_swz(int r) { return W == 128 ? 4 * (r & 3) : 4 * ((r >> 1) & 1); }

// 8 consecutive m of column `col` of a row-major [BRB][W] bf16 tile (m = mrow .. mrow + 7), as one MFMA operand.
// lane: t = lane & 15 supplies the address of row mrow + (t >> 2) (+4 for the second read), columns col16 + 4 (t & 3)..+3 of the
// 16-column group col16 that contains `col` (col = col16 + t).
template <int W>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int mrow, int col16, int t) {
    const int r0 = mrow + (t >> 2);
    const int c = col16 + 4 * (t & 3);             // element column of this lane's 8-byte piece
    const int chunk = c >> 3, within = (c & 7) * 2;
    const unsigned char* a0 = tile + r0 * (W * 2) + ((chunk ^ tn_swz<W>(r0)) << 4) + within;
    const int r1 = r0 + 4;
    const unsigned char* a1 = tile + r1 * (W * 2) + ((chunk ^ tn_swz<W>(r1)) << 4) + within;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(const_cast<unsigned char*>(a0)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(const_cast<unsigned char*>(a1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BN, int BKo, int WN, int WK, int YM = 0, bool XG = false>   // YM: 0 plain Y, 1 implicit 3x3 (yconv), 2 gathered 2x2 cells (yg2); XG: xg2
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const GemmTNB p) {
    static_assert(WN * WK == 4, "4 waves");
    constexpr bool YCONV = YM == 1, YG = YM == 2;
    constexpr int TN = BN / (WN * 32), TK = BKo / (WK * 32);
    constexpr int XB = BRB * BN * 2, YB = BRB * BKo * 2;     // bytes per tile
    constexpr int X_IT = XB / 4096, Y_IT = YB / 4096;        // 256 threads x 16 bytes per pass
    constexpr int XCH = BN / 8, YCH = BKo / 8;               // 16-byte chunks per row
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][XB];
    __shared__ __attribute__((aligned(16))) unsigned char Ys[2][YB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int tilesK = (p.K + BKo - 1) / BKo, tilesN = (p.N + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / (tilesN * tilesK);
    const int tile = lin % (tilesN * tilesK);
    const int tile_n = tile / tilesK, tile_k = tile % tilesK;
    const int n0 = tile_n * BN, k0 = tile_k * BKo;
    const int64_t mbeg = (int64_t)split * p.rows_per_split;
    int64_t mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const int64_t mw = mbeg < p.M ? mbeg : 0;
    const int64_t xbase = XG ? fine_elem(p, mw) : mw * (int64_t)p.ldx;   // elements
    const i32x4 rsX = make_rsrc_dma(p.X + xbase);
    int64_t ypix0 = mw;   // gathered Y: the window starts one image row + one pixel before the chunk (clipped at the tensor start)
    if constexpr (YCONV) {
        ypix0 -= p.gW + 1;
        if (ypix0 < 0) ypix0 = 0;
    }
    const int64_t ybase = YG ? fine_elem(p, mw) : ypix0 * (int64_t)(YCONV ? p.gC : p.ldy);
    const i32x4 rsY = make_rsrc_dma(p.Y + ybase);

    // staging: pass i, thread tid writes tile bytes [4096 i + 16 tid, +16): row = (256 i + tid) / chunks-per-row, chunk POSITION
    // cp = (256 i + tid) % chunks-per-row, which holds source chunk cp ^ swz(row)
    int xrow[X_IT], yrow[Y_IT];
    uint32_t xfix[X_IT], yfix[Y_IT];
    int yky[Y_IT], ykx[Y_IT];   // gathered Y: the tap of this thread's chunk in pass i (fixed for the whole launch)
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int e = 256 * i + tid;
        xrow[i] = e / XCH;
        const int c = (e % XCH) ^ tn_swz<BN>(xrow[i]);
        if constexpr (XG) {   // element offset of the chunk's (i, j) cell and channel relative to the row's fine pixel
            const int col = n0 + 8 * c, ij = col / p.gC, ch = col - ij * p.gC;
            xfix[i] = (col < p.N) ? (uint32_t)(((ij >> 1) * (2 * p.gW) + (ij & 1)) * p.gC + ch) : COL_SENT;
        } else {
            xfix[i] = (n0 + 8 * c < p.N) ? ((uint32_t)xrow[i] * (uint32_t)p.ldx + (uint32_t)(n0 + 8 * c)) * 2u : COL_SENT;
        }
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
        const int e = 256 * i + tid;
        yrow[i] = e / YCH;
        const int c = (e % YCH) ^ tn_swz<BKo>(yrow[i]);
        if constexpr (YCONV) {
            const int k = k0 + 8 * c;
            const int tap = k / p.gC, ch = k - tap * p.gC;
            yky[i] = tap / 3;
            ykx[i] = tap - 3 * yky[i];
            // offset of (tap, ch) relative to the row's own pixel, in elements (may be negative: added to the pixel offset below)
            yfix[i] = (k < p.K) ? (uint32_t)(((yky[i] - 1) * p.gW + (ykx[i] - 1)) * p.gC + ch) : COL_SENT;
        } else if constexpr (YG) {
            const int col = k0 + 8 * c, ij = col / p.gC, ch = col - ij * p.gC;
            yfix[i] = (col < p.K) ? (uint32_t)(((ij >> 1) * (2 * p.gW) + (ij & 1)) * p.gC + ch) : COL_SENT;
        } else {
            yfix[i] = (k0 + 8 * c < p.K) ? ((uint32_t)yrow[i] * (uint32_t)p.ldy + (uint32_t)(k0 + 8 * c)) * 2u : COL_SENT;
        }
    }
    const uint32_t lds_x = lds_addr(reinterpret_cast<const float*>(&Xs[0][0])) + wave * 1024;
    const uint32_t lds_y = lds_addr(reinterpret_cast<const float*>(&Ys[0][0])) + wave * 1024;
    auto gload = [&](int64_t mt, int buf) {
        const int left = (int)(mend - mt);
        const uint32_t step = (uint32_t)(mt - mbeg);
        if constexpr (XG) {
#pragma unroll
            for (int i = 0; i < X_IT; ++i) {
                const bool ok = xrow[i] < left && xfix[i] != COL_SENT;
                dma16(rsX, lds_x + buf * XB + i * 4096, ok ? (uint32_t)((fine_elem(p, mt + xrow[i]) - xbase + xfix[i]) * 2) : ROW_SENT, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < X_IT; ++i)
                dma16(rsX, lds_x + buf * XB + i * 4096, (xrow[i] < left) ? xfix[i] : ROW_SENT, step * (uint32_t)p.ldx * 2u);
        }
        if constexpr (YG) {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i) {
                const bool ok = yrow[i] < left && yfix[i] != COL_SENT;
                dma16(rsY, lds_y + buf * YB + i * 4096, ok ? (uint32_t)((fine_elem(p, mt + yrow[i]) - ybase + yfix[i]) * 2) : ROW_SENT, 0);
            }
        } else if constexpr (YCONV) {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i) {
                const int64_t m = mt + yrow[i];
                const int w = (int)(m % p.gW), h = (int)((m / p.gW) % p.gH);
                const int hh = h + yky[i] - 1, ww = w + ykx[i] - 1;
                const bool ok = yrow[i] < left && yfix[i] != COL_SENT && hh >= 0 && hh < p.gH && ww >= 0 && ww < p.gW;
                dma16(rsY, lds_y + buf * YB + i * 4096, ok ? (uint32_t)(((int64_t)(m - ypix0) * p.gC + (int)yfix[i]) * 2) : ROW_SENT, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < Y_IT; ++i)
                dma16(rsY, lds_y + buf * YB + i * 4096, (yrow[i] < left) ? yfix[i] : ROW_SENT, step * (uint32_t)p.ldy * 2u);
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
    const bool do_cs = (p.colsum != nullptr) && (tid < BN);

    const int64_t nmt = (mend - mbeg + BRB - 1) / BRB;
    if (nmt > 0) gload(mbeg, 0);
    dma_wait_all();
    __syncthreads();
    const int t16 = lane & 15, g16 = (lane >> 4) & 1, fh = lane >> 5;
    for (int64_t t = 0; t < nmt; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < nmt) gload(mbeg + (t + 1) * BRB, buf ^ 1);
        const unsigned char* xs = &Xs[buf][0];
        const unsigned char* ys = &Ys[buf][0];
#pragma unroll
        for (int st = 0; st < BRB / 16; ++st) {
            bf16x8 a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = tr_frag<BN>(xs, 16 * st + 8 * fh, (wn * TN + i) * 32 + 16 * g16, t16);
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = tr_frag<BKo>(ys, 16 * st + 8 * fh, (wk * TK + j) * 32 + 16 * g16, t16);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_cs && (int)(t % tilesK) == tile_k) {
            float s = 0.f;
            const int ch = tid >> 3, wi = (tid & 7) * 2;
#pragma unroll 8
            for (int r = 0; r < BRB; ++r) {
                const uint16_t h = *reinterpret_cast<const uint16_t*>(&xs[r * (BN * 2) + ((ch ^ tn_swz<BN>(r)) << 4) + wi]);
                s += __builtin_bit_cast(float, (uint32_t)h << 16);
            }
            cs += s;
        }
        dma_wait_all();
        __syncthreads();
    }

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
}

void tn_shape(int N, int K, int* bn, int* bk) {
    *bn = (N <= 64) ? 64 : 128;
    *bk = (K <= 64) ? 64 : 128;
}

}  // namespace

int launch_gemm_nt_bf16(const GemmNTB& pin, int epi, hipStream_t s) {
    GemmNTB p = pin;
    if (p.nb < 1) p.nb = 1;
    DCPT_CHECK_ARG(p.A && p.Bw && (p.C || epi == EB_BIASGATE) && p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt_bf16: null operand or empty problem");
    DCPT_CHECK_ARG(p.K % 8 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldc % 8 == 0 && p.ldres % 8 == 0,
                   "gemm_nt_bf16: K=%d, N=%d and the row strides must be multiples of 8 (16-byte rows)", p.K, p.N);
    if (p.conv3)
        DCPT_CHECK_ARG((epi == EB_PLAIN || epi == EB_LNFWD || epi == EB_LNBWDM) && p.gC % 8 == 0 && p.K == 9 * p.gC && p.nb == 1 &&
                           (double)(130 + 2 * p.gW + 2) * p.gC * 2.0 < 1.0e9,
                       "gemm_nt_bf16: conv3 needs the plain or a LayerNorm epilogue, K == 9 * gC, gC %% 8 == 0");
    if (p.gather2)
        DCPT_CHECK_ARG((epi == EB_PLAIN || epi == EB_BIAS) && !p.conv3 && p.gC % 8 == 0 && p.K == 4 * p.gC && p.nb == 1 &&
                           (double)(130 + 2 * p.gW) * 4.0 * p.gC * 2.0 < 1.0e9,
                       "gemm_nt_bf16: gather2 needs the plain / bias epilogue, K == 4 * gC, gC %% 8 == 0");
    if (epi == EB_SCATTER || epi == EB_SCATTER_ADD)
        DCPT_CHECK_ARG(p.gC % 8 == 0 && p.N == 4 * p.gC && p.nb == 1 && !p.conv3 && !p.gather2 && (epi == EB_SCATTER || p.res) &&
                           (double)(130 + 2 * p.gW) * 4.0 * p.gC * 2.0 < 1.0e9,
                       "gemm_nt_bf16: scatter epilogue needs N == 4 * gC, gC %% 8 == 0 (and res for the adding form)");
    DCPT_CHECK_ARG(p.K < (1 << 20) && p.N < (1 << 20) && p.lda < (1 << 20) && p.ldc < (1 << 20) && (double)p.N * p.K * 2.0 < 1.0e9,
                   "gemm_nt_bf16: K/N/row strides out of the 32-bit window range");
    DCPT_CHECK_ARG(cdiv64(p.M, 128) * cdiv(p.N, 32) < (1ll << 31), "gemm_nt_bf16: grid too large");
    if (epi == EB_BIASGATE) DCPT_CHECK_ARG(p.gate && p.N % 16 == 0, "gemm_nt_bf16: gate epilogue needs gate != null, N %% 16 == 0");
    if (epi == EB_DOTCOL) DCPT_CHECK_ARG(p.colpart && p.res && p.nb == 1, "gemm_nt_bf16: column-dot epilogue needs colpart and res");
    if (epi == EB_RESID) DCPT_CHECK_ARG(p.res, "gemm_nt_bf16: residual epilogue needs res");
    if (epi == EB_SGBWD) DCPT_CHECK_ARG(p.aux && p.ldc == 2 * p.N && (!p.rowpart || (p.uvec && p.cvec)), "gemm_nt_bf16: SimpleGate-backward epilogue needs aux and ldc == 2N");
    if (epi == EB_LNBWD2)
        DCPT_CHECK_ARG(p.res && p.mu && p.rstd && p.lnw && p.colpart && p.rowpart && p.rowparts >= 1 && p.rowparts <= 8 && p.nb == 1,
                       "gemm_nt_bf16: LayerNorm-backward epilogue needs res / mu / rstd / lnw / colpart / rowpart (<= 8 partials per row)");
    if (epi == EB_LNFWD)
        DCPT_CHECK_ARG(p.y2 && p.lnw && p.lnb && p.mu_out && p.rstd_out && p.ldc == p.N && p.nb == 1 && !p.gather2 &&
                           gemm_nt_bf16_ln_epi_ok(p.M, p.N, p.K, p.conv3, p.gC),
                       "gemm_nt_bf16: LayerNorm-forward epilogue needs y2 / lnw / lnb / mu_out / rstd_out, dense rows and N=%d inside one column tile", p.N);
    if (epi == EB_LNBWDM)
        DCPT_CHECK_ARG(p.aux && p.mu && p.rstd && p.lnw && p.colpart && !p.res && (p.ymask || !p.relu || p.lnb) && p.ldc == p.N && p.nb == 1 && !p.gather2 &&
                           gemm_nt_bf16_ln_epi_ok(p.M, p.N, p.K, p.conv3, p.gC),
                       "gemm_nt_bf16: masked LayerNorm-backward epilogue needs aux / mu / rstd / lnw / colpart, dense rows and N=%d inside one column tile", p.N);
    const double mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    double bytes = mk + mn * (epi == EB_SGBWD ? 4 : epi == EB_BIASGATE ? 1.5 : 1) + (double)p.N * p.K;
    if (epi == EB_RESID || epi == EB_DOTCOL || epi == EB_SCATTER_ADD) bytes += mn;
    if (epi == EB_LNBWD2) bytes += 2 * mn;
    if (epi == EB_LNFWD) bytes += mn * (p.res ? 2 : 1);
    if (epi == EB_LNBWDM) bytes += mn * (1 + (p.ymask ? 1 : 0) + (p.y2 ? 1 : 0));
    ProfScope prof(s, PROF_NT + 256 + epi, p.M, p.N, p.K, 2.0 * mn * p.K * p.nb, bytes * 2.0 * p.nb);
    static const int use256 = dcpt_tuning("DCPT_NT256", 1);   // (A/B switch while the kernel is being tuned)
    static const int usetall = dcpt_tuning("DCPT_NT_TALL", 1);   // (A/B switch)
    if (use256 && usetall && gemm_nt_bf16_tall_ok(p, epi, usetall == 2 ? 1 : 192)) {
        trace_tag(p.conv3 ? "nt_bf16.tall512_conv3" : "nt_bf16.tall512");
        return launch_gemm_nt_bf16_tall(p, epi, s);
    }
    if (use256 && gemm_nt_bf16_256_ok(p, epi, use256 == 2 ? 1 : 192)) {
        trace_tag(p.conv3 ? "nt_bf16.256_conv3" : "nt_bf16.256");
        return launch_gemm_nt_bf16_256(p, epi, s);
    }
    trace_tag(p.conv3 ? "nt_bf16.128_conv3" : p.gather2 ? "nt_bf16.128_gather2" : "nt_bf16.128");
    switch (epi) {
        case EB_PLAIN: return launch_nt<EB_PLAIN>(p, s);
        case EB_BIAS: return launch_nt<EB_BIAS>(p, s);
        case EB_RESID: return launch_nt<EB_RESID>(p, s);
        case EB_SGBWD: return launch_nt<EB_SGBWD>(p, s);
        case EB_BIASGATE: return launch_nt<EB_BIASGATE>(p, s);
        case EB_DOTCOL: return launch_nt<EB_DOTCOL>(p, s);
        case EB_LNBWD2: return launch_nt<EB_LNBWD2>(p, s);
        case EB_SCATTER: return launch_nt<EB_SCATTER>(p, s);
        case EB_SCATTER_ADD: return launch_nt<EB_SCATTER_ADD>(p, s);
        case EB_LNFWD: return launch_nt<EB_LNFWD>(p, s);
        case EB_LNBWDM: return launch_nt<EB_LNBWDM>(p, s);
    }
    dcpt_set_error("gemm_nt_bf16: unknown epilogue %d", epi);
    return DCPT_ERR_ARG;
}

// A LayerNorm epilogue needs the whole row in one column tile: N <= 128 on the 128-row kernel, N == 256 on the 256-row kernel (where that
// kernel takes the launch at all: gemm_nt_bf16_256_ok)
bool gemm_nt_bf16_ln_epi_ok(int64_t M, int N, int K, int conv3, int gC) {
    if (N % 8 != 0 || N < 8) return false;
    if (N <= 128) return true;
    if (N != 256) return false;
    static const int use256 = dcpt_tuning("DCPT_NT256", 1);
    GemmNTB q{};
    q.M = M; q.N = N; q.K = K; q.conv3 = conv3; q.gC = gC; q.nb = 1;
    return use256 && gemm_nt_bf16_256_ok(q, EB_LNFWD, use256 == 2 ? 1 : 192);
}

int gemm_nt_bf16_tiles_n(const GemmNTB& p, int epi) {
    const bool gate = epi == EB_BIASGATE;
    const int ncols = gate ? p.N / 2 : p.N;
    return ncols <= (gate ? 32 : 64) ? cdiv(ncols, gate ? 32 : 64) : cdiv(ncols, gate ? 64 : 128);
}

int gemm_tn_bf16_tiles_k(int N, int K) {
    int bn, bk;
    tn_shape(N, K, &bn, &bk);
    return cdiv(K, bk);
}

void gemm_tn_bf16_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split) {
    int bn, bk;
    tn_shape(N, K, &bn, &bk);
    const int64_t tiles = (int64_t)cdiv(N, bn) * cdiv(K, bk);
    // one resident round of the 512 block slots (2 blocks per CU), at least 256 pixels per split
    int64_t want = 512 / tiles;
    const int64_t max_by_rows = cdiv64(M, 256);
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    const int64_t rps = cdiv64(cdiv64(M, want), BRB) * BRB;
    *rows_per_split = rps;
    *splits = (int)cdiv64(M, rps);
}

bool gemm_tn_bf16_plan_images(int64_t M, int N, int K, int P, int* splits, int64_t* rows_per_split) {
    int sp;
    int64_t rps;
    gemm_tn_bf16_plan(M, N, K, &sp, &rps);
    if (M % P != 0) return false;
    int64_t best = 0;
    for (int64_t d = 1; d <= P; ++d)
        if (P % d == 0 && d <= rps + rps / 2) best = d;   // largest divisor of the image not much above the target chunk
    if (best == 0 || best * 4 < rps) return false;
    *rows_per_split = best;
    *splits = (int)(M / best);
    return *splits <= 65535;
}

int launch_gemm_tn_bf16(const GemmTNB& p, hipStream_t s) {
    trace_tag(p.yconv ? "tn_bf16.128_conv3" : "tn_bf16.128");
    DCPT_CHECK_ARG(p.X && p.Y && p.slab && p.M > 0 && p.N > 0 && p.K > 0, "gemm_tn_bf16: null operand or empty problem");
    DCPT_CHECK_ARG(p.N % 8 == 0 && p.K % 8 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0, "gemm_tn_bf16: N=%d, K=%d, strides must be multiples of 8", p.N,
                   p.K);
    DCPT_CHECK_ARG(p.splits >= 1 && p.splits <= 65535 && p.rows_per_split >= 1, "gemm_tn_bf16: bad split plan");
    DCPT_CHECK_ARG((double)(p.rows_per_split + 64) * (double)(p.ldx > p.ldy ? p.ldx : p.ldy) * 2.0 < 1.0e9 && (double)p.N * p.K < 2.0e8,
                   "gemm_tn_bf16: pixel chunk or slab too large for 32-bit window offsets");
    if (p.yconv)
        DCPT_CHECK_ARG(p.gC % 8 == 0 && p.K == 9 * p.gC && p.colsum == nullptr && (double)(p.rows_per_split + 66 + 2 * p.gW) * p.gC * 2.0 < 1.0e9,
                       "gemm_tn_bf16: gathered Y needs K == 9 * gC, gC %% 8 == 0, no column sums");
    if (p.xg2 || p.yg2)
        DCPT_CHECK_ARG(!(p.xg2 && p.yg2) && !p.yconv && p.gC % 8 == 0 && (p.xg2 ? p.N : p.K) == 4 * p.gC && (!p.xg2 || p.colsum == nullptr) &&
                           (double)(p.rows_per_split + 66 + 2 * p.gW) * 4.0 * p.gC * 2.0 < 1.0e9,
                       "gemm_tn_bf16: a gathered operand needs 4 * gC columns, gC %% 8 == 0 (and no column sums of a gathered X)");
    const double bytes = ((double)p.M * p.N + (double)p.M * p.K) * 2.0 + (double)p.splits * p.N * p.K * 4.0;
    ProfScope prof(s, PROF_TN + 256, p.M, p.N, p.K, 2.0 * (double)p.M * p.N * p.K, bytes);
    int bn, bk;
    tn_shape(p.N, p.K, &bn, &bk);
    const int tiles = cdiv(p.N, bn) * cdiv(p.K, bk);
    const dim3 grid((unsigned)(tiles * p.splits));
    if (p.yconv) {
        if (bn == 64) gemm_tn_bf16_kernel<64, 128, 1, 4, 1><<<grid, dim3(256), 0, s>>>(p);   // (K = 9 gC >= 72: the 128-wide k tile)
        else gemm_tn_bf16_kernel<128, 128, 2, 2, 1><<<grid, dim3(256), 0, s>>>(p);
        DCPT_CHECK_LAUNCH("gemm_tn_bf16 conv3");
        return DCPT_OK;
    }
    if (p.yg2 || p.xg2) {   // N resp. K = 4 gC >= 32 on the gathered side
#define TNG(BN_, BK_, WN_, WK_)                                                                         \
    do {                                                                                                \
        if (p.yg2) gemm_tn_bf16_kernel<BN_, BK_, WN_, WK_, 2, false><<<grid, dim3(256), 0, s>>>(p);     \
        else gemm_tn_bf16_kernel<BN_, BK_, WN_, WK_, 0, true><<<grid, dim3(256), 0, s>>>(p);           \
    } while (0)
        if (bn == 64 && bk == 64) TNG(64, 64, 2, 2);
        else if (bk == 64) TNG(128, 64, 4, 1);
        else if (bn == 64) TNG(64, 128, 1, 4);
        else TNG(128, 128, 2, 2);
#undef TNG
        DCPT_CHECK_LAUNCH("gemm_tn_bf16 gather2");
        return DCPT_OK;
    }
    if (bn == 64 && bk == 64) gemm_tn_bf16_kernel<64, 64, 2, 2><<<grid, dim3(256), 0, s>>>(p);
    else if (bk == 64) gemm_tn_bf16_kernel<128, 64, 4, 1><<<grid, dim3(256), 0, s>>>(p);
    else if (bn == 64) gemm_tn_bf16_kernel<64, 128, 1, 4><<<grid, dim3(256), 0, s>>>(p);
    else gemm_tn_bf16_kernel<128, 128, 2, 2><<<grid, dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("gemm_tn_bf16");
    return DCPT_OK;
}
