// NT GEMM on fp32 MFMA:  C[m][n] = sum_k A'(m,k) * Bw[n][k]  with fused operand loader + epilogue.
//
// Tile: BM x BN x 32, 256 threads = 4 waves (64 lanes each), every wave owns TM x TN MFMA tiles of
// 32x32 (v_mfma_f32_32x32x2_f32).  A and Bw tiles live in LDS as [rows][32] floats, UNPADDED, with the
// eight 16-byte k-quads of row r stored at slot q ^ ((r >> 1) & 7): ds_read_b128 of 32 consecutive rows at
// one k-quad is conflict-free, and a wave's 64 lanes (8 rows x 8 quads) cover 1 KiB of contiguous LDS.
// That is exactly the image an LDS-DMA (buffer_load_dwordx4 ... lds) writes, so operands that need no math
// (weights, plain / gathered / implicit-conv activations) go global -> LDS without touching VGPRs, the swizzle
// being applied on the SOURCE side (each lane fetches k-quad slot ^ f(row)).  Operands with a fused transform
// (LayerNorm, SimpleGate, SCA scale) are loaded to registers before the current tile's MFMAs and written
// (transformed) to the same image after them.  Double buffered, one barrier per k-tile.
// Lane (i = lane&31, h = lane>>5) reads the 4 consecutive k at quad 2j+h of row i as one ds_read_b128 and feeds
// four MFMAs (k order inside a block of 8 is {0,4},{1,5},{2,6},{3,7}; A and B use the same order).
//
// Block id -> tile: XCD-aware (dcpt_common.h xcd_remap); consecutive logical ids walk the N tiles of
// one M panel so an A panel is fetched from HBM once per XCD and re-read from that XCD's L2.
#include "gemm_nt_epi.h"
#include "prof.h"

#ifdef DCPT_TIMELINE
// Diagnostic build only (tools/timeline.py): per-block phase stamps, 100 MHz wall clock + HW id.
__device__ unsigned long long g_tl[1 << 15][12];
extern "C" int dcpt_timeline_read(unsigned long long* host, int nblk) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * 12 * (size_t)nblk);
}
#define TL_STAMP(i) if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < (1 << 15)) { g_tl[blockIdx.x][i] = wall_clock64(); if (i == 1) g_tl[blockIdx.x][6] = clock64(); if (i == 2) g_tl[blockIdx.x][7] = clock64(); }
#define TL_HWID() if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < (1 << 15)) { unsigned hw, xcc; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    g_tl[blockIdx.x][4] = hw; g_tl[blockIdx.x][5] = xcc; }
#if DCPT_TIMELINE >= 2
#define TL_LOOP_DECL unsigned long long tl_c = 0, tl_v = 0, tl_b = 0, tl_1, tl_2, tl_3, tl_4;
#define TL_T(n) tl_##n = __builtin_readcyclecounter();
#define TL_ACC() tl_c += tl_2 - tl_1; tl_v += tl_3 - tl_2; tl_b += tl_4 - tl_3;
#define TL_LOOP_END() if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < (1 << 15)) { g_tl[blockIdx.x][8] = tl_c; g_tl[blockIdx.x][9] = tl_v; g_tl[blockIdx.x][10] = tl_b; }
#else
#define TL_LOOP_DECL
#define TL_T(n)
#define TL_ACC()
#define TL_LOOP_END()
#endif
#else
#define TL_STAMP(i)
#define TL_HWID()
#define TL_LOOP_DECL
#define TL_T(n)
#define TL_ACC()
#define TL_LOOP_END()
#endif

namespace {

template <int BM, int BN, int WM, int WN, int AK, int EK, int BK>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const GemmNT pin) {
    static_assert(BK == 32 && WM * WN == 4, "32-deep k-tiles, 4 waves");
    constexpr int NTHR = 256;
    constexpr bool A_DMA = (AK == A_PLAIN || AK == A_GATHER || AK == A_CONV3);  // pure data movement
    GemmNT p = pin;
    TL_STAMP(0) TL_HWID()
    int lin, batch;
    xcd_remap_batched(lin, batch);
    if (gridDim.y > 1) {  // batched: shift the base pointers of this problem
        const int b1 = batch / p.nb2, b2 = batch % p.nb2;
        p.A += b1 * p.sA1 + b2 * p.sA2;
        p.Bw += b1 * p.sB1 + b2 * p.sB2;
        p.C += b1 * p.sC1 + b2 * p.sC2;
        if (p.res) p.res += b1 * p.sR1 + b2 * p.sR2;
        if (p.cscale) p.cscale += b1 * p.sS1 + b2 * p.sS2;
    }
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = BM / 32, B_IT = BN / 32;      // 256 threads cover 32 rows x 8 quads per pass
    constexpr int A_SZ = BM * 32, B_SZ = BN * 32;
    constexpr int OP_SZ = 2 * (A_SZ + B_SZ);
    static_assert(OP_SZ >= BM * BN, "epilogue staging does not fit");
    __shared__ __attribute__((aligned(16))) float smem[OP_SZ];
    float* const As0 = smem;               // [2][A_SZ]
    float* const Bs0 = smem + 2 * A_SZ;    // [2][B_SZ]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    constexpr bool GATE = (EK == E_BIASGATE);
    const int tilesN = GATE ? (p.N / 2 + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    const int64_t m0 = (int64_t)(lin / tilesN) * BM;
    const int n0 = (lin % tilesN) * (GATE ? BN / 2 : BN);   // GATE: first column of the tile inside each half

    Operand oa;
    oa.ptr = p.A; oa.M = p.M; oa.ncols = p.K; oa.ld = p.lda;
    oa.mu = p.mu; oa.rstd = p.rstd; oa.lnw = p.lnw; oa.lnb = p.lnb;
    oa.simg = p.simg; oa.P = p.P; oa.gH = p.gH; oa.gW = p.gW; oa.gC = p.gC;
    open_window<AK>(oa, m0 < p.M ? m0 : 0);
    const i32x4 rsB = make_rsrc_dma(p.Bw + (GATE ? 0 : (int64_t)n0 * p.K));
    rsrc_t rsW = oa.rs, rsWb = oa.rs;
    if constexpr (AK == A_LN || AK == A_LNBF) rsW = make_rsrc(p.lnw, (uint32_t)p.K * 4u);
    if constexpr (AK == A_LN) rsWb = make_rsrc(p.lnb, (uint32_t)p.K * 4u);

    // staging map: thread -> row (tid >> 3) + 32 * pass, LDS slot tid & 7, i.e. logical k-quad slot ^ f(row)
    const int lrow = tid >> 3;
    const int lk = 4 * ((tid & 7) ^ ((tid >> 4) & 7));
    RowCtx rca[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) make_row<AK>(oa, m0 + lrow + 32 * i, rca[i]);
    uint32_t boff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int nl = lrow + 32 * i;
        if constexpr (GATE) {  // tile row nl < BN/2: column n0 + nl of the first half, else the partner column of the second half
            const int hl = nl % (BN / 2), Ch = p.N / 2;
            const int n = (nl < BN / 2 ? 0 : Ch) + n0 + hl;
            boff[i] = (n0 + hl < Ch) ? (uint32_t)n * (uint32_t)p.K * 4u + 4u * (uint32_t)lk : ROW_SENT;
        } else {
            boff[i] = (n0 + nl < p.N) ? (uint32_t)nl * (uint32_t)p.K * 4u + 4u * (uint32_t)lk : ROW_SENT;
        }
    }
    const uint32_t lds_a = lds_addr(As0) + wave * 1024, lds_b = lds_addr(Bs0) + wave * 1024;   // this wave's 8 rows inside a 32-row pass
    const int lds_t = tid * 4;          // this thread's quad inside a 32-row pass (floats)

    RawVec ra[A_IT];
    float4 kw = f4_zero(), kb = f4_zero();  // A_LN / A_LNBF: per-column weight/bias of this thread's k quad (same for all its rows)
    auto gload = [&](int kt, int buf) {
        const int k = kt * BK + lk;
        const uint32_t ksent = (k < p.K) ? 0u : COL_SENT;   // only the last k-tile of a ragged K
#pragma unroll
        for (int i = 0; i < B_IT; ++i) dma16(rsB, lds_b + (buf * B_SZ + i * 1024) * 4, boff[i] + ksent, (uint32_t)kt * (BK * 4));
        if constexpr (A_DMA) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) dma16(oa.rsd, lds_a + (buf * A_SZ + i * 1024) * 4, elem_voff<AK>(oa, rca[i], k), 0);
        } else {
            if constexpr (AK == A_LN || AK == A_LNBF) {
                kw = buf_ld4(rsW, 4u * (uint32_t)k);
                if constexpr (AK == A_LN) kb = buf_ld4(rsWb, 4u * (uint32_t)k);
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) load_raw<AK>(oa, rca[i], k, ra[i]);
        }
    };
    auto lstore = [&](int buf) {
        if constexpr (!A_DMA) {
            if constexpr (AK == A_LN || AK == A_LNBF) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    ra[i].b = kw;
                    ra[i].c = kb;
                }
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4*>(&As0[buf * A_SZ + i * 1024 + lds_t]) = finish<AK>(rca[i], ra[i]);
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    gload(0, 0);
    lstore(0);
    dma_wait_all();
    __syncthreads();
    TL_STAMP(1)
    // fragment addresses: row (lane & 31) of the wave's tile, slot (2j + h) ^ f(row) for k-group j
    const int fi = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    const int a_row = (wm * TM * 32 + (lane & 31)) * 32;
    const int b_row = (wn * TN * 32 + (lane & 31)) * 32;
    int slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = ((2 * j + fh) ^ fi) * 4;
    TL_LOOP_DECL
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        TL_T(1)
        if (kt + 1 < nkt) gload(kt + 1, buf ^ 1);
        const float* as = As0 + buf * A_SZ + a_row;
        const float* bs = Bs0 + buf * B_SZ + b_row;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 1024 + slot[j]);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) bf[jn] = *reinterpret_cast<const float4*>(bs + jn * 1024 + slot[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[jn].x, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[jn].y, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[jn].z, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[jn].w, acc[i][jn], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        TL_T(2)
        dma_wait_all();   // this wave's LDS-DMA pieces of tile kt+1 have landed; the barrier publishes them
        TL_T(3)
        __syncthreads();
        TL_T(4) TL_ACC()
    }
    TL_LOOP_END()

    TL_STAMP(2)
    // park the accumulators in LDS (the last k-tile's barrier already separates us from the MFMA reads)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = (wm * TM + i) * 32;   // first row of this MFMA tile inside the block tile
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = rb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                smem[ml * BN + nl] = acc[i][j][r];
            }
        }
    }
    __syncthreads();
    if constexpr (GATE) epilogue_gate<BM, BN, NTHR>(p, smem, m0, n0, tid);
    else epilogue_rows<EK, BM, BN, NTHR>(p, smem, m0, n0, tid);
    TL_STAMP(3)
}

// Tile width of a launch.  Row-spanning epilogues (E_LNBWD / E_RESIDLN) need the whole row in one tile.  Otherwise: 96-wide
// tiles when they pad N less than 128-wide ones (Restormer's 96 / 192 / 288 / 576-channel layers); 128 x 64 tiles (48 KB of LDS:
// three resident blocks per CU instead of two) whenever the grid is only a few rounds of the resident slots -- a grid of 580
// 128 x 128 tiles is 1.13 rounds of 512 slots and runs as two, the same work as 1160 narrower tiles is 1.5 rounds of 768 slots,
// and three co-resident blocks hide each other's prologue / epilogue better (measured: 2K tiled inference -9 %, the training
// step -0.9 %: its 1024- and 2048-tile launches at the deepest level); grids of many rounds keep the wider tile (fewer LDS
// reads per flop).
int nt_pick_bn(const GemmNT& p, int epi) {
    const bool gate = (epi == E_BIASGATE);
    const unsigned nbatch = (unsigned)((p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1));
    if (epi == E_LNBWD || epi == E_RESIDLN) return p.N <= 64 ? 64 : 128;
    static const int smallk = dcpt_tuning("DCPT_NT_SMALLK", 0);
    static const int use96 = dcpt_tuning("DCPT_NT_96", 1);
    static const int use64_below = dcpt_tuning("DCPT_NT_64_BELOW", 2100);
    if (!gate && use96 && p.N > 64 && cdiv(p.N, 96) * 96 < cdiv(p.N, 128) * 128 && cdiv64(p.M, 128) * cdiv(p.N, 96) * nbatch > 256) return 96;
    const int64_t tiles128 = cdiv64(p.M, 128) * (gate ? cdiv(p.N / 2, 64) : cdiv(p.N, 128));
    if (p.N <= 64 || tiles128 * nbatch < use64_below || p.K <= smallk) return 64;
    return 128;
}

template <int AK, int EK>
int launch_cfg(const GemmNT& p, hipStream_t s) {
    const unsigned nbatch = (unsigned)((p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1));
    constexpr bool GATE = (EK == E_BIASGATE);
    const int bn = nt_pick_bn(p, EK);
    const int64_t tiles = cdiv64(p.M, 128) * (GATE ? cdiv(p.N / 2, bn / 2) : cdiv(p.N, bn));
    const dim3 grid((unsigned)tiles, nbatch);
    if (bn == 64) {
        gemm_nt_kernel<128, 64, 4, 1, AK, EK, 32><<<grid, dim3(256), 0, s>>>(p);
    } else if (bn == 96) {
        if constexpr (!GATE && EK != E_LNBWD && EK != E_RESIDLN) gemm_nt_kernel<128, 96, 4, 1, AK, EK, 32><<<grid, dim3(256), 0, s>>>(p);
    } else {
        gemm_nt_kernel<128, 128, 2, 2, AK, EK, 32><<<grid, dim3(256), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_nt");
    return DCPT_OK;
}

}  // namespace

int gemm_nt_tiles_n(const GemmNT& pin, int aload, int epi) {
    (void)aload;
    GemmNT p = pin;
    const int bn = nt_pick_bn(p, epi);
    return epi == E_BIASGATE ? cdiv(p.N / 2, bn / 2) : cdiv(p.N, bn);
}

int launch_gemm_nt(const GemmNT& pin, int aload, int epi, hipStream_t s) {
    GemmNT p = pin;
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;

    DCPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%lld N=%d K=%d", (long long)p.M, p.N, p.K);
    DCPT_CHECK_ARG(p.K % 4 == 0, "gemm_nt: K=%d must be a multiple of 4", p.K);
    DCPT_CHECK_ARG(cdiv64(p.M, 128) * cdiv(p.N, 64) < (1ll << 31), "gemm_nt: grid too large");
    // 32-bit offsets inside a tile's buffer window (gemm_operand.h)
    DCPT_CHECK_ARG(p.K < (1 << 20) && p.N < (1 << 20) && p.lda < (1 << 20) && p.ldc < (1 << 20) && p.ldres < (1 << 20),
                   "gemm_nt: K/N/row strides must be below 2^20");
    if (aload == A_GATHER) DCPT_CHECK_ARG(p.gC % 4 == 0 && p.K == 4 * p.gC, "gemm_nt: gather needs K == 4*gC, gC %% 4 == 0");
    if (epi == E_BIASGATE)
        DCPT_CHECK_ARG(p.gate && p.N % 8 == 0 && (double)p.N * p.K * 4.0 < 1.0e9, "gemm_nt: gate epilogue needs gate != null, N %% 8 == 0");
    if (epi == E_DOTCOL) DCPT_CHECK_ARG(p.colpart && p.res && p.nb1 * p.nb2 == 1, "gemm_nt: column-dot epilogue needs colpart and res");
    if (epi == E_RESIDLN)
        DCPT_CHECK_ARG(p.res && p.ln_out && p.ln_mu && p.ln_rstd && p.lnw && p.N <= 128 && p.N % 4 == 0 && p.nb1 * p.nb2 == 1 && p.ldc == p.N,
                       "gemm_nt: residual + LayerNorm epilogue needs N <= 128, ldc == N and res / lnw / ln_out / ln_mu / ln_rstd");
    if (epi == E_LNBWD)
        DCPT_CHECK_ARG(p.colpart && p.res && p.mu && p.rstd && p.lnw && p.N <= 128 && p.N % 4 == 0 && p.nb1 * p.nb2 == 1,
                       "gemm_nt: LayerNorm-backward epilogue needs N <= 128 and res / mu / rstd / lnw / colpart");
    if (aload == A_CONV3) DCPT_CHECK_ARG(p.gC % 4 == 0 && p.K == 9 * p.gC, "gemm_nt: conv3 needs K == 9*gC, gC %% 4 == 0");
    // algorithmic work of this launch (for the live roofline in bench.py)
    const double mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    double bytes = mk * (aload == A_SG ? 2 : 1) + mn * (epi == E_SGBWD ? 4 : epi == E_BIASGATE ? 1.5 : 1) + (double)p.N * p.K;
    if (epi == E_RESID || epi == E_SCATTER_ADD || epi == E_DOTCOL) bytes += mn;
    if (epi == E_LNBWD || epi == E_LNBWD2) bytes += 2 * mn;
    if (epi == E_LNBWD2)
        DCPT_CHECK_ARG(p.colpart && p.res && p.mu && p.rstd && p.lnw && p.rowpart && p.rowparts >= 1 && p.rowparts <= 16 && p.N % 4 == 0 &&
                           p.nb1 * p.nb2 == 1, "gemm_nt: LayerNorm-backward (row sums supplied) epilogue needs res / mu / rstd / lnw / colpart / rowpart, <= 16 row partials");
    if (epi == E_SGBWD && p.rowpart) DCPT_CHECK_ARG(p.uvec && p.cvec, "gemm_nt: SimpleGate-backward row partials need uvec / cvec");
    if (epi == E_RESIDLN) bytes += 2 * mn;
    const double nbat = (double)((p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1));
    ProfScope prof(s, PROF_NT + aload * 16 + epi, p.M, p.N, p.K, 2.0 * mn * p.K * nbat, bytes * 4.0 * nbat);
    if (gemm_nt_x3_ok(p, aload, epi)) return launch_gemm_nt_x3(p, aload, epi, s);   // (opt-in mode; off unless dcpt_set_gemm_x3 was called)
#define CASE(AK, EK) \
    if (aload == AK && epi == EK) return launch_cfg<AK, EK>(p, s);
    CASE(A_LN, E_BIAS)
    CASE(A_SCALE, E_RESID)
    CASE(A_SG, E_RESID)
    CASE(A_PLAIN, E_SGBWD)
    CASE(A_PLAIN, E_PLAIN)
    CASE(A_PLAIN, E_BIAS)
    CASE(A_GATHER, E_BIAS)
    CASE(A_PLAIN, E_SCATTER)
    CASE(A_PLAIN, E_SCATTER_ADD)
    CASE(A_GATHER, E_PLAIN)
    CASE(A_CONV3, E_PLAIN)
    CASE(A_LNBF, E_PLAIN)
    CASE(A_LN, E_PLAIN)
    CASE(A_PLAIN, E_RESID)
    CASE(A_PLAIN, E_ADDSCALED)
    CASE(A_PLAIN, E_MUL)
    CASE(A_PLAIN, E_BIASGATE)
    CASE(A_PLAIN, E_DOTCOL)
    CASE(A_PLAIN, E_LNBWD)
    CASE(A_PLAIN, E_LNBWD2)
    CASE(A_SCALE, E_RESIDLN)
#undef CASE
    dcpt_set_error("gemm_nt: unsupported loader/epilogue combination %d/%d", aload, epi);
    return DCPT_ERR_ARG;
}

