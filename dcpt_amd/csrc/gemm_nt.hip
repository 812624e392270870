// NT GEMM on fp32 MFMA:  C[m][n] = sum_k A'(m,k) * Bw[n][k]  with fused operand loader + epilogue.
//
// Tile: BM x BN x 32, 256 threads = 4 waves (64 lanes each), every wave owns TM x TN MFMA tiles of
// 32x32 (v_mfma_f32_32x32x2_f32).  A and Bw tiles are staged through LDS ([rows][32+4] floats, the
// +4 keeps ds_read_b128 conflict-free for 32 consecutive rows), double buffered: the next tile's
// global loads are issued into registers before the current tile's MFMAs and written to the other
// LDS buffer after them, one barrier per k-tile.  Lane (i = lane&31, h = lane>>5) reads 4
// consecutive k of row i at k-offset 4h as one ds_read_b128 and feeds four MFMAs (the k order inside
// a block of 8 is {0,4},{1,5},{2,6},{3,7}; A and B use the same order so the sum is unchanged).
//
// Block id -> tile: XCD-aware (dcpt_common.h xcd_remap); consecutive logical ids walk the N tiles of
// one M panel so an A panel is fetched from HBM once per XCD and re-read from that XCD's L2.
#include "gemm_operand.h"
#include "prof.h"

namespace {


// Epilogue: the accumulators are first parked in LDS ([BM][BN+4] floats, reusing the operand tiles'
// space), then every thread owns one float4 column group of 16 rows: global loads (residual / gate
// inputs) and stores are 16 B per lane and 512 B contiguous per row, and all the loads of a thread are
// issued before the LDS round trip so their latency overlaps it.
template <int EK, int BM, int BN, int NT_>
__device__ __forceinline__ void epilogue_rows(const GemmNT& p, const float* __restrict__ Cs, int64_t m0, int n0, int tid) {
    constexpr int NTHR = NT_;
    const int ldres = p.ldres ? p.ldres : p.ldc;
    constexpr int LDC = BN + 4;
    constexpr int Q = BN / 4;          // float4 groups per row
    constexpr int RPP = NTHR / Q;      // rows per pass
    constexpr int IT = BM / RPP;
    const int q = tid % Q, r0 = tid / Q;
    const int n = n0 + 4 * q;
    if (n >= p.N) return;
    float4 bias = f4_zero(), cs = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (EK == E_BIAS || EK == E_RESID || EK == E_MUL) {
        if (p.bias) bias = ldg4(p.bias + n);
    }
    if constexpr (EK == E_RESID || EK == E_ADDSCALED) {
        if (p.cscale) cs = ldg4(p.cscale + n);
    }
    int si = 0, sj = 0, ch = 0;
    if constexpr (EK == E_SCATTER || EK == E_SCATTER_ADD) {
        const int ij = n / p.gC;
        ch = n - ij * p.gC;
        si = ij >> 1;
        sj = ij & 1;
    }
    constexpr int HALF = (EK == E_SGBWD) ? 2 : 1;   // SGBWD needs two loads per row: do it in two halves
    constexpr int ITH = IT / HALF;
#pragma unroll
    for (int hh = 0; hh < HALF; ++hh) {
        float4 pre1[ITH], pre2[ITH];
        int64_t addr[ITH];
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int64_t m = m0 + r0 + (int64_t)(hh * ITH + it) * RPP;
            const bool ok = m < p.M;
            pre1[it] = f4_zero();
            pre2[it] = f4_zero();
            if constexpr (EK == E_RESID || EK == E_ADDSCALED || EK == E_MUL) {
                addr[it] = m * p.ldc + n;
                if (ok) pre1[it] = ldg4(p.res + m * ldres + n);
            } else if constexpr (EK == E_SGBWD) {
                addr[it] = m * p.ldc + n;
                if (ok) {
                    pre1[it] = ldg4(p.aux + m * (2 * (int64_t)p.N) + n);
                    pre2[it] = ldg4(p.aux + m * (2 * (int64_t)p.N) + p.N + n);
                }
            } else if constexpr (EK == E_SCATTER || EK == E_SCATTER_ADD) {
                const int64_t mm = ok ? m : 0;
                const int w = (int)(mm % p.gW);
                const int64_t t = mm / p.gW;
                const int h = (int)(t % p.gH);
                const int64_t b = t / p.gH;
                addr[it] = ((b * (2 * p.gH) + 2 * h + si) * (int64_t)(2 * p.gW) + 2 * w + sj) * p.gC + ch;
                if constexpr (EK == E_SCATTER_ADD) {
                    if (ok) pre1[it] = ldg4(p.res + addr[it]);
                }
            } else {
                addr[it] = m * p.ldc + n;
            }
        }
#pragma unroll
        for (int it = 0; it < ITH; ++it) {
            const int rl = r0 + (hh * ITH + it) * RPP;
            const int64_t m = m0 + rl;
            if (m >= p.M) continue;
            const float4 v = *reinterpret_cast<const float4*>(&Cs[rl * LDC + 4 * q]);
            if constexpr (EK == E_PLAIN || EK == E_SCATTER) {
                stg4(p.C + addr[it], v);
            } else if constexpr (EK == E_BIAS) {
                stg4(p.C + addr[it], f4_add(v, bias));
            } else if constexpr (EK == E_RESID) {
                stg4(p.C + addr[it], f4_fma(f4_add(v, bias), cs, pre1[it]));
            } else if constexpr (EK == E_ADDSCALED) {
                stg4(p.C + addr[it], f4_fma(cs, pre1[it], v));
            } else if constexpr (EK == E_MUL) {
                stg4(p.C + addr[it], f4_mul(f4_add(v, bias), pre1[it]));
            } else if constexpr (EK == E_SGBWD) {
                stg4(p.C + addr[it], f4_mul(v, pre2[it]));
                stg4(p.C + addr[it] + p.N, f4_mul(v, pre1[it]));
            } else {  // E_SCATTER_ADD
                stg4(p.C + addr[it], f4_add(v, pre1[it]));
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AK, int EK, int BK>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const GemmNT pin) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int LDS_LD = BK + 4;
    constexpr int TPR = BK / 4;        // threads per tile row (one float4 each)
    constexpr int RPP = NTHR / TPR;    // rows per pass
    GemmNT p = pin;
    if (gridDim.y > 1) {  // batched: shift the base pointers of this problem
        const int b1 = blockIdx.y / p.nb2, b2 = blockIdx.y % p.nb2;
        p.A += b1 * p.sA1 + b2 * p.sA2;
        p.Bw += b1 * p.sB1 + b2 * p.sB2;
        p.C += b1 * p.sC1 + b2 * p.sC2;
        if (p.res) p.res += b1 * p.sR1 + b2 * p.sR2;
        if (p.cscale) p.cscale += b1 * p.sS1 + b2 * p.sS2;
    }
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    constexpr int A_SZ = BM * LDS_LD, B_SZ = BN * LDS_LD;
    constexpr int OP_SZ = 2 * (A_SZ + B_SZ);
    // the accumulators are parked in the operand area for the epilogue; if it is too small for the whole tile
    // (16-deep k-tiles) the epilogue runs in two row halves
    constexpr int C_ROWS = (OP_SZ >= BM * (BN + 4)) ? BM : BM / 2;
    static_assert(OP_SZ >= C_ROWS * (BN + 4), "epilogue staging does not fit");
    __shared__ __attribute__((aligned(16))) float smem[OP_SZ];
    float* const As0 = smem;               // [2][A_SZ]
    float* const Bs0 = smem + 2 * A_SZ;    // [2][B_SZ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tilesN = (p.N + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(lin / tilesN) * BM;
    const int n0 = (lin % tilesN) * BN;

    Operand oa;
    oa.ptr = p.A; oa.M = p.M; oa.ncols = p.K; oa.ld = p.lda;
    oa.mu = p.mu; oa.rstd = p.rstd; oa.lnw = p.lnw; oa.lnb = p.lnb;
    oa.simg = p.simg; oa.P = p.P; oa.gH = p.gH; oa.gW = p.gW; oa.gC = p.gC;

    const int lrow = tid / TPR, lk = (tid % TPR) * 4;
    RowCtx rca[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) make_row<AK>(oa, m0 + lrow + RPP * i, rca[i]);
    const float* brow[B_IT];
    bool bval[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + lrow + RPP * i;
        bval[i] = n < p.N;
        brow[i] = p.Bw + (int64_t)(bval[i] ? n : 0) * p.K;
    }

    RawVec ra[A_IT];
    float4 rb[B_IT];
    float4 kw = f4_zero(), kb = f4_zero();  // A_LN / A_LNBF: per-column weight/bias of this thread's k quad (same for all its rows)
    auto gload = [&](int kt) {
        const int k = kt * BK + lk;
        if constexpr (AK == A_LN || AK == A_LNBF) {
            kw = (k < p.K) ? ldg4(p.lnw + k) : f4_zero();
            if constexpr (AK == A_LN) kb = (k < p.K) ? ldg4(p.lnb + k) : f4_zero();
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i].a = (rca[i].valid && k < p.K) ? ldg4(p.A + rca[i].off + k) : f4_zero();
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) load_raw<AK>(oa, rca[i], k, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = (bval[i] && k < p.K) ? ldg4(brow[i] + k) : f4_zero();
    };
    auto lstore = [&](int buf) {
        if constexpr (AK == A_LN || AK == A_LNBF) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                ra[i].b = kw;
                ra[i].c = kb;
            }
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<float4*>(&As0[buf * A_SZ + (lrow + RPP * i) * LDS_LD + lk]) = finish<AK>(rca[i], ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs0[buf * B_SZ + (lrow + RPP * i) * LDS_LD + lk]) = rb[i];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int a_off = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int b_off = (wn * TN * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const float* as = As0 + buf * A_SZ + a_off;
        const float* bs = Bs0 + buf * B_SZ + b_off;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_LD + kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // park the accumulators in LDS (the last k-tile's barrier already separates us from the MFMA reads)
    constexpr int LDC = BN + 4;
#pragma unroll
    for (int ps = 0; ps < BM / C_ROWS; ++ps) {
        if (ps > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = (wm * TM + i) * 32;   // first row of this MFMA tile inside the block tile
            if (rb / C_ROWS != ps) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = rb - ps * C_ROWS + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    smem[ml * LDC + nl] = acc[i][j][r];
                }
            }
        }
        __syncthreads();
        epilogue_rows<EK, C_ROWS, BN, NTHR>(p, smem, m0 + ps * C_ROWS, n0, tid);
    }
}

template <int AK, int EK>
int launch_cfg(const GemmNT& p, hipStream_t s) {
    const unsigned nbatch = (unsigned)((p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1));
    if (p.N <= 64) {
        constexpr int BM = 128, BN = 64;
        const int64_t tiles = cdiv64(p.M, BM) * cdiv(p.N, BN);
        gemm_nt_kernel<BM, BN, 4, 1, AK, EK, 32><<<dim3((unsigned)tiles, nbatch), dim3(256), 0, s>>>(p);
    } else {
        constexpr int BM = 128, BN = 128;
        const int64_t tiles = cdiv64(p.M, BM) * cdiv(p.N, BN);
        gemm_nt_kernel<BM, BN, 2, 2, AK, EK, 32><<<dim3((unsigned)tiles, nbatch), dim3(256), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_nt");
    return DCPT_OK;
}

}  // namespace

int launch_gemm_nt(const GemmNT& pin, int aload, int epi, hipStream_t s) {
    GemmNT p = pin;
    if (p.nb1 < 1) p.nb1 = 1;
    if (p.nb2 < 1) p.nb2 = 1;

    DCPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%lld N=%d K=%d", (long long)p.M, p.N, p.K);
    DCPT_CHECK_ARG(p.K % 4 == 0, "gemm_nt: K=%d must be a multiple of 4", p.K);
    DCPT_CHECK_ARG(cdiv64(p.M, 128) * cdiv(p.N, 64) < (1ll << 31), "gemm_nt: grid too large");
    if (aload == A_GATHER) DCPT_CHECK_ARG(p.gC % 4 == 0 && p.K == 4 * p.gC, "gemm_nt: gather needs K == 4*gC, gC %% 4 == 0");
    if (aload == A_CONV3) DCPT_CHECK_ARG(p.gC % 4 == 0 && p.K == 9 * p.gC, "gemm_nt: conv3 needs K == 9*gC, gC %% 4 == 0");
    // algorithmic work of this launch (for the live roofline in bench.py)
    const double mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    double bytes = mk * (aload == A_SG ? 2 : 1) + mn * (epi == E_SGBWD ? 4 : 1) + (double)p.N * p.K;
    if (epi == E_RESID || epi == E_SCATTER_ADD) bytes += mn;
    const double nbat = (double)((p.nb1 > 0 ? p.nb1 : 1) * (p.nb2 > 0 ? p.nb2 : 1));
    ProfScope prof(s, PROF_NT + aload * 8 + epi, p.M, p.N, p.K, 2.0 * mn * p.K * nbat, bytes * 4.0 * nbat);
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
#undef CASE
    dcpt_set_error("gemm_nt: unsupported loader/epilogue combination %d/%d", aload, epi);
    return DCPT_ERR_ARG;
}

