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

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int EK>
__device__ __forceinline__ void epilogue_tile(const GemmNT& p, const floatx16& acc, int64_t mbase, int n, int lane) {
    if (n >= p.N) return;
    float bias = 0.f, cs = 1.f;
    if constexpr (EK == E_BIAS || EK == E_RESID) bias = p.bias ? p.bias[n] : 0.f;
    if constexpr (EK == E_RESID) cs = p.cscale[n];
    int si = 0, sj = 0, ch = 0;
    if constexpr (EK == E_SCATTER || EK == E_SCATTER_ADD) {
        const int ij = n / p.gC;
        ch = n - ij * p.gC;
        si = ij >> 1;
        sj = ij & 1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        const float v = acc[r];
        if constexpr (EK == E_PLAIN) {
            p.C[m * p.ldc + n] = v;
        } else if constexpr (EK == E_BIAS) {
            p.C[m * p.ldc + n] = v + bias;
        } else if constexpr (EK == E_RESID) {
            p.C[m * p.ldc + n] = fmaf(v + bias, cs, p.res[m * p.ldc + n]);
        } else if constexpr (EK == E_SGBWD) {
            const float a1 = p.aux[m * (2 * (int64_t)p.N) + n];
            const float a2 = p.aux[m * (2 * (int64_t)p.N) + p.N + n];
            p.C[m * p.ldc + n] = v * a2;
            p.C[m * p.ldc + p.N + n] = v * a1;
        } else {  // scatter to the fine image
            const int w = (int)(m % p.gW);
            const int64_t t = m / p.gW;
            const int h = (int)(t % p.gH);
            const int64_t b = t / p.gH;
            const int64_t a = ((b * (2 * p.gH) + 2 * h + si) * (int64_t)(2 * p.gW) + 2 * w + sj) * p.gC + ch;
            if constexpr (EK == E_SCATTER) p.C[a] = v;
            else p.C[a] = v + p.res[a];
        }
    }
}

template <int BM, int BN, int WM, int WN, int AK, int EK>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNT p) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];

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

    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    RowCtx rca[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) make_row<AK>(oa, m0 + lrow + 32 * i, rca[i]);
    const float* brow[B_IT];
    bool bval[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + lrow + 32 * i;
        bval[i] = n < p.N;
        brow[i] = p.Bw + (int64_t)(bval[i] ? n : 0) * p.K;
    }

    RawVec ra[A_IT];
    float4 rb[B_IT];
    auto gload = [&](int kt) {
        const int k = kt * BK + lk;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) load_raw<AK>(oa, rca[i], k, ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = (bval[i] && k < p.K) ? ldg4(brow[i] + k) : f4_zero();
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<float4*>(&As[buf][(lrow + 32 * i) * LDS_LD + lk]) = finish<AK>(rca[i], ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs[buf][(lrow + 32 * i) * LDS_LD + lk]) = rb[i];
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
        const float* as = &As[buf][a_off];
        const float* bs = &Bs[buf][b_off];
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

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            epilogue_tile<EK>(p, acc[i][j], m0 + (wm * TM + i) * 32, n0 + (wn * TN + j) * 32 + (lane & 31), lane);
}

template <int AK, int EK>
int launch_cfg(const GemmNT& p, hipStream_t s) {
    if (p.N <= 64) {
        constexpr int BM = 128, BN = 64;
        const int64_t tiles = cdiv64(p.M, BM) * cdiv(p.N, BN);
        gemm_nt_kernel<BM, BN, 4, 1, AK, EK><<<dim3((unsigned)tiles), dim3(256), 0, s>>>(p);
    } else {
        constexpr int BM = 128, BN = 128;
        const int64_t tiles = cdiv64(p.M, BM) * cdiv(p.N, BN);
        gemm_nt_kernel<BM, BN, 2, 2, AK, EK><<<dim3((unsigned)tiles), dim3(256), 0, s>>>(p);
    }
    DCPT_CHECK_LAUNCH("gemm_nt");
    return DCPT_OK;
}

}  // namespace

int launch_gemm_nt(const GemmNT& p, int aload, int epi, hipStream_t s) {
    DCPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: empty problem M=%lld N=%d K=%d", (long long)p.M, p.N, p.K);
    DCPT_CHECK_ARG(p.K % 4 == 0, "gemm_nt: K=%d must be a multiple of 4", p.K);
    DCPT_CHECK_ARG(cdiv64(p.M, 128) * cdiv(p.N, 64) < (1ll << 31), "gemm_nt: grid too large");
    if (aload == A_GATHER) DCPT_CHECK_ARG(p.gC % 4 == 0 && p.K == 4 * p.gC, "gemm_nt: gather needs K == 4*gC, gC %% 4 == 0");
    // algorithmic work of this launch (for the live roofline in bench.py)
    const double mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    double bytes = mk * (aload == A_SG ? 2 : 1) + mn * (epi == E_SGBWD ? 4 : 1) + (double)p.N * p.K;
    if (epi == E_RESID || epi == E_SCATTER_ADD) bytes += mn;
    ProfScope prof(s, PROF_NT + aload * 8 + epi, p.M, p.N, p.K, 2.0 * mn * p.K, bytes * 4.0);
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
#undef CASE
    dcpt_set_error("gemm_nt: unsupported loader/epilogue combination %d/%d", aload, epi);
    return DCPT_ERR_ARG;
}
