// Depthwise 3x3 (zero pad 1) + bias + SimpleGate on NHWC, forward and backward
// (reference basicsr/archs/nafnet_arch.py:96-104 conv2, :77-80 SimpleGate, :171-172).
//
// Thread = one channel quad (float4, coalesced along C) x one pixel column; it walks a band of RH
// rows, streaming each input row once and scattering it into three running output-row accumulators
// (kernel rows 2/1/0), so the only re-reads are the left/right neighbours (L1/L2 hits) and the two
// halo rows per band.  A block is QB quads x PB columns (QB*PB = 256) and loops over several
// (band, column-chunk) items of one image so that per-channel sums (pooling for SCA, depthwise
// weight gradients) are reduced in registers, then once per block in LDS in a fixed order, and
// written as per-block partials: deterministic, no float atomics.
#include "kernels.h"

namespace {

constexpr int RH = 8;  // rows per band

struct DwMap {
    int QW;   // quads handled per pixel
    int QB;   // quads per block (power of two)
    int PB;   // columns per block
    int nqc;  // quad chunks
    int nwc;  // column chunks
    int nbands;
    int items;  // per image
};

__host__ __device__ inline DwMap dw_map(int H, int W, int quads) {
    DwMap m;
    m.QW = quads;
    // at most 32 quads (512 B of channels) per block row, so a block spans >= 8 pixel columns and the
    // left/right halo columns are mostly served by the same CU's L1 instead of another XCD's HBM fetch
    int qb = 1;
    while (qb < quads && qb < 32) qb <<= 1;
    m.QB = qb;
    m.PB = 256 / qb;
    m.nqc = (quads + qb - 1) / qb;
    m.nwc = (W + m.PB - 1) / m.PB;
    m.nbands = (H + RH - 1) / RH;
    m.items = m.nwc * m.nbands;
    return m;
}

struct DwP {
    const float* in0;   // fwd/bwd_a: t1 [M][2C];  bwd_b: da [M][2C]
    const float* in1;   // bwd_a: dts [M][C];      bwd_b: t1 [M][2C]
    const float* w2p;   // [9][2C]
    const float* b2;    // [2C]
    const float* simg;  // bwd_a: [B][C]
    const float* dpool; // bwd_a: [B][C]
    float* out;         // fwd: t2 [M][C]; bwd_a: da [M][2C]; bwd_b: dt1 [M][2C]
    float* part;        // fwd: pool_part [B][NBLK][C]; bwd_b: wpart [B*NBLK][10][2C]
    int B, H, W, C;
    int Ctot;  // bwd_b / plain: total channel count of the depthwise conv
};

__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.39894228040143268f * expf(-0.5f * a * a);
}
__device__ __forceinline__ float4 gelu4(float4 a) { return make_float4(gelu_f(a.x), gelu_f(a.y), gelu_f(a.z), gelu_f(a.w)); }
__device__ __forceinline__ float4 gelu_d4(float4 a) { return make_float4(gelu_df(a.x), gelu_df(a.y), gelu_df(a.z), gelu_df(a.w)); }

__device__ __forceinline__ float4 ld_or_zero(const float* p, bool ok) { return ok ? ldg4(p) : f4_zero(); }

// MODE 0: forward (t2 + pool partials);  MODE 1: backward-a (da)
// GATE 0: SimpleGate g1*g2 (NAFNet);  GATE 1: gelu(g1)*g2 (Restormer GDFN, exact erf GELU)
template <int MODE, int GATE>
__global__ __launch_bounds__(256) void dw_gate_kernel(const DwP p) {
    __shared__ float4 red[256];
    const DwMap mp = dw_map(p.H, p.W, p.C / 4);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = blockIdx.x * mp.QB + ql;
    const int b = blockIdx.z;
    const bool qok = q < mp.QW;
    const int C = p.C, C2 = 2 * p.C;
    const int c1 = 4 * q, c2 = C + 4 * q;

    float4 w1[9], w2[9], bias1 = f4_zero(), bias2 = f4_zero();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w1[t] = ld_or_zero(p.w2p + t * C2 + c1, qok);
        w2[t] = ld_or_zero(p.w2p + t * C2 + c2, qok);
    }
    if (qok && p.b2) {
        bias1 = ldg4(p.b2 + c1);
        bias2 = ldg4(p.b2 + c2);
    }
    float4 sv = make_float4(1.f, 1.f, 1.f, 1.f), dpv = f4_zero();
    if (MODE == 1 && qok && p.simg) {
        sv = ldg4(p.simg + (int64_t)b * C + c1);
        dpv = ldg4(p.dpool + (int64_t)b * C + c1);
    }
    float4 pool = f4_zero();

    for (int item = blockIdx.y; item < mp.items; item += gridDim.y) {
        const int band = item / mp.nwc, wc = item % mp.nwc;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = band * RH;
        const int h1 = (h0 + RH < p.H) ? h0 + RH : p.H;
        float4 a0_1 = f4_zero(), a0_2 = f4_zero(), a1_1 = f4_zero(), a1_2 = f4_zero();
        for (int r = h0 - 1; r <= h1; ++r) {
            const bool rin = ok && r >= 0 && r < p.H;
            const int64_t rowbase = ((int64_t)b * p.H + r) * p.W;
            float4 xl1, xc1, xr1, xl2, xc2, xr2;
            {
                const float* pc = p.in0 + (rowbase + x) * C2;
                const bool okl = rin && x > 0, okr = rin && x + 1 < p.W;
                xl1 = ld_or_zero(pc - C2 + c1, okl);
                xl2 = ld_or_zero(pc - C2 + c2, okl);
                xc1 = ld_or_zero(pc + c1, rin);
                xc2 = ld_or_zero(pc + c2, rin);
                xr1 = ld_or_zero(pc + C2 + c1, okr);
                xr2 = ld_or_zero(pc + C2 + c2, okr);
            }
            // kernel row 2 completes output row r-1
            a0_1 = f4_fma(w1[6], xl1, f4_fma(w1[7], xc1, f4_fma(w1[8], xr1, a0_1)));
            a0_2 = f4_fma(w2[6], xl2, f4_fma(w2[7], xc2, f4_fma(w2[8], xr2, a0_2)));
            // kernel row 1 -> output row r
            a1_1 = f4_fma(w1[3], xl1, f4_fma(w1[4], xc1, f4_fma(w1[5], xr1, a1_1)));
            a1_2 = f4_fma(w2[3], xl2, f4_fma(w2[4], xc2, f4_fma(w2[5], xr2, a1_2)));
            // kernel row 0 starts output row r+1
            const float4 a2_1 = f4_fma(w1[0], xl1, f4_fma(w1[1], xc1, f4_mul(w1[2], xr1)));
            const float4 a2_2 = f4_fma(w2[0], xl2, f4_fma(w2[1], xc2, f4_mul(w2[2], xr2)));
            const int y = r - 1;
            if (ok && y >= h0) {
                const float4 g1 = f4_add(a0_1, bias1), g2 = f4_add(a0_2, bias2);
                const int64_t pix = ((int64_t)b * p.H + y) * p.W + x;
                if (MODE == 0) {
                    const float4 t = f4_mul(GATE == 0 ? g1 : gelu4(g1), g2);
                    stg4(p.out + pix * C + c1, t);
                    pool = f4_add(pool, t);
                } else {
                    const float4 dts = ldg4(p.in1 + pix * C + c1);
                    const float4 dt2 = f4_fma(dts, sv, dpv);
                    if (GATE == 0) {
                        stg4(p.out + pix * C2 + c1, f4_mul(dt2, g2));
                        stg4(p.out + pix * C2 + c2, f4_mul(dt2, g1));
                    } else {
                        stg4(p.out + pix * C2 + c1, f4_mul(f4_mul(dt2, g2), gelu_d4(g1)));
                        stg4(p.out + pix * C2 + c2, f4_mul(dt2, gelu4(g1)));
                    }
                }
            }
            a0_1 = a1_1; a0_2 = a1_2;
            a1_1 = a2_1; a1_2 = a2_2;
        }
    }
    if (MODE == 0 && p.part != nullptr) {
        red[tid] = pool;
        __syncthreads();
        if (pl == 0 && qok) {
            float4 s = red[ql];
            for (int j = 1; j < mp.PB; ++j) s = f4_add(s, red[j * mp.QB + ql]);
            stg4(p.part + ((int64_t)b * gridDim.y + blockIdx.y) * C + c1, s);
        }
    }
}

// backward-b: dt1 = dw3x3^T(da), plus per-block partial sums of dw2[ch][tap] and db2[ch].
// Thread = one quad of the 2C channels.
__global__ __launch_bounds__(256) void dw_bwd_b_kernel(const DwP p) {
    __shared__ float4 red[256];
    const int C2 = p.Ctot;
    const DwMap mp = dw_map(p.H, p.W, C2 / 4);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = blockIdx.x * mp.QB + ql;
    const int b = blockIdx.z;
    const bool qok = q < mp.QW;
    const int c0 = 4 * q;
    float4 w[9], wa[10];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = ld_or_zero(p.w2p + t * C2 + c0, qok);
#pragma unroll
    for (int t = 0; t < 10; ++t) wa[t] = f4_zero();

    for (int item = blockIdx.y; item < mp.items; item += gridDim.y) {
        const int band = item / mp.nwc, wc = item % mp.nwc;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = band * RH;
        const int h1 = (h0 + RH < p.H) ? h0 + RH : p.H;
        float4 a0 = f4_zero(), a1 = f4_zero();
        // t1 rows r-1, r at column x (for the weight gradient); row r+1 is loaded in the loop
        float4 tm = f4_zero(), tc = f4_zero();
        {
            const int r = h0 - 1;
            if (ok && r - 1 >= 0) tm = ldg4(p.in1 + (((int64_t)b * p.H + r - 1) * p.W + x) * C2 + c0);
            if (ok && r >= 0) tc = ldg4(p.in1 + (((int64_t)b * p.H + r) * p.W + x) * C2 + c0);
        }
        for (int r = h0 - 1; r <= h1; ++r) {
            const bool rin = ok && r >= 0 && r < p.H;
            const float* pc = p.in0 + (((int64_t)b * p.H + r) * p.W + x) * C2 + c0;
            const float4 dl = ld_or_zero(pc - C2, rin && x > 0);
            const float4 dc = ld_or_zero(pc, rin);
            const float4 dr = ld_or_zero(pc + C2, rin && x + 1 < p.W);
            const bool tpin = ok && r + 1 >= 0 && r + 1 < p.H;
            const float4 tp = ld_or_zero(p.in1 + (((int64_t)b * p.H + r + 1) * p.W + x) * C2 + c0, tpin);
            // dt1[y][x] = sum_{ky,kx} da[y+1-ky][x+1-kx] * w[ky][kx];  da row r feeds y = r-1+ky
            // d{l,c,r} = da[r][x-1+j], j=0,1,2  <->  kx = 2-j
            a0 = f4_fma(w[0 * 3 + 2], dl, f4_fma(w[0 * 3 + 1], dc, f4_fma(w[0 * 3 + 0], dr, a0)));  // ky=0 -> y=r-1
            a1 = f4_fma(w[1 * 3 + 2], dl, f4_fma(w[1 * 3 + 1], dc, f4_fma(w[1 * 3 + 0], dr, a1)));  // ky=1 -> y=r
            const float4 a2 = f4_fma(w[2 * 3 + 2], dl, f4_fma(w[2 * 3 + 1], dc, f4_mul(w[2 * 3 + 0], dr)));  // ky=2 -> y=r+1
            const int y = r - 1;
            if (ok && y >= h0) stg4(p.out + (((int64_t)b * p.H + y) * p.W + x) * C2 + c0, a0);
            a0 = a1;
            a1 = a2;
            // weight gradient: each da row counted once (rows of this band only)
            if (rin && r >= h0 && r < h1) {
                // dw[ky][kx=2-j] += da[r][x-1+j] * t1[r-1+ky][x]
                wa[0 * 3 + 2] = f4_fma(dl, tm, wa[0 * 3 + 2]);
                wa[0 * 3 + 1] = f4_fma(dc, tm, wa[0 * 3 + 1]);
                wa[0 * 3 + 0] = f4_fma(dr, tm, wa[0 * 3 + 0]);
                wa[1 * 3 + 2] = f4_fma(dl, tc, wa[1 * 3 + 2]);
                wa[1 * 3 + 1] = f4_fma(dc, tc, wa[1 * 3 + 1]);
                wa[1 * 3 + 0] = f4_fma(dr, tc, wa[1 * 3 + 0]);
                wa[2 * 3 + 2] = f4_fma(dl, tp, wa[2 * 3 + 2]);
                wa[2 * 3 + 1] = f4_fma(dc, tp, wa[2 * 3 + 1]);
                wa[2 * 3 + 0] = f4_fma(dr, tp, wa[2 * 3 + 0]);
                wa[9] = f4_add(wa[9], dc);
            }
            tm = tc;
            tc = tp;
        }
    }
    float* part = p.part + ((int64_t)b * gridDim.y + blockIdx.y) * 10 * C2;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        __syncthreads();
        red[tid] = wa[t];
        __syncthreads();
        if (pl == 0 && qok) {
            float4 s = red[ql];
            for (int j = 1; j < mp.PB; ++j) s = f4_add(s, red[j * mp.QB + ql]);
            stg4(part + t * C2 + c0, s);
        }
    }
}

// Plain depthwise 3x3 (no bias, no gate) over Ctot channels (Restormer MDTA qkv_dwconv), plus per-block partial
// sums of out^2 for the first nsq channels (the L2 norms of q and k over the pixels).
__global__ __launch_bounds__(256) void dw_plain_kernel(const DwP p, int nsq) {
    __shared__ float4 red[256];
    const int C = p.Ctot;
    const DwMap mp = dw_map(p.H, p.W, C / 4);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = blockIdx.x * mp.QB + ql;
    const int b = blockIdx.z;
    const bool qok = q < mp.QW;
    const int c0 = 4 * q;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = ld_or_zero(p.w2p + t * C + c0, qok);
    float4 sq = f4_zero();
    for (int item = blockIdx.y; item < mp.items; item += gridDim.y) {
        const int band = item / mp.nwc, wc = item % mp.nwc;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = band * RH;
        const int h1 = (h0 + RH < p.H) ? h0 + RH : p.H;
        float4 a0 = f4_zero(), a1 = f4_zero();
        for (int r = h0 - 1; r <= h1; ++r) {
            const bool rin = ok && r >= 0 && r < p.H;
            const float* pc = p.in0 + (((int64_t)b * p.H + r) * p.W + x) * C + c0;
            const float4 xl = ld_or_zero(pc - C, rin && x > 0);
            const float4 xc = ld_or_zero(pc, rin);
            const float4 xr = ld_or_zero(pc + C, rin && x + 1 < p.W);
            a0 = f4_fma(w[6], xl, f4_fma(w[7], xc, f4_fma(w[8], xr, a0)));
            a1 = f4_fma(w[3], xl, f4_fma(w[4], xc, f4_fma(w[5], xr, a1)));
            const float4 a2 = f4_fma(w[0], xl, f4_fma(w[1], xc, f4_mul(w[2], xr)));
            const int y = r - 1;
            if (ok && y >= h0) {
                stg4(p.out + (((int64_t)b * p.H + y) * p.W + x) * C + c0, a0);
                sq = f4_fma(a0, a0, sq);
            }
            a0 = a1;
            a1 = a2;
        }
    }
    if (p.part != nullptr) {
        red[tid] = sq;
        __syncthreads();
        if (pl == 0 && qok && c0 < nsq) {
            float4 s = red[ql];
            for (int j = 1; j < mp.PB; ++j) s = f4_add(s, red[j * mp.QB + ql]);
            stg4(p.part + ((int64_t)b * gridDim.y + blockIdx.y) * nsq + c0, s);
        }
    }
}

__global__ void dw_pack_kernel(const float* __restrict__ w2, float* __restrict__ w2p, int C2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C2 * 9) {
        const int ch = i / 9, t = i % 9;
        w2p[t * C2 + ch] = w2[i];
    }
}

// dw2[ch*9+tap] = sum_r wpart[r][tap][ch];  db2[ch] = sum_r wpart[r][9][ch]
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ wpart, int R, int C2,
                                                              float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float red[8][32];
    const int t = blockIdx.y;
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C2)
        for (int r = rg; r < R; r += 8) s += wpart[((int64_t)r * 10 + t) * C2 + c];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C2) {
        float v = red[0][cl];
#pragma unroll
        for (int i = 1; i < 8; ++i) v += red[i][cl];
        if (t < 9) dw2[c * 9 + t] = v;
        else db2[c] = v;
    }
}

int nblk_for(const DwGeom& g, int quads) {
    const DwMap mp = dw_map(g.H, g.W, quads);
    int64_t want = cdiv64(1024, (int64_t)g.B * mp.nqc);
    if (want > mp.items) want = mp.items;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

int dw_num_blocks_per_image(const DwGeom& g) { return nblk_for(g, g.C / 4); }
int dw_num_blocks_per_image_b(const DwGeom& g) { return nblk_for(g, g.C / 2); }

int launch_dw_pack_weights(const float* w2, float* w2p, int C2, hipStream_t s) {
    dw_pack_kernel<<<dim3(cdiv(C2 * 9, 256)), dim3(256), 0, s>>>(w2, w2p, C2);
    DCPT_CHECK_LAUNCH("dw_pack");
    return DCPT_OK;
}

int launch_dw_fwd(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g,
                  hipStream_t s) {
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_fwd: C=%d must be a multiple of 4, B<=65535", g.C);
    DwP p{};
    p.in0 = t1; p.w2p = w2p; p.b2 = b2; p.out = t2; p.part = pool_part;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const DwMap mp = dw_map(g.H, g.W, g.C / 4);
    dw_gate_kernel<0, 0><<<dim3(mp.nqc, dw_num_blocks_per_image(g), g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_fwd");
    return DCPT_OK;
}

int launch_dw_bwd_a(const float* dts, const float* t1, const float* w2p, const float* b2, const float* simg,
                    const float* dpool, float* da, const DwGeom& g, hipStream_t s) {
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_bwd_a: C=%d must be a multiple of 4", g.C);
    DwP p{};
    p.in0 = t1; p.in1 = dts; p.w2p = w2p; p.b2 = b2; p.simg = simg; p.dpool = dpool; p.out = da;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const DwMap mp = dw_map(g.H, g.W, g.C / 4);
    dw_gate_kernel<1, 0><<<dim3(mp.nqc, dw_num_blocks_per_image(g), g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_bwd_a");
    return DCPT_OK;
}

int launch_dw_bwd_b(const float* da, const float* t1, const float* w2p, float* dt1, float* wpart, const DwGeom& g,
                    hipStream_t s) {
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_bwd_b: C=%d must be a multiple of 4", g.C);
    DwP p{};
    p.in0 = da; p.in1 = t1; p.w2p = w2p; p.out = dt1; p.part = wpart;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C; p.Ctot = 2 * g.C;
    const DwMap mp = dw_map(g.H, g.W, g.C / 2);
    dw_bwd_b_kernel<<<dim3(mp.nqc, dw_num_blocks_per_image_b(g), g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_bwd_b");
    return DCPT_OK;
}

// ---- generic entry points used by the Restormer blocks ---------------------------------------------
int dw_num_blocks_generic(int B, int H, int W, int Ctot) {
    DwGeom g{B, H, W, Ctot};
    return nblk_for(g, Ctot / 4);
}

int launch_dw_gelu_fwd(const float* u, const float* w2p, float* t, int B, int H, int W, int Ch, hipStream_t s) {
    DCPT_CHECK_ARG(Ch % 4 == 0 && B <= 65535, "dw_gelu_fwd: Ch=%d must be a multiple of 4", Ch);
    DwP p{};
    p.in0 = u; p.w2p = w2p; p.out = t; p.B = B; p.H = H; p.W = W; p.C = Ch;
    DwGeom g{B, H, W, Ch};
    const DwMap mp = dw_map(H, W, Ch / 4);
    dw_gate_kernel<0, 1><<<dim3(mp.nqc, dw_num_blocks_per_image(g), B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_gelu_fwd");
    return DCPT_OK;
}

int launch_dw_gelu_bwd_a(const float* dt, const float* u, const float* w2p, float* da, int B, int H, int W, int Ch, hipStream_t s) {
    DCPT_CHECK_ARG(Ch % 4 == 0 && B <= 65535, "dw_gelu_bwd_a: Ch=%d must be a multiple of 4", Ch);
    DwP p{};
    p.in0 = u; p.in1 = dt; p.w2p = w2p; p.out = da; p.B = B; p.H = H; p.W = W; p.C = Ch;
    DwGeom g{B, H, W, Ch};
    const DwMap mp = dw_map(H, W, Ch / 4);
    dw_gate_kernel<1, 1><<<dim3(mp.nqc, dw_num_blocks_per_image(g), B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_gelu_bwd_a");
    return DCPT_OK;
}

int launch_dw_plain_fwd(const float* x, const float* w2p, float* y, float* sq_part, int nsq, int B, int H, int W, int Ctot,
                        hipStream_t s) {
    DCPT_CHECK_ARG(Ctot % 4 == 0 && nsq % 4 == 0 && B <= 65535, "dw_plain_fwd: Ctot=%d", Ctot);
    DwP p{};
    p.in0 = x; p.w2p = w2p; p.out = y; p.part = sq_part; p.B = B; p.H = H; p.W = W; p.Ctot = Ctot;
    const DwMap mp = dw_map(H, W, Ctot / 4);
    dw_plain_kernel<<<dim3(mp.nqc, dw_num_blocks_generic(B, H, W, Ctot), B), dim3(256), 0, s>>>(p, nsq);
    DCPT_CHECK_LAUNCH("dw_plain_fwd");
    return DCPT_OK;
}

// dx = dw^T(dy) over Ctot channels, wpart[B*nblk][10][Ctot] (nblk = dw_num_blocks_generic)
int launch_dw_generic_bwd(const float* dy, const float* x, const float* w2p, float* dx, float* wpart, int B, int H, int W, int Ctot,
                          hipStream_t s) {
    DCPT_CHECK_ARG(Ctot % 4 == 0 && B <= 65535, "dw_generic_bwd: Ctot=%d", Ctot);
    DwP p{};
    p.in0 = dy; p.in1 = x; p.w2p = w2p; p.out = dx; p.part = wpart; p.B = B; p.H = H; p.W = W; p.Ctot = Ctot;
    const DwMap mp = dw_map(H, W, Ctot / 4);
    dw_bwd_b_kernel<<<dim3(mp.nqc, dw_num_blocks_generic(B, H, W, Ctot), B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_generic_bwd");
    return DCPT_OK;
}

int launch_dw_wgrad_reduce(const float* wpart, int R, int C2, float* dw2, float* db2, hipStream_t s) {
    dw_wgrad_reduce_kernel<<<dim3(cdiv(C2, 32), 10), dim3(256), 0, s>>>(wpart, R, C2, dw2, db2);
    DCPT_CHECK_LAUNCH("dw_wgrad_reduce");
    return DCPT_OK;
}
