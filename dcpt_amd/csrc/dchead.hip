// Degradation-classifier head building blocks (reference basicsr/archs/degrad_classify_arch.py):
//   conv(1x1 | dense 3x3, no bias) -> LayerNorm over channels -> [+shortcut] -> [ReLU]     (:69-103, :227-243)
//   conv1x1 -> MaxPool2d(2,2) -> ReLU                                                       (:596-602)
//   lq_feats + softmax(mixing_weights)[i] * feature                                         (:632-637)
//   mean over H,W -> Linear                                                                 (:639-640)
// The convolutions are MFMA GEMMs (gemm_nt/gemm_tn with the implicit-GEMM 3x3 loader); channels-first
// LayerNorm on an NCHW tensor is a per-pixel row LayerNorm in NHWC, i.e. the ln.hip kernels.
#include "bf16.h"
#include "gemm.h"
#include "kernels.h"
#include "../../include/dcpt_hip.h"

namespace {

// ---- maxpool 2x2 + relu -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_relu_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, int B, int H,
                                                            int W, int C) {
    const int nq = C / 4, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        int64_t t = i / nq;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int64_t b = t / Ho;
        const float* p = z + ((b * H + 2 * ho) * (int64_t)W + 2 * wo) * C + 4 * q;
        const float4 a = ldg4(p), bb = ldg4(p + C), c = ldg4(p + (int64_t)W * C), d = ldg4(p + (int64_t)W * C + C);
        float4 m;
        m.x = fmaxf(fmaxf(fmaxf(a.x, bb.x), fmaxf(c.x, d.x)), 0.f);
        m.y = fmaxf(fmaxf(fmaxf(a.y, bb.y), fmaxf(c.y, d.y)), 0.f);
        m.z = fmaxf(fmaxf(fmaxf(a.z, bb.z), fmaxf(c.z, d.z)), 0.f);
        m.w = fmaxf(fmaxf(fmaxf(a.w, bb.w), fmaxf(c.w, d.w)), 0.f);
        stg4(y + i * 4, m);
    }
}

// dz gets dy at the FIRST maximum of each window in scan order (torch MaxPool2d), if that maximum is > 0
__device__ __forceinline__ void route4(float a, float b, float c, float d, float g, float& oa, float& ob, float& oc, float& od) {
    int idx = 0;
    float m = a;
    if (b > m) { m = b; idx = 1; }
    if (c > m) { m = c; idx = 2; }
    if (d > m) { m = d; idx = 3; }
    const float v = (m > 0.f) ? g : 0.f;
    oa = idx == 0 ? v : 0.f;
    ob = idx == 1 ? v : 0.f;
    oc = idx == 2 ? v : 0.f;
    od = idx == 3 ? v : 0.f;
}

__global__ __launch_bounds__(256) void pool_relu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                            float* __restrict__ dz, int B, int H, int W, int C) {
    const int nq = C / 4, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        int64_t t = i / nq;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int64_t b = t / Ho;
        const int64_t o = ((b * H + 2 * ho) * (int64_t)W + 2 * wo) * C + 4 * q;
        const float4 a = ldg4(z + o), bb = ldg4(z + o + C), c = ldg4(z + o + (int64_t)W * C), d = ldg4(z + o + (int64_t)W * C + C);
        const float4 g = ldg4(dy + i * 4);
        float4 oa, ob, oc, od;
        route4(a.x, bb.x, c.x, d.x, g.x, oa.x, ob.x, oc.x, od.x);
        route4(a.y, bb.y, c.y, d.y, g.y, oa.y, ob.y, oc.y, od.y);
        route4(a.z, bb.z, c.z, d.z, g.z, oa.z, ob.z, oc.z, od.z);
        route4(a.w, bb.w, c.w, d.w, g.w, oa.w, ob.w, oc.w, od.w);
        stg4(dz + o, oa);
        stg4(dz + o + C, ob);
        stg4(dz + o + (int64_t)W * C, oc);
        stg4(dz + o + (int64_t)W * C + C, od);
    }
}

// ---- mixing -------------------------------------------------------------------------------------
// (the mixing step and the mean + Linear are templated on the STORAGE type of the feature maps: float, or bf16_t for the all-bf16 head --
// arithmetic and reductions are fp32 either way, so the bf16 entry points give exactly what the fp32 ones gave behind casts)
template <typename ST>
__device__ __forceinline__ float4 ld4s(const ST* p) {
    if constexpr (sizeof(ST) == 2) return bf4_unpack(*reinterpret_cast<const u32x2*>(p));
    else return ldg4(p);
}
template <typename ST>
__device__ __forceinline__ void st4s(ST* p, float4 v) {
    if constexpr (sizeof(ST) == 2) *reinterpret_cast<u32x2*>(p) = bf4_pack(v);
    else stg4(p, v);
}
__device__ __forceinline__ float softmax_i(const float* w, int n, int i) {
    float m = w[0];
    for (int j = 1; j < n; ++j) m = fmaxf(m, w[j]);
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += expf(w[j] - m);
    return expf(w[i] - m) / s;
}

template <typename ST>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const ST* __restrict__ prev, const ST* __restrict__ feat,
                                                      const float* __restrict__ mw, int n, int idx, ST* __restrict__ out,
                                                      int64_t nq) {
    const float s = softmax_i(mw, n, idx);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
        const float4 f = ld4s(feat + 4 * i);
        float4 o = f4_scale(f, s);
        if (prev) o = f4_add(o, ld4s(prev + 4 * i));
        st4s(out + 4 * i, o);
    }
}

// dfeat = s*dout ; part[block] = sum dout*feat
template <typename ST>
__global__ __launch_bounds__(256) void mix_bwd_kernel(const ST* __restrict__ dout, const ST* __restrict__ feat,
                                                      const float* __restrict__ mw, int n, int idx, ST* __restrict__ dfeat,
                                                      float* __restrict__ part, int64_t nq) {
    __shared__ float red[256];
    const float s = softmax_i(mw, n, idx);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
        const float4 g = ld4s(dout + 4 * i), f = ld4s(feat + 4 * i);
        acc += f4_sum(f4_mul(g, f));
        st4s(dfeat + 4 * i, f4_scale(g, s));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// dmix[j] = ds * s_i * (delta_ij - s_j),  ds = sum_blocks part
__global__ __launch_bounds__(256) void mix_bwd_final_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ mw,
                                                            int n, int idx, float* __restrict__ dmix) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < n) {
        const float ds = red[0];
        const float si = softmax_i(mw, n, idx), sj = softmax_i(mw, n, threadIdx.x);
        dmix[threadIdx.x] = ds * si * ((threadIdx.x == idx ? 1.f : 0.f) - sj);
    }
}

// ---- mean over pixels + linear ------------------------------------------------------------------
// part[b][j][c] = sum over pixel slice j
template <typename ST>
__global__ __launch_bounds__(256) void meanpool_part_kernel(const ST* __restrict__ x, float* __restrict__ part, int C, int P,
                                                            int nsl) {
    __shared__ float4 red[256];
    const int b = blockIdx.z, j = blockIdx.y;
    const int nq = C / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    const int tid = threadIdx.x, ql = tid % qb, pl = tid / qb;
    const int q = blockIdx.x * qb + ql;
    const bool qok = q < nq;
    const int per = (P + nsl - 1) / nsl;
    const int pbeg = j * per;
    int pend = pbeg + per;
    if (pend > P) pend = P;
    float4 acc = f4_zero();
    if (qok)
        for (int px = pbeg + pl; px < pend; px += pb) acc = f4_add(acc, ld4s(x + ((int64_t)b * P + px) * C + 4 * q));
    red[tid] = acc;
    __syncthreads();
    if (pl == 0 && qok) {
        float4 s = red[ql];
        for (int i = 1; i < pb; ++i) s = f4_add(s, red[i * qb + ql]);
        stg4(part + ((int64_t)b * nsl + j) * C + 4 * q, s);
    }
}

// pooled[b][c] = mean; logits[b][n] = sum_c fw[n][c]*pooled[b][c] + fb[n]      one block per image
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ part, int nsl, const float* __restrict__ fw,
                                                     const float* __restrict__ fb, float* __restrict__ pooled,
                                                     float* __restrict__ logits, int C, int NC, float invP) {
    extern __shared__ __attribute__((aligned(16))) float pl[];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < nsl; ++j) s += part[((int64_t)b * nsl + j) * C + c];
        s *= invP;
        pl[c] = s;
        pooled[(int64_t)b * C + c] = s;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int n = wave; n < NC; n += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(fw[(int64_t)n * C + c], pl[c], s);
        s = group_sum(s, 64);
        if (lane == 0) logits[(int64_t)b * NC + n] = s + (fb ? fb[n] : 0.f);
    }
}

// role 0: dpooled[b][c] = sum_n g[b][n]*fw[n][c] (scaled by invP -> dxrow) ; role 1: dfw[n][c] = sum_b g[b][n]*pooled[b][c];
// role 2: dfb[n] = sum_b g[b][n]
__global__ __launch_bounds__(256) void fc_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pooled,
                                                     const float* __restrict__ fw, float* __restrict__ dxrow,
                                                     float* __restrict__ dfw, float* __restrict__ dfb, int B, int C, int NC,
                                                     float invP) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {
        if (i >= (int64_t)B * C) return;
        const int b = (int)(i / C), c = (int)(i % C);
        float s = 0.f;
        for (int n = 0; n < NC; ++n) s = fmaf(g[(int64_t)b * NC + n], fw[(int64_t)n * C + c], s);
        dxrow[i] = s * invP;
    } else if (blockIdx.y == 1) {
        if (i >= (int64_t)NC * C) return;
        const int n = (int)(i / C), c = (int)(i % C);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(g[(int64_t)b * NC + n], pooled[(int64_t)b * C + c], s);
        dfw[i] = s;
    } else {
        if (i >= NC) return;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += g[(int64_t)b * NC + i];
        dfb[i] = s;
    }
}

// dx[b][p][c] = dxrow[b][c]
template <typename ST>
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ dxrow, ST* __restrict__ dx, int B, int P,
                                                         int C) {
    const int nq = C / 4;
    const int64_t total = (int64_t)B * P * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        const int64_t b = i / nq / P;
        st4s(dx + 4 * i, ldg4(dxrow + b * C + 4 * q));
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t nb = cdiv64(n, 256);
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

struct ConvWs {
    float* wp;      // packed / transposed weights
    float* dz;      // [M][Cout]  (backward)
    float* slab;
    float* lnpart;
    int splits;
    int64_t rps;
    int ln_nblk;
};

size_t conv_layout(int B, int H, int W, int Cin, int Cout, int ks, int backward, bool with_ln, void* base, size_t bytes,
                   ConvWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    ConvWs w{};
    const int K = ks * ks * Cin;
    const int64_t M = (int64_t)B * H * W;
    w.wp = a.get<float>((size_t)Cout * K);
    if (backward) {
        w.dz = a.get<float>((size_t)M * Cout);
        gemm_tn_plan(M, Cout, K, &w.splits, &w.rps);
        w.slab = a.get<float>((size_t)w.splits * Cout * K);
        if (with_ln) {
            w.ln_nblk = ln_bwd_num_blocks(M, Cout);
            w.lnpart = a.get<float>((size_t)w.ln_nblk * 3 * Cout);
        }
    }
    if (out) *out = w;
    return a.off;
}

int conv_fwd(const float* x, const float* w, float* z, const ConvWs& cw, int B, int H, int W, int Cin, int Cout, int ks,
             hipStream_t s) {
    GemmNT g{};
    g.M = (int64_t)B * H * W; g.A = x; g.N = Cout; g.C = z; g.ldc = Cout;
    if (ks == 1) {
        g.lda = Cin; g.K = Cin; g.Bw = w;  // [Cout][Cin][1][1] is already [N][K]
        return launch_gemm_nt(g, A_PLAIN, E_PLAIN, s);
    }
    DCPT_TRY(launch_wpack(w, cw.wp, nullptr, Cout, 9 * Cin, WP_CONV3, s));
    g.K = 9 * Cin; g.gH = H; g.gW = W; g.gC = Cin; g.Bw = cw.wp;
    return launch_gemm_nt(g, A_CONV3, E_PLAIN, s);
}

// dx = conv^T(dz), dw = wgrad(dz, x)
int conv_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, const ConvWs& cw, int B, int H, int W, int Cin,
             int Cout, int ks, hipStream_t s, const float* dx_add = nullptr) {
    const int64_t M = (int64_t)B * H * W;
    GemmNT g{};
    g.M = M; g.A = dz; g.N = Cin; g.C = dx; g.ldc = Cin; g.Bw = cw.wp;
    g.res = dx_add; g.ldres = Cin;
    const int EDX = dx_add ? E_RESID : E_PLAIN;   // dx = dx_add + dz W (the shortcut gradient of a bottleneck block rides in the epilogue)
    GemmTN t{};
    t.M = M; t.X = dz; t.ldx = Cout; t.N = Cout; t.Y = x; t.slab = cw.slab; t.colsum = nullptr; t.splits = cw.splits;
    t.rows_per_split = cw.rps;
    if (ks == 1) {
        DCPT_TRY(launch_wpack(w, cw.wp, nullptr, Cout, Cin, WP_TRANSPOSE, s));
        g.lda = Cout; g.K = Cout;
        if (dx) DCPT_TRY(launch_gemm_nt(g, A_PLAIN, EDX, s));
        t.ldy = Cin; t.K = Cin;
        DCPT_TRY(launch_gemm_tn(t, A_PLAIN, A_PLAIN, s));
        DCPT_TRY(launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr,
                                     WR_PLAIN, s));
    } else {
        DCPT_TRY(launch_wpack(w, cw.wp, nullptr, Cout, 9 * Cin, WP_CONV3_T, s));
        g.K = 9 * Cout; g.gH = H; g.gW = W; g.gC = Cout;
        if (dx) DCPT_TRY(launch_gemm_nt(g, A_CONV3, EDX, s));
        t.K = 9 * Cin; t.gH = H; t.gW = W; t.gC = Cin;
        DCPT_TRY(launch_gemm_tn(t, A_PLAIN, A_CONV3, s));
        DCPT_TRY(launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, 9 * Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr,
                                     WR_CONV3, s));
    }
    return DCPT_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" size_t dcpt_conv_ln_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward) {
    return conv_layout(B, H, W, Cin, Cout, ksize, backward, true, nullptr, 0, nullptr);
}

extern "C" int dcpt_conv_ln_fwd(const float* x, const float* w, const float* lnw, const float* lnb, const float* res, int relu,
                                float* z, float* y, float* mu, float* rstd, void* ws, size_t ws_bytes, int B, int H, int W,
                                int Cin, int Cout, int ksize, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && lnw && lnb && z && y && mu && rstd, "conv_ln_fwd: null argument");
    DCPT_CHECK_ARG((ksize == 1 || ksize == 3) && Cin % 4 == 0 && Cout % 4 == 0, "conv_ln_fwd: ksize=%d Cin=%d Cout=%d", ksize, Cin, Cout);
    ConvWs cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 0, true, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_ln_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(conv_fwd(x, w, z, cw, B, H, W, Cin, Cout, ksize, s));
    return launch_ln_act_fwd(z, lnw, lnb, res, relu, y, mu, rstd, (int64_t)B * H * W, Cout, 1e-6f, s);  // eps: degrad_classify_arch.py:24
}

extern "C" int dcpt_conv_ln_bwd_acc(const float* dy, const float* x, const float* w, const float* lnw, const float* z, const float* y,
                                    const float* mu, const float* rstd, const float* dx_add, float* dx, float* dw, float* dlnw, float* dlnb,
                                    float* dres, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu,
                                    dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && lnw && z && mu && rstd && dw && dlnw && dlnb, "conv_ln_bwd: null argument");
    DCPT_CHECK_ARG(!relu || y, "conv_ln_bwd: relu needs the saved output y");
    DCPT_CHECK_ARG(!dx_add || (dx && ksize == 1), "conv_ln_bwd: dx_add needs dx and a 1 x 1 conv (the block's conv1)");
    DCPT_CHECK_ARG((ksize == 1 || ksize == 3) && Cin % 4 == 0 && Cout % 4 == 0, "conv_ln_bwd: bad shape");
    ConvWs cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 1, true, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_ln_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    DCPT_TRY(launch_ln_act_bwd(dy, z, mu, rstd, lnw, nullptr, relu ? y : nullptr, dres, cw.dz, cw.lnpart, cw.ln_nblk, M, Cout, s));
    DCPT_TRY(launch_colpart_reduce(cw.lnpart, cw.ln_nblk, 3, Cout, dlnw, dlnb, nullptr, s));
    return conv_bwd(cw.dz, x, w, dx, dw, cw, B, H, W, Cin, Cout, ksize, s, dx_add);
}

extern "C" int dcpt_conv_ln_bwd(const float* dy, const float* x, const float* w, const float* lnw, const float* z, const float* y,
                                const float* mu, const float* rstd, float* dx, float* dw, float* dlnw, float* dlnb, float* dres,
                                void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu,
                                dcpt_stream_t stream) {
    return dcpt_conv_ln_bwd_acc(dy, x, w, lnw, z, y, mu, rstd, nullptr, dx, dw, dlnw, dlnb, dres, ws, ws_bytes, B, H, W, Cin, Cout, ksize, relu, stream);
}

extern "C" size_t dcpt_conv_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward) {
    return conv_layout(B, H, W, Cin, Cout, ksize, backward, false, nullptr, 0, nullptr);
}

extern "C" int dcpt_conv_fwd(const float* x, const float* w, float* y, void* ws, size_t ws_bytes, int B, int H, int W, int Cin,
                             int Cout, int ksize, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && w && y, "conv_fwd: null argument");
    DCPT_CHECK_ARG((ksize == 1 || ksize == 3) && Cin % 4 == 0 && Cout % 4 == 0, "conv_fwd: ksize=%d Cin=%d Cout=%d", ksize, Cin, Cout);
    ConvWs cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 0, false, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    return conv_fwd(x, w, y, cw, B, H, W, Cin, Cout, ksize, (hipStream_t)stream);
}

extern "C" int dcpt_conv_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, void* ws, size_t ws_bytes,
                             int B, int H, int W, int Cin, int Cout, int ksize, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(dy && x && w && dw, "conv_bwd: null argument");
    DCPT_CHECK_ARG((ksize == 1 || ksize == 3) && Cin % 4 == 0 && Cout % 4 == 0, "conv_bwd: bad shape");
    ConvWs cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 1, false, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    return conv_bwd(dy, x, w, dx, dw, cw, B, H, W, Cin, Cout, ksize, (hipStream_t)stream);
}

extern "C" size_t dcpt_conv1x1_pool_relu_ws_bytes(int B, int H, int W, int Cin, int Cout, int backward) {
    return conv_layout(B, H, W, Cin, Cout, 1, backward, false, nullptr, 0, nullptr);
}

extern "C" int dcpt_conv1x1_pool_relu_fwd(const float* x, const float* w, float* z, float* y, void* ws, size_t ws_bytes, int B,
                                          int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && z && y, "conv1x1_pool_relu_fwd: null argument");
    DCPT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && Cin % 4 == 0 && Cout % 4 == 0, "conv1x1_pool_relu_fwd: bad shape");
    ConvWs cw{};
    DCPT_TRY(conv_fwd(x, w, z, cw, B, H, W, Cin, Cout, 1, s));
    pool_relu_fwd_kernel<<<dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (Cout / 4))), dim3(256), 0, s>>>(z, y, B, H, W, Cout);
    DCPT_CHECK_LAUNCH("pool_relu_fwd");
    return DCPT_OK;
}

extern "C" int dcpt_conv1x1_pool_relu_bwd(const float* dy, const float* x, const float* w, const float* z, float* dx, float* dw,
                                          void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && z && dx && dw, "conv1x1_pool_relu_bwd: null argument");
    ConvWs cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, 1, 1, false, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv1x1_pool_relu_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    pool_relu_bwd_kernel<<<dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (Cout / 4))), dim3(256), 0, s>>>(z, dy, cw.dz, B, H, W, Cout);
    DCPT_CHECK_LAUNCH("pool_relu_bwd");
    return conv_bwd(cw.dz, x, w, dx, dw, cw, B, H, W, Cin, Cout, 1, s);
}

// ---------------------------------------------------------------------------------------------
template <typename ST>
static int mix_fwd_t(const ST* prev, const ST* feat, const float* mixing_weights, int n, int idx, ST* out, int64_t numel, hipStream_t s) {
    DCPT_CHECK_ARG(feat && mixing_weights && out && n >= 1 && n <= 64 && idx >= 0 && idx < n && numel % 4 == 0, "mix_fwd: bad argument");
    mix_fwd_kernel<ST><<<dim3(grid_for(numel / 4)), dim3(256), 0, s>>>(prev, feat, mixing_weights, n, idx, out, numel / 4);
    DCPT_CHECK_LAUNCH("mix_fwd");
    return DCPT_OK;
}
extern "C" int dcpt_mix_fwd(const float* prev, const float* feat, const float* mixing_weights, int n, int idx, float* out,
                            int64_t numel, dcpt_stream_t stream) {
    return mix_fwd_t<float>(prev, feat, mixing_weights, n, idx, out, numel, (hipStream_t)stream);
}
extern "C" int dcpt_mix_fwd_bf16(const uint16_t* prev, const uint16_t* feat, const float* mixing_weights, int n, int idx, uint16_t* out,
                                 int64_t numel, dcpt_stream_t stream) {
    return mix_fwd_t<bf16_t>(prev, feat, mixing_weights, n, idx, out, numel, (hipStream_t)stream);
}

extern "C" size_t dcpt_mix_bwd_ws_bytes(int64_t numel) { return align_up((size_t)grid_for(numel / 4) * sizeof(float), 256); }

template <typename ST>
static int mix_bwd_t(const ST* dout, const ST* feat, const float* mixing_weights, int n, int idx, ST* dfeat, float* dmix, void* ws,
                     size_t ws_bytes, int64_t numel, hipStream_t s) {
    DCPT_CHECK_ARG(dout && feat && mixing_weights && dfeat && dmix && n >= 1 && n <= 64 && numel % 4 == 0, "mix_bwd: bad argument");
    if (ws == nullptr || ws_bytes < dcpt_mix_bwd_ws_bytes(numel)) {
        dcpt_set_error("mix_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const unsigned nb = grid_for(numel / 4);
    mix_bwd_kernel<ST><<<dim3(nb), dim3(256), 0, s>>>(dout, feat, mixing_weights, n, idx, dfeat, (float*)ws, numel / 4);
    DCPT_CHECK_LAUNCH("mix_bwd");
    mix_bwd_final_kernel<<<dim3(1), dim3(256), 0, s>>>((float*)ws, (int)nb, mixing_weights, n, idx, dmix);
    DCPT_CHECK_LAUNCH("mix_bwd_final");
    return DCPT_OK;
}
extern "C" int dcpt_mix_bwd(const float* dout, const float* feat, const float* mixing_weights, int n, int idx, float* dfeat,
                            float* dmix, void* ws, size_t ws_bytes, int64_t numel, dcpt_stream_t stream) {
    return mix_bwd_t<float>(dout, feat, mixing_weights, n, idx, dfeat, dmix, ws, ws_bytes, numel, (hipStream_t)stream);
}
extern "C" int dcpt_mix_bwd_bf16(const uint16_t* dout, const uint16_t* feat, const float* mixing_weights, int n, int idx, uint16_t* dfeat,
                                 float* dmix, void* ws, size_t ws_bytes, int64_t numel, dcpt_stream_t stream) {
    return mix_bwd_t<bf16_t>(dout, feat, mixing_weights, n, idx, dfeat, dmix, ws, ws_bytes, numel, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
static int pool_slices(int P) {
    int n = P / 64;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

extern "C" size_t dcpt_meanpool_fc_ws_bytes(int B, int P, int C) {
    return align_up((size_t)B * pool_slices(P) * C * sizeof(float), 256) + align_up((size_t)B * C * sizeof(float), 256);
}

template <typename ST>
static int meanpool_fc_fwd_t(const ST* x, const float* fw, const float* fb, float* pooled, float* logits, void* ws, size_t ws_bytes, int B,
                             int P, int C, int NC, hipStream_t s) {
    DCPT_CHECK_ARG(x && fw && pooled && logits && C % 4 == 0 && B <= 65535 && C * 4 <= 65536, "meanpool_fc_fwd: bad argument");
    if (ws == nullptr || ws_bytes < dcpt_meanpool_fc_ws_bytes(B, P, C)) {
        dcpt_set_error("meanpool_fc_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int nsl = pool_slices(P), nq = C / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    meanpool_part_kernel<ST><<<dim3(cdiv(nq, qb), nsl, B), dim3(256), 0, s>>>(x, (float*)ws, C, P, nsl);
    DCPT_CHECK_LAUNCH("meanpool_part");
    fc_fwd_kernel<<<dim3(B), dim3(256), C * sizeof(float), s>>>((float*)ws, nsl, fw, fb, pooled, logits, C, NC, 1.0f / (float)P);
    DCPT_CHECK_LAUNCH("fc_fwd");
    return DCPT_OK;
}
extern "C" int dcpt_meanpool_fc_fwd(const float* x, const float* fw, const float* fb, float* pooled, float* logits, void* ws,
                                    size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream) {
    return meanpool_fc_fwd_t<float>(x, fw, fb, pooled, logits, ws, ws_bytes, B, P, C, NC, (hipStream_t)stream);
}
extern "C" int dcpt_meanpool_fc_fwd_bf16(const uint16_t* x, const float* fw, const float* fb, float* pooled, float* logits, void* ws,
                                         size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream) {
    return meanpool_fc_fwd_t<bf16_t>(x, fw, fb, pooled, logits, ws, ws_bytes, B, P, C, NC, (hipStream_t)stream);
}

template <typename ST>
static int meanpool_fc_bwd_t(const float* dlogits, const float* pooled, const float* fw, ST* dx, float* dfw, float* dfb, void* ws,
                             size_t ws_bytes, int B, int P, int C, int NC, hipStream_t s) {
    DCPT_CHECK_ARG(dlogits && pooled && fw && dx && dfw && dfb && C % 4 == 0, "meanpool_fc_bwd: bad argument");
    if (ws == nullptr || ws_bytes < dcpt_meanpool_fc_ws_bytes(B, P, C)) {
        dcpt_set_error("meanpool_fc_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    float* dxrow = (float*)ws;
    const int64_t mx = (int64_t)B * C > (int64_t)NC * C ? (int64_t)B * C : (int64_t)NC * C;
    fc_bwd_kernel<<<dim3((unsigned)cdiv64(mx, 256), 3), dim3(256), 0, s>>>(dlogits, pooled, fw, dxrow, dfw, dfb, B, C, NC, 1.0f / (float)P);
    DCPT_CHECK_LAUNCH("fc_bwd");
    bcast_rows_kernel<ST><<<dim3(grid_for((int64_t)B * P * (C / 4))), dim3(256), 0, s>>>(dxrow, dx, B, P, C);
    DCPT_CHECK_LAUNCH("bcast_rows");
    return DCPT_OK;
}
extern "C" int dcpt_meanpool_fc_bwd(const float* dlogits, const float* pooled, const float* fw, float* dx, float* dfw, float* dfb,
                                    void* ws, size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream) {
    return meanpool_fc_bwd_t<float>(dlogits, pooled, fw, dx, dfw, dfb, ws, ws_bytes, B, P, C, NC, (hipStream_t)stream);
}
extern "C" int dcpt_meanpool_fc_bwd_bf16(const float* dlogits, const float* pooled, const float* fw, uint16_t* dx, float* dfw, float* dfb,
                                         void* ws, size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream) {
    return meanpool_fc_bwd_t<bf16_t>(dlogits, pooled, fw, dx, dfw, dfb, ws, ws_bytes, B, P, C, NC, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Image embedding of PromptIR_DC (reference basicsr/archs/degrad_classify_arch.py:491-494: Conv2d(3, dim, 7, stride 2, pad 3)
// + bias -> LayerNorm): the strided 7x7 over a 3-channel NCHW image is unfolded into patch rows
//   A[m][k],  m = (b, oy, ox),  k = (c * ks + ky) * ks + kx  (the weight's own (Cin, ks, ks) order),
// followed by one column of ones (so the conv bias is one more weight column) and zero padding to a multiple of 4 columns;
// the product runs on the fp32 MFMA GEMM + LayerNorm path of the head (dcpt_conv_ln_*, 1x1 over the patch rows).
namespace {

__global__ __launch_bounds__(256) void patch_unfold_kernel(const float* __restrict__ x, float* __restrict__ A, int B, int Cin, int H, int W,
                                                           int Ho, int Wo, int ks, int stride, int pad, int Kp) {
    const int nq = Kp / 4;
    const int64_t total = (int64_t)B * Ho * Wo * nq;
    const int kreal = Cin * ks * ks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        const int64_t m = i / nq;
        const int ox = (int)(m % Wo);
        const int oy = (int)((m / Wo) % Ho);
        const int b = (int)(m / ((int64_t)Wo * Ho));
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float t = 0.f;
            if (k < kreal) {
                const int kx = k % ks, ky = (k / ks) % ks, c = k / (ks * ks);
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) t = x[(((int64_t)b * Cin + c) * H + iy) * W + ix];
            } else if (k == kreal) {
                t = 1.f;
            }
            v[j] = t;
        }
        stg4(A + m * Kp + 4 * q, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// gradient of the image: every input element gathers the patch entries it was copied to (fixed order, no atomics)
__global__ __launch_bounds__(256) void patch_fold_kernel(const float* __restrict__ dA, float* __restrict__ dx, int B, int Cin, int H, int W,
                                                         int Ho, int Wo, int ks, int stride, int pad, int Kp) {
    const int64_t total = (int64_t)B * Cin * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i % W);
        const int iy = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % Cin);
        const int b = (int)(i / ((int64_t)W * H * Cin));
        float s = 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride != 0 || ty / stride >= Ho) continue;
            const int oy = ty / stride;
            for (int kx = 0; kx < ks; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride != 0 || tx / stride >= Wo) continue;
                const int ox = tx / stride;
                s += dA[(((int64_t)b * Ho + oy) * Wo + ox) * Kp + (c * ks + ky) * ks + kx];
            }
        }
        dx[i] = s;
    }
}

}  // namespace

extern "C" int dcpt_patch_unfold(const float* x, float* A, int B, int Cin, int H, int W, int ksize, int stride, int pad, int Kp,
                                 dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && A && B > 0 && Cin > 0 && H > 0 && W > 0 && ksize > 0 && stride > 0 && pad >= 0, "patch_unfold: bad argument");
    DCPT_CHECK_ARG(Kp % 4 == 0 && Kp >= Cin * ksize * ksize + 1, "patch_unfold: Kp=%d must be a multiple of 4 and >= Cin*k*k + 1", Kp);
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    DCPT_CHECK_ARG(Ho > 0 && Wo > 0, "patch_unfold: image smaller than the kernel");
    patch_unfold_kernel<<<dim3(grid_for((int64_t)B * Ho * Wo * (Kp / 4))), dim3(256), 0, (hipStream_t)stream>>>(x, A, B, Cin, H, W, Ho, Wo, ksize,
                                                                                                          stride, pad, Kp);
    DCPT_CHECK_LAUNCH("patch_unfold");
    return DCPT_OK;
}

extern "C" int dcpt_patch_fold(const float* dA, float* dx, int B, int Cin, int H, int W, int ksize, int stride, int pad, int Kp,
                               dcpt_stream_t stream) {
    DCPT_CHECK_ARG(dA && dx && B > 0 && Cin > 0 && H > 0 && W > 0 && ksize > 0 && stride > 0 && pad >= 0, "patch_fold: bad argument");
    DCPT_CHECK_ARG(Kp % 4 == 0 && Kp >= Cin * ksize * ksize + 1, "patch_fold: Kp=%d must be a multiple of 4 and >= Cin*k*k + 1", Kp);
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    DCPT_CHECK_ARG(Ho > 0 && Wo > 0, "patch_fold: image smaller than the kernel");
    patch_fold_kernel<<<dim3(grid_for((int64_t)B * Cin * H * W)), dim3(256), 0, (hipStream_t)stream>>>(dA, dx, B, Cin, H, W, Ho, Wo, ksize, stride,
                                                                                                   pad, Kp);
    DCPT_CHECK_LAUNCH("patch_fold");
    return DCPT_OK;
}
