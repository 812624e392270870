// Degradation-classifier head with bf16 STORAGE (BASELINE.json configs[2]; reference basicsr/archs/degrad_classify_arch.py):
//   conv(1x1 | dense 3x3, no bias) -> channels-first LayerNorm -> [+shortcut] -> [ReLU]     (:69-103, :227-243)
//   conv1x1 -> MaxPool2d(2,2) -> ReLU                                                       (:596-602)
// with bf16 activations (x, conv output z, LayerNorm output y and their gradients), fp32 parameters / parameter gradients /
// LayerNorm statistics / accumulation -- the same contract as the bf16 NAFBlock (nafblock_bf16.hip).  The convolutions are the
// bf16 MFMA GEMMs of gemm_bf16.hip; the dense 3x3 is an implicit GEMM (A gathered tap by tap by LDS-DMA, zero padding by the
// range check; weight gradient: the transposing TN kernel with the gathered operand).  The mixing step and the mean + Linear at
// the end stay on the fp32 kernels of dchead.hip behind casts (they touch each tensor once).
#include "bf16_ops.h"
#include "../../include/dcpt_hip.h"

namespace {

inline unsigned grid_for(int64_t n) {
    int64_t nb = cdiv64(n, 256);
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

__device__ __forceinline__ float4 ldb4(const bf16_t* p) { return bf4_unpack(*reinterpret_cast<const u32x2*>(p)); }
__device__ __forceinline__ void stb4(bf16_t* p, float4 v) { *reinterpret_cast<u32x2*>(p) = bf4_pack(v); }

__global__ __launch_bounds__(256) void pool_relu_fwd_bf16_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ y, int B, int H, int W, int C) {
    const int nq = C / 4, Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        int64_t t = i / nq;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int64_t b = t / Ho;
        const bf16_t* p = z + ((b * H + 2 * ho) * (int64_t)W + 2 * wo) * C + 4 * q;
        const float4 a = ldb4(p), bb = ldb4(p + C), c = ldb4(p + (int64_t)W * C), d = ldb4(p + (int64_t)W * C + C);
        float4 m;
        m.x = fmaxf(fmaxf(fmaxf(a.x, bb.x), fmaxf(c.x, d.x)), 0.f);
        m.y = fmaxf(fmaxf(fmaxf(a.y, bb.y), fmaxf(c.y, d.y)), 0.f);
        m.z = fmaxf(fmaxf(fmaxf(a.z, bb.z), fmaxf(c.z, d.z)), 0.f);
        m.w = fmaxf(fmaxf(fmaxf(a.w, bb.w), fmaxf(c.w, d.w)), 0.f);
        stb4(y + i * 4, m);
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

__global__ __launch_bounds__(256) void pool_relu_bwd_bf16_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dz,
                                                                 int B, int H, int W, int C) {
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
        const float4 a = ldb4(z + o), bb = ldb4(z + o + C), c = ldb4(z + o + (int64_t)W * C), d = ldb4(z + o + (int64_t)W * C + C);
        const float4 g = ldb4(dy + i * 4);
        float4 ra, rb, rc, rd;
        route4(a.x, bb.x, c.x, d.x, g.x, ra.x, rb.x, rc.x, rd.x);
        route4(a.y, bb.y, c.y, d.y, g.y, ra.y, rb.y, rc.y, rd.y);
        route4(a.z, bb.z, c.z, d.z, g.z, ra.z, rb.z, rc.z, rd.z);
        route4(a.w, bb.w, c.w, d.w, g.w, ra.w, rb.w, rc.w, rd.w);
        stb4(dz + o, ra);
        stb4(dz + o + C, rb);
        stb4(dz + o + (int64_t)W * C, rc);
        stb4(dz + o + (int64_t)W * C + C, rd);
    }
}

struct ConvWsB {
    bf16_t* wp;     // packed / transposed bf16 weights
    bf16_t* dz;     // [M][Cout]  (backward)
    float* slab;
    float* lnpart;
    int splits;
    int64_t rps;
    int ln_nblk;
    bool tn256;     // weight gradient on the 256 x 256-tile kernel + finisher (gemm_tn_bf16_256.hip)
    GemmTNG wg;
};

// the wide layers' weight gradient: N = Cout, K = ks * ks * Cin multiples of 256 (a dense 3 x 3 also needs 128-channel taps)
bool conv_tn256(int Cin, int Cout, int ks) { return gemm_tn_bf16_256_ok(Cout, ks * ks * Cin) && (ks == 1 || Cin % 128 == 0); }

size_t conv_layout(int B, int H, int W, int Cin, int Cout, int ks, int backward, bool with_ln, void* base, size_t bytes, ConvWsB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    ConvWsB w{};
    const int K = ks * ks * Cin;
    const int64_t M = (int64_t)B * H * W;
    w.wp = a.get<bf16_t>((size_t)Cout * K);
    if (backward) {
        w.dz = a.get<bf16_t>((size_t)M * Cout);
        w.tn256 = conv_tn256(Cin, Cout, ks);
        if (w.tn256) {
            w.wg.n = 1;
            TnProb& q = w.wg.p[0];
            q.M = M; q.N = q.ldx = Cout; q.K = K; q.ldy = ks == 1 ? Cin : 0;
            if (ks == 3) {
                q.yconv = 1; q.gH = H; q.gW = W; q.gC = Cin;
            }
            gemm_tn_bf16_256_plan(w.wg);
            q.slab = a.get<float>(gemm_tn_bf16_256_slab_floats(q));
        } else {
            gemm_tn_bf16_plan(M, Cout, K, &w.splits, &w.rps);
            w.slab = a.get<float>((size_t)w.splits * Cout * K);
        }
        if (with_ln) {
            w.ln_nblk = ln_bwd_bf16_num_blocks(M, Cout);
            w.lnpart = a.get<float>((size_t)w.ln_nblk * 2 * Cout);
        }
    }
    if (out) *out = w;
    return a.off;
}

int pack(const float* w, bf16_t* out, int N, int K, int mode, hipStream_t s) {
    WpackBJobs j{};
    j.n = 1;
    j.in[0] = w; j.out[0] = out; j.N[0] = N; j.K[0] = K; j.transpose[0] = mode;
    return launch_wpack_bf16(j, s);
}

// A conv's two operand images as ONE cached buffer (dcpt_conv_wpack_bf16_multi, ABI 14): the forward image [Cout][ks ks Cin] at offset 0, the
// data gradient's (transposed; flipped taps for the dense 3 x 3) at wpack_half_bytes.  `pk` below = that buffer or nullptr (pack in the call).
size_t wpack_half_bytes(int Cin, int Cout, int ks) { return (((size_t)Cout * ks * ks * Cin * sizeof(bf16_t)) + 255) & ~(size_t)255; }
inline const bf16_t* pk_fwd(const void* pk) { return static_cast<const bf16_t*>(pk); }
inline const bf16_t* pk_bwd(const void* pk, int Cin, int Cout, int ks) {
    return reinterpret_cast<const bf16_t*>(static_cast<const char*>(pk) + wpack_half_bytes(Cin, Cout, ks));
}
inline int pk_check(const void* pk, size_t pk_bytes, int Cin, int Cout, int ks, const char* who) {
    DCPT_CHECK_ARG(pk == nullptr || pk_bytes >= 2 * wpack_half_bytes(Cin, Cout, ks), "%s: packed weights too small (dcpt_conv_wpack_bf16_bytes)", who);
    return DCPT_OK;
}

int conv_fwd(const bf16_t* x, const float* w, bf16_t* z, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout, int ks, hipStream_t s,
             const void* pk = nullptr) {
    GemmNTB g{};
    g.M = (int64_t)B * H * W; g.A = x; g.N = Cout; g.C = z; g.ldc = Cout; g.Bw = pk ? pk_fwd(pk) : cw.wp;
    if (ks == 1) {
        if (!pk) DCPT_TRY(pack(w, cw.wp, Cout, Cin, 0, s));
        g.lda = Cin; g.K = Cin;
    } else {
        if (!pk) DCPT_TRY(pack(w, cw.wp, Cout, 9 * Cin, 2, s));
        g.K = 9 * Cin; g.conv3 = 1; g.gH = H; g.gW = W; g.gC = Cin;
    }
    return launch_gemm_nt_bf16(g, EB_PLAIN, s);
}

// dx = conv^T(dz), dw = wgrad(dz, x); `ln` (optional): the LayerNorm's column partials, reduced here as well
int conv_bwd(const bf16_t* dz, const bf16_t* x, const float* w, bf16_t* dx, float* dw, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout,
             int ks, hipStream_t s, const FinCols* ln = nullptr, const bf16_t* dx_add = nullptr, const void* pk = nullptr) {
    const int64_t M = (int64_t)B * H * W;
    GemmNTB g{};
    g.M = M; g.A = dz; g.N = Cin; g.C = dx; g.ldc = Cin; g.Bw = pk ? pk_bwd(pk, Cin, Cout, ks) : cw.wp;
    g.res = dx_add; g.ldres = Cin;
    const int EDX = dx_add ? EB_RESID : EB_PLAIN;   // dx = dx_add + dz W (the shortcut gradient of a bottleneck block rides in the epilogue)
    if (cw.tn256) {   // data gradient as before; the weight gradient as one 256-tile launch + one finisher launch (which also takes the LN sums)
        if (!pk && dx) DCPT_TRY(pack(w, cw.wp, Cout, ks * ks * Cin, ks == 1 ? 1 : 3, s));
        if (ks == 1) {
            g.lda = Cout; g.K = Cout;
        } else {
            g.K = 9 * Cout; g.conv3 = 1; g.gH = H; g.gW = W; g.gC = Cout;
        }
        if (dx) DCPT_TRY(launch_gemm_nt_bf16(g, EDX, s));
        GemmTNG wg = cw.wg;
        wg.p[0].X = dz; wg.p[0].Y = x;
        DCPT_TRY(launch_gemm_tn_bf16_256(wg, s));
        FinJobs f{};
        f.nslab = 1;
        f.slab[0].slab = wg.p[0].slab; f.slab[0].N = Cout; f.slab[0].K = wg.p[0].K; f.slab[0].splits = wg.p[0].slots; f.slab[0].cs_rows = wg.p[0].splits;
        f.slab[0].tiles_k = wg.p[0].tiles_k; f.slab[0].ks_div = 1; f.slab[0].dW = dw; f.slab[0].conv3 = ks == 3;
        if (ln) {
            f.ncols = 1;
            f.cols[0] = *ln;
        }
        return launch_wgrad_finish(f, s);
    }
    if (ln) DCPT_TRY(launch_colpart_reduce(ln->part, ln->R, 2, ln->C, ln->out0, ln->out1, nullptr, s));
    GemmTNB t{};
    t.M = M; t.X = dz; t.ldx = Cout; t.N = Cout; t.Y = x; t.slab = cw.slab; t.colsum = nullptr; t.splits = cw.splits; t.rows_per_split = cw.rps;
    if (ks == 1) {
        if (!pk && dx) DCPT_TRY(pack(w, cw.wp, Cout, Cin, 1, s));
        g.lda = Cout; g.K = Cout;
        if (dx) DCPT_TRY(launch_gemm_nt_bf16(g, EDX, s));
        t.ldy = Cin; t.K = Cin;
        DCPT_TRY(launch_gemm_tn_bf16(t, s));
        return launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_PLAIN, s);
    }
    if (!pk && dx) DCPT_TRY(pack(w, cw.wp, Cout, 9 * Cin, 3, s));
    g.K = 9 * Cout; g.conv3 = 1; g.gH = H; g.gW = W; g.gC = Cout;
    if (dx) DCPT_TRY(launch_gemm_nt_bf16(g, EDX, s));
    t.K = 9 * Cin; t.yconv = 1; t.gH = H; t.gW = W; t.gC = Cin; t.ldy = Cin;
    DCPT_TRY(launch_gemm_tn_bf16(t, s));
    return launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, 9 * Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_CONV3, s);
}

bool conv_shape_ok(int Cin, int Cout, int ks) { return (ks == 1 || ks == 3) && Cin % 8 == 0 && Cout % 8 == 0 && Cout <= 1024; }

}  // namespace

extern "C" size_t dcpt_conv_ln_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward) {
    return conv_layout(B, H, W, Cin, Cout, ksize, backward, true, nullptr, 0, nullptr);
}

extern "C" int dcpt_conv_ln_fwd_bf16_packed(const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes, const float* lnw,
                                            const float* lnb, const uint16_t* res, int relu, uint16_t* z, uint16_t* y, float* mu, float* rstd,
                                            void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && (w || wpacked) && lnw && lnb && z && y && mu && rstd, "conv_ln_fwd_bf16: null argument");
    DCPT_TRY(pk_check(wpacked, wpacked_bytes, Cin, Cout, ksize, "conv_ln_fwd_bf16"));
    DCPT_CHECK_ARG(conv_shape_ok(Cin, Cout, ksize), "conv_ln_fwd_bf16: ksize=%d Cin=%d Cout=%d (channels %% 8 == 0, Cout <= 1024)", ksize, Cin, Cout);
    ConvWsB cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 0, true, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_ln_fwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(conv_fwd(x, w, z, cw, B, H, W, Cin, Cout, ksize, s, wpacked));
    return launch_ln_act_fwd_bf16(z, lnw, lnb, res, relu, y, mu, rstd, (int64_t)B * H * W, Cout, 1e-6f, s);   // eps: degrad_classify_arch.py:24
}

extern "C" int dcpt_conv_ln_fwd_bf16(const uint16_t* x, const float* w, const float* lnw, const float* lnb, const uint16_t* res, int relu,
                                     uint16_t* z, uint16_t* y, float* mu, float* rstd, void* ws, size_t ws_bytes, int B, int H, int W, int Cin,
                                     int Cout, int ksize, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(w, "conv_ln_fwd_bf16: null argument");
    return dcpt_conv_ln_fwd_bf16_packed(x, w, nullptr, 0, lnw, lnb, res, relu, z, y, mu, rstd, ws, ws_bytes, B, H, W, Cin, Cout, ksize, stream);
}

extern "C" int dcpt_conv_ln_bwd_acc_bf16_packed(const uint16_t* dy, const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes,
                                                const float* lnw, const uint16_t* z, const uint16_t* y, const float* mu, const float* rstd,
                                                const uint16_t* dx_add, uint16_t* dx, float* dw, float* dlnw, float* dlnb, uint16_t* dres, void* ws,
                                                size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && (w || wpacked) && lnw && z && mu && rstd && dw && dlnw && dlnb, "conv_ln_bwd_bf16: null argument");
    DCPT_TRY(pk_check(wpacked, wpacked_bytes, Cin, Cout, ksize, "conv_ln_bwd_bf16"));
    DCPT_CHECK_ARG(!relu || y, "conv_ln_bwd_bf16: relu needs the saved output y");
    DCPT_CHECK_ARG(!dx_add || (dx && ksize == 1), "conv_ln_bwd_bf16: dx_add needs dx and a 1 x 1 conv (the block's conv1)");
    DCPT_CHECK_ARG(conv_shape_ok(Cin, Cout, ksize), "conv_ln_bwd_bf16: bad shape");
    ConvWsB cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, ksize, 1, true, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv_ln_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    DCPT_TRY(launch_ln_act_bwd_bf16(dy, z, mu, rstd, lnw, relu ? y : nullptr, dres, cw.dz, cw.lnpart, cw.ln_nblk, M, Cout, s));
    const FinCols ln{cw.lnpart, dlnw, dlnb, cw.ln_nblk, 2, Cout, 0};
    return conv_bwd(cw.dz, x, w, dx, dw, cw, B, H, W, Cin, Cout, ksize, s, &ln, dx_add, wpacked);
}

extern "C" int dcpt_conv_ln_bwd_acc_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const float* lnw, const uint16_t* z, const uint16_t* y,
                                         const float* mu, const float* rstd, const uint16_t* dx_add, uint16_t* dx, float* dw, float* dlnw, float* dlnb,
                                         uint16_t* dres, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu,
                                         dcpt_stream_t stream) {
    DCPT_CHECK_ARG(w, "conv_ln_bwd_bf16: null argument");
    return dcpt_conv_ln_bwd_acc_bf16_packed(dy, x, w, nullptr, 0, lnw, z, y, mu, rstd, dx_add, dx, dw, dlnw, dlnb, dres, ws, ws_bytes, B, H, W, Cin,
                                            Cout, ksize, relu, stream);
}

extern "C" int dcpt_conv_ln_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const float* lnw, const uint16_t* z, const uint16_t* y,
                                     const float* mu, const float* rstd, uint16_t* dx, float* dw, float* dlnw, float* dlnb, uint16_t* dres,
                                     void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream) {
    return dcpt_conv_ln_bwd_acc_bf16(dy, x, w, lnw, z, y, mu, rstd, nullptr, dx, dw, dlnw, dlnb, dres, ws, ws_bytes, B, H, W, Cin, Cout, ksize, relu, stream);
}

extern "C" size_t dcpt_conv1x1_pool_relu_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int backward) {
    return conv_layout(B, H, W, Cin, Cout, 1, backward, false, nullptr, 0, nullptr);
}

extern "C" int dcpt_conv1x1_pool_relu_fwd_bf16_packed(const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes, uint16_t* z,
                                                      uint16_t* y, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout,
                                                      dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && (w || wpacked) && z && y, "conv1x1_pool_relu_fwd_bf16: null argument");
    DCPT_TRY(pk_check(wpacked, wpacked_bytes, Cin, Cout, 1, "conv1x1_pool_relu_fwd_bf16"));
    DCPT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && conv_shape_ok(Cin, Cout, 1), "conv1x1_pool_relu_fwd_bf16: bad shape");
    ConvWsB cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, 1, 0, false, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv1x1_pool_relu_fwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(conv_fwd(x, w, z, cw, B, H, W, Cin, Cout, 1, s, wpacked));
    pool_relu_fwd_bf16_kernel<<<dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (Cout / 4))), dim3(256), 0, s>>>(z, y, B, H, W, Cout);
    DCPT_CHECK_LAUNCH("pool_relu_fwd_bf16");
    return DCPT_OK;
}

extern "C" int dcpt_conv1x1_pool_relu_fwd_bf16(const uint16_t* x, const float* w, uint16_t* z, uint16_t* y, void* ws, size_t ws_bytes, int B, int H,
                                               int W, int Cin, int Cout, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(w, "conv1x1_pool_relu_fwd_bf16: null argument");
    return dcpt_conv1x1_pool_relu_fwd_bf16_packed(x, w, nullptr, 0, z, y, ws, ws_bytes, B, H, W, Cin, Cout, stream);
}

extern "C" int dcpt_conv1x1_pool_relu_bwd_bf16_packed(const uint16_t* dy, const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes,
                                                      const uint16_t* z, uint16_t* dx, float* dw, void* ws, size_t ws_bytes, int B, int H, int W,
                                                      int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && (w || wpacked) && z && dx && dw, "conv1x1_pool_relu_bwd_bf16: null argument");
    DCPT_TRY(pk_check(wpacked, wpacked_bytes, Cin, Cout, 1, "conv1x1_pool_relu_bwd_bf16"));
    DCPT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && conv_shape_ok(Cin, Cout, 1), "conv1x1_pool_relu_bwd_bf16: bad shape");
    ConvWsB cw;
    const size_t need = conv_layout(B, H, W, Cin, Cout, 1, 1, false, ws, ws_bytes, &cw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("conv1x1_pool_relu_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    pool_relu_bwd_bf16_kernel<<<dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (Cout / 4))), dim3(256), 0, s>>>(z, dy, cw.dz, B, H, W, Cout);
    DCPT_CHECK_LAUNCH("pool_relu_bwd_bf16");
    return conv_bwd(cw.dz, x, w, dx, dw, cw, B, H, W, Cin, Cout, 1, s, nullptr, nullptr, wpacked);
}

extern "C" int dcpt_conv1x1_pool_relu_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const uint16_t* z, uint16_t* dx, float* dw,
                                               void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(w, "conv1x1_pool_relu_bwd_bf16: null argument");
    return dcpt_conv1x1_pool_relu_bwd_bf16_packed(dy, x, w, nullptr, 0, z, dx, dw, ws, ws_bytes, B, H, W, Cin, Cout, stream);
}

// ---- the two operand images of n convs in a few launches (ABI 14) ----
// One WpackBJobsL launch holds 40 convs (80 jobs); the dense 3 x 3 convs go first in launches of their own with a wider grid (their largest has
// 16 384 gather tiles), the 1 x 1 convs share a 256-wide one.  Bit-identical to the packs the unpacked entry points make per call.
extern "C" size_t dcpt_conv_wpack_bf16_bytes(int Cin, int Cout, int ksize) { return 2 * wpack_half_bytes(Cin, Cout, ksize); }

extern "C" int dcpt_conv_wpack_bf16_multi(const float* const* w, void* const* packed, const size_t* packed_bytes, const int* Cin, const int* Cout,
                                          const int* ksize, int n, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(w && packed && packed_bytes && Cin && Cout && ksize && n >= 1, "conv_wpack_bf16_multi: null argument");
    for (int i = 0; i < n; ++i) {
        DCPT_CHECK_ARG(w[i] && packed[i], "conv_wpack_bf16_multi: null argument (conv %d)", i);
        DCPT_CHECK_ARG(conv_shape_ok(Cin[i], Cout[i], ksize[i]), "conv_wpack_bf16_multi: conv %d: ksize=%d Cin=%d Cout=%d", i, ksize[i], Cin[i], Cout[i]);
        DCPT_CHECK_ARG(packed_bytes[i] >= 2 * wpack_half_bytes(Cin[i], Cout[i], ksize[i]), "conv_wpack_bf16_multi: conv %d: buffer too small", i);
    }
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: the 3 x 3 convs, pass 1: the 1 x 1 convs
        WpackBJobsL j{};
        for (int i = 0; i < n; ++i) {
            if ((ksize[i] == 3) != (pass == 0)) continue;
            const int K = ksize[i] * ksize[i] * Cin[i];
            for (int f = 0; f < 2; ++f) {
                j.in[j.n] = w[i];
                j.out[j.n] = f == 0 ? const_cast<bf16_t*>(pk_fwd(packed[i])) : const_cast<bf16_t*>(pk_bwd(packed[i], Cin[i], Cout[i], ksize[i]));
                j.N[j.n] = Cout[i]; j.K[j.n] = K;
                j.transpose[j.n] = (ksize[i] == 1 ? 0 : 2) + f;
                ++j.n;
            }
            if (j.n == WPACKB_MAX_JOBS_L) {
                DCPT_TRY(launch_wpack_bf16(j, s, pass == 0 ? 1024 : 256));
                j = WpackBJobsL{};
            }
        }
        if (j.n) DCPT_TRY(launch_wpack_bf16(j, s, pass == 0 ? 1024 : 256));
    }
    return DCPT_OK;
}
