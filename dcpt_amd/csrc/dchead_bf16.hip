// Degradation-classifier head with bf16 STORAGE (BASELINE.json configs[2]; reference basicsr/archs/degrad_classify_arch.py):
//   conv(1x1 | dense 3x3, no bias) -> channels-first LayerNorm -> [+shortcut] -> [ReLU]     (:69-103, :227-243)
//   conv1x1 -> MaxPool2d(2,2) -> ReLU                                                       (:596-602)
// with bf16 activations (x, conv output z, LayerNorm output y and their gradients), fp32 parameters / parameter gradients /
// LayerNorm statistics / accumulation -- the same contract as the bf16 NAFBlock (nafblock_bf16.hip).  The convolutions are the
// bf16 MFMA GEMMs of gemm_bf16.hip; the dense 3x3 is an implicit GEMM (A gathered tap by tap by LDS-DMA, zero padding by the
// range check; weight gradient: the transposing TN kernel with the gathered operand).  The mixing step and the mean + Linear at
// the end stay on the fp32 kernels of dchead.hip behind casts (they touch each tensor once).
#include "bf16_ops.h"
#include "prof.h"
#include "side.h"
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
    int ln_nblk;     // blocks of the stand-alone LayerNorm backward (rows of lnpart it writes)
    int ln_tiles;    // 128-row tiles of the GEMM whose epilogue does that LayerNorm backward instead (rows of lnpart it writes)
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
            w.ln_tiles = (int)cdiv64(M, 128);
            w.lnpart = a.get<float>((size_t)(w.ln_nblk > w.ln_tiles ? w.ln_nblk : w.ln_tiles) * 2 * Cout);
        }
    }
    if (out) *out = w;
    return a.off;
}

int pack(const float* w, bf16_t* out, int N, int K, int mode, hipStream_t s) {
    trace_tag("head.wpack_per_call");   // (not reached with the per-step cached images: dcpt_conv_wpack_bf16_multi)
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

// the forward GEMM of a conv (weights packed here unless the caller holds the cached images)
int conv_fwd_problem(GemmNTB& g, const bf16_t* x, const float* w, bf16_t* z, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout, int ks,
                     hipStream_t s, const void* pk) {
    g = GemmNTB{};
    g.M = (int64_t)B * H * W; g.A = x; g.N = Cout; g.C = z; g.ldc = Cout; g.Bw = pk ? pk_fwd(pk) : cw.wp;
    if (ks == 1) {
        if (!pk) DCPT_TRY(pack(w, cw.wp, Cout, Cin, 0, s));
        g.lda = Cin; g.K = Cin;
    } else {
        if (!pk) DCPT_TRY(pack(w, cw.wp, Cout, 9 * Cin, 2, s));
        g.K = 9 * Cin; g.conv3 = 1; g.gH = H; g.gW = W; g.gC = Cin;
    }
    return DCPT_OK;
}

int conv_fwd(const bf16_t* x, const float* w, bf16_t* z, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout, int ks, hipStream_t s,
             const void* pk = nullptr) {
    GemmNTB g;
    DCPT_TRY(conv_fwd_problem(g, x, w, z, cw, B, H, W, Cin, Cout, ks, s, pk));
    return launch_gemm_nt_bf16(g, EB_PLAIN, s);
}

// LayerNorm inside the conv GEMMs' epilogues (EB_LNFWD / EB_LNBWDM, bf16.h) wherever a row fits one column tile; 0 in diagnostic builds restores
// the GEMM + stand-alone LayerNorm pair everywhere (bit-identical forward; the backward differs by the order of the LN parameter sums)
bool ln_epi_on() {
    static const int on = dcpt_tuning("DCPT_HEAD_LN_EPI", 1);
    return on != 0;
}

// conv -> channels-first LayerNorm -> [+res] -> [ReLU] of one group: z (conv output), y, mu / rstd written
int conv_ln_fwd_group(const bf16_t* x, const float* w, const void* pk, const float* lnw, const float* lnb, const bf16_t* res, int relu, bf16_t* z,
                      bf16_t* y, float* mu, float* rstd, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout, int ks, hipStream_t s) {
    const int64_t M = (int64_t)B * H * W;
    GemmNTB g;
    DCPT_TRY(conv_fwd_problem(g, x, w, z, cw, B, H, W, Cin, Cout, ks, s, pk));
    if (ln_epi_on() && gemm_nt_bf16_ln_epi_ok(M, Cout, g.K, g.conv3, g.gC)) {
        trace_tag(ks == 3 ? "head.conv3x3+ln_fwd_epilogue" : "head.conv1x1+ln_fwd_epilogue");
        g.y2 = y; g.lnw = lnw; g.lnb = lnb; g.res = res; g.relu = relu; g.eps = 1e-6f; g.mu_out = mu; g.rstd_out = rstd;   // eps: degrad_classify_arch.py:24
        return launch_gemm_nt_bf16(g, EB_LNFWD, s);
    }
    trace_tag(ks == 3 ? "head.conv3x3,ln_fwd_kernel" : "head.conv1x1,ln_fwd_kernel");
    DCPT_TRY(launch_gemm_nt_bf16(g, EB_PLAIN, s));
    return launch_ln_act_fwd_bf16(z, lnw, lnb, res, relu, y, mu, rstd, M, Cout, 1e-6f, s);
}

// The LayerNorm of the group BELOW a conv (the one whose output the conv reads): its masked backward can ride in the epilogue of this conv's
// data-gradient GEMM -- dz_below = LN'(relu'(conv^T(dz))) -- instead of a pass of its own over the gradient just written.
struct LnBelow {
    const bf16_t* z;       // its input (the conv output below); the ReLU mask is recomputed from it (lnb: the LayerNorm's bias; null: no ReLU)
    const float *mu, *rstd, *lnw, *lnb;
    bf16_t* dz;            // out: the gradient of z
    float* colpart;        // out: [cdiv(M, 128)][2][C] partial sums of the LN weight / bias gradients
};

// dx = conv^T(dz), dw = wgrad(dz, x); `ln` (optional): the LayerNorm's column partials, reduced here as well
bool conv_bwd_below_ok(int64_t M, int Cin, int Cout, int ks) {   // may the LayerNorm below this conv ride in its data-gradient epilogue?
    return ln_epi_on() && gemm_nt_bf16_ln_epi_ok(M, Cin, ks * ks * Cout, ks == 3, Cout);
}

int conv_bwd(const bf16_t* dz, const bf16_t* x, const float* w, bf16_t* dx, float* dw, const ConvWsB& cw, int B, int H, int W, int Cin, int Cout,
             int ks, hipStream_t s, const FinCols* ln = nullptr, const bf16_t* dx_add = nullptr, const void* pk = nullptr,
             const LnBelow* below = nullptr, hipStream_t sw = nullptr) {
    // sw: the stream of the weight-gradient GEMM and the parameter-gradient reductions (a side stream forked by the caller once dz and the
    // LayerNorm partials are ready; default: s)
    if (!sw) sw = s;
    const int64_t M = (int64_t)B * H * W;
    GemmNTB g{};
    g.M = M; g.A = dz; g.N = Cin; g.C = dx; g.ldc = Cin; g.Bw = pk ? pk_bwd(pk, Cin, Cout, ks) : cw.wp;
    g.res = dx_add; g.ldres = Cin;
    int EDX = dx_add ? EB_RESID : EB_PLAIN;   // dx = dx_add + dz W (the shortcut gradient of a bottleneck block rides in the epilogue)
    if (below) {   // the data gradient never reaches memory: the epilogue turns it into the gradient of the conv output below
        DCPT_CHECK_ARG(!dx_add && !dx && below->z && below->mu && below->rstd && below->lnw && below->dz && below->colpart,
                       "conv_bwd: LayerNorm-below epilogue: null argument");
        trace_tag(ks == 3 ? "head.conv3x3_dgrad+ln_bwd_epilogue" : "head.conv1x1_dgrad+ln_bwd_epilogue");
        dx = below->dz;   // (so that the `if (dx)` launches below run)
        g.C = below->dz; g.aux = below->z; g.relu = below->lnb != nullptr; g.lnb = below->lnb; g.mu = below->mu; g.rstd = below->rstd; g.lnw = below->lnw; g.colpart = below->colpart;
        EDX = EB_LNBWDM;
    }
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
        DCPT_TRY(launch_gemm_tn_bf16_256(wg, sw));
        FinJobs f{};
        f.nslab = 1;
        f.slab[0].slab = wg.p[0].slab; f.slab[0].N = Cout; f.slab[0].K = wg.p[0].K; f.slab[0].splits = wg.p[0].slots; f.slab[0].cs_rows = wg.p[0].splits;
        f.slab[0].tiles_k = wg.p[0].tiles_k; f.slab[0].ks_div = 1; f.slab[0].dW = dw; f.slab[0].conv3 = ks == 3;
        if (ln) {
            f.ncols = 1;
            f.cols[0] = *ln;
        }
        return launch_wgrad_finish(f, sw);
    }
    if (ln) DCPT_TRY(launch_colpart_reduce(ln->part, ln->R, 2, ln->C, ln->out0, ln->out1, nullptr, sw));
    GemmTNB t{};
    t.M = M; t.X = dz; t.ldx = Cout; t.N = Cout; t.Y = x; t.slab = cw.slab; t.colsum = nullptr; t.splits = cw.splits; t.rows_per_split = cw.rps;
    if (ks == 1) {
        if (!pk && dx) DCPT_TRY(pack(w, cw.wp, Cout, Cin, 1, s));
        g.lda = Cout; g.K = Cout;
        if (dx) DCPT_TRY(launch_gemm_nt_bf16(g, EDX, s));
        t.ldy = Cin; t.K = Cin;
        DCPT_TRY(launch_gemm_tn_bf16(t, sw));
        return launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_PLAIN, sw);
    }
    if (!pk && dx) DCPT_TRY(pack(w, cw.wp, Cout, 9 * Cin, 3, s));
    g.K = 9 * Cout; g.conv3 = 1; g.gH = H; g.gW = W; g.gC = Cout;
    if (dx) DCPT_TRY(launch_gemm_nt_bf16(g, EDX, s));
    t.K = 9 * Cin; t.yconv = 1; t.gH = H; t.gW = W; t.gC = Cin; t.ldy = Cin;
    DCPT_TRY(launch_gemm_tn_bf16(t, sw));
    return launch_wgrad_reduce(cw.slab, nullptr, cw.splits, 0, Cout, 9 * Cin, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_CONV3, sw);
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
    return conv_ln_fwd_group(x, w, wpacked, lnw, lnb, res, relu, z, y, mu, rstd, cw, B, H, W, Cin, Cout, ksize, s);
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

// ---- the whole BottleneckBlock in one call (ABI 15) -------------------------------------------------------------------------------------
// relu(LN(conv1 1x1 C -> 2C)) -> relu(LN(conv2 3x3 2C -> 2C)) -> relu(LN(conv3 1x1 2C -> C) + x)   (degrad_classify_arch.py:132-243, identity
// shortcut).  Forward: every LayerNorm in its conv's epilogue where the row fits a column tile.  Backward: the block's last LayerNorm takes its
// gradient from outside and stays a kernel; the two inner ones ride in the epilogues of the data-gradient GEMMs above them (conv3^T, conv2^T), so
// the gradients of y2 and y1 never reach memory; the shortcut's gradient rides in conv1^T's epilogue as before.
namespace {

struct BneckGeom {
    int Cin[3], Cout[3], ks[3];
};
inline BneckGeom bneck_geom(int C) { return BneckGeom{{C, 2 * C, 2 * C}, {2 * C, 2 * C, C}, {1, 3, 1}}; }

struct BneckWs {
    ConvWsB cw[3];
    bf16_t* dshort;   // [M][C]   the masked output gradient = the shortcut's gradient
    bf16_t* dybuf;    // [M][2C]  a data gradient that had to be written (its LayerNorm is not in the epilogue)
};

size_t bneck_layout(int B, int H, int W, int C, int backward, void* base, size_t bytes, BneckWs* out) {
    const BneckGeom gm = bneck_geom(C);
    const int64_t M = (int64_t)B * H * W;
    BneckWs w{};
    size_t off = 0;
    for (int k = 0; k < 3; ++k) {
        char* b = base ? static_cast<char*>(base) + off : nullptr;
        const size_t left = base ? (bytes > off ? bytes - off : 0) : 0;
        off += align_up(conv_layout(B, H, W, gm.Cin[k], gm.Cout[k], gm.ks[k], backward, true, b, left, &w.cw[k]), 256);
    }
    if (backward) {
        WsAlloc a(base ? static_cast<char*>(base) + off : nullptr, base ? (bytes > off ? bytes - off : 0) : (size_t)-1);
        w.dshort = a.get<bf16_t>((size_t)M * C);
        const bool all_fused = conv_bwd_below_ok(M, gm.Cin[2], gm.Cout[2], 1) && conv_bwd_below_ok(M, gm.Cin[1], gm.Cout[1], 3);
        if (!all_fused) w.dybuf = a.get<bf16_t>((size_t)M * 2 * C);
        off += a.off;
    }
    if (out) *out = w;
    return off;
}

int bneck_check(const dcpt_bneck_group_t* g, int C, bool backward, const char* who) {
    const BneckGeom gm = bneck_geom(C);
    DCPT_CHECK_ARG(g && C >= 8 && C % 8 == 0 && 2 * C <= 1024, "%s: C=%d (a multiple of 8, at most 512)", who, C);
    for (int k = 0; k < 3; ++k) {
        DCPT_CHECK_ARG((g[k].w || g[k].wpacked) && g[k].lnw && g[k].z && g[k].y && g[k].mu && g[k].rstd && (g[k].lnb || (backward && k == 2)), "%s: null argument (group %d)", who, k);
        DCPT_CHECK_ARG(!backward || (g[k].dw && g[k].dlnw && g[k].dlnb), "%s: null gradient output (group %d)", who, k);
        DCPT_TRY(pk_check(g[k].wpacked, g[k].wpacked_bytes, gm.Cin[k], gm.Cout[k], gm.ks[k], who));
    }
    return DCPT_OK;
}

}  // namespace

extern "C" size_t dcpt_bottleneck_bf16_ws_bytes(int B, int H, int W, int C, int backward) { return bneck_layout(B, H, W, C, backward, nullptr, 0, nullptr); }

extern "C" int dcpt_bottleneck_fwd_bf16(const uint16_t* x, const dcpt_bneck_group_t* g, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                                        dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && B > 0 && H > 0 && W > 0, "bottleneck_fwd_bf16: null argument");
    DCPT_TRY(bneck_check(g, C, false, "bottleneck_fwd_bf16"));
    BneckWs bw;
    const size_t need = bneck_layout(B, H, W, C, 0, ws, ws_bytes, &bw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("bottleneck_fwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const BneckGeom gm = bneck_geom(C);
    const bf16_t* cur = x;
    for (int k = 0; k < 3; ++k) {
        DCPT_TRY(conv_ln_fwd_group(cur, g[k].w, g[k].wpacked, g[k].lnw, g[k].lnb, k == 2 ? x : nullptr, 1, g[k].z, g[k].y, g[k].mu, g[k].rstd, bw.cw[k], B,
                                   H, W, gm.Cin[k], gm.Cout[k], gm.ks[k], s));
        cur = g[k].y;
    }
    return DCPT_OK;
}

extern "C" int dcpt_bottleneck_bwd_bf16(const uint16_t* dout, const uint16_t* x, const dcpt_bneck_group_t* g, uint16_t* dx, void* ws, size_t ws_bytes,
                                        int B, int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dout && x && dx && B > 0 && H > 0 && W > 0, "bottleneck_bwd_bf16: null argument");
    DCPT_TRY(bneck_check(g, C, true, "bottleneck_bwd_bf16"));
    BneckWs bw;
    const size_t need = bneck_layout(B, H, W, C, 1, ws, ws_bytes, &bw);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("bottleneck_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const BneckGeom gm = bneck_geom(C);
    const int64_t M = (int64_t)B * H * W;
    // the block's last LayerNorm: gradient from outside, masked by the block's output; the masked gradient is the shortcut's
    trace_tag("head.ln_bwd_kernel");
    DCPT_TRY(launch_ln_act_bwd_bf16(dout, g[2].z, g[2].mu, g[2].rstd, g[2].lnw, g[2].y, bw.dshort, bw.cw[2].dz, bw.cw[2].lnpart, bw.cw[2].ln_nblk, M, C, s));
    int ln_rows = bw.cw[2].ln_nblk;   // rows of the current group's lnpart
    // The three weight-gradient GEMMs and their reductions are off the chain dout -> dx and CAN run on the side stream (side.hip), forked as each
    // dz is ready -- measured and left off: head alone 27.7 / 28.1 ms with, 28.1 / 28.1 without; DCPT step 91.5 / 92.3 with, 92.0 / 92.0 without
    // (profiles/r6/head_ln_epilogues/).  Every GEMM here owns whole CUs (one 256-row tile = 128-160 KB of LDS), so blocks of the two streams take
    // turns on a CU instead of sharing it: the kernels overlap in time (sum of durations 43 ms in a 29-ms span) and slow each other down by the
    // same amount.
    static const int use_side = dcpt_tuning("DCPT_HEAD_SIDE", 0);
    Side* sd = use_side ? side_for(s) : nullptr;
    hipStream_t sw = side_stream(sd, s);
    for (int k = 2; k >= 0; --k) {
        const ConvWsB& cw = bw.cw[k];
        DCPT_TRY(side_fork(sd, 2 - k, s));   // dz of group k and its LayerNorm partials are ready
        const FinCols ln{cw.lnpart, g[k].dlnw, g[k].dlnb, ln_rows, 2, gm.Cout[k], 0};
        const bf16_t* xin = k == 0 ? x : g[k - 1].y;
        if (k == 0) {
            DCPT_TRY(conv_bwd(cw.dz, xin, g[k].w, dx, g[k].dw, cw, B, H, W, gm.Cin[k], gm.Cout[k], gm.ks[k], s, &ln, bw.dshort, g[k].wpacked, nullptr, sw));
        } else if (conv_bwd_below_ok(M, gm.Cin[k], gm.Cout[k], gm.ks[k])) {
            const LnBelow lb{g[k - 1].z, g[k - 1].mu, g[k - 1].rstd, g[k - 1].lnw, g[k - 1].lnb, bw.cw[k - 1].dz, bw.cw[k - 1].lnpart};
            DCPT_TRY(conv_bwd(cw.dz, xin, g[k].w, nullptr, g[k].dw, cw, B, H, W, gm.Cin[k], gm.Cout[k], gm.ks[k], s, &ln, nullptr, g[k].wpacked, &lb, sw));
            ln_rows = bw.cw[k - 1].ln_tiles;
        } else {
            DCPT_TRY(conv_bwd(cw.dz, xin, g[k].w, bw.dybuf, g[k].dw, cw, B, H, W, gm.Cin[k], gm.Cout[k], gm.ks[k], s, &ln, nullptr, g[k].wpacked, nullptr, sw));
            trace_tag("head.ln_bwd_kernel");
            DCPT_TRY(launch_ln_act_bwd_bf16(bw.dybuf, g[k - 1].z, g[k - 1].mu, g[k - 1].rstd, g[k - 1].lnw, g[k - 1].y, nullptr, bw.cw[k - 1].dz,
                                            bw.cw[k - 1].lnpart, bw.cw[k - 1].ln_nblk, M, gm.Cout[k - 1], s));
            ln_rows = bw.cw[k - 1].ln_nblk;
        }
    }
    return side_join(sd, s);   // the caller's stream continues only after every parameter gradient is written
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
    trace_tag("head.wpack_multi");
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
