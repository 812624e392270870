// bf16-storage forms of the layers BETWEEN the NAFBlock groups (reference basicsr/archs/nafnet_arch.py:202-219 intro / ending,
// :230 down = Conv2d(C, 2C, 2, 2), :238-242 up = Conv2d(C, 2C, 1, bias=False) + PixelShuffle(2), :264-265 skip add), so that a
// NAFNet with act_dtype = "bf16" keeps every feature map in bf16 from the intro conv's output to the ending conv's input: no cast
// kernels, no fp32 gather / scatter GEMMs.  Same algorithms as the fp32 entry points in capi.hip (the 2x2 convolution and the
// pixel shuffle are address maps of the GEMM operand loaders / epilogues, never a materialised tensor); activations and their
// gradients bf16 (rounded once, on store), weights / biases and their gradients fp32, accumulation fp32.
#include "bf16.h"
#include "bf16_ops.h"
#include "../../include/dcpt_hip.h"
#include "kernels.h"

namespace {

struct EdgeWsB {
    bf16_t* wp;      // packed bf16 weight of the launch
    float* slab;
    float* colsum;
    int splits;
    int64_t rps;
};

// down: weight [2C][C][2][2]; forward pack [2C][4C], backward pack [4C][2C]; wgrad slabs [splits][2C][4C]
size_t down_layout_b(int B, int H, int W, int C, int backward, void* base, size_t bytes, EdgeWsB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    EdgeWsB w{};
    w.wp = a.get<bf16_t>((size_t)8 * C * C);
    if (backward) {
        const int64_t Mc = (int64_t)B * (H / 2) * (W / 2);
        gemm_tn_bf16_plan(Mc, 2 * C, 4 * C, &w.splits, &w.rps);
        w.slab = a.get<float>((size_t)w.splits * 8 * C * C);
        w.colsum = a.get<float>((size_t)w.splits * gemm_tn_bf16_tiles_k(2 * C, 4 * C) * 2 * C);
    }
    if (out) *out = w;
    return a.off;
}

// up: weight [2C][C]; forward pack [2C][C] (rows permuted to (i, j)-major), backward pack [C][2C]; wgrad slabs [splits][2C][C]
size_t up_layout_b(int B, int H, int W, int C, int backward, void* base, size_t bytes, EdgeWsB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    EdgeWsB w{};
    w.wp = a.get<bf16_t>((size_t)2 * C * C);
    if (backward) {
        gemm_tn_bf16_plan((int64_t)B * H * W, 2 * C, C, &w.splits, &w.rps);
        w.slab = a.get<float>((size_t)w.splits * 2 * C * C);
    }
    if (out) *out = w;
    return a.off;
}

int pack1(const float* w, bf16_t* out, int N, int K, int mode, hipStream_t s) {
    WpackBJobs j{};
    j.in[0] = w; j.out[0] = out; j.N[0] = N; j.K[0] = K; j.nimg[0] = 1; j.transpose[0] = mode; j.n = 1;
    return launch_wpack_bf16(j, s);
}

}  // namespace

// ---- down: x [B][H][W][C] -> y [B][H/2][W/2][2C] ------------------------------------------------------------------------------
extern "C" size_t dcpt_down2x2_bf16_ws_bytes(int B, int H, int W, int C, int backward) {
    return down_layout_b(B, H, W, C, backward, nullptr, 0, nullptr);
}

extern "C" int dcpt_down2x2_fwd_bf16(const uint16_t* x, const float* w, const float* bias, uint16_t* y, void* ws, size_t ws_bytes, int B, int H,
                                     int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && y, "down2x2_fwd_bf16: null argument");
    DCPT_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "down2x2_fwd_bf16: H=%d W=%d must be even, C=%d %% 8", H, W, C);
    EdgeWsB d;
    const size_t need = down_layout_b(B, H, W, C, 0, ws, ws_bytes, &d);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("down2x2_fwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(pack1(w, d.wp, 2 * C, 4 * C, 4, s));
    GemmNTB g{};
    g.M = (int64_t)B * (H / 2) * (W / 2);
    g.A = x; g.K = 4 * C; g.gather2 = 1; g.gH = H / 2; g.gW = W / 2; g.gC = C;
    g.Bw = d.wp; g.N = 2 * C; g.C = y; g.ldc = 2 * C; g.bias = bias;
    return launch_gemm_nt_bf16(g, EB_BIAS, s);
}

extern "C" int dcpt_down2x2_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, float* dbias, void* ws,
                                     size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    return dcpt_down2x2_bwd_acc_bf16(dy, x, w, nullptr, dx, dw, dbias, ws, ws_bytes, B, H, W, C, stream);
}

extern "C" int dcpt_down2x2_bwd_acc_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const uint16_t* dx_add, uint16_t* dx, float* dw,
                                         float* dbias, void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw && dbias, "down2x2_bwd_bf16: null argument");
    DCPT_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "down2x2_bwd_bf16: bad shape");
    EdgeWsB d;
    const size_t need = down_layout_b(B, H, W, C, 1, ws, ws_bytes, &d);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("down2x2_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t Mc = (int64_t)B * (H / 2) * (W / 2);
    // dx (fine) = scatter(dy [Mc][2C] x Wp^T):  Bw [N = 4C][K = 2C]
    DCPT_TRY(pack1(w, d.wp, 2 * C, 4 * C, 5, s));
    GemmNTB g{};
    g.M = Mc; g.A = dy; g.lda = 2 * C; g.K = 2 * C; g.Bw = d.wp; g.N = 4 * C; g.C = dx; g.ldc = 4 * C;
    g.gH = H / 2; g.gW = W / 2; g.gC = C;
    g.res = dx_add;   // (the skip connection's gradient joins in the scatter epilogue)
    DCPT_TRY(launch_gemm_nt_bf16(g, dx_add ? EB_SCATTER_ADD : EB_SCATTER, s));
    // dW packed [2C][4C] = sum_m dy[m][oc] * gather(x)[m][k'],  db = column sums of dy
    GemmTNB t{};
    t.M = Mc; t.X = dy; t.ldx = 2 * C; t.N = 2 * C; t.Y = x; t.ldy = 4 * C; t.K = 4 * C; t.yg2 = 1; t.gH = H / 2; t.gW = W / 2; t.gC = C;
    t.slab = d.slab; t.colsum = d.colsum; t.splits = d.splits; t.rows_per_split = d.rps;
    DCPT_TRY(launch_gemm_tn_bf16(t, s));
    return launch_wgrad_reduce(d.slab, d.colsum, d.splits, d.splits * gemm_tn_bf16_tiles_k(2 * C, 4 * C), 2 * C, 4 * C, nullptr, nullptr, nullptr, dw,
                               nullptr, dbias, WR_DOWN, s);
}

// ---- up: x [B][H][W][C] -> y [B][2H][2W][C/2] = PixelShuffle2(conv1x1(x)) + skip ---------------------------------------------
extern "C" size_t dcpt_up_ps_bf16_ws_bytes(int B, int H, int W, int C, int backward) { return up_layout_b(B, H, W, C, backward, nullptr, 0, nullptr); }

extern "C" int dcpt_up_ps_fwd_bf16(const uint16_t* x, const float* w, const uint16_t* skip, uint16_t* y, void* ws, size_t ws_bytes, int B, int H,
                                   int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && y, "up_ps_fwd_bf16: null argument");
    DCPT_CHECK_ARG(B > 0 && H > 0 && W > 0 && C % 16 == 0, "up_ps_fwd_bf16: C=%d must be a multiple of 16", C);
    EdgeWsB u;
    const size_t need = up_layout_b(B, H, W, C, 0, ws, ws_bytes, &u);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("up_ps_fwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(pack1(w, u.wp, 2 * C, C, 6, s));
    GemmNTB g{};
    g.M = (int64_t)B * H * W; g.A = x; g.lda = C; g.K = C; g.Bw = u.wp; g.N = 2 * C; g.C = y; g.ldc = 2 * C;
    g.gH = H; g.gW = W; g.gC = C / 2; g.res = skip;
    return launch_gemm_nt_bf16(g, skip ? EB_SCATTER_ADD : EB_SCATTER, s);
}

extern "C" int dcpt_up_ps_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, void* ws, size_t ws_bytes, int B,
                                   int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw, "up_ps_bwd_bf16: null argument");
    DCPT_CHECK_ARG(B > 0 && H > 0 && W > 0 && C % 16 == 0, "up_ps_bwd_bf16: C=%d must be a multiple of 16", C);
    EdgeWsB u;
    const size_t need = up_layout_b(B, H, W, C, 1, ws, ws_bytes, &u);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("up_ps_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    // dx[m][ic] = sum_n' gather(dy)[m][n'] * Wp[n'][ic]:  Bw [N = C][K = 2C] = Wp^T
    DCPT_TRY(pack1(w, u.wp, 2 * C, C, 7, s));
    GemmNTB g{};
    g.M = M; g.A = dy; g.K = 2 * C; g.gather2 = 1; g.gH = H; g.gW = W; g.gC = C / 2; g.Bw = u.wp; g.N = C; g.C = dx; g.ldc = C;
    DCPT_TRY(launch_gemm_nt_bf16(g, EB_PLAIN, s));
    // dW packed [2C][C] = sum_m gather(dy)[m][n'] * x[m][ic]
    GemmTNB t{};
    t.M = M; t.X = dy; t.ldx = 2 * C; t.N = 2 * C; t.xg2 = 1; t.Y = x; t.ldy = C; t.K = C; t.gH = H; t.gW = W; t.gC = C / 2;
    t.slab = u.slab; t.colsum = nullptr; t.splits = u.splits; t.rows_per_split = u.rps;
    DCPT_TRY(launch_gemm_tn_bf16(t, s));
    return launch_wgrad_reduce(u.slab, nullptr, u.splits, 0, 2 * C, C, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_UP, s);
}

// ---- intro-type conv: image NCHW fp32 (Cin <= 4) -> features NHWC bf16 ---------------------------------------------------------
extern "C" int dcpt_conv3x3_in_fwd_bf16(const float* x, const float* w, const float* bias, uint16_t* y, int B, int H, int W, int Cin, int Cout,
                                        dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && w && y, "conv3x3_in_fwd_bf16: null argument");
    return launch_conv3x3_s2b_bf16(x, w, bias, y, B, H, W, Cin, Cout, 0, (hipStream_t)stream);
}

extern "C" int dcpt_conv3x3_in_bwd_bf16(const uint16_t* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws,
                                        size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dw && dbias, "conv3x3_in_bwd_bf16: null argument");
    if (ws == nullptr || ws_bytes < dcpt_conv3x3_in_bwd_ws_bytes(B, H, W, Cin, Cout)) {
        dcpt_set_error("conv3x3_in_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    const int nblk = conv3x3_wgrad_num_blocks(B, H, W, Cout);
    DCPT_TRY(launch_conv3x3_wgrad_bf16(dy, x, (float*)ws, nblk, dw, dbias, B, H, W, Cin, Cout, 0, s));
    if (dx) DCPT_TRY(launch_conv3x3_b2s_bf16(dy, w, nullptr, nullptr, dx, B, H, W, Cin, Cout, 1, s));
    return DCPT_OK;
}

// ---- ending-type conv: features NHWC bf16 -> image NCHW fp32 (Cout <= 4) (+ residual image) ------------------------------------
extern "C" int dcpt_conv3x3_out_fwd_bf16(const uint16_t* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W,
                                         int Cin, int Cout, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && w && y, "conv3x3_out_fwd_bf16: null argument");
    return launch_conv3x3_b2s_bf16(x, w, bias, res, y, B, H, W, Cout, Cin, 0, (hipStream_t)stream);
}

extern "C" int dcpt_conv3x3_out_bwd_bf16(const float* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, float* dbias, void* ws,
                                         size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw && dbias, "conv3x3_out_bwd_bf16: null argument");
    if (ws == nullptr || ws_bytes < dcpt_conv3x3_out_bwd_ws_bytes(B, H, W, Cin, Cout)) {
        dcpt_set_error("conv3x3_out_bwd_bf16: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(launch_conv3x3_s2b_bf16(dy, w, nullptr, dx, B, H, W, Cout, Cin, 1, s));
    const int nblk = conv3x3_wgrad_num_blocks(B, H, W, Cin);
    DCPT_TRY(launch_conv3x3_wgrad_bf16(x, dy, (float*)ws, nblk, dw, nullptr, B, H, W, Cout, Cin, 1, s));
    float* cspart = (float*)((char*)ws + align_up((size_t)nblk * (Cout * 9 + 1) * Cin * sizeof(float), 256));
    return launch_nchw_channel_sum(dy, cspart, dbias, B, Cout, H * W, s);
}
