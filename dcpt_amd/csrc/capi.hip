// C ABI entry points other than the NAFBlock pair (see include/dcpt_hip.h for the contract).
#include <stdarg.h>
#include <string.h>

#include "gemm.h"
#include "kernels.h"
#include "../../include/dcpt_hip.h"

static thread_local char g_err[512] = "";

void dcpt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dcpt_last_error(void) { return g_err; }
extern "C" int dcpt_abi_version(void) { return 15; }   // 15: + dcpt_bottleneck_fwd_bf16 / _bwd_bf16 (the head's BottleneckBlock in one call, LayerNorms in GEMM epilogues), dcpt_trace_enable / dcpt_trace_read (launch trace for the tests); 14: + dcpt_conv_wpack_bf16_multi and the *_packed forms of the bf16 head's conv entry points (cached operand copies of the weights); 13: + dcpt_conv_ln_bwd_acc / _bf16 (the BottleneckBlock's shortcut gradient summed in the data-gradient GEMM), dcpt_down2x2_bwd_acc / _bf16 (the skip connection's gradient summed in the down layer's scatter epilogue); 12: + dcpt_adamw_step (multi-tensor AdamW); 11: + dcpt_mix_*_bf16 / dcpt_meanpool_fc_*_bf16 (the all-bf16 head without cast passes), dcpt_nafblock_bf16_fused_ffn may return 2 (chain kernel of the wide levels); 10: + dcpt_nafblock_wpack_bf16_multi; 9: + dcpt_conv1x1_wgrad_bf16 (grouped 256 x 256-tile weight-gradient GEMM + finisher); 8: + dcpt_nafblock_fused_ffn / dcpt_nafblock_bf16_fused_ffn (fused 1 x 1 chains of the narrowest level; saved tensors that become optional); 7: + per-block packed weights for the bf16 NAFBlock (dcpt_nafblock_wpack_bf16, *_packed); 6: + bf16-storage intro / ending / down / up layers; 5: + bf16-storage classifier-head groups; 4: + bf16-storage NAFBlock and casts; 3: + dcpt_allreduce_flat; 2: dcpt_nafblock_saved / mdta / gdfn saved structs carry the kept LN (and gate) tensors

// ---------------------------------------------------------------------------------------------
extern "C" int dcpt_ln2d_fwd(const float* x, const float* weight, const float* bias, float* y, float* mu, float* rstd,
                             int64_t M, int C, float eps, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && weight && bias && y && mu && rstd, "ln2d_fwd: null argument");
    return launch_ln_fwd(x, weight, bias, y, mu, rstd, M, C, eps, (hipStream_t)stream);
}

extern "C" size_t dcpt_ln2d_bwd_ws_bytes(int64_t M, int C) { return align_up((size_t)ln_bwd_num_blocks(M, C) * 3 * C * sizeof(float), 256); }

extern "C" int dcpt_ln2d_bwd(const float* dy, const float* x, const float* mu, const float* rstd, const float* weight,
                             float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, int64_t M, int C,
                             dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && mu && rstd && weight && dx, "ln2d_bwd: null argument");
    DCPT_CHECK_ARG(C > 0 && C % 4 == 0 && M > 0, "ln2d_bwd: bad shape");
    const int nblk = ln_bwd_num_blocks(M, C);
    if (ws == nullptr || ws_bytes < dcpt_ln2d_bwd_ws_bytes(M, C)) {
        dcpt_set_error("ln2d_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(launch_ln_bwd(dy, x, mu, rstd, weight, nullptr, dx, (float*)ws, nblk, M, C, s));
    DCPT_TRY(launch_colpart_reduce((float*)ws, nblk, 3, C, dweight, dbias, nullptr, s));
    return DCPT_OK;
}

// ---------------------------------------------------------------------------------------------
// intro-type conv: image NCHW (Cin small) -> features NHWC (Cout)
extern "C" int dcpt_conv3x3_in_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                                   int Cout, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && w && y, "conv3x3_in_fwd: null argument");
    return launch_conv3x3_s2b(x, w, bias, y, B, H, W, Cin, Cout, 0, (hipStream_t)stream);
}

extern "C" size_t dcpt_conv3x3_in_bwd_ws_bytes(int B, int H, int W, int Cin, int Cout) {
    return align_up((size_t)conv3x3_wgrad_num_blocks(B, H, W, Cout) * (Cin * 9 + 1) * Cout * sizeof(float), 256);
}

extern "C" int dcpt_conv3x3_in_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias,
                                   void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dw && dbias, "conv3x3_in_bwd: null argument");
    if (ws == nullptr || ws_bytes < dcpt_conv3x3_in_bwd_ws_bytes(B, H, W, Cin, Cout)) {
        dcpt_set_error("conv3x3_in_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int nblk = conv3x3_wgrad_num_blocks(B, H, W, Cout);
    // dW[c][s][tap] = sum_p dy[p][c] * x[p+off][s];  db[c] = sum_p dy[p][c]
    DCPT_TRY(launch_conv3x3_wgrad(dy, x, (float*)ws, nblk, dw, dbias, B, H, W, Cin, Cout, 0, s));
    if (dx) DCPT_TRY(launch_conv3x3_b2s(dy, w, nullptr, nullptr, dx, B, H, W, Cin, Cout, 1, s));
    return DCPT_OK;
}

// ending-type conv: features NHWC (Cin) -> image NCHW (Cout small) (+ residual image)
extern "C" int dcpt_conv3x3_out_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int B,
                                    int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && w && y, "conv3x3_out_fwd: null argument");
    return launch_conv3x3_b2s(x, w, bias, res, y, B, H, W, Cout, Cin, 0, (hipStream_t)stream);
}

extern "C" size_t dcpt_conv3x3_out_bwd_ws_bytes(int B, int H, int W, int Cin, int Cout) {
    return align_up((size_t)conv3x3_wgrad_num_blocks(B, H, W, Cin) * (Cout * 9 + 1) * Cin * sizeof(float), 256) +
           align_up((size_t)Cout * 128 * sizeof(float), 256);
}

extern "C" int dcpt_conv3x3_out_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias,
                                    void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw && dbias, "conv3x3_out_bwd: null argument");
    if (ws == nullptr || ws_bytes < dcpt_conv3x3_out_bwd_ws_bytes(B, H, W, Cin, Cout)) {
        dcpt_set_error("conv3x3_out_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    // dx[p][c] = sum_{s,tap} dy[p-off][s] * w[s][c][tap]
    DCPT_TRY(launch_conv3x3_s2b(dy, w, nullptr, dx, B, H, W, Cout, Cin, 1, s));
    // dW[s][c][tap] = sum_p dy[p][s]*x[p+off][c] = sum_p' x[p'][c]*dy[p'-off][s]  (flipped-tap form)
    const int nblk = conv3x3_wgrad_num_blocks(B, H, W, Cin);
    DCPT_TRY(launch_conv3x3_wgrad(x, dy, (float*)ws, nblk, dw, nullptr, B, H, W, Cout, Cin, 1, s));
    float* cspart = (float*)((char*)ws + align_up((size_t)nblk * (Cout * 9 + 1) * Cin * sizeof(float), 256));
    DCPT_TRY(launch_nchw_channel_sum(dy, cspart, dbias, B, Cout, H * W, s));
    return DCPT_OK;
}

// ---------------------------------------------------------------------------------------------
// down: x [B][H][W][C] -> y [B][H/2][W/2][2C]
namespace {
struct DownWs {
    float* wp;      // fwd: packed [2C][4C]; bwd: packed-transposed [4C][2C]
    float* slab;
    float* colsum;
    int splits;
    int64_t rps;
};
size_t down_layout(int B, int H, int W, int C, int backward, void* base, size_t bytes, DownWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    DownWs w{};
    w.wp = a.get<float>((size_t)8 * C * C);
    if (backward) {
        const int64_t Mc = (int64_t)B * (H / 2) * (W / 2);
        gemm_tn_plan(Mc, 2 * C, 4 * C, &w.splits, &w.rps);
        w.slab = a.get<float>((size_t)w.splits * 8 * C * C);
        w.colsum = a.get<float>((size_t)w.splits * gemm_tn_tiles_k(2 * C, 4 * C) * 2 * C);
    }
    if (out) *out = w;
    return a.off;
}
}  // namespace

extern "C" size_t dcpt_down2x2_ws_bytes(int B, int H, int W, int C, int backward) {
    return down_layout(B, H, W, C, backward, nullptr, 0, nullptr);
}

extern "C" int dcpt_down2x2_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                                int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && y, "down2x2_fwd: null argument");
    DCPT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "down2x2_fwd: H=%d W=%d must be even, C=%d %% 4", H, W, C);
    DownWs d;
    const size_t need = down_layout(B, H, W, C, 0, ws, ws_bytes, &d);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("down2x2_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(launch_wpack(w, d.wp, nullptr, 2 * C, 4 * C, WP_DOWN, s));
    GemmNT g{};
    g.M = (int64_t)B * (H / 2) * (W / 2);
    g.A = x; g.K = 4 * C; g.gH = H / 2; g.gW = W / 2; g.gC = C;
    g.Bw = d.wp; g.N = 2 * C; g.C = y; g.ldc = 2 * C; g.bias = bias;
    return launch_gemm_nt(g, A_GATHER, E_BIAS, s);
}

extern "C" int dcpt_down2x2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws,
                                size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    return dcpt_down2x2_bwd_acc(dy, x, w, nullptr, dx, dw, dbias, ws, ws_bytes, B, H, W, C, stream);
}

extern "C" int dcpt_down2x2_bwd_acc(const float* dy, const float* x, const float* w, const float* dx_add, float* dx, float* dw, float* dbias,
                                    void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw && dbias, "down2x2_bwd: null argument");
    DCPT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "down2x2_bwd: bad shape");
    DownWs d;
    const size_t need = down_layout(B, H, W, C, 1, ws, ws_bytes, &d);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("down2x2_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t Mc = (int64_t)B * (H / 2) * (W / 2);
    // dx (fine) = scatter( dy [Mc][2C] x Wp^T )  ;  Bw [N=4C][K=2C]
    DCPT_TRY(launch_wpack(w, d.wp, nullptr, 2 * C, 4 * C, WP_DOWN_T, s));
    GemmNT g{};
    g.M = Mc; g.A = dy; g.lda = 2 * C; g.K = 2 * C; g.Bw = d.wp; g.N = 4 * C; g.C = dx;
    g.gH = H / 2; g.gW = W / 2; g.gC = C;
    g.res = dx_add;   // (the encoder output's second consumer is the skip connection: its gradient joins in the scatter epilogue)
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, dx_add ? E_SCATTER_ADD : E_SCATTER, s));
    // dW packed [2C][4C] = sum_m dy[m][oc] * gather(x)[m][k']
    GemmTN t{};
    t.M = Mc; t.X = dy; t.ldx = 2 * C; t.N = 2 * C; t.Y = x; t.K = 4 * C;
    t.gH = H / 2; t.gW = W / 2; t.gC = C;
    t.slab = d.slab; t.colsum = d.colsum; t.splits = d.splits; t.rows_per_split = d.rps;
    DCPT_TRY(launch_gemm_tn(t, A_PLAIN, A_GATHER, s));
    DCPT_TRY(launch_wgrad_reduce(d.slab, d.colsum, d.splits, d.splits * gemm_tn_tiles_k(2 * C, 4 * C), 2 * C, 4 * C, nullptr, nullptr,
                                 nullptr, dw, nullptr, dbias, WR_DOWN, s));
    return DCPT_OK;
}

// ---------------------------------------------------------------------------------------------
// up: x [B][H][W][C] -> y [B][2H][2W][C/2] = PixelShuffle2(conv1x1(x)) + skip
namespace {
struct UpWs {
    float* wp;  // packed [2C][C] (fwd) or [C][2C] (bwd)
    float* slab;
    int splits;
    int64_t rps;
};
size_t up_layout(int B, int H, int W, int C, int backward, void* base, size_t bytes, UpWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    UpWs w{};
    w.wp = a.get<float>((size_t)2 * C * C);
    if (backward) {
        gemm_tn_plan((int64_t)B * H * W, 2 * C, C, &w.splits, &w.rps);
        w.slab = a.get<float>((size_t)w.splits * 2 * C * C);
    }
    if (out) *out = w;
    return a.off;
}
}  // namespace

extern "C" size_t dcpt_up_ps_ws_bytes(int B, int H, int W, int C, int backward) {
    return up_layout(B, H, W, C, backward, nullptr, 0, nullptr);
}

extern "C" int dcpt_up_ps_fwd(const float* x, const float* w, const float* skip, float* y, void* ws, size_t ws_bytes, int B,
                              int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(x && w && y, "up_ps_fwd: null argument");
    DCPT_CHECK_ARG(C % 8 == 0, "up_ps_fwd: C=%d must be a multiple of 8", C);
    UpWs u;
    const size_t need = up_layout(B, H, W, C, 0, ws, ws_bytes, &u);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("up_ps_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    DCPT_TRY(launch_wpack(w, u.wp, nullptr, 2 * C, C, WP_UP, s));
    GemmNT g{};
    g.M = (int64_t)B * H * W; g.A = x; g.lda = C; g.K = C; g.Bw = u.wp; g.N = 2 * C; g.C = y;
    g.gH = H; g.gW = W; g.gC = C / 2; g.res = skip;
    return launch_gemm_nt(g, A_PLAIN, skip ? E_SCATTER_ADD : E_SCATTER, s);
}

extern "C" int dcpt_up_ps_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, void* ws, size_t ws_bytes,
                              int B, int H, int W, int C, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dy && x && w && dx && dw, "up_ps_bwd: null argument");
    DCPT_CHECK_ARG(C % 8 == 0, "up_ps_bwd: C=%d must be a multiple of 8", C);
    UpWs u;
    const size_t need = up_layout(B, H, W, C, 1, ws, ws_bytes, &u);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("up_ps_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    // dx[m][ic] = sum_n' gather(dy)[m][n'] * Wp[n'][ic]  ->  Bw [N=C][K=2C] = Wp^T
    DCPT_TRY(launch_wpack(w, u.wp, nullptr, 2 * C, C, WP_UP_T, s));
    GemmNT g{};
    g.M = M; g.A = dy; g.K = 2 * C; g.gH = H; g.gW = W; g.gC = C / 2; g.Bw = u.wp; g.N = C; g.C = dx; g.ldc = C;
    DCPT_TRY(launch_gemm_nt(g, A_GATHER, E_PLAIN, s));
    // dW packed [2C][C] = sum_m gather(dy)[m][n'] * x[m][ic]
    GemmTN t{};
    t.M = M; t.X = dy; t.N = 2 * C; t.Y = x; t.ldy = C; t.K = C; t.gH = H; t.gW = W; t.gC = C / 2;
    t.slab = u.slab; t.colsum = nullptr; t.splits = u.splits; t.rows_per_split = u.rps;
    DCPT_TRY(launch_gemm_tn(t, A_GATHER, A_PLAIN, s));
    DCPT_TRY(launch_wgrad_reduce(u.slab, nullptr, u.splits, 0, 2 * C, C, nullptr, nullptr, nullptr, dw, nullptr, nullptr, WR_UP, s));
    return DCPT_OK;
}

// ---------------------------------------------------------------------------------------------
namespace {
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ ref,
                                      float* __restrict__ y, int64_t n, int size_b, int64_t step_b, int act, int grad,
                                      float alpha, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (bias) v += bias[(i / step_b) % size_b];
        const float r = ref ? ref[i] : 0.f;
        float o;
        if (act == 3) {  // leaky relu
            if (grad == 0) o = (v > 0.f) ? v : v * alpha;
            else if (grad == 1) o = (r > 0.f) ? v : v * alpha;
            else o = 0.f;
        } else {  // linear
            o = (grad == 2) ? 0.f : v;
        }
        y[i] = o * scale;
    }
}
}  // namespace

extern "C" int dcpt_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, int64_t n, int size_b,
                                   int64_t step_b, int act, int grad, float alpha, float scale, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && y && n >= 0, "fused_bias_act: null argument");
    DCPT_CHECK_ARG(act == 1 || act == 3, "fused_bias_act: act=%d (1 linear, 3 leaky relu)", act);
    DCPT_CHECK_ARG(!bias || (size_b > 0 && step_b > 0), "fused_bias_act: bad bias geometry");
    if (n == 0) return DCPT_OK;
    int64_t nb = cdiv64(n, 256);
    if (nb > 4096) nb = 4096;
    fused_bias_act_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(x, bias, ref, y, n, size_b, step_b, act,
                                                                                     grad, alpha, scale);
    DCPT_CHECK_LAUNCH("fused_bias_act");
    return DCPT_OK;
}

extern "C" int dcpt_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, dcpt_stream_t stream) {
    return launch_nchw_to_nhwc(x, y, B, C, HW, (hipStream_t)stream);
}
extern "C" int dcpt_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, dcpt_stream_t stream) {
    return launch_nhwc_to_nchw(x, y, B, C, HW, (hipStream_t)stream);
}
