// PromptIR's prompt generation (reference basicsr/archs/promptir_arch.py:237-262, PromptGenBlock) between the linear layer and
// the 3x3 conv:   prompt[b] = bilinear_{(S,S)->(H,W)}( sum_l softmax(logits[b])[l] * param[l] ),   written as NHWC.
// The mean over the pixels + linear layer in front of it is dcpt_meanpool_fc_*, the conv behind it dcpt_conv_* (dense 3x3).
// Bilinear = F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * S / H - 0.5 clamped at 0, neighbours
// x0 = floor(src), x1 = min(x0 + 1, S - 1), weights (1 - l, l).
#include "dcpt_common.h"
#include "../../include/dcpt_hip.h"

namespace {

constexpr int MAXL = 8;   // prompt_len (5 in the reference)

struct Src {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Src src_index(int dst, float scale, int S) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    Src r;
    r.i0 = (int)s;
    if (r.i0 > S - 1) r.i0 = S - 1;
    r.i1 = r.i0 + (r.i0 < S - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ void prompt_softmax_kernel(const float* __restrict__ logits, float* __restrict__ w, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
    for (int l = 0; l < L; ++l) m = fmaxf(m, logits[b * L + l]);
    float e[MAXL], s = 0.f;
    for (int l = 0; l < L; ++l) {
        e[l] = expf(logits[b * L + l] - m);
        s += e[l];
    }
    for (int l = 0; l < L; ++l) w[b * L + l] = e[l] / s;
}

// thread = (b, y, x, 4 channels)
__global__ __launch_bounds__(256) void prompt_mix_fwd_kernel(const float* __restrict__ w, const float* __restrict__ param,
                                                             float* __restrict__ out, int B, int L, int D, int S, int H, int W) {
    const int nq = D / 4;
    const int64_t total = (int64_t)B * H * W * nq;
    const float sy = (float)S / (float)H, sx = (float)S / (float)W;
    const int64_t SS = (int64_t)S * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % nq);
        const int64_t pix = i / nq;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        const Src ry = src_index(y, sy, S), rx = src_index(x, sx, S);
        float wl[MAXL];
        for (int l = 0; l < L; ++l) wl[l] = w[b * L + l];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 4 * q + j;
            float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
            for (int l = 0; l < L; ++l) {
                const float* pp = param + ((int64_t)l * D + d) * SS;
                p00 = fmaf(wl[l], pp[ry.i0 * S + rx.i0], p00);
                p01 = fmaf(wl[l], pp[ry.i0 * S + rx.i1], p01);
                p10 = fmaf(wl[l], pp[ry.i1 * S + rx.i0], p10);
                p11 = fmaf(wl[l], pp[ry.i1 * S + rx.i1], p11);
            }
            o[j] = ry.l0 * (rx.l0 * p00 + rx.l1 * p01) + ry.l1 * (rx.l0 * p10 + rx.l1 * p11);
        }
        stg4(out + pix * D + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// dP[b][sy][sx][d] = sum over the output pixels whose bilinear footprint contains (sy, sx); gather, fixed order
__global__ __launch_bounds__(256) void prompt_mix_bwd_dp_kernel(const float* __restrict__ dout, float* __restrict__ dP, int B, int D, int S,
                                                                int H, int W) {
    const int64_t total = (int64_t)B * S * S * D;
    const float scy = (float)S / (float)H, scx = (float)S / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % D);
        const int sx = (int)((i / D) % S), sy = (int)((i / ((int64_t)D * S)) % S), b = (int)(i / ((int64_t)D * S * S));
        // candidate output rows / columns: src in (s - 1, s + 1)  (+ a safety margin; membership is re-checked exactly)
        int ylo = (int)floorf(((float)sy - 0.5f) / scy - 0.5f) - 1, yhi = (int)ceilf(((float)sy + 1.5f) / scy - 0.5f) + 1;
        int xlo = (int)floorf(((float)sx - 0.5f) / scx - 0.5f) - 1, xhi = (int)ceilf(((float)sx + 1.5f) / scx - 0.5f) + 1;
        if (sy == 0) ylo = 0;           // clamped sources (src < 0 -> 0)
        if (sx == 0) xlo = 0;
        if (sy == S - 1) yhi = H - 1;   // i1 clamped at S - 1
        if (sx == S - 1) xhi = W - 1;
        ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
        yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) {
            const Src ry = src_index(y, scy, S);
            float cy = 0.f;
            if (ry.i0 == sy) cy += ry.l0;
            if (ry.i1 == sy) cy += ry.l1;
            if (cy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                const Src rx = src_index(x, scx, S);
                float cx = 0.f;
                if (rx.i0 == sx) cx += rx.l0;
                if (rx.i1 == sx) cx += rx.l1;
                if (cx == 0.f) continue;
                acc = fmaf(cy * cx, dout[(((int64_t)b * H + y) * W + x) * D + d], acc);
            }
        }
        dP[i] = acc;
    }
}

// dparam[l][d][sy][sx] = sum_b w[b][l] * dP[b][sy][sx][d]
__global__ __launch_bounds__(256) void prompt_mix_bwd_param_kernel(const float* __restrict__ dP, const float* __restrict__ w,
                                                                   float* __restrict__ dparam, int B, int L, int D, int S) {
    const int64_t total = (int64_t)L * D * S * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int sx = (int)(i % S), sy = (int)((i / S) % S), d = (int)((i / ((int64_t)S * S)) % D), l = (int)(i / ((int64_t)S * S * D));
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(w[b * L + l], dP[(((int64_t)b * S + sy) * S + sx) * D + d], acc);
        dparam[i] = acc;
    }
}

// one block per image: dw[l] = <dP[b], param[l]>, then through the softmax: dlogits[l] = w[l] * (dw[l] - sum_l' dw[l'] w[l'])
__global__ __launch_bounds__(256) void prompt_mix_bwd_logits_kernel(const float* __restrict__ dP, const float* __restrict__ param,
                                                                    const float* __restrict__ w, float* __restrict__ dlogits, int L, int D,
                                                                    int S) {
    __shared__ float red[MAXL][256];
    const int b = blockIdx.x;
    const int64_t n = (int64_t)D * S * S;
    float acc[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) acc[l] = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {   // i = (sy, sx, d) in dP order
        const int d = (int)(i % D);
        const int64_t ss = i / D;
        const float g = dP[(int64_t)b * n + i];
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
            if (l < L) acc[l] = fmaf(g, param[((int64_t)l * D + d) * S * S + ss], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) red[l][threadIdx.x] = acc[l];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
#pragma unroll
            for (int l = 0; l < MAXL; ++l) red[l][threadIdx.x] += red[l][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float dot = 0.f;
        for (int l = 0; l < L; ++l) dot += red[l][0] * w[b * L + l];
        for (int l = 0; l < L; ++l) dlogits[b * L + l] = w[b * L + l] * (red[l][0] - dot);
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t nb = (n + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

}  // namespace

extern "C" size_t dcpt_prompt_mix_bwd_ws_bytes(int B, int D, int S) { return align_up((size_t)B * D * S * S * sizeof(float), 256); }

extern "C" int dcpt_prompt_mix_fwd(const float* logits, const float* param, float* weights, float* out, int B, int L, int D, int S, int H,
                                   int W, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(logits && param && weights && out, "prompt_mix_fwd: null argument");
    DCPT_CHECK_ARG(B > 0 && L >= 1 && L <= MAXL && D > 0 && D % 4 == 0 && S > 0 && H > 0 && W > 0,
                   "prompt_mix_fwd: bad shape (prompt_len <= %d, prompt_dim %% 4 == 0)", MAXL);
    prompt_softmax_kernel<<<dim3(cdiv(B, 64)), dim3(64), 0, s>>>(logits, weights, B, L);
    DCPT_CHECK_LAUNCH("prompt_softmax");
    prompt_mix_fwd_kernel<<<dim3(grid_for((int64_t)B * H * W * (D / 4))), dim3(256), 0, s>>>(weights, param, out, B, L, D, S, H, W);
    DCPT_CHECK_LAUNCH("prompt_mix_fwd");
    return DCPT_OK;
}

extern "C" int dcpt_prompt_mix_bwd(const float* dout, const float* param, const float* weights, float* dlogits, float* dparam, void* ws,
                                   size_t ws_bytes, int B, int L, int D, int S, int H, int W, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dout && param && weights && dlogits && dparam, "prompt_mix_bwd: null argument");
    DCPT_CHECK_ARG(B > 0 && L >= 1 && L <= MAXL && D > 0 && D % 4 == 0 && S > 0 && H > 0 && W > 0, "prompt_mix_bwd: bad shape");
    if (ws == nullptr || ws_bytes < dcpt_prompt_mix_bwd_ws_bytes(B, D, S)) {
        dcpt_set_error("prompt_mix_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    float* dP = (float*)ws;
    prompt_mix_bwd_dp_kernel<<<dim3(grid_for((int64_t)B * S * S * D)), dim3(256), 0, s>>>(dout, dP, B, D, S, H, W);
    DCPT_CHECK_LAUNCH("prompt_mix_bwd_dp");
    prompt_mix_bwd_param_kernel<<<dim3(grid_for((int64_t)L * D * S * S)), dim3(256), 0, s>>>(dP, weights, dparam, B, L, D, S);
    DCPT_CHECK_LAUNCH("prompt_mix_bwd_param");
    prompt_mix_bwd_logits_kernel<<<dim3(B), dim3(256), 0, s>>>(dP, param, weights, dlogits, L, D, S);
    DCPT_CHECK_LAUNCH("prompt_mix_bwd_logits");
    return DCPT_OK;
}
