// AdamW over a LIST of fp32 parameter tensors in a few launches (the optimizer step that closes every training step of the path:
// reference basicsr/models/base_model.py:70-93 builds torch.optim.AdamW, sr_model.py:118 / degradation_classification_pretrain_model.py:170-173
// call optimizer.step()).  The update is elementwise and purely bandwidth-bound -- 4 reads + 3 writes of 4 bytes per parameter, 1.9 GB for
// the 67.9 M parameters of NAFNet-64 [1,1,1,28] -- and the network has 664 parameter tensors, most of them 64 .. 2048 elements: torch's
// fused kernel takes 36 tensors and 320 blocks per launch (19-23 launches, 1.7 ms for that network); here a launch takes AW_MAX tensors
// whatever their sizes, their pointers travel in the kernel arguments (no device-side table, no copy), a block owns one 4096-element chunk
// of one tensor and finds it by a binary search over the per-tensor block offsets.
//
// Arithmetic (per element, fp32, the order of torch's _fused_adamw_ kernel):
//     p -= lr wd p;   m = lerp(m, g, 1 - b1);   v = b2 v + (1 - b2) g g;   p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t computed by the caller in double precision.
#include "dcpt_common.h"
#include "../../include/dcpt_hip.h"
#include "prof.h"

namespace {

constexpr int AW_MAX = 80;       // tensors per launch: 80 x (4 pointers + size + offset) = 3.2 KB of kernel arguments
constexpr int AW_CHUNK = 4096;   // elements per block: 256 threads x 4 float4

struct AdamWArgs {
    float* p[AW_MAX];
    const float* g[AW_MAX];
    float* m[AW_MAX];
    float* v[AW_MAX];
    uint32_t n[AW_MAX];
    uint32_t blk0[AW_MAX + 1];   // first block of tensor i; blk0[cnt] = number of blocks
    int cnt;
    float lr_wd, w1, b2, omb2, step_size, bc2_sqrt, eps, gsign;
};

__device__ __forceinline__ float lerp_t(float a, float b, float w) {   // at::native::lerp (weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w))
    const float d = b - a;
    return w < 0.5f ? a + w * d : b - d * (1.0f - w);
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWArgs& a) {
    g *= a.gsign;
    p -= a.lr_wd * p;
    m = lerp_t(m, g, a.w1);
    v = a.b2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;   // (a division, as torch's fused kernel: the kernel is bandwidth-bound, and a resumed run reproduces bit for bit)
    p -= a.step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamWArgs a) {
    // tensor of this block: the last i with blk0[i] <= blockIdx.x (wave-uniform: scalar loads from the kernel arguments)
    int lo = 0, hi = a.cnt - 1;
    const uint32_t b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.blk0[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    const uint32_t n = a.n[lo];
    const uint32_t e0 = (b - a.blk0[lo]) * (uint32_t)AW_CHUNK;
    float* __restrict__ P = a.p[lo];
    const float* __restrict__ G = a.g[lo];
    float* __restrict__ M = a.m[lo];
    float* __restrict__ V = a.v[lo];
    const bool vec = (((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0;
    if (vec && e0 + AW_CHUNK <= n) {
        float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // all sixteen loads in flight before the first use
            const uint32_t e = e0 + (uint32_t)(j * 256 + threadIdx.x) * 4u;
            pv[j] = *reinterpret_cast<const float4*>(P + e);
            gv[j] = *reinterpret_cast<const float4*>(G + e);
            mv[j] = *reinterpret_cast<const float4*>(M + e);
            vv[j] = *reinterpret_cast<const float4*>(V + e);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t e = e0 + (uint32_t)(j * 256 + threadIdx.x) * 4u;
            adamw_one(pv[j].x, gv[j].x, mv[j].x, vv[j].x, a);
            adamw_one(pv[j].y, gv[j].y, mv[j].y, vv[j].y, a);
            adamw_one(pv[j].z, gv[j].z, mv[j].z, vv[j].z, a);
            adamw_one(pv[j].w, gv[j].w, mv[j].w, vv[j].w, a);
            *reinterpret_cast<float4*>(P + e) = pv[j];
            *reinterpret_cast<float4*>(M + e) = mv[j];
            *reinterpret_cast<float4*>(V + e) = vv[j];
        }
    } else {   // a tensor's last chunk, tensors shorter than a chunk, views that are not 16-byte aligned
        for (uint32_t e = e0 + threadIdx.x; e < n && e < e0 + AW_CHUNK; e += 256) {
            float p = P[e], m = M[e], v = V[e];
            adamw_one(p, G[e], m, v, a);
            P[e] = p;
            M[e] = m;
            V[e] = v;
        }
    }
}

}  // namespace

extern "C" int dcpt_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                               const int64_t* numel, const dcpt_adamw_hparams* h, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(n >= 0 && (n == 0 || (params && grads && exp_avg && exp_avg_sq && numel)) && h, "adamw_step: null argument");
    DCPT_CHECK_ARG(h->bias_correction1 > 0.0 && h->bias_correction2 > 0.0 && h->beta1 >= 0.0 && h->beta1 < 1.0 && h->beta2 >= 0.0 && h->beta2 < 1.0,
                   "adamw_step: betas in [0, 1) and positive bias corrections (step >= 1) expected");
    int64_t total = 0;
    for (int k = 0; k < n; ++k) total += numel[k] > 0 ? numel[k] : 0;
    ProfScope prof(s, PROF_OTHER + 4, total, 0, 0, 12.0 * (double)total, 28.0 * (double)total);
    AdamWArgs a{};
    a.lr_wd = (float)(h->lr * h->weight_decay);
    a.w1 = (float)(1.0 - h->beta1);
    a.b2 = (float)h->beta2;
    a.omb2 = (float)(1.0 - h->beta2);
    a.step_size = (float)(h->lr / h->bias_correction1);
    a.bc2_sqrt = (float)sqrt(h->bias_correction2);
    a.eps = (float)h->eps;
    a.gsign = h->maximize ? -1.0f : 1.0f;
    int i = 0;
    while (i < n) {
        a.cnt = 0;
        uint32_t blocks = 0;
        for (; i < n && a.cnt < AW_MAX; ++i) {
            if (numel[i] == 0) continue;
            DCPT_CHECK_ARG(numel[i] > 0 && numel[i] < ((int64_t)1 << 32) - AW_CHUNK, "adamw_step: tensor %d has %lld elements (1 .. 2^32 - 4097)", i,
                           (long long)numel[i]);
            DCPT_CHECK_ARG(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], "adamw_step: tensor %d has a null pointer", i);
            a.p[a.cnt] = params[i]; a.g[a.cnt] = grads[i]; a.m[a.cnt] = exp_avg[i]; a.v[a.cnt] = exp_avg_sq[i];
            a.n[a.cnt] = (uint32_t)numel[i];
            a.blk0[a.cnt] = blocks;
            blocks += (uint32_t)((numel[i] + AW_CHUNK - 1) / AW_CHUNK);
            ++a.cnt;
        }
        if (a.cnt == 0) break;
        a.blk0[a.cnt] = blocks;
        adamw_kernel<<<dim3(blocks), dim3(256), 0, s>>>(a);
        DCPT_CHECK_LAUNCH("adamw");
    }
    return DCPT_OK;
}
