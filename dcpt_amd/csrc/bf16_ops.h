// Launchers of the bf16-path bandwidth kernels (bf16_ops.hip, dwconv.hip).
#pragma once
#include "bf16.h"
#include "kernels.h"

int launch_ln_fwd_bf16(const bf16_t* x, const float* w, const float* b, bf16_t* y, float* mu, float* rstd, int64_t M, int C, float eps,
                       hipStream_t s);
int ln_bwd_bf16_num_blocks(int64_t M, int C);
// part: [nblk][2][C] fp32 (0: sum gy*xhat -> dweight, 1: sum gy -> dbias); reduce with launch_colpart_reduce(part, nblk, 2, C, ...)
int launch_ln_bwd_bf16(const bf16_t* gy, const bf16_t* x, const float* mu, const float* rstd, const float* w, const bf16_t* dres, bf16_t* dx,
                       float* part, int nblk, int64_t M, int C, hipStream_t s);
// channels-first LayerNorm of the classifier head with its fused tail / head: y = [relu](LN(x) * w + b [+ res]);
// backward: gy masked by (ymask > 0) (and copied to gmasked for the shortcut branch) before the LayerNorm backward
int launch_ln_act_fwd_bf16(const bf16_t* x, const float* w, const float* b, const bf16_t* res, int relu, bf16_t* y, float* mu, float* rstd,
                           int64_t M, int C, float eps, hipStream_t s);
int launch_ln_act_bwd_bf16(const bf16_t* gy, const bf16_t* x, const float* mu, const float* rstd, const float* w, const bf16_t* ymask,
                           bf16_t* gmasked, bf16_t* dx, float* part, int nblk, int64_t M, int C, hipStream_t s);
int launch_cast_f32_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t s);
int launch_cast_bf16_f32(const bf16_t* x, float* y, int64_t n, hipStream_t s);

constexpr int WPACKB_MAX_JOBS = 12;
constexpr int WPACKB_MAX_JOBS_L = 80;   // the multi-block form (8 NAFBlocks per launch; 80 x 48 B of kernel arguments)
template <int MAXJ>
struct WpackBJobsT {   // job j: in [N][K] fp32 -> transpose 0: out[img][n][k] = in[n][k] * kscale[img][k]; 1: out[k][n] = in[n][k] * rs[n];
                       // 2 / 3: the dense-3x3 packs [Co][9 Ci] / [Ci][9 Co] (flipped taps) of misc.hip's WP_CONV3 / WP_CONV3_T (N = Co, K = 9 Ci)
                       // 4 / 5 / 6 / 7: misc.hip's WP_DOWN / WP_DOWN_T / WP_UP / WP_UP_T (2x2 stride-2 conv, 1x1 conv + PixelShuffle(2))
                       // 8: depthwise taps [N = 2C][K = 9] -> FP32 [9][2C] (dw_pack layout; `out` points at floats)
                       // 10: the chain kernels' stream of one transposed, row-scaled [C][C] matrix (in[c][r] * rs[c]): N = K = C (chain_bf16.h)
                       // 9: the chain kernel's fragment-order stream of conv4 (in, [2C][C]) + conv5 (rs, [C][C]): N = 3 C, K = C (chain_bf16.h)
    const float* in[MAXJ];
    bf16_t* out[MAXJ];
    const float* rs[MAXJ];
    const float* kscale[MAXJ];
    int N[MAXJ], K[MAXJ], nimg[MAXJ], transpose[MAXJ];
    int n;
};
using WpackBJobs = WpackBJobsT<WPACKB_MAX_JOBS>;
using WpackBJobsL = WpackBJobsT<WPACKB_MAX_JOBS_L>;
int launch_wpack_bf16(const WpackBJobs& jobs, hipStream_t s);
int launch_wpack_bf16(const WpackBJobsL& jobs, hipStream_t s, int grid_cap = 256);   // grid.x <= grid_cap (x up to 80 jobs in grid.y)
int launch_scale_rows_bf16(const bf16_t* x, const float* simg, bf16_t* out, int64_t M, int C, int P, hipStream_t s);
int launch_sca_ds_part_bf16(const bf16_t* dts, const bf16_t* t2, float* ds_part, int B, int C, int P, int nslices, hipStream_t s);

// dwconv.hip
int dw_num_blocks_per_image_bf16(const DwGeom& g);
int dw_num_blocks_per_image_fused_bf16(const DwGeom& g);
int launch_dw_fwd_bf16(const bf16_t* t1, const float* w2p, const float* b2, bf16_t* t2, float* pool_part, const DwGeom& g, hipStream_t s);
int dw_fused_row_chunks_bf16(const DwGeom& g);
int launch_dw_bwd_fused_bf16(const bf16_t* dts, const bf16_t* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                             bf16_t* dt1, float* wpart, const DwGeom& g, hipStream_t s, float* rowpart = nullptr, const float* uvec = nullptr,
                             const float* cvec = nullptr);
// dwring.hip
int launch_dw_ring_fwd_bf16(const bf16_t* t1, const float* w2p, const float* b2, bf16_t* t2, float* pool_part, const DwGeom& g, hipStream_t s);
int launch_dw_ring_bwd_fused_bf16(const bf16_t* dts, const bf16_t* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                                  bf16_t* dt1, float* wpart, const DwGeom& g, hipStream_t s);

// conv3x3.hip: the network-edge 3x3 convs with the feature side in bf16 storage
int launch_conv3x3_s2b_bf16(const float* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int Cs, int Cb, int wmode, hipStream_t s);
int launch_conv3x3_b2s_bf16(const bf16_t* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cs, int Cb,
                            int wmode, hipStream_t s);
int launch_conv3x3_wgrad_bf16(const bf16_t* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H, int W, int Cs,
                              int Cb, int omode, hipStream_t s);
