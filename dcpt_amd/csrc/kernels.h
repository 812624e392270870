// Internal launchers for the non-GEMM kernels (all NHWC fp32; every launcher enqueues on `s`,
// never synchronises, returns DCPT_OK or sets the error string).
#pragma once
#include "dcpt_common.h"

// ---- ln.hip -------------------------------------------------------------------------------
int launch_ln_stats(const float* x, float* mu, float* rstd, int64_t M, int C, float eps, hipStream_t s);
int launch_ln_fwd(const float* x, const float* w, const float* b, float* y, float* mu, float* rstd, int64_t M, int C,
                  float eps, hipStream_t s);
// y = [relu]( LN(x)*w + b [+ res] )   (channels-first LayerNorm of the classifier head, fused tail)
int launch_ln_act_fwd(const float* x, const float* w, const float* b, const float* res, int relu, float* y, float* mu,
                      float* rstd, int64_t M, int C, float eps, hipStream_t s);
// as launch_ln_bwd, with the incoming gradient first masked by (ymask > 0) and optionally copied to gmasked
int launch_ln_act_bwd(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                      const float* ymask, float* gmasked, float* dx, float* part, int nblk, int64_t M, int C, hipStream_t s);
// general form: biasfree = 1 selects Restormer's BiasFree_LayerNorm backward (part row 1 is then meaningless)
int launch_ln_bwd_ex(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                     const float* ymask, float* gmasked, int biasfree, float* dx, float* part, int nblk, int64_t M, int C,
                     hipStream_t s);
// dx = LN-backward(gy; x, mu, rstd, w) (+ dres if non-null).  Column sums go to part[nblk][3][C]
// (0: sum gy*xhat, 1: sum gy, 2: sum dx_total); returns nblk through *nblk_out.
int ln_bwd_num_blocks(int64_t M, int C);
int launch_ln_bwd(const float* gy, const float* x, const float* mu, const float* rstd, const float* w, const float* dres,
                  float* dx, float* part, int nblk, int64_t M, int C, hipStream_t s);
// out[j][c] = sum_r part[r][j][c], j < nj  (deterministic order); out rows may be null to skip
int launch_colpart_reduce(const float* part, int R, int nj, int C, float* out0, float* out1, float* out2, hipStream_t s);

// ---- dwconv.hip ---------------------------------------------------------------------------
struct DwGeom {
    int B, H, W, C;  // C = gated channels (t1 has 2C)
};
int dw_num_blocks_per_image(const DwGeom& g);            // NBLK for fwd / bwd_a (quads of C)
int launch_dw_pack_weights(const float* w2, float* w2p, int C2, hipStream_t s);  // [C2][9] -> [9][C2]
// t2 = SG(dw3x3(t1) + b2); pool_part[B][NBLK][C] partial sums of t2
int launch_dw_fwd(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g,
                  hipStream_t s);
// SimpleGate-backward + transposed depthwise + tap-gradient partials in one pass (da = gate-backward(dts * s + dpool) is never
// written): dt1 and wpart[B*NBLKf][10][2C], NBLKf = dw_num_blocks_per_image_fused
int dw_num_blocks_per_image_fused(const DwGeom& g);
// optional rowpart[pixel][dw_fused_row_chunks][2]: per-pixel partials of dt1 . uvec and dt1 . (t1 - cvec) (gemm.h, E_LNBWD2)
int dw_fused_row_chunks(const DwGeom& g);
int launch_dw_bwd_fused(const float* dts, const float* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                        float* dt1, float* wpart, const DwGeom& g, hipStream_t s, float* rowpart = nullptr, const float* uvec = nullptr,
                        const float* cvec = nullptr);
// LayerNorm-through-conv vectors of the row-sum identities (gemm.h, E_LNBWD2), for a 1x1 conv W [N2][C] (+ bias bz) applied to
// LN(x) * w + b:   u[j] = sum_c W[j][c] w[c],   cvec[j] = bz[j] + sum_c W[j][c] b[c];   grid.y = job (two LayerNorms per block)
struct LnVecJobs {
    const float* W[2];
    const float* bz[2];
    const float* lnw[2];
    const float* lnb[2];
    float* u[2];
    float* cvec[2];
    int N2, C, n;
    int round_bf16;   // the bf16 path: W enters as its bf16-rounded GEMM operand copy
};
int launch_lnvec(const LnVecJobs& jobs, hipStream_t s);
// dw2[ch*9+tap] and db2[ch] from wpart[R][10][C2]
int launch_dw_wgrad_reduce(const float* wpart, int R, int C2, float* dw2, float* db2, hipStream_t s);

// dwring.hip: the same forward on an LDS-DMA row ring (fp32 / bf16 storage); pool_part[B][dw_ring_num_blocks_per_image][C]
bool dw_ring_usable(const DwGeom& g, int elem_bytes);
int dw_ring_num_blocks_per_image(const DwGeom& g, int elem_bytes);
int launch_dw_ring_fwd_f32(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g, hipStream_t s);
// launch_dw_bwd_fused on the ring (no row partials); wpart[B][dw_ring_bwd_num_blocks_per_image][10][2C]
bool dw_ring_bwd_usable(const DwGeom& g, int elem_bytes);
int dw_ring_bwd_num_blocks_per_image(const DwGeom& g);
int launch_dw_ring_bwd_fused_f32(const float* dts, const float* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                                 float* dt1, float* wpart, const DwGeom& g, hipStream_t s);
// Restormer forms on the same ring kernel: the GDFN gate gelu(a_1) a_2 backward (launch_dw_gelu_bwd_a + launch_dw_generic_bwd in one
// pass, da never written) and the plain depthwise backward; block counts from dw_ring_bwd_num_blocks_per_image({B, H, W, Ch | Ctot / 2})
int launch_dw_ring_gelu_fwd_f32(const float* u, const float* w2p, float* t, int B, int H, int W, int Ch, hipStream_t s);
// tout != null: also writes the gate product gelu(a_1) a_2 [M][Ch] (bit-identical to launch_dw_ring_gelu_fwd_f32) for callers that did not keep it
int launch_dw_ring_bwd_gelu_f32(const float* dt, const float* u, const float* w2p, float* du, float* wpart, int B, int H, int W, int Ch, hipStream_t s,
                                float* tout = nullptr);
int launch_dw_ring_bwd_plain_f32(const float* dy, const float* x, const float* w2p, float* dx, float* wpart, int B, int H, int W, int Ctot, hipStream_t s);

// generic depthwise pieces (Restormer): w2p is the [9][Ctot] packed weight (launch_dw_pack_weights)
int dw_num_blocks_generic(int B, int H, int W, int Ctot);
// t[M][Ch] = gelu(dw(u)[:, :Ch]) * dw(u)[:, Ch:]   (u has 2*Ch channels, no bias)
int launch_dw_gelu_fwd(const float* u, const float* w2p, float* t, int B, int H, int W, int Ch, hipStream_t s);
int launch_dw_gelu_bwd_a(const float* dt, const float* u, const float* w2p, float* da, int B, int H, int W, int Ch, hipStream_t s);
// y = dw(x) over Ctot channels; sq_part[B][nblk][nsq] partial sums of y^2 for the first nsq channels (may be null)
int launch_dw_plain_fwd(const float* x, const float* w2p, float* y, float* sq_part, int nsq, int B, int H, int W, int Ctot,
                        hipStream_t s);
// dx = dw^T(dy); wpart[B*nblk][10][Ctot] partials of the tap gradients (rows 0..8) -- reduce with launch_dw_wgrad_reduce
int launch_dw_generic_bwd(const float* dy, const float* x, const float* w2p, float* dx, float* wpart, int B, int H, int W, int Ctot,
                          hipStream_t s);

// ---- misc.hip -----------------------------------------------------------------------------
// pooled[b][k] = (sum_blk pool_part[b][blk][k]) / P ;  s[b][n] = sum_k Wsca[n][k]*pooled[b][k] + bsca[n]
int launch_sca_fwd(const float* pool_part, int nblk, const float* Wsca, const float* bsca, float* pooled, float* simg,
                   int B, int C, int P, hipStream_t s);
// ds[b][k] = sum_{m in image b} dts[m][k] * t2[m][k]   (two-stage, deterministic)
int sca_ds_num_blocks(int P);
int sca_ds_fused_slices(int P);   // P / 128 when the E_DOTCOL form applies, else 0
// backward: part[b][j][k] = sum over the j-th pixel slice of image b of dts*t2  (j < sca_ds_num_blocks(P))
int launch_sca_ds_part(const float* dts, const float* t2, float* ds_part, int B, int C, int P, hipStream_t s);
// (the slices may also come out of the dts GEMM's E_DOTCOL epilogue: 128-pixel slices, nslices = P / 128)
// critical path: dpool[b][k] = (1/P) sum_n Wsca[n][k] * sum_j part[b][j][n]
// (ds_out != null: also stores ds[b][n] = sum_j part[b][j][n], which the parameter-gradient side needs)
int launch_sca_dpool(const float* ds_part, int nslices, const float* Wsca, float* dpool, int B, int C, int P, hipStream_t s, float* ds_out = nullptr);
// parameter gradients (off the critical path): ds = sum_j part, dWsca = ds^T pooled, dbsca = sum_b ds
int launch_sca_wgrad(const float* ds_part, int nslices, float* ds, const float* pooled, float* dWsca, float* dbsca, int B, int C,
                     hipStream_t s);
// dpool[b][k] = (sum_n Wsca[n][k]*ds[b][n]) / P ; dWsca[n][k] = sum_b ds[b][n]*pooled[b][k] ; dbsca[n] = sum_b ds[b][n]

// TLSC box mean (arch_util.py:378-396): out[M][C] local k1 x k2 mean of in, replicate-padded; rowsum: [B][H][W-k2+1][C] scratch
int launch_box_mean(const float* in, float* rowsum, float* out, int B, int H, int W, int C, int k1, int k2, hipStream_t s);

enum WPackMode {
    WP_TRANSPOSE = 0,   // out[k][n] = in[n][k] * (rs ? rs[n] : 1)             in: [N][K]
    WP_DOWN = 1,        // out[oc][ij*C + ic] = in[oc][ic][ij]                  in: [N=2C][C][2][2]  (K = 4C)
    WP_DOWN_T = 2,      // out[ij*C + ic][oc] = in[oc][ic][ij]
    WP_UP = 3,          // out[ij*G + kk][ic] = in[4kk+ij][ic]                  in: [N=4G][K]
    WP_UP_T = 4,        // out[ic][ij*G + kk] = in[4kk+ij][ic]
    WP_CONV3 = 5,       // out[oc][tap*Ci + ic] = in[oc][ic][tap]              in: [N=Co][Ci][3][3]  (K = 9*Ci)
    WP_CONV3_T = 6      // out[ic][tap*Co + oc] = in[oc][ic][8-tap]            (dgrad: N' = Ci rows, K' = 9*Co)
};
int launch_wpack(const float* in, float* out, const float* rs, int N, int K, int mode, hipStream_t s);
// up to WPACK_MAX_JOBS WP_TRANSPOSE jobs (out[k][n] = in[n][k] * rs[n]) in one launch
constexpr int WPACK_MAX_JOBS = 6;
struct WpackJobs {
    const float* in[WPACK_MAX_JOBS];
    float* out[WPACK_MAX_JOBS];
    const float* rs[WPACK_MAX_JOBS];
    int N[WPACK_MAX_JOBS], K[WPACK_MAX_JOBS];
    int n;
};
int launch_wpack_multi(const WpackJobs& jobs, hipStream_t s);

enum WReduceMode { WR_PLAIN = 0, WR_DOWN = 1, WR_UP = 2, WR_CONV3 = 3 };
// dW = rowscale[n] * sum_s slab[s][n][k] (layout per mode)
// optional: dgain[n] = sum_k W[n][k]*G[n][k] + wbias[n]*cs[n];  dbias[n] = rowscale[n]*cs[n]
//           (cs = sum over the cs_rows partial rows colsum[r][n])
int launch_wgrad_reduce(const float* slab, const float* colsum, int splits, int cs_rows, int N, int K, const float* rowscale,
                        const float* W, const float* wbias, float* dW, float* dgain, float* dbias, int mode, hipStream_t s);
// as above, with slab s weighted by kscale[(s / splits_per_image)][k] (every split inside one image; SCA scale folded into the reduce)
int launch_wgrad_reduce_scaled(const float* slab, const float* colsum, int splits, int cs_rows, int N, int K, const float* rowscale,
                               const float* W, const float* wbias, float* dW, float* dgain, float* dbias, int mode,
                               const float* kscale, int splits_per_image, hipStream_t s);

// ---- conv3x3.hip ----------------------------------------------------------------------------
// small (NCHW, Cs <= 4) -> big (NHWC, Cb % 4 == 0):  y[p][c] = sum_{s,tap} x[p+off(tap)][s] * W(c,s,tap) (+ bias[c])
//   wmode 0: W(c,s,tap) = w[(c*Cs+s)*9+tap]      (forward of a Cs->Cb conv, weight [Cb][Cs][3][3])
//   wmode 1: W(c,s,tap) = w[(s*Cb+c)*9+(8-tap)]  (dgrad of a Cb->Cs conv, weight [Cs][Cb][3][3])
int launch_conv3x3_s2b(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cs, int Cb,
                       int wmode, hipStream_t s);
// big (NHWC) -> small (NCHW): y[p][s] = sum_{c,tap} x[p+off(tap)][c] * W(s,c,tap) (+ bias[s]) (+ res[p][s])
//   wmode 0: W(s,c,tap) = w[(s*Cb+c)*9+tap]      (forward of a Cb->Cs conv)
//   wmode 1: W(s,c,tap) = w[(c*Cs+s)*9+(8-tap)]  (dgrad of a Cs->Cb conv)
int launch_conv3x3_b2s(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W,
                       int Cs, int Cb, int wmode, hipStream_t s);
// G[c][s][tap] = sum_p big[p][c] * small[p+off(tap)][s];  bsum[c] = sum_p big[p][c]
//   omode 0: dW[(c*Cs+s)*9+tap] = G      omode 1: dW[(s*Cb+c)*9+(8-tap)] = G
int conv3x3_wgrad_num_blocks(int B, int H, int W, int Cb);
int launch_conv3x3_wgrad(const float* big, const float* small, float* part, int nblk, float* dW, float* bsum, int B, int H,
                         int W, int Cs, int Cb, int omode, hipStream_t s);
// out[c] = sum_{b,h,w} x[b][c][h][w]   (NCHW, tiny C); part: [C][128] scratch
int launch_nchw_channel_sum(const float* x, float* part, float* out, int B, int C, int HW, hipStream_t s);
// layout converters for arbitrary C
int launch_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, hipStream_t s);
int launch_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, hipStream_t s);
