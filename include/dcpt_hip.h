/* dcpt_hip.h -- C ABI of libdcpt_hip.so: the MI355X (gfx950) implementation of the DCPT
 * restoration-encoder hot path (NAFNet blocks and the layers between them).
 *
 * Drop-in boundary.  The reference (MILab-PKU/dcpt) has no FFI of its own on this path: its
 * archs call ATen ops from Python (basicsr/archs/nafnet_arch.py) and its only native bindings
 * are the unused JIT extensions under basicsr/ops (layernorm: ops/layernorm/src/layernorm_kernel.cpp
 * :57-60 pybind `forward`/`backward`; fused_act: ops/fused_act/src/fused_bias_act.cpp:14-26).  The
 * entry points below are what a maintainer would bind from the arch modules instead of those ATen
 * calls; each one cites the reference code it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless it says otherwise; the caller (PyTorch)
 *    owns all buffers, including workspaces and tensors saved for backward;
 *  - feature maps are NHWC ([B][H][W][C], C contiguous; torch `channels_last` memory); only the
 *    network's image input/output is NCHW (3 channels);
 *  - weights keep the reference's state-dict layouts (conv: [out][in][kh][kw]);
 *  - every call enqueues work on `stream` (a hipStream_t) and returns immediately: 0 on success,
 *    non-zero on error (dcpt_last_error() gives the message, per host thread).  No compute call
 *    synchronises with the host or allocates device memory, and none keeps DATA between calls.
 *    Process-wide state that does exist, all of it listed here: (1) two global toggles,
 *    dcpt_set_side_stream() and dcpt_prof_enable() (the latter also owns the event pairs it recorded
 *    until dcpt_prof_read() drains them -- dcpt_prof_read is the one call that waits on the device);
 *    (2) a cache of one internal low-priority HIP stream + 8 events per (device, caller stream),
 *    created on the first backward call that uses it and kept for the life of the process;
 *    (3) the lazily resolved RCCL entry point of dcpt_allreduce_flat().  Calls on different
 *    streams / devices are re-entrant; the toggles are not meant to be flipped concurrently with
 *    launches.
 *  - channel counts must be multiples of 4.
 */
#ifndef DCPT_HIP_H
#define DCPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dcpt_stream_t; /* hipStream_t */

const char* dcpt_last_error(void);
int dcpt_abi_version(void);

/* ---- LayerNorm2d --------------------------------------------------------------------------
 * replaces nafnet_arch.py:25-64 (LayerNormFunction / LayerNorm2d) and ops/layernorm
 * (layernorm_kernel.cpp:14-55).  x,y: [M][C] rows = pixels.  mu/rstd: [M] saved statistics. */
int dcpt_ln2d_fwd(const float* x, const float* weight, const float* bias, float* y, float* mu, float* rstd,
                  int64_t M, int C, float eps, dcpt_stream_t stream);
size_t dcpt_ln2d_bwd_ws_bytes(int64_t M, int C);
int dcpt_ln2d_bwd(const float* dy, const float* x, const float* mu, const float* rstd, const float* weight,
                  float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, int64_t M, int C,
                  dcpt_stream_t stream);

/* ---- NAFBlock -----------------------------------------------------------------------------
 * replaces nafnet_arch.py:83-186 (NAFBlock.forward and its autograd backward). */
typedef struct {
    const float* norm1_w; const float* norm1_b;   /* [C] */
    const float* conv1_w; const float* conv1_b;   /* [2C][C][1][1], [2C] */
    const float* conv2_w; const float* conv2_b;   /* [2C][1][3][3], [2C]  (depthwise) */
    const float* conv3_w; const float* conv3_b;   /* [C][C][1][1], [C] */
    const float* sca_w;   const float* sca_b;     /* [C][C][1][1], [C] */
    const float* norm2_w; const float* norm2_b;   /* [C] */
    const float* conv4_w; const float* conv4_b;   /* [2C][C][1][1], [2C] */
    const float* conv5_w; const float* conv5_b;   /* [C][C][1][1], [C] */
    const float* beta;    const float* gamma;     /* [1][C][1][1] */
} dcpt_nafblock_params;

typedef struct {   /* same shapes as dcpt_nafblock_params; all written (not accumulated) */
    float* norm1_w; float* norm1_b;
    float* conv1_w; float* conv1_b;
    float* conv2_w; float* conv2_b;
    float* conv3_w; float* conv3_b;
    float* sca_w;   float* sca_b;
    float* norm2_w; float* norm2_b;
    float* conv4_w; float* conv4_b;
    float* conv5_w; float* conv5_b;
    float* beta;    float* gamma;
} dcpt_nafblock_grads;

typedef struct {   /* activations kept for backward (M = B*H*W pixels) */
    float* t1;      /* [M][2C]  conv1(LN1(inp)) */
    float* t2;      /* [M][C]   SimpleGate(dwconv(t1)) */
    float* y;       /* [M][C]   inp + conv3(t2*s)*beta */
    float* v;       /* [M][2C]  conv4(LN2(y)) */
    float* mu1; float* rstd1; float* mu2; float* rstd2;   /* [M] */
    float* pooled;  /* [B][C]   mean_{h,w} t2 */
    float* s;       /* [B][C]   SCA scale */
    float* xn1;     /* [M][C]   LN1(inp): read back by conv1's forward and weight-gradient GEMMs as a plain operand */
    float* xn2;     /* [M][C]   LN2(y) */
    float* g;       /* [M][C]   SimpleGate(v), written by conv4's GEMM epilogue; conv5's operand in forward and weight gradient */
} dcpt_nafblock_saved;

size_t dcpt_nafblock_fwd_ws_bytes(int B, int H, int W, int C);
size_t dcpt_nafblock_bwd_ws_bytes(int B, int H, int W, int C);
/* inp,out: [B][H][W][C].  `saved` is the forward's scratch for ITS OWN backward (inference callers may recycle the buffers); which of
 * the tensors hold data afterwards is the library's business.  Where dcpt_nafblock_fused_ffn(C) is 1 (C = 64: the forward 1 x 1 chains
 * run as one pass each, ffn_f32.hip) saved->xn1 / xn2 / g are never read or written in either pass and may be NULL (dcpt_nafblock_bwd takes
 * them from its GEMMs' operand loaders), and a caller that will not run the backward pass may also pass v = mu1 = rstd1 = mu2 = rstd2 =
 * NULL: the forward then skips those writes (the second half moves 2 tensor passes instead of 4).  Elsewhere every pointer is required. */
int dcpt_nafblock_fused_ffn(int C);
int dcpt_nafblock_fwd(const dcpt_nafblock_params* p, const float* inp, float* out, const dcpt_nafblock_saved* saved,
                      void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);
/* dout,dinp: [B][H][W][C]; dinp may alias dout. */
int dcpt_nafblock_bwd(const dcpt_nafblock_params* p, const dcpt_nafblock_grads* g, const float* inp,
                      const dcpt_nafblock_saved* saved, const float* dout, float* dinp, void* ws, size_t ws_bytes,
                      int B, int H, int W, int C, dcpt_stream_t stream);

/* ---- Opt-in GEMM precision mode "bf16x3" (gemm_x3.hip): while a scratch buffer is registered, the wide fp32 NT GEMMs of every entry point
 * (plain or per-image-scaled A operand, N % 256 == 0, K % 16 == 0, >= 192 tiles of 256 x 256) run on the bf16 matrix pipe with both operands
 * split into three bfloat16 pieces (x = x0 + x1 + x2 exactly), six piece products per element pair, fp32 accumulation: fp32-CLASS results
 * (dropped terms <= 2^-24 relative; not bit-identical to the default fp32-MFMA kernels).  PROCESS-WIDE state: the scratch (>= 6 bytes per
 * element of the largest A operand + weights) is shared by all launches, which must therefore all be on ONE stream while the mode is on.
 * scratch == NULL switches the mode off (the default).  The reference computes in fp32; this mode is reported separately. */
/* min_tiles: launches with fewer 256 x 256 tiles keep the fp32-MFMA kernels (<= 0: the default, 192 -- most of the 256 CUs busy). */
int dcpt_set_gemm_x3(void* scratch, size_t bytes, int min_tiles);
/* launches since process start that were eligible for the mode but ran on the fp32 kernels because `bytes` was too small for their weight
 * images (a mode that is silently only partly on would be a wrong label on a measurement: callers check this stays 0) */
long long dcpt_gemm_x3_scratch_misses(void);

/* ---- NAFBlock, bf16 storage (BASELINE.json configs[2]) --------------------------------------------------------------
 * Same block (nafnet_arch.py:83-186), activations and saved tensors as bfloat16 (raw uint16_t, upper half of an fp32, stored
 * round-to-nearest-even), fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 parameters / parameter gradients / LayerNorm
 * statistics / reductions.  The reference has no reduced-precision mode (its AMP / TF32 switches are commented out,
 * basicsr/test.py:26-27): this is new behaviour with its own tolerance (tests/test_gpu_bf16.py, oracle bf16 mode).
 * C must be a multiple of 8 (16-byte rows), at most 1024.  dinp must NOT alias dout (weight-gradient GEMMs on the side
 * stream still read dout when dinp is written). */
typedef struct {
    uint16_t* t1;   /* [M][2C] bf16 */
    uint16_t* t2;   /* [M][C]  */
    uint16_t* y;    /* [M][C]  */
    uint16_t* v;    /* [M][2C] */
    float* mu1; float* rstd1; float* mu2; float* rstd2;   /* [M] fp32 */
    float* pooled;  /* [B][C] fp32 */
    float* s;       /* [B][C] fp32 */
    uint16_t* xn1;  /* [M][C]  LN1(inp) */
    uint16_t* xn2;  /* [M][C]  LN2(y) */
    uint16_t* g;    /* [M][C]  SimpleGate(v) */
} dcpt_nafblock_saved_bf16;
size_t dcpt_nafblock_fwd_bf16_ws_bytes(int B, int H, int W, int C);
size_t dcpt_nafblock_bwd_bf16_ws_bytes(int B, int H, int W, int C);
int dcpt_nafblock_fwd_bf16(const dcpt_nafblock_params* p, const uint16_t* inp, uint16_t* out, const dcpt_nafblock_saved_bf16* saved,
                           void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);
int dcpt_nafblock_bwd_bf16(const dcpt_nafblock_params* p, const dcpt_nafblock_grads* g, const uint16_t* inp,
                           const dcpt_nafblock_saved_bf16* saved, const uint16_t* dout, uint16_t* dinp, void* ws, size_t ws_bytes,
                           int B, int H, int W, int C, dcpt_stream_t stream);
/* Per-block operand copies of the weights (ABI 7).  Everything the block's GEMMs and depthwise kernels read of the PARAMETERS --
 * bf16 [N][K] copies of conv1 / conv4 / conv5, transposed (beta / gamma-scaled) copies of conv5 / conv4 / conv3 / conv1 for the data
 * gradients, the depthwise taps as [9][2C] fp32 -- depends on the parameters only: a caller that keeps one buffer of
 * dcpt_nafblock_wpack_bf16_bytes(C) per block and re-runs dcpt_nafblock_wpack_bf16 (ONE launch) whenever the block's parameters
 * changed (once per optimizer step) hands it to the *_packed forms below, which then launch no pack kernels except the per-image
 * SCA-scaled conv3 weights (5 launches fewer per block and step).  Results are bit-identical to dcpt_nafblock_fwd/bwd_bf16. */
size_t dcpt_nafblock_wpack_bf16_bytes(int C);
/* 1 when dcpt_nafblock_fwd_bf16 at width C runs the second half of the block (LayerNorm2 -> conv4 -> SimpleGate -> conv5 -> residual,
 * nafnet_arch.py:180-186) as ONE kernel (ffn_bf16.hip, the narrow levels): a caller that will not run the backward pass (inference)
 * may then pass saved->v = xn2 = g = mu2 = rstd2 = NULL -- all five or none -- and the forward skips v as well (4 -> 2 tensor passes for
 * that half).  With 1 the library never reads or writes saved->xn2 / g / mu2 / rstd2 in either pass (its backward kernels recompute
 * LayerNorm2, the gate and the statistics from y and v): in training they only have to be non-null.  0: every saved buffer is used --
 * except that saved->v ALONE may be NULL at any width when no backward pass follows: conv4's bias + gate epilogue then writes
 * SimpleGate(v) only (one tensor pass instead of three for that launch).
 * 2 (ABI 11; C = 256 / 512, chain_bf16.hip: the same chain per 128-pixel tile with the weights streamed past it): the forward WRITES
 * xn2 / g / mu2 / rstd2 (the unfused backward reads them), but a caller that will not run the backward pass may pass all five of
 * v / xn2 / g / mu2 / rstd2 as NULL exactly as with 1. */
int dcpt_nafblock_bf16_fused_ffn(int C);
int dcpt_nafblock_wpack_bf16(const dcpt_nafblock_params* p, void* packed, size_t packed_bytes, int C, dcpt_stream_t stream);
/* The same for n blocks at once (ABI 10; ps / packed / packed_bytes / C are arrays of n, any mix of widths): ceil(n / 8) launches
 * instead of n -- a network's packs once per optimizer step.  Bit-identical to n calls of dcpt_nafblock_wpack_bf16. */
int dcpt_nafblock_wpack_bf16_multi(const dcpt_nafblock_params* ps, void* const* packed, const size_t* packed_bytes, const int* C, int n,
                                   dcpt_stream_t stream);
/* Weight (and bias) gradient of a 1 x 1 convolution in bf16 storage (ABI 9; the conv1 / conv3 / conv4 / conv5 weight gradients of
 * nafnet_arch.py:170-186 as an operator of its own): dW[n][k] = sum_m dY[m][n] * X[m][k], db[n] = sum_m dY[m][n] (db may be NULL).
 * dY: [M][N] bf16, X: [M][K] bf16, dW: [N][K] fp32.  N, K multiples of 8; where both are multiples of 256 the 256 x 256-tile grouped
 * kernel + finisher of gemm_tn_bf16_256.hip run (what dcpt_nafblock_bwd_bf16 uses at the wide levels), else the 128-wide kernel.
 * Deterministic (fixed summation order). */
size_t dcpt_conv1x1_wgrad_bf16_ws_bytes(int64_t M, int N, int K);
int dcpt_conv1x1_wgrad_bf16(const uint16_t* dY, const uint16_t* X, float* dW, float* db, void* ws, size_t ws_bytes, int64_t M, int N, int K,
                            dcpt_stream_t stream);

int dcpt_nafblock_fwd_bf16_packed(const dcpt_nafblock_params* p, const void* packed, size_t packed_bytes, const uint16_t* inp, uint16_t* out,
                                  const dcpt_nafblock_saved_bf16* saved, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                                  dcpt_stream_t stream);
int dcpt_nafblock_bwd_bf16_packed(const dcpt_nafblock_params* p, const void* packed, size_t packed_bytes, const dcpt_nafblock_grads* g,
                                  const uint16_t* inp, const dcpt_nafblock_saved_bf16* saved, const uint16_t* dout, uint16_t* dinp, void* ws,
                                  size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);
/* fp32 <-> bf16 (RNE) on n contiguous elements, n % 8 == 0: the edges of the bf16 path */
int dcpt_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, dcpt_stream_t stream);
int dcpt_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, dcpt_stream_t stream);

/* ---- the layers between the NAFBlock groups, bf16 storage (edge_bf16.hip): same algorithms and argument meaning as
 * dcpt_conv3x3_in / _out, dcpt_down2x2 and dcpt_up_ps below (reference basicsr/archs/nafnet_arch.py:202-219, :230, :238-242,
 * :252-272); feature maps (NHWC) and their gradients are bf16, the 3-channel images (NCHW), weights, biases and all parameter
 * gradients fp32.  With these a NAFNet in bf16 storage has no cast kernels between its layers.  down: C % 8 == 0; up: C % 16 == 0.
 * Workspaces: dcpt_conv3x3_in_bwd_ws_bytes / dcpt_conv3x3_out_bwd_ws_bytes (unchanged), dcpt_down2x2_bf16_ws_bytes, dcpt_up_ps_bf16_ws_bytes. */
int dcpt_conv3x3_in_fwd_bf16(const float* x, const float* w, const float* bias, uint16_t* y, int B, int H, int W, int Cin, int Cout,
                             dcpt_stream_t stream);
int dcpt_conv3x3_in_bwd_bf16(const uint16_t* dy, const float* x, const float* w, float* dx /* may be NULL */, float* dw, float* dbias, void* ws,
                             size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
int dcpt_conv3x3_out_fwd_bf16(const uint16_t* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cin,
                              int Cout, dcpt_stream_t stream);
int dcpt_conv3x3_out_bwd_bf16(const float* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                              int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
size_t dcpt_down2x2_bf16_ws_bytes(int B, int H, int W, int C, int backward);
int dcpt_down2x2_fwd_bf16(const uint16_t* x, const float* w, const float* bias, uint16_t* y, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                          dcpt_stream_t stream);
int dcpt_down2x2_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                          int B, int H, int W, int C, dcpt_stream_t stream);
/* ABI 13: dx = dx_add + the down layer's data gradient (dx_add [B][H][W][C] or NULL): see dcpt_down2x2_bwd_acc */
int dcpt_down2x2_bwd_acc_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const uint16_t* dx_add, uint16_t* dx, float* dw, float* dbias,
                              void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);
size_t dcpt_up_ps_bf16_ws_bytes(int B, int H, int W, int C, int backward);
int dcpt_up_ps_fwd_bf16(const uint16_t* x, const float* w, const uint16_t* skip /* may be NULL */, uint16_t* y, void* ws, size_t ws_bytes, int B,
                        int H, int W, int C, dcpt_stream_t stream);
int dcpt_up_ps_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, uint16_t* dx, float* dw, void* ws, size_t ws_bytes, int B, int H,
                        int W, int C, dcpt_stream_t stream);

/* ---- classifier head, bf16 storage: the conv -> channels-first LayerNorm -> [+shortcut] -> [ReLU] groups and the
 * conv1x1 -> MaxPool2d(2,2) -> ReLU downsamples of degrad_classify_arch.py (:69-103, :227-243, :596-602) with bf16 activations
 * (x, z = conv output, y and their gradients), fp32 parameters / gradients / statistics; same argument meaning as dcpt_conv_ln_* and
 * dcpt_conv1x1_pool_relu_* below.  Channel counts must be multiples of 8, Cout <= 1024. */
size_t dcpt_conv_ln_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward);
int dcpt_conv_ln_fwd_bf16(const uint16_t* x, const float* w, const float* lnw, const float* lnb, const uint16_t* res, int relu, uint16_t* z,
                          uint16_t* y, float* mu, float* rstd, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize,
                          dcpt_stream_t stream);
int dcpt_conv_ln_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const float* lnw, const uint16_t* z, const uint16_t* y,
                          const float* mu, const float* rstd, uint16_t* dx, float* dw, float* dlnw, float* dlnb, uint16_t* dres, void* ws,
                          size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream);
/* ABI 13: as dcpt_conv_ln_bwd_bf16 with dx = dx_add + conv^T(dz) (dx_add [B][H][W][Cin] or NULL; may alias dx).  The BottleneckBlock's input
 * feeds conv1 AND the shortcut (degrad_classify_arch.py:227-243): conv3's dres goes in as conv1's dx_add and the two gradients are summed
 * in the data-gradient GEMM's epilogue instead of by an extra pass over the feature map. */
int dcpt_conv_ln_bwd_acc_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const float* lnw, const uint16_t* z, const uint16_t* y,
                              const float* mu, const float* rstd, const uint16_t* dx_add, uint16_t* dx, float* dw, float* dlnw, float* dlnb,
                              uint16_t* dres, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu,
                              dcpt_stream_t stream);
size_t dcpt_conv1x1_pool_relu_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int backward);
int dcpt_conv1x1_pool_relu_fwd_bf16(const uint16_t* x, const float* w, uint16_t* z, uint16_t* y, void* ws, size_t ws_bytes, int B, int H, int W,
                                    int Cin, int Cout, dcpt_stream_t stream);
int dcpt_conv1x1_pool_relu_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* w, const uint16_t* z, uint16_t* dx, float* dw, void* ws,
                                    size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
/* ABI 14: cached operand copies of the head's conv weights.  The entry points above pack their bf16 operand image of `w` on every call (one small
 * launch in front of each GEMM: 68 per step of the DCPT head).  A caller that keeps one buffer of dcpt_conv_wpack_bf16_bytes(Cin, Cout, ksize) bytes
 * per conv (device memory, 256-byte aligned) fills all of them with dcpt_conv_wpack_bf16_multi (n convs: a launch per 40 convs, the 3 x 3 and the
 * 1 x 1 convs apart) once per optimizer step -- whenever the weights may have changed -- and passes it as `wpacked` to the *_packed forms: same
 * results bit for bit, no pack launches.  A buffer holds the forward image ([Cout][ksize ksize Cin]) and the data gradient's (transposed; taps
 * flipped for the 3 x 3).  wpacked == NULL: packs in the call, exactly the unpacked entry point (`w` may be NULL only when wpacked is given). */
size_t dcpt_conv_wpack_bf16_bytes(int Cin, int Cout, int ksize);
int dcpt_conv_wpack_bf16_multi(const float* const* w, void* const* packed, const size_t* packed_bytes, const int* Cin, const int* Cout,
                               const int* ksize, int n, dcpt_stream_t stream);
int dcpt_conv_ln_fwd_bf16_packed(const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes, const float* lnw, const float* lnb,
                                 const uint16_t* res, int relu, uint16_t* z, uint16_t* y, float* mu, float* rstd, void* ws, size_t ws_bytes, int B,
                                 int H, int W, int Cin, int Cout, int ksize, dcpt_stream_t stream);
int dcpt_conv_ln_bwd_acc_bf16_packed(const uint16_t* dy, const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes,
                                     const float* lnw, const uint16_t* z, const uint16_t* y, const float* mu, const float* rstd,
                                     const uint16_t* dx_add, uint16_t* dx, float* dw, float* dlnw, float* dlnb, uint16_t* dres, void* ws,
                                     size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream);
int dcpt_conv1x1_pool_relu_fwd_bf16_packed(const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes, uint16_t* z, uint16_t* y,
                                           void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
int dcpt_conv1x1_pool_relu_bwd_bf16_packed(const uint16_t* dy, const uint16_t* x, const float* w, const void* wpacked, size_t wpacked_bytes,
                                           const uint16_t* z, uint16_t* dx, float* dw, void* ws, size_t ws_bytes, int B, int H, int W, int Cin,
                                           int Cout, dcpt_stream_t stream);

/* ABI 15: the classifier head's BottleneckBlock as ONE call (reference basicsr/archs/degrad_classify_arch.py:132-243 as the DCPT head builds it:
 * identity shortcut, LN norm, bias-free convs -- relu(LN(conv1 1x1 C -> 2C)) -> relu(LN(conv2 3x3 2C -> 2C)) -> relu(LN(conv3 1x1 2C -> C) + x)),
 * bf16 activations.  Same arithmetic as three dcpt_conv_ln_fwd_bf16 / dcpt_conv_ln_bwd_acc_bf16 calls (forward: bit-identical), fewer passes:
 * a channels-first LayerNorm (:17-44) runs in the epilogue of the GEMM that produces its input wherever a row fits one column tile (<= 128
 * channels, or 256 on the 256-row kernel), and in backward the two inner LayerNorms ride in the epilogues of the data-gradient GEMMs above
 * them, so the gradients of the two inner activations are never written.  group[0..2] = conv1, conv2, conv3:
 *   w [Cout][Cin][k][k] fp32 and/or its cached operand images (wpacked, dcpt_conv_wpack_bf16_bytes; NULL, 0: packed in the call);
 *   lnw / lnb [Cout] fp32 (lnb of conv1 / conv2 is read by backward too: the inner ReLU masks are recomputed from z);  z (conv output), y (group output; y of group 2 = the block's output) [B][H][W][Cout] bf16 and mu / rstd [B H W]
 *   fp32 are written by forward and read by backward;  dw / dlnw / dlnb: backward outputs (unused in forward).
 * dout, x, dx: [B][H][W][C] bf16.  Workspace: dcpt_bottleneck_bf16_ws_bytes. */
typedef struct {
    const float* w;
    const void* wpacked;
    size_t wpacked_bytes;
    const float* lnw;
    const float* lnb;
    uint16_t* z;
    uint16_t* y;
    float* mu;
    float* rstd;
    float* dw;
    float* dlnw;
    float* dlnb;
} dcpt_bneck_group_t;
size_t dcpt_bottleneck_bf16_ws_bytes(int B, int H, int W, int C, int backward);
int dcpt_bottleneck_fwd_bf16(const uint16_t* x, const dcpt_bneck_group_t* group, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                             dcpt_stream_t stream);
int dcpt_bottleneck_bwd_bf16(const uint16_t* dout, const uint16_t* x, const dcpt_bneck_group_t* group, uint16_t* dx, void* ws, size_t ws_bytes,
                             int B, int H, int W, int C, dcpt_stream_t stream);

/* ABI 15: launch trace for tests.  dcpt_trace_enable(1) clears and starts it: from then on every dispatch decision of the library counts the
 * kernel family it launched under a fixed name ("head.conv3x3+ln_fwd_epilogue", "nt_bf16.256", ...); dcpt_trace_read writes "name count\n"
 * lines into buf (NUL-terminated, truncated to cap) and returns the bytes the whole text needs.  Off (the default) it costs one load per launch.
 * A parity test uses it to assert WHICH kernel produced the result it compared (tests/kernel_trace.py). */
int dcpt_trace_enable(int on);
size_t dcpt_trace_read(char* buf, size_t cap);

/* TLSC variant (nafnet_arch.py:277-288 `NAFNet`, arch_util.py:313-455): inference-only forward where SCA's global
 * mean is a k1 x k2 local box mean (replicate-padded), i.e. a per-pixel attention map.  Callers use the plain
 * dcpt_nafblock_fwd when the window covers the whole map (arch_util.py:352-353). */
size_t dcpt_nafblock_local_ws_bytes(int B, int H, int W, int C, int k1, int k2);
int dcpt_nafblock_local_fwd(const dcpt_nafblock_params* p, const float* inp, float* out, void* ws, size_t ws_bytes, int B, int H,
                            int W, int C, int k1, int k2, dcpt_stream_t stream);

/* ---- network-edge 3x3 convs ------------------------------------------------------------------
 * intro: nafnet_arch.py:202-210,252  x NCHW [B][Cin][H][W] -> y NHWC [B][H][W][Cout]
 * ending: nafnet_arch.py:211-219,271-272  x NHWC -> y NCHW [B][Cout][H][W] (+ res NCHW, may be NULL) */
int dcpt_conv3x3_in_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                        int Cout, dcpt_stream_t stream);
size_t dcpt_conv3x3_in_bwd_ws_bytes(int B, int H, int W, int Cin, int Cout);
/* dx (NCHW) may be NULL when the image does not need a gradient */
int dcpt_conv3x3_in_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws,
                        size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
int dcpt_conv3x3_out_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H,
                         int W, int Cin, int Cout, dcpt_stream_t stream);
size_t dcpt_conv3x3_out_bwd_ws_bytes(int B, int H, int W, int Cin, int Cout);
int dcpt_conv3x3_out_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);

/* ---- down: Conv2d(C, 2C, 2, 2) (nafnet_arch.py:230) ------------------------------------------
 * x [B][H][W][C] -> y [B][H/2][W/2][2C];  w [2C][C][2][2] */
size_t dcpt_down2x2_ws_bytes(int B, int H, int W, int C, int backward);
int dcpt_down2x2_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B,
                     int H, int W, int C, dcpt_stream_t stream);
int dcpt_down2x2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws,
                     size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);
/* ABI 13: dx = dx_add + the data gradient (dx_add NHWC [B][H][W][C] or NULL).  An encoder group's output feeds the down layer AND the skip
 * connection (nafnet_arch.py:255-258, :264-265): the skip's gradient (= the up layer's dy) is summed in the scatter epilogue of the down
 * layer's data-gradient GEMM instead of by a pass of autograd's own over the feature map. */
int dcpt_down2x2_bwd_acc(const float* dy, const float* x, const float* w, const float* dx_add, float* dx, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream);

/* ---- up: Conv2d(C, 2C, 1, bias=False) + PixelShuffle(2) + skip add (nafnet_arch.py:238-242,264-265)
 * x [B][H][W][C], skip/y [B][2H][2W][C/2];  w [2C][C][1][1].  The skip gradient equals dy. */
size_t dcpt_up_ps_ws_bytes(int B, int H, int W, int C, int backward);
int dcpt_up_ps_fwd(const float* x, const float* w, const float* skip, float* y, void* ws, size_t ws_bytes, int B, int H,
                   int W, int C, dcpt_stream_t stream);
int dcpt_up_ps_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, void* ws, size_t ws_bytes,
                   int B, int H, int W, int C, dcpt_stream_t stream);

/* ---- degradation-classifier head (basicsr/archs/degrad_classify_arch.py) -------------------------
 * conv (ksize 1 or dense 3x3 / pad 1, no bias, weight [Cout][Cin][k][k]) -> channels-first LayerNorm (eps 1e-6,
 * :17-44) -> [+ res] -> [ReLU]: the `Conv2d` wrapper :69-103 with norm="LN" and the BottleneckBlock tail :227-243.
 * z = conv output (saved for backward), y = result; x,z,y,res NHWC. */
size_t dcpt_conv_ln_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward);
int dcpt_conv_ln_fwd(const float* x, const float* w, const float* lnw, const float* lnb, const float* res, int relu, float* z,
                     float* y, float* mu, float* rstd, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout,
                     int ksize, dcpt_stream_t stream);
/* dres (may be NULL) receives the gradient of the residual input; dx may be NULL */
int dcpt_conv_ln_bwd(const float* dy, const float* x, const float* w, const float* lnw, const float* z, const float* y,
                     const float* mu, const float* rstd, float* dx, float* dw, float* dlnw, float* dlnb, float* dres, void* ws,
                     size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream);
/* ABI 13: dx = dx_add + conv^T(dz) (dx_add NHWC [B][H][W][Cin] or NULL; may alias dx) -- the shortcut gradient of a BottleneckBlock
 * (:227-243) summed in the data-gradient GEMM's epilogue */
int dcpt_conv_ln_bwd_acc(const float* dy, const float* x, const float* w, const float* lnw, const float* z, const float* y,
                         const float* mu, const float* rstd, const float* dx_add, float* dx, float* dw, float* dlnw, float* dlnb, float* dres,
                         void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ksize, int relu, dcpt_stream_t stream);
/* downsample layer :596-602: Conv2d(Cin, Cout, 1, bias=False) -> MaxPool2d(2,2) -> ReLU; y [B][H/2][W/2][Cout] */
size_t dcpt_conv1x1_pool_relu_ws_bytes(int B, int H, int W, int Cin, int Cout, int backward);
int dcpt_conv1x1_pool_relu_fwd(const float* x, const float* w, float* z, float* y, void* ws, size_t ws_bytes, int B, int H, int W,
                               int Cin, int Cout, dcpt_stream_t stream);
int dcpt_conv1x1_pool_relu_bwd(const float* dy, const float* x, const float* w, const float* z, float* dx, float* dw, void* ws,
                               size_t ws_bytes, int B, int H, int W, int Cin, int Cout, dcpt_stream_t stream);
/* :632-637: out = prev + softmax(mixing_weights)[idx] * feat  (prev may be NULL); the gradient of prev equals dout */
int dcpt_mix_fwd(const float* prev, const float* feat, const float* mixing_weights, int n, int idx, float* out, int64_t numel,
                 dcpt_stream_t stream);
size_t dcpt_mix_bwd_ws_bytes(int64_t numel);
int dcpt_mix_bwd(const float* dout, const float* feat, const float* mixing_weights, int n, int idx, float* dfeat, float* dmix,
                 void* ws, size_t ws_bytes, int64_t numel, dcpt_stream_t stream);
/* :639-640: mean over the P pixels of each image, then Linear(C, NC).  x [B][P][C] */
size_t dcpt_meanpool_fc_ws_bytes(int B, int P, int C);
int dcpt_meanpool_fc_fwd(const float* x, const float* fw, const float* fb, float* pooled, float* logits, void* ws, size_t ws_bytes,
                         int B, int P, int C, int NC, dcpt_stream_t stream);
int dcpt_meanpool_fc_bwd(const float* dlogits, const float* pooled, const float* fw, float* dx, float* dfw, float* dfb, void* ws,
                         size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream);
/* ABI 11: the same four with the feature maps (prev / feat / out, dout / dfeat; x, dx) in bf16 storage -- the all-bf16 head
 * (degrad_classify_arch.py:632-640 with act_dtype "bf16") mixes and pools without cast passes.  Arithmetic and reductions are fp32, one
 * rounding on store: the results equal the fp32 entry points' behind casts, bit for bit.  Workspaces as above. */
int dcpt_mix_fwd_bf16(const uint16_t* prev, const uint16_t* feat, const float* mixing_weights, int n, int idx, uint16_t* out, int64_t numel,
                      dcpt_stream_t stream);
int dcpt_mix_bwd_bf16(const uint16_t* dout, const uint16_t* feat, const float* mixing_weights, int n, int idx, uint16_t* dfeat, float* dmix,
                      void* ws, size_t ws_bytes, int64_t numel, dcpt_stream_t stream);
int dcpt_meanpool_fc_fwd_bf16(const uint16_t* x, const float* fw, const float* fb, float* pooled, float* logits, void* ws, size_t ws_bytes,
                              int B, int P, int C, int NC, dcpt_stream_t stream);
int dcpt_meanpool_fc_bwd_bf16(const float* dlogits, const float* pooled, const float* fw, uint16_t* dx, float* dfw, float* dfb, void* ws,
                              size_t ws_bytes, int B, int P, int C, int NC, dcpt_stream_t stream);
/* PromptIR_DC.conv_embed (:491-494, Conv2d(3, dim, 7, 2, 3) + bias -> LayerNorm): the strided conv over the NCHW image is
 * unfolded into patch rows A[(b,oy,ox)][Kp], column k = (c*ksize + ky)*ksize + kx (the weight's own order), column
 * Cin*ksize*ksize = 1 (the bias column), the rest 0 up to Kp (a multiple of 4); the product + LayerNorm then run through
 * dcpt_conv_ln_* as a 1x1 conv over the patch rows.  dcpt_patch_fold is the gradient of the image (gather, fixed order). */
int dcpt_patch_unfold(const float* x, float* A, int B, int Cin, int H, int W, int ksize, int stride, int pad, int Kp,
                      dcpt_stream_t stream);
int dcpt_patch_fold(const float* dA, float* dx, int B, int Cin, int H, int W, int ksize, int stride, int pad, int Kp,
                    dcpt_stream_t stream);

/* plain conv (ksize 1 or dense 3x3 / pad 1, no bias), NHWC -> NHWC: Restormer Downsample/Upsample convs
 * (restormer_arch.py:179-185,194-200) and reduce_chan_level{2,3} (:300-302,:315-317) */
size_t dcpt_conv_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int backward);
int dcpt_conv_fwd(const float* x, const float* w, float* y, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout,
                  int ksize, dcpt_stream_t stream);
int dcpt_conv_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, void* ws, size_t ws_bytes, int B, int H,
                  int W, int Cin, int Cout, int ksize, dcpt_stream_t stream);

/* ---- Restormer blocks (basicsr/archs/restormer_arch.py) -------------------------------------------------
 * MDTA half of a TransformerBlock (:148-159 with Attention :103-145):  y = x + project_out(attn(LN(x))).
 * flags: DCPT_LN_BIASFREE = BiasFree_LayerNorm (:26-40, norm_b ignored), else WithBias_LayerNorm (:43-59); 0 / 1 are the
 * Restormer settings of this repo's reference (LayerNorm eps 1e-6, ReLU attention :134-136).  DCPT_LN_EPS_1E5 and
 * DCPT_ATTN_SOFTMAX select the PromptIR variants of the same blocks (basicsr/archs/promptir_arch.py:39-40,57-59: eps 1e-5;
 * :136: softmax over the last dimension).  heads: C % heads == 0 and (C / heads) % 4 == 0 (<= 256 for softmax).
 * F.normalize over the pixels (eps 1e-12). */
#define DCPT_LN_BIASFREE 1
#define DCPT_LN_EPS_1E5 2
#define DCPT_ATTN_SOFTMAX 4
typedef struct {
    const float* norm_w; const float* norm_b;   /* [C] */
    const float* qkv_w;                         /* [3C][C][1][1] */
    const float* dw_w;                          /* [3C][1][3][3] */
    const float* proj_w;                        /* [C][C][1][1] */
    const float* temperature;                   /* [heads][1][1] */
} dcpt_mdta_params;
typedef struct {
    float* norm_w; float* norm_b; float* qkv_w; float* dw_w; float* proj_w; float* temperature;
} dcpt_mdta_params_grads;
typedef struct {   /* saved for backward; M = B*H*W, ch = C/heads */
    float* mu; float* rstd;       /* [M] */
    float* qkv1;                  /* [M][3C] conv1x1(LN(x)) */
    float* qkv;                   /* [M][3C] after the depthwise 3x3 */
    float* nrm;                   /* [B][2C] L2 norms of q and k over the pixels */
    float* ghat; float* attn; float* attnT;   /* [B][heads][ch][ch] */
    float* out_att;               /* [M][C] attn @ v */
    float* xn;                    /* [M][C] LN(x): plain operand of the qkv conv in forward and weight gradient */
    /* LEAN MODE: qkv1, out_att and xn may be NULL (pass the same NULLs to forward and backward): the forward pass then keeps
     * them in the workspace only and the backward pass recomputes them (one LayerNorm pass, the qkv GEMM, the attn @ v GEMM)
     * -- 5 of the 9 [M][C] units this half keeps, for ~8 % more flops. */
} dcpt_mdta_saved;
size_t dcpt_mdta_ws_bytes(int B, int H, int W, int C, int heads, int backward);
int dcpt_mdta_fwd(const dcpt_mdta_params* p, const float* x, float* y, const dcpt_mdta_saved* saved, void* ws, size_t ws_bytes,
                  int B, int H, int W, int C, int heads, int flags, dcpt_stream_t stream);
int dcpt_mdta_bwd(const dcpt_mdta_params* p, const dcpt_mdta_params_grads* g, const float* x, const dcpt_mdta_saved* saved,
                  const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int H, int W, int C, int heads, int flags,
                  dcpt_stream_t stream);
/* GDFN half (:75-100):  y = x + project_out(gelu(x1) * x2), (x1,x2) = dwconv(project_in(LN(x))).chunk(2); the erf GELU of
 * torch.nn.functional.gelu (not the tanh form), evaluated branch-free with |error| <= 5e-7 absolute (Abramowitz-Stegun 7.1.26).
 * hidden = int(C * ffn_expansion_factor) may be any positive integer (padded to a multiple of 4 internally).
 * flags: DCPT_LN_BIASFREE | DCPT_LN_EPS_1E5 as for dcpt_mdta_fwd. */
typedef struct {
    const float* norm_w; const float* norm_b;   /* [C] */
    const float* in_w;                          /* [2*hidden][C][1][1] */
    const float* dw_w;                          /* [2*hidden][1][3][3] */
    const float* out_w;                         /* [C][hidden][1][1] */
} dcpt_gdfn_params;
typedef struct { float* norm_w; float* norm_b; float* in_w; float* dw_w; float* out_w; } dcpt_gdfn_params_grads;
typedef struct {   /* hp = hidden rounded up to a multiple of 4 */
    float* mu; float* rstd;   /* [M] */
    float* u;                 /* [M][2hp] project_in(LN(x)) */
    float* t;                 /* [M][hp]  gelu(x1)*x2 */
    float* xn;                /* [M][C]   LN(x): plain operand of project_in in forward and weight gradient */
    /* LEAN MODE: t and xn may be NULL (in forward AND backward): recomputed in backward from u and x. */
} dcpt_gdfn_saved;
size_t dcpt_gdfn_ws_bytes(int B, int H, int W, int C, int hidden, int backward);
int dcpt_gdfn_fwd(const dcpt_gdfn_params* p, const float* x, float* y, const dcpt_gdfn_saved* saved, void* ws, size_t ws_bytes,
                  int B, int H, int W, int C, int hidden, int flags, dcpt_stream_t stream);
int dcpt_gdfn_bwd(const dcpt_gdfn_params* p, const dcpt_gdfn_params_grads* g, const float* x, const dcpt_gdfn_saved* saved,
                  const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int H, int W, int C, int hidden, int flags,
                  dcpt_stream_t stream);
/* PromptIR's PromptGenBlock (basicsr/archs/promptir_arch.py:237-262) between its linear layer and its 3x3 conv:
 *   out[b] = bilinear_{(S,S)->(H,W), align_corners=False}( sum_l softmax(logits[b])[l] * param[l] )   as NHWC [B][H][W][D].
 * logits [B][L] (= dcpt_meanpool_fc_fwd of the block input), param [L][D][S][S] (the (1,L,D,S,S) parameter), weights [B][L]
 * receives the softmax (saved for backward).  L <= 8, D % 4 == 0. */
int dcpt_prompt_mix_fwd(const float* logits, const float* param, float* weights, float* out, int B, int L, int D, int S, int H, int W,
                        dcpt_stream_t stream);
size_t dcpt_prompt_mix_bwd_ws_bytes(int B, int D, int S);
int dcpt_prompt_mix_bwd(const float* dout, const float* param, const float* weights, float* dlogits, float* dparam, void* ws,
                        size_t ws_bytes, int B, int L, int D, int S, int H, int W, dcpt_stream_t stream);
/* NHWC PixelUnshuffle(2): x [B][H][W][C] -> y [B][H/2][W/2][4C];  PixelShuffle(2): x [B][H][W][C4] -> y [B][2H][2W][C4/4] */
int dcpt_pixel_unshuffle(const float* x, float* y, int B, int H, int W, int C, dcpt_stream_t stream);
int dcpt_pixel_shuffle(const float* x, float* y, int B, int H, int W, int C4, dcpt_stream_t stream);
/* channel concat of two NHWC maps ([M][Ca], [M][Cb] -> [M][Ca+Cb]) and its inverse */
int dcpt_concat_channels(const float* a, const float* b, float* out, int64_t M, int Ca, int Cb, dcpt_stream_t stream);
int dcpt_split_channels(const float* cat, float* a, float* b, int64_t M, int Ca, int Cb, dcpt_stream_t stream);

/* ---- fused bias + leaky-ReLU (API parity with basicsr/ops/fused_act/src/fused_bias_act.cpp:14-26,
 * kernel fused_bias_act_kernel.cu:20-50): y = act(x + bias[(i / step_b) % size_b]) * scale, act in
 * {1: linear, 3: leaky relu(alpha)}; grad 0: forward, 1: first derivative w.r.t. x using `ref` sign. */
int dcpt_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, int64_t n, int size_b,
                        int64_t step_b, int act, int grad, float alpha, float scale, dcpt_stream_t stream);

/* ---- layout helpers (arbitrary C) ----------------------------------------------------------- */
int dcpt_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, dcpt_stream_t stream);
int dcpt_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, dcpt_stream_t stream);

/* ---- optional launch profiling (bench.py) -------------------------------------------------------
 * While enabled, MFMA GEMM launches are bracketed by HIP events on their own stream: every launch for on = 1, every
 * on-th launch for on > 1 (sampling keeps the perturbation of a timed region small: an event pair costs ~3 us of queue time).
 * dcpt_prof_read waits for them and writes rows of 8 doubles {class id, M, N, K, launches, total ms,
 * algorithmic flops, algorithmic bytes}, one per (class, shape); class id = 16*loaderA + epilogue (NT) or
 * 512 + 8*loaderX + loaderY (TN). */
int dcpt_prof_enable(int on);
int dcpt_prof_read(double* out, int max_rows);

/* ---- weight-gradient side stream ------------------------------------------------------------------
 * dcpt_nafblock_bwd enqueues its four weight-gradient GEMMs (+ slab reductions) on an internal low-priority
 * HIP stream, forked from and joined back into `stream` with events inside the call (callers see ordinary
 * stream semantics).  on = 0 keeps everything on the caller's stream (also: DCPT_SIDE_STREAM=0 in the
 * environment, and automatically while `stream` is being captured into a graph).  Returns the previous setting. */
int dcpt_set_side_stream(int on);

/* ---- optimizer step --------------------------------------------------------------------------------------
 * AdamW (decoupled weight decay) over a LIST of fp32 tensors, the update that closes every training step of the path: replaces
 * torch.optim.AdamW(...).step() as built by basicsr/models/base_model.py:70-93 and called by sr_model.py:118 and
 * degradation_classification_pretrain_model.py:170-173.  Per element, in fp32 and in the order of torch's fused kernel:
 *     p -= lr wd p;  m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
 * `params / grads / exp_avg / exp_avg_sq / numel` are HOST arrays of n entries (device pointers and element counts, tensor i dense in
 * any layout shared by its four buffers; entries with numel 0 are skipped); bias_correction{1,2} = 1 - beta{1,2}^step for the step
 * being taken (step >= 1).  maximize != 0 negates the gradients.  A few launches on `stream` for any n (80 tensors per launch, the
 * pointers travel as kernel arguments: nothing is copied or allocated). */
typedef struct {
    double lr, beta1, beta2, eps, weight_decay;   /* doubles as the host holds them: 1 - beta is formed in double (1 - 0.999 in fp32 is off by 1e-5) */
    double bias_correction1, bias_correction2;
    int maximize;
} dcpt_adamw_hparams;
int dcpt_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numel, const dcpt_adamw_hparams* h, dcpt_stream_t stream);

/* ---- gradient all-reduce (data-parallel step) -------------------------------------------------------
 * replaces what torch DistributedDataParallel does for the reference (basicsr/models/base_model.py:108-115: bucketed
 * all-reduce(SUM) / world of the gradients; :448 the loss reduce) for hosts that drive the collective themselves:
 * in-place `ncclAllReduce(SUM, fp32)` of buf[0..n) over the CALLER-PROVIDED RCCL communicator (an ncclComm_t created by the
 * launcher, one rank per GPU), enqueued on `stream`, followed by buf *= scale (pass 1/world for the mean, 1 to skip).
 * The library does not create communicators or bootstrap ranks, and it does not link RCCL: the entry point is resolved on
 * first use from the copy already loaded in the process (torch's) or the system's librccl.so.1.  Overlap with backward
 * is the caller's choice of stream (a side stream + events, as DDP's bucket hooks do).
 * basicsr/ in this repository keeps the reference's own mechanism (torch DDP over backend "nccl" = RCCL) for the training
 * step; this entry point is the C-ABI form of the same collective (SURVEY 8b minimum export set). */
int dcpt_allreduce_flat(float* buf, size_t n, void* rccl_comm, float scale, dcpt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCPT_HIP_H */
