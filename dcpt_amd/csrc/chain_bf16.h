// Fused 1x1-conv chains of a NAFBlock at the WIDE levels (C = 256 / 512) in bf16 storage (chain_bf16.hip).  The narrow level's chains
// (C = 64) keep their weights in registers (ffn_bf16.hip); here a weight matrix is 0.5 - 1 MB, so the roles are turned around: the
// PIXEL TILE stays on the CU (128 pixels x C channels in LDS) and the weights stream past it, L2 -> registers, every wave its own rows.
#pragma once
#include "bf16.h"

// Second half of the block, reference basicsr/archs/nafnet_arch.py:180-186:
//   out = y + gamma * (conv5(SimpleGate(conv4(LayerNorm2(y)))) + b5)
struct ChainFwdB {
    const bf16_t* y;         // [M][C]
    const float *lnw, *lnb;  // [C]
    const bf16_t* Wf;        // fragment-order weight stream of conv4 + conv5 (chain_wstream_elems(C) elements, launch_wpack_bf16 mode 9)
    const float *b4, *b5, *gamma;
    bf16_t* out;             // [M][C]
    bf16_t* v;               // [M][2C] conv4 output, or null (no backward pass follows)
    bf16_t *xn2, *g;         // [M][C] LayerNorm2(y), SimpleGate(v): operands of the weight-gradient GEMMs, or null (both)
    float *mu, *rstd;        // [M] or null (both)
    int64_t M;
    float eps;
    // conv3 in front of the chain (t2 != null; nafnet_arch.py:174-178): y = inp + beta (conv3(t2 s) + b3) is computed HERE (y is written,
    // not read): t2 [M][C] SimpleGate output, inp [M][C] the block input, W3f the per-image fragment-order streams of W3[n][k] s[img][k]
    // (B x C^2 elements, pack mode 11), b3 / beta [C], P pixels per image (a multiple of 128: a tile lies inside one image)
    const bf16_t *t2, *inp, *W3f;
    const float *b3, *beta;
    int P;
};
bool chain_fwd_bf16_ok(int C, int64_t M);
size_t chain_wstream_elems(int C);   // bf16 elements of Wf (3 C^2), 0 if the width has no chain kernel
int launch_chain_fwd_bf16(const ChainFwdB& p, int C, hipStream_t s);
// First two links of the block on the same kernel, t1 = conv1(LayerNorm1(inp)) + b1 (nafnet_arch.py:169-170): y = inp, Wf = conv1's stream
// (chain_head_wstream_elems(C) = 2 C^2 elements, pack mode 9 with no second matrix), b4 = conv1's bias, v = t1, xn2 / mu / rstd =
// LayerNorm1's output and statistics (null: inference); b5 / gamma / out / g unused.
size_t chain_head_wstream_elems(int C);
int launch_chain_head_bf16(const ChainFwdB& p, int C, hipStream_t s);

// Backward, the middle of the block (nafnet_arch.py:178-180 under autograd): dy = dout + LayerNorm2'(gln; y) and dts = dy (beta W3)^T with
// SCA's per-tile channel sums, one kernel per 128-pixel tile (replaces ln_bwd_bf16 + the conv3^T data-gradient GEMM with the column-dot epilogue).
struct ChainMidB {
    const bf16_t *gln, *y, *dout, *t2;   // [M][C]: gradient of LN2's output, LN2's input, the residual path's gradient, SimpleGate output
    const float *mu, *rstd, *lnw;        // [M], [M], [C]
    const bf16_t* Wf;                    // fragment-order stream of (beta W3)^T (chain_mid_wstream_elems(C) = C^2 elements, pack mode 10)
    bf16_t *dy, *dts;                    // [M][C]
    float* lnpart;                       // [ceil(M / 128)][2][C]: per-tile sum_rows gln xhat, sum_rows gln
    float* dspart;                       // [ceil(M / 128)][C]:   per-tile sum_rows dts t2
    int64_t M;
};
size_t chain_mid_wstream_elems(int C);
int launch_chain_bwd_mid_bf16(const ChainMidB& p, int C, hipStream_t s);

// geometry shared with the pack (bf16_ops.hip, mode 9): 8 waves per block, wave w owns gate channels [w C/8, (w+1) C/8)
constexpr int CHAIN_NW = 8;
