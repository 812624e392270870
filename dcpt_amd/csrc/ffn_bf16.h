// Fused LayerNorm2 -> conv4 -> SimpleGate -> conv5 -> residual of a NAFBlock in bf16 storage for the narrow levels (ffn_bf16.hip).
#pragma once
#include "bf16.h"

struct FfnFwdB {
    const bf16_t* y;      // [M][C] the block's mid-point (input of LayerNorm2 and of the residual)
    const float *lnw, *lnb;      // [C]
    const bf16_t *W4, *W5;       // bf16 operand copies [2C][C], [C][C] (k contiguous)
    const float *b4, *b5, *gamma;
    bf16_t* out;          // [M][C]  y + (conv5(SG(conv4(LN2(y)))) + b5) * gamma
    bf16_t* v;            // [M][2C] conv4 output (saved for backward) or null (inference)
    bf16_t *xn2, *g;      // [M][C] LN2(y), SimpleGate(v): both or neither (null: the fused backward recomputes them)
    float *mu, *rstd;     // [M] LayerNorm statistics or null
    int64_t M;
    float eps;
};
bool ffn_fwd_bf16_ok(int C);
int launch_ffn_fwd_bf16(const FfnFwdB& p, int C, hipStream_t s);
// the first two links only, t1 = conv1(LN1(inp)) + b1:  y = inp, W4 / b4 = conv1, v = t1, xn2 = LN1(inp) (mu / rstd optional)
int launch_ln_conv_bf16(const FfnFwdB& p, int C, hipStream_t s);

struct FfnBwdB {
    const bf16_t *dout, *v, *y;   // [M][C], [M][2C], [M][C]
    const bf16_t *wT5, *wT4;      // dgrad operand copies: wT5[k][n] = W5[n][k] gamma[n]  ([C][C]),  wT4[c][j] = W4[j][c]  ([C][2C])
    const float* lnw;             // [C]
    bf16_t *dv, *dy;              // [M][2C] (operand of conv4's weight-gradient GEMM; null: not written), [M][C] = dout + LayerNorm2 backward
    float* lnpart;                // [ffn_bwd_bf16_waves(M)][2][C]: sum_rows dxn2 * xhat, sum_rows dxn2
    int64_t M;
    float eps;
};
int ffn_bwd_bf16_waves(int64_t M);   // rows of lnpart the launch writes (one per wave)
int launch_ffn_bwd_bf16(const FfnBwdB& p, int C, hipStream_t s);
// the last two links only, dx = dres + LN'(dz W; x):  v = dz [M][2C], wT4 = W^T, y = x, dout = dres, dy = dx
int launch_conv_ln_bwd_tail_bf16(const FfnBwdB& p, int C, hipStream_t s);

// weight gradients of conv5 and conv4 of the same half, the operands (gate, LN2(y), dv) recomputed per 32 pixels: one fp32 slab and one
// column-sum row per wave (ffn_bwd_bf16_waves(M) of them) for launch_wgrad_reduce (splits = waves, one partial column-sum row per split)
struct FfnWgB {
    const bf16_t *dout, *v, *y;
    const bf16_t* wT5;            // as FfnBwdB
    const float *lnw, *lnb;
    float *g5, *g4;               // [waves][C][C]: sum_m dout[m][n] g[m][k];  [waves][2C][C]: sum_m dv[m][j] LN2(y)[m][k]
    float *cs5, *cs4;             // [waves][C]: sum_m dout[m][n];  [waves][2C]: sum_m dv[m][j]
    int64_t M;
    float eps;
};
int launch_ffn_wgrad_bf16(const FfnWgB& p, int C, hipStream_t s);
