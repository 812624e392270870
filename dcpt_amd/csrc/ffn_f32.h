// Forward 1 x 1 chains of a narrow-level NAFBlock in fp32 storage as one pass each (ffn_f32.hip), C = 64.
#pragma once
#include "dcpt_common.h"

struct FfnFwdF {
    const float* y;              // [M][C] input of the LayerNorm (FFN: also of the residual)
    const float *lnw, *lnb;      // [C]
    const float *W4, *W5;        // the reference's [2C][C] / [C][C] 1 x 1 conv weights as they are (W5: FFN only)
    const float *b4, *b5, *gamma;   // [2C], [C], [C] (null: 0 / 0 / 1; b5 and gamma: FFN only)
    float* out;                  // FFN: [M][C]  y + (conv5(SimpleGate(conv4(LN(y)) + b4)) + b5) * gamma
    float* v;                    // [M][2C] conv4(LN(y)) + b4  (FFN: may be null in inference; HEAD: the output)
    float *mu, *rstd;            // [M] LayerNorm statistics, or null
    int64_t M;
    float eps;
};
bool ffn_fwd_f32_ok(int C);
int launch_ffn_fwd_f32(const FfnFwdF& p, int C, hipStream_t s);   // FFN
int launch_ln_conv_f32(const FfnFwdF& p, int C, hipStream_t s);   // HEAD: v = conv(LN(y)) + b4

// backward data path of the second half in one pass: dv (operand of conv4's weight-gradient GEMM) and dy = dout + LayerNorm2 backward
struct FfnBwdF {
    const float *dout, *v, *y;   // [M][C], [M][2C], [M][C]
    const float *wT5, *wT4;      // wT5[k][n] = W5[n][k] gamma[n]  ([C][C]),  wT4[c][j] = W4[j][c]  ([C][2C])
    const float* lnw;            // [C]
    float *dv, *dy;              // [M][2C], [M][C]
    float* lnpart;               // [ffn_bwd_f32_waves(M)][2][C]: sum_rows dxn2 * xhat, sum_rows dxn2
    int64_t M;
    float eps;
};
int ffn_bwd_f32_waves(int64_t M);
int launch_ffn_bwd_f32(const FfnBwdF& p, int C, hipStream_t s);
