// fp32 MFMA GEMM family for NHWC activations (rows = pixels, columns = channels).
//
//   NT : C[m][n]  = sum_k A'(m,k) * Bw[n][k]        (1x1-conv forward; dgrad with Bw = W^T)
//   TN : G[n][k]  = sum_m X'(m,n) * Y'(m,k)          (wgrad, split over m into slabs)
//
// Both run on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 peak of gfx950).
// A'/X'/Y' are produced by fused operand loaders, C is consumed by fused epilogues.
#pragma once
#include "dcpt_common.h"

enum GemmALoad { A_PLAIN = 0, A_LN = 1, A_SCALE = 2, A_SG = 3, A_GATHER = 4, A_CONV3 = 5, A_LNBF = 6 };
enum GemmEpi { E_PLAIN = 0, E_BIAS = 1, E_RESID = 2, E_SGBWD = 3, E_SCATTER = 4, E_SCATTER_ADD = 5, E_ADDSCALED = 6, E_MUL = 7, E_BIASGATE = 8, E_DOTCOL = 9, E_LNBWD = 10, E_RESIDLN = 11, E_LNBWD2 = 12 };

struct GemmNT {
    const float* A;   // [M][lda]   (A_SG: 2K columns; A_GATHER: fine NHWC image, see g*)
    const float* Bw;  // [N][K] row-major (K contiguous)
    float* C;         // [M][ldc]   (E_SGBWD: 2N columns; E_SCATTER*: fine NHWC image)
    int64_t M;
    int N, K;
    int lda, ldc;
    // A_LN: a = (x - mu[m]) * rstd[m] * lnw[k] + lnb[k];   A_LNBF (Restormer BiasFree): a = x * rstd[m] * lnw[k]
    const float* mu;
    const float* rstd;
    const float* lnw;
    const float* lnb;
    // A_SCALE: a = x * simg[(m / P) * K + k]   (P = pixels per image)
    const float* simg;
    int P;
    // A_GATHER / E_SCATTER*: coarse grid gH x gW per image, fine image is 2gH x 2gW with gC channels;
    // column index = (2*i + j) * gC + ch  <->  fine pixel (2h+i, 2w+j), channel ch
    // A_CONV3 (implicit GEMM of a dense 3x3, zero pad 1): image gH x gW with gC channels, K = 9*gC,
    // column index = tap * gC + ch, tap = 3*ky + kx  <->  pixel (h+ky-1, w+kx-1)
    int gH, gW, gC;
    // epilogues
    const float* bias;    // [N] (may be null)
    const float* res;     // E_RESID / E_ADDSCALED: [M][ldres]; E_SCATTER_ADD: fine image like C
    const float* cscale;  // E_RESID: C = res + (acc+bias)*cscale[n] (null = 1); E_ADDSCALED: C = acc + cscale[n]*res
                          // E_MUL: C = (acc+bias)*res
    const float* aux;     // E_SGBWD: v [M][2N]
    // E_LNBWD (N <= 128, one tile spans the row): the accumulator row g is the gradient of a LayerNorm output whose input was
    // res (x); C = rstd*(g*lnw - mean_c(g*lnw) - xhat*mean_c(g*lnw*xhat)) + aux (the residual gradient, may be null), and
    // colpart[m / 128][0][n] = sum_rows g*xhat (-> dweight), colpart[m / 128][1][n] = sum_rows g (-> dbias); uses mu, rstd, lnw
    // E_RESIDLN (N <= 128): C as E_RESID AND the LayerNorm of that output row: ln_out[m][n] = (C - mean) * rstd * lnw[n] + lnb[n],
    // ln_mu[m], ln_rstd[m] (biased variance about the mean, eps inside the sqrt) -- the next LayerNorm's forward pass
    float* ln_out;
    float* ln_mu;
    float* ln_rstd;
    float ln_eps;
    // E_LNBWD2 (any N): LayerNorm backward WITHOUT a row reduction in the epilogue.  The two row sums of the backward formula,
    //   C s1 = sum_c g_c w_c  and  C s2 = sum_c g_c w_c xhat_c   (g = this GEMM's accumulator row = dZ W, Z = W (w * xhat + b) + bz),
    // are linear in dZ:  C s1 = dZ . u,  C s2 = dZ . (Z - cvec)  with u_j = sum_c W[j][c] w_c,  cvec_j = bz_j + sum_c W[j][c] b_c,
    // so whoever PRODUCED dZ (the SimpleGate-backward epilogue E_SGBWD, the fused depthwise backward) already left them as
    // per-row partials rowpart[m][np][2] (np column chunks).  The epilogue is then elementwise:
    //   C = rstd (g w - xhat s2 - s1) + aux,   colpart[m / 128][0][n] = sum_rows g xhat,  colpart[m / 128][1][n] = sum_rows g.
    // E_SGBWD writes such partials when rowpart != null (np = its number of column tiles; uvec / cvec have 2N entries).
    float* rowpart;
    int rowparts;         // E_LNBWD2: np of the producer
    const float* uvec;    // E_SGBWD with rowpart
    const float* cvec;
    float* colpart;       // E_DOTCOL: C = acc as E_PLAIN AND colpart[m / 128][n] = sum over the tile's 128 rows of acc * res[m][n]
                          // (fixed order; SCA backward's per-image channel sums come out of the producing GEMM)
    float* gate;          // E_BIASGATE (N = 2*Ch, SimpleGate input): C = acc + bias as usual AND gate[m][c] = C[m][c] * C[m][Ch + c],
                          // [M][Ch] row-major; a block's tile then holds BN/2 columns of each half
    int ldres;            // row stride of res (0 = ldc)
    // batching: grid.y = nb1*nb2 problems; pointer offsets b1*s?1 + b2*s?2 (elements)
    int nb1, nb2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2, sS1, sS2;
};

int launch_gemm_nt(const GemmNT& p, int aload, int epi, hipStream_t stream);
// opt-in split-operand mode (gemm_x3.hip, dcpt_set_gemm_x3): fp32-class results on the bf16 matrix pipe; launch_gemm_nt routes eligible
// launches there while the mode is on
bool gemm_nt_x3_ok(const GemmNT& p, int aload, int epi);
int launch_gemm_nt_x3(const GemmNT& p, int aload, int epi, hipStream_t stream);
// number of column tiles the launch of this problem uses (= row partials per row written by E_SGBWD with rowpart)
int gemm_nt_tiles_n(const GemmNT& p, int aload, int epi);

struct GemmTN {
    const float* X;  // [M][ldx]  -> rows of G (n index)
    const float* Y;  // [M][ldy]  -> cols of G (k index)
    float* slab;     // [batch][splits][N][K]
    float* colsum;   // [splits * gemm_tn_tiles_k(N, K)][N] partial column sums of X' (may be null)
    int64_t M;
    int N, K;
    int ldx, ldy;
    int splits;        // number of m-chunks (grid.y)
    int64_t rows_per_split;  // multiple of 32
    // Y loader (GemmALoad kinds)
    const float* mu;
    const float* rstd;
    const float* lnw;
    const float* lnb;
    const float* simg;
    int P;
    int gH, gW, gC;  // gather geometry for whichever operand is gathered
    // batching: grid.y = nb1*nb2 problems; X/Y offsets b1*s?1 + b2*s?2; slab of batch z at z*splits*N*K
    int nb1, nb2;
    int64_t sX1, sX2, sY1, sY2;
};

// xload in {A_PLAIN, A_GATHER}; yload any GemmALoad kind
int launch_gemm_tn(const GemmTN& p, int xload, int yload, hipStream_t stream);
// choose a split count / rows_per_split for a TN problem
void gemm_tn_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split);
// same, with every split inside one image of P pixels; false if no such plan is near the occupancy target
bool gemm_tn_plan_images(int64_t M, int N, int K, int P, int* splits, int64_t* rows_per_split);
// number of k-tile columns of the TN grid for this shape (= partial colsum rows per split)
int gemm_tn_tiles_k(int N, int K);
// opt-in split-operand mode for the weight-gradient GEMM (gemm_x3.hip)
bool gemm_tn_x3_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split);   // false: not a shape / size for the mode
bool gemm_tn_x3_ok(const GemmTN& p, int xload, int yload);
int launch_gemm_tn_x3(const GemmTN& p, int fp32_tiles_k, hipStream_t stream);
