// bf16-storage path (BASELINE.json configs[2]): activations and saved tensors are bfloat16 in HBM, every kernel computes in
// fp32 (MFMA accumulators, LayerNorm statistics, reductions), parameters and their gradients stay fp32.  A bf16 value is the
// upper half of an fp32; conversion to bf16 is round-to-nearest-even (v_cvt_pk_bf16_f32), conversion back is a shift.
#pragma once
#include "bufops.h"
#include "dcpt_common.h"

typedef uint16_t bf16_t;   // raw storage
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef __HIPCC__
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float lo, float hi) {   // two fp32 -> {bf16 lo, bf16 hi}, RNE
    floatx2 v;
    v.x = lo;
    v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_round(float x) { return bf_lo(bf_pack(x, 0.f)); }   // fp32 -> bf16 -> fp32

__device__ __forceinline__ float4 bf4_unpack(u32x2 w) { return make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y)); }
__device__ __forceinline__ u32x2 bf4_pack(float4 f) {
    u32x2 w;
    w.x = bf_pack(f.x, f.y);
    w.y = bf_pack(f.z, f.w);
    return w;
}
// 4 consecutive bf16 (8 bytes) through a range-checked buffer window (sentinel offsets load 0 / drop the store)
__device__ __forceinline__ float4 bbuf_ld4(rsrc_t r, uint32_t voff) {
    return bf4_unpack(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0)));
}
__device__ __forceinline__ void bbuf_st4(rsrc_t r, uint32_t voff, float4 f) {
    __builtin_amdgcn_raw_buffer_store_b64(bf4_pack(f), r, voff, 0, 0);
}
// 8 consecutive bf16 (16 bytes)
struct f8 {
    float4 lo, hi;
};
__device__ __forceinline__ f8 bbuf_ld8(rsrc_t r, uint32_t voff) {
    const u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    f8 o;
    o.lo = make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y));
    o.hi = make_float4(bf_lo(w.z), bf_hi(w.z), bf_lo(w.w), bf_hi(w.w));
    return o;
}
__device__ __forceinline__ void bbuf_st8(rsrc_t r, uint32_t voff, f8 v) {
    u32x4 w;
    w.x = bf_pack(v.lo.x, v.lo.y);
    w.y = bf_pack(v.lo.z, v.lo.w);
    w.z = bf_pack(v.hi.x, v.hi.y);
    w.w = bf_pack(v.hi.z, v.hi.w);
#ifdef DCPT_ABL_NOSTORE   // ablation builds (tools/build_variant.sh): the store is issued but dropped by the range check
    voff |= ROW_SENT;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
}
// 8 fp32 rounded to bf16 (RNE): the packed words `w` (what a store writes) and the values a reload of them gives
__device__ __forceinline__ f8 f8_round_bf16(f8 v, u32x4& w) {
    w.x = bf_pack(v.lo.x, v.lo.y);
    w.y = bf_pack(v.lo.z, v.lo.w);
    w.z = bf_pack(v.hi.x, v.hi.y);
    w.w = bf_pack(v.hi.z, v.hi.w);
    f8 o;
    o.lo = make_float4(bf_lo(w.x), bf_hi(w.x), bf_lo(w.y), bf_hi(w.y));
    o.hi = make_float4(bf_lo(w.z), bf_hi(w.z), bf_lo(w.w), bf_hi(w.w));
    return o;
}
__device__ __forceinline__ void bbuf_st8_raw(rsrc_t r, uint32_t voff, u32x4 w) {
#ifdef DCPT_ABL_NOSTORE
    voff |= ROW_SENT;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
}
__device__ __forceinline__ f8 f8_zero() { return f8{f4_zero(), f4_zero()}; }
__device__ __forceinline__ f8 f8_ld(const float* p) { return f8{ldg4(p), ldg4(p + 4)}; }   // 8 fp32 (parameters)
__device__ __forceinline__ f8 f8_add(f8 a, f8 b) { return f8{f4_add(a.lo, b.lo), f4_add(a.hi, b.hi)}; }
__device__ __forceinline__ f8 f8_mul(f8 a, f8 b) { return f8{f4_mul(a.lo, b.lo), f4_mul(a.hi, b.hi)}; }
__device__ __forceinline__ f8 f8_fma(f8 a, f8 b, f8 c) { return f8{f4_fma(a.lo, b.lo, c.lo), f4_fma(a.hi, b.hi, c.hi)}; }
__device__ __forceinline__ float f8_sum(f8 a) { return f4_sum(a.lo) + f4_sum(a.hi); }
#endif

// ---- launchers of the bf16 kernels (gemm_bf16.hip, bf16_ops.hip) ----------------------------------------------------------
enum GemmEpiB { EB_PLAIN = 0, EB_BIAS = 1, EB_RESID = 2, EB_SGBWD = 3, EB_BIASGATE = 4, EB_DOTCOL = 5, EB_LNBWD2 = 6, EB_SCATTER = 7, EB_SCATTER_ADD = 8,
                EB_LNFWD = 9, EB_LNBWDM = 10 };

// C[m][n] = sum_k A[m][k] * Bw[n][k]  on v_mfma_f32_32x32x16_bf16; A, Bw, C, res, aux, gate bf16; bias / cscale / colpart fp32.
struct GemmNTB {
    const bf16_t* A;    // [M][lda]
    const bf16_t* Bw;   // [N][K], K contiguous, K % 8 == 0
    bf16_t* C;          // [M][ldc]  (EB_SGBWD: 2N columns)
    int64_t M;
    int N, K, lda, ldc;
    const float* bias;      // [N] or null
    const bf16_t* res;      // EB_RESID: C = res + (acc + bias) * cscale[n];  EB_DOTCOL: colpart[m/128][n] = sum_rows acc * res
    int ldres;              // 0 = ldc
    const float* cscale;    // [N] or null (= 1)
    const bf16_t* aux;      // EB_SGBWD: v [M][2N]:  C[:, n] = acc * v[:, N + n],  C[:, N + n] = acc * v[:, n]
    bf16_t* gate;           // EB_BIASGATE (N = 2 Ch): C = acc + bias AND gate[m][c] = C[m][c] * C[m][Ch + c]   ([M][Ch])
    float* colpart;         // EB_DOTCOL
    // LayerNorm backward with the row sums supplied by the producer of dZ (gemm.h, E_LNBWD2 -- in bf16 the GEMMs have MFMA slack,
    // so this is a pure saving of two tensor passes and one launch per LayerNorm):
    //   EB_SGBWD with rowpart != null also writes rowpart[m][np][2] = per-tile partials of  dZ . uvec  and  dZ . (Z - cvec)
    //   (np = column tiles of the launch, Z = aux, uvec / cvec fp32 [2N]);
    //   EB_LNBWD2:  C = rstd (g w - xhat s2 - s1) + aux,  xhat from res (the LayerNorm input), s1 / s2 = sum of the `rowparts`
    //   partials / N;  colpart[m / 128][0][n] = sum_rows g xhat,  colpart[m / 128][1][n] = sum_rows g.
    float* rowpart;
    int rowparts;
    const float *uvec, *cvec, *mu, *rstd, *lnw;
    // grid.y = nb independent problems (one per image): element offsets b * s?
    int nb;
    int64_t sA, sB, sC, sR;
    // conv3 != 0: implicit GEMM of a dense 3x3 (zero pad 1): A is the NHWC image [.][gH][gW][gC], K = 9 * gC, column index
    // = tap * gC + ch, tap = 3 ky + kx  <->  pixel (h + ky - 1, w + kx - 1); gC % 8 == 0 (a 16-byte chunk never straddles taps)
    int conv3, gH, gW, gC;
    // gather2 != 0 (the 2x2 stride-2 conv and the pixel-shuffle backward, as gemm.h's A_GATHER): A is the FINE NHWC image
    // [.][2 gH][2 gW][gC], a row is a coarse pixel (b, h, w) of the gH x gW grid, K = 4 * gC, column (2 i + j) * gC + ch  <->  fine
    // pixel (2 h + i, 2 w + j), channel ch.  EB_SCATTER / EB_SCATTER_ADD: C (and res) are such a fine image, N = 4 * gC, the same
    // column map; the adding form computes C = acc + res.   gC % 8 == 0.
    int gather2;
    // The classifier head's channels-first LayerNorm inside the conv GEMMs (degrad_classify_arch.py:17-44, :69-103, :227-243), for N <= one
    // column tile (N <= 128, or N == 256 on the 256-row kernel), rows dense (ldc == N):
    //   EB_LNFWD :  C = z = bf16(acc);  y2 = [relu](LN(z) * lnw + lnb [+ res]);  mu_out / rstd_out [M]  -- statistics two-pass in fp32 over the
    //               ROUNDED z, lane -> channel map and reduction tree of ln_fwd_bf16_kernel: bit-identical to GEMM + launch_ln_act_fwd_bf16
    //   EB_LNBWDM:  acc = the gradient g of an LN'd, [ReLU'd] tensor;  g = bf16(acc) masked by the ReLU (ymask > 0 where ymask is given, else --
    //               relu != 0 -- by the sign of the forward's own fma(xhat, lnw, lnb) recomputed from z: one tensor read less), optionally stored to y2;  C = dz = rstd (g w - xhat mean(g w xhat) - mean(g w)),  xhat from aux (the LN input z) / mu / rstd;
    //               colpart[m / 128][0][n] = sum_rows g xhat, [1][n] = sum_rows g  (-> LN weight / bias gradients)
    const float* lnb;
    float *mu_out, *rstd_out;
    bf16_t* y2;
    const bf16_t* ymask;
    int relu;
    float eps;
};
// N (= row length) for which the two LayerNorm epilogues exist on a launch of M rows; conv3: the implicit 3 x 3 form of the GEMM
bool gemm_nt_bf16_ln_epi_ok(int64_t M, int N, int K, int conv3, int gC);
int launch_gemm_nt_bf16(const GemmNTB& p, int epi, hipStream_t s);
// 256 x 256-tile kernel for the wide levels (gemm_bf16_256.hip); launch_gemm_nt_bf16 routes eligible launches to it
bool gemm_nt_bf16_256_ok(const GemmNTB& p, int epi, int min_tiles = 192);   // min_tiles: fill most of the 256 CUs
int launch_gemm_nt_bf16_256(const GemmNTB& p, int epi, hipStream_t s);
// 512 x 128 tiles of the same kernel for N == 128 (the head's stage-0 convs); launch_gemm_nt_bf16 routes eligible launches to it
bool gemm_nt_bf16_tall_ok(const GemmNTB& p, int epi, int min_tiles = 192);
int launch_gemm_nt_bf16_tall(const GemmNTB& p, int epi, hipStream_t s);
int gemm_nt_bf16_tiles_n(const GemmNTB& p, int epi);   // column tiles of the launch (row partials per row written by EB_SGBWD)

// G[n][k] = sum_m X[m][n] * Y[m][k]  (weight gradients): X, Y bf16 row-major, fp32 slabs [splits][N][K] + column sums of X as in
// the fp32 TN kernel (gemm.h); operands are transposed on the way out of LDS by ds_read_b64_tr_b16.
struct GemmTNB {
    const bf16_t* X;
    const bf16_t* Y;
    float* slab;
    float* colsum;   // [splits * tiles_k][N] or null
    int64_t M;
    int N, K, ldx, ldy;
    int splits;
    int64_t rows_per_split;
    // yconv != 0: Y is gathered like GemmNTB's conv3 operand (weight gradient of a dense 3x3): K = 9 * gC
    // xg2 / yg2 != 0: X / Y is the fine image gathered in 2x2 cells like GemmNTB's gather2 operand (N resp. K = 4 * gC)
    int yconv, gH, gW, gC;
    int xg2, yg2;
};
int launch_gemm_tn_bf16(const GemmTNB& p, hipStream_t s);
int gemm_tn_bf16_tiles_k(int N, int K);
void gemm_tn_bf16_plan(int64_t M, int N, int K, int* splits, int64_t* rows_per_split);
bool gemm_tn_bf16_plan_images(int64_t M, int N, int K, int P, int* splits, int64_t* rows_per_split);

// ---- grouped weight-gradient GEMM with 256 x 256 tiles + the finisher (gemm_tn_bf16_256.hip) ------------------------------
// G_p[n][k] = sum_m X_p[m][n] * Y_p[m][k] for up to TNG_MAX problems in one launch; N, K multiples of 256; every (problem, tile,
// pixel range) is one block, partial sums go to fp32 slabs in the accumulator layout ([splits][tiles][65536], read back only by
// launch_wgrad_finish), column sums of X to colsum[splits][N] (optional).
constexpr int TNG_MAX = 4;
struct TnProb {
    const bf16_t* X;
    const bf16_t* Y;
    float* slab;
    float* colsum;
    int64_t M;
    int N, K, ldx, ldy;
    // filled by gemm_tn_bf16_256_plan: `splits` blocks per tile of rows_per_split pixels each; seg_rows > 0: a block's pixels are cut into
    // pieces of seg_rows (one image each; rows_per_split is a multiple) with ONE PARTIAL-SUM SLOT PER PIECE, so that the finisher can
    // weight every image's sum on its own; `slots` = partial sums per tile (= splits, or the number of pieces)
    int splits, tiles_k, blk0, seg_rows, slots;
    int64_t rows_per_split;
    // yconv != 0 (every problem of the launch, or none): Y is the NHWC image [.][gH][gW][gC] gathered like GemmNTB's conv3 operand -- the
    // weight gradient of a dense 3 x 3, K = 9 * gC, column tap * gC + ch <-> pixel (h + ky - 1, w + kx - 1); gC % 128 == 0, no seg_rows
    int yconv, gH, gW, gC;
};
struct GemmTNG {
    TnProb p[TNG_MAX];
    int n;
    // XCD-aligned block order (filled by gemm_tn_bf16_256_plan; xcd_slots == 0: blocks in problem / pixel-range / tile order through
    // xcd_remap).  The tiles of one pixel range share their operand columns, and they share them through ONE XCD's L2 only: a UNIT is
    // min(tiles, 4) consecutive tiles of a pixel range, a problem's units in order are dealt to the 8 XCDs in contiguous runs --
    // xq[i] packs, 6 bits each, the first unit of problem i on XCD 0..7 and (entry 8) its unit count; hardware block b runs slot b >> 3
    // of XCD b & 7 (slots count through the problems' runs; past them the block is idle); the grid is 8 * xcd_slots blocks.
    int xcd_slots;
    int only;   // diagnostic builds (DCPT_TUNING): >= 0 = only this problem's blocks run; -1 otherwise
    unsigned long long xq[TNG_MAX];
};
bool gemm_tn_bf16_256_ok(int N, int K);
// img_P[i] > 0: problem i needs partial sums per image of img_P[i] pixels (false if that is not possible: no plan is made)
bool gemm_tn_bf16_256_plan(GemmTNG& g, const int* img_P = nullptr, int target_blocks = 256);
size_t gemm_tn_bf16_256_slab_floats(const TnProb& p);
size_t gemm_tn_bf16_256_colsum_floats(const TnProb& p);
int launch_gemm_tn_bf16_256(const GemmTNG& g, hipStream_t s);

// Finisher: one launch for every parameter-gradient reduction of a block.
//   slab jobs: dW[n][k] = rowscale[n] * sum_s (kscale ? kscale[s / ks_div][k] : 1) * slab_s[n][k];
//              dgain[n] = sum_k W[n][k] G[n][k] + wbias[n] cs[n];  dbias[n] = rowscale[n] cs[n]   (cs = sum_s colsum[s][n])
//   cols jobs: out_j[c] = sum_r part[r][j][c]  (mode 0: j < nj <= 2 -> out0 / out1;  mode 1: nj = 10 depthwise taps -> out0[c][9], out1[c])
//   sca job  : dW[n][k] = sum_b ds[b][n] pooled[b][k],  db[n] = sum_b ds[b][n]   (ds == null: none)
constexpr int FIN_MAX_COLS = 4;
struct FinSlab {
    const float *slab, *colsum, *rowscale, *W, *wbias, *kscale;
    float *dW, *dgain, *dbias;
    int N, K, splits, tiles_k, ks_div, cs_rows;   // splits = partial sums per tile; cs_rows = rows of colsum (blocks per tile)
    int conv3;                                    // != 0: K = 9 * Ci packed as tap * Ci + ic, written as dW[n][ic][tap] (a dense 3 x 3's weight layout)
};
struct FinCols {
    const float* part;
    float *out0, *out1;
    int R, nj, C, mode;
};
struct FinSca {
    const float *ds, *pooled;
    float *dW, *db;
    int B, C;
};
struct FinJobs {
    FinSlab slab[TNG_MAX];
    FinCols cols[FIN_MAX_COLS];
    FinSca sca;
    int nslab, ncols;
    int slab_blk0[TNG_MAX], cols_blk0[FIN_MAX_COLS], slab_end, cols_end;   // filled by launch_wgrad_finish
};
int launch_wgrad_finish(FinJobs& j, hipStream_t s);
