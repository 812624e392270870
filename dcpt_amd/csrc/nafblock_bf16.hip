// NAFBlock forward / backward with bf16 STORAGE (BASELINE.json configs[2]; reference basicsr/archs/nafnet_arch.py:165-186; the
// reference itself has no reduced-precision mode -- its AMP / TF32 switches are commented out, basicsr/test.py:26-27 -- so this
// is new behaviour with its own parity bar: tests/test_gpu_bf16.py against oracle/nafnet_oracle.py's bf16 mode, which rounds to
// bf16 at exactly the points listed here).
//
// What is bf16: every [M][.] activation that crosses HBM -- the block's input / output, LN1(inp), t1, t2, y, LN2(y), v,
// SimpleGate(v) and in backward dv, the two LayerNorm input gradients, dy, dts, dt1, dinp -- and the operand copies of the
// weights made per call.  What stays fp32: parameters and parameter gradients, MFMA accumulators (v_mfma_f32_32x32x16_bf16),
// LayerNorm statistics, every reduction (pooling sums, SCA, weight-gradient slabs, LayerNorm / depthwise / bias column sums).
// Epilogue values are computed from the fp32 accumulator and rounded once on store (RNE); SimpleGate(v) is the product of the
// UNROUNDED conv4 outputs, rounded once; SCA's pooling sums the unrounded SimpleGate outputs.
//
// Schedule = the fp32 block's (nafblock.hip) with two changes that remove every register-staged operand loader:
//   * conv3's SCA scale is folded into per-image weights  W3s[b][n][k] = W3[n][k] * s[b][k]  (a [B][C][C] bf16 pack, <= 16 MB),
//     and conv3 runs as one batched GEMM per image -- both operands by LDS-DMA;
//   * LayerNorm always runs as its own bandwidth kernel (no C <= 128 epilogue variants yet).
// Weight gradients go through the transposing-LDS-read TN GEMM into fp32 slabs and the fp32 path's deterministic reducers, on
// the same side stream as the fp32 block.
#include "bf16_ops.h"
#include "ffn_bf16.h"
#include "chain_bf16.h"
#include "side.h"
#include "../../include/dcpt_hip.h"

namespace {

// fused second half of the block at the narrow levels (ffn_bf16.hip)
bool ffn_fused(int C) {
    static const int on = dcpt_tuning("DCPT_FFN_FUSED", 1);
    return on && ffn_fwd_bf16_ok(C);
}

// wide levels: LayerNorm2 -> conv4 -> gate -> conv5 -> residual as one kernel per 128-pixel tile, weights streamed (chain_bf16.hip)
// ffn_chain(C): the width has the kernel (what dcpt_nafblock_bf16_fused_ffn reports: a caller may then leave xn2 / g / mu2 / rstd2 NULL when
// no backward follows); ffn_chain_use(C, M): this launch takes it -- a block owns a whole CU for a 128-pixel tile, so a pixel count that
// fills less than 3/4 of the last round of 256 tiles (tiled inference: 4 tiles of 528 x 528 are 137 tiles at level 3; DCPT at 128 x 128:
// 128) keeps the three-kernel chain, which scales with M (measured: 2K tiled inference 25.9 -> 58 ms with the chain kernel forced).
bool ffn_chain(int C) { return !ffn_fused(C) && chain_fwd_bf16_ok(C, 1 << 20); }
bool ffn_chain_use(int C, int64_t M) { return !ffn_fused(C) && chain_fwd_bf16_ok(C, M); }
bool chain_conv3_on() {
    // OFF: built, parity-green and measured (profiles/r5/bf16_block_level3_kernels_conv3_in_chain_negative.txt): the kernel with conv3 in front
    // takes 120-123 us where the chain kernel + the separate conv3 launch take 86.5 + 35.4 -- one tile per CU serialises the extra tile load, GEMM,
    // two barriers and the accumulator-layout round trip through LDS that the 256 x 256-tile GEMM overlaps across its tiles; same-box step +0.2 ms.
    static const int on = dcpt_tuning("DCPT_CHAIN_CONV3", 0);
    return on != 0;
}
bool chain_mid_on() {
    static const int on = dcpt_tuning("DCPT_CHAIN_MID", 1);
    return on != 0;
}
bool chain_head_on() {
    static const int on = dcpt_tuning("DCPT_CHAIN_HEAD", 1);
    return on != 0;
}

struct FwdWsB {
    float* w2p;
    float* pool_part;
    bf16_t *W1, *W4, *W5, *W3s, *W3, *t2s, *Wf, *Wf1;
    bf16_t *xn2, *g;   // stand-ins for saved->xn2 / g / mu2 / rstd2 where a caller left them NULL (allowed at the chain widths) and the launch
    float *mu2, *rstd2;   // runs the three-kernel chain after all
    int nblk_pool;
    bool scale_act;   // conv3's SCA scale on the activations (t2 * s, one GEMM) instead of in per-image weights: images smaller than 2 C pixels
};

// Per-image weight copies W3 * s[b] are C x C each; an image of P pixels brings P x C activations: below P = 2 C the weight copies (written
// by the pack, read by the GEMM) are the larger traffic and the per-image GEMMs have fewer rows than a tile -- scale the activations instead.
bool conv3_scale_activations(int P, int C) { return P < 2 * C; }

size_t fwd_layout(int B, int H, int W, int C, void* base, size_t bytes, FwdWsB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    DwGeom g{B, H, W, C};
    FwdWsB w{};
    w.nblk_pool = dw_ring_usable(g, 2) ? dw_ring_num_blocks_per_image(g, 2) : dw_num_blocks_per_image_bf16(g);
    w.w2p = a.get<float>((size_t)18 * C);
    w.pool_part = a.get<float>((size_t)B * w.nblk_pool * C);
    w.W1 = a.get<bf16_t>((size_t)2 * C * C);
    w.W4 = a.get<bf16_t>((size_t)2 * C * C);
    w.W5 = a.get<bf16_t>((size_t)C * C);
    w.Wf = ffn_chain(C) ? a.get<bf16_t>(chain_wstream_elems(C)) : nullptr;
    w.Wf1 = ffn_chain(C) ? a.get<bf16_t>(chain_head_wstream_elems(C)) : nullptr;   // conv1 for the LayerNorm1 -> conv1 form of the kernel
    w.xn2 = w.g = nullptr;
    w.mu2 = w.rstd2 = nullptr;
    if (ffn_chain(C) && !ffn_chain_use(C, (int64_t)B * H * W)) {
        w.xn2 = a.get<bf16_t>((size_t)B * H * W * C);
        w.g = a.get<bf16_t>((size_t)B * H * W * C);
        w.mu2 = a.get<float>((size_t)B * H * W);
        w.rstd2 = a.get<float>((size_t)B * H * W);
    }
    w.scale_act = conv3_scale_activations(H * W, C);
    w.W3s = w.scale_act ? nullptr : a.get<bf16_t>((size_t)B * C * C);
    w.W3 = w.scale_act ? a.get<bf16_t>((size_t)C * C) : nullptr;
    w.t2s = w.scale_act ? a.get<bf16_t>((size_t)B * H * W * C) : nullptr;
    if (out) *out = w;
    return a.off;
}

struct BwdWsB {
    bf16_t *wT5, *wT4, *wT3, *wT1, *WfT3;
    bool mid;   // LayerNorm2 backward + the conv3^T data gradient + SCA's sums as one kernel per pixel tile (chain_bf16.hip)
    float* w2p;
    bf16_t *dv, *gln, *dy, *dts, *dt1, *t2s;
    float *slab, *colsum;
    size_t slab_elems, colsum_elems;
    float *lnpart, *lnpart2;
    float *ds_part, *ds, *dpool, *wpart;
    int ln_nblk, nblk_b, ds_slices;
    bool ds_fused;
    // LayerNorm backward inside the dgrad GEMM epilogues (row sums supplied by the producers of dv / dt1: bf16.h EB_LNBWD2)
    float *u4, *c4, *u1, *c1, *rowpart;
    int rp_sg, rp_dw;
    bool lrs;
    float *ffn_g5, *ffn_g4, *ffn_cs5, *ffn_cs4;   // per-wave slabs / column sums of ffn_wgrad_bf16
    float *ffn_part, *ffn_part1;   // LayerNorm2 / LayerNorm1 column partials of the fused narrow-level backward kernels: [waves][2][C] each
    // wide levels (C % 256 == 0): the four weight gradients as two grouped 256 x 256-tile launches + one finisher (gemm_tn_bf16_256.hip)
    bool tn256, conv3_images;
    GemmTNG wg;   // conv5, conv4, conv3, conv1: shapes, pixel ranges, partial-sum / column-sum buffers
};

// The four weight gradients of a wide block as ONE grouped launch (after dt1 exists): the more tiles a launch has, the fewer partial sums
// per output element its 256 blocks make -- 24 tiles at C = 512 are 10 per element and 86 MB of slabs per block backward where two
// launches of 12 tiles made 21 and 132 MB, four single launches 32 and 256 MB -- and the two streams time-slice rather than overlap
// (profiles/r4/), so what counts is the bytes, not how early a weight gradient can start.
bool plan_wg256(int64_t M, int C, int P, GemmTNG* g, bool* conv3_images) {
    if (!gemm_tn_bf16_256_ok(C, C)) return false;
    *g = GemmTNG{};
    g->n = 4;
    const int Ns[4] = {C, 2 * C, C, 2 * C};   // conv5, conv4, conv3, conv1
    for (int i = 0; i < 4; ++i) {
        g->p[i].M = M;
        g->p[i].N = g->p[i].ldx = Ns[i];
        g->p[i].K = g->p[i].ldy = C;
    }
    const int imgP[4] = {0, 0, P, 0};
    *conv3_images = gemm_tn_bf16_256_plan(*g, imgP);   // conv3's partial sums per image: the SCA scale goes into the finisher
    if (!*conv3_images) gemm_tn_bf16_256_plan(*g);     // otherwise its Y operand is the scaled copy t2 * s
    return true;
}

size_t bwd_layout(int B, int H, int W, int C, void* base, size_t bytes, BwdWsB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W;
    DwGeom g{B, H, W, C};
    BwdWsB w{};
    w.wT5 = a.get<bf16_t>((size_t)C * C);
    w.wT4 = a.get<bf16_t>((size_t)2 * C * C);
    w.wT3 = a.get<bf16_t>((size_t)C * C);
    w.wT1 = a.get<bf16_t>((size_t)2 * C * C);
    w.WfT3 = ffn_chain(C) ? a.get<bf16_t>(chain_mid_wstream_elems(C)) : nullptr;
    w.w2p = a.get<float>((size_t)18 * C);
    w.dv = a.get<bf16_t>((size_t)M * 2 * C);
    w.gln = a.get<bf16_t>((size_t)M * C);
    w.dy = a.get<bf16_t>((size_t)M * C);
    w.dts = a.get<bf16_t>((size_t)M * C);
    w.dt1 = a.get<bf16_t>((size_t)M * 2 * C);
    w.t2s = a.get<bf16_t>((size_t)M * C);
    int sp1, sp2, sp3;
    int64_t r1, r2, r3;
    gemm_tn_bf16_plan(M, 2 * C, C, &sp1, &r1);
    gemm_tn_bf16_plan(M, C, C, &sp2, &r2);
    size_t slab1 = (size_t)sp1 * 2 * C * C, slab2 = (size_t)sp2 * C * C;
    size_t cs1 = (size_t)sp1 * gemm_tn_bf16_tiles_k(2 * C, C) * 2 * C, cs2 = (size_t)sp2 * gemm_tn_bf16_tiles_k(C, C) * C;
    if (gemm_tn_bf16_plan_images(M, C, C, P, &sp3, &r3)) {
        const size_t s3 = (size_t)sp3 * C * C, c3 = (size_t)sp3 * gemm_tn_bf16_tiles_k(C, C) * C;
        if (s3 > slab2) slab2 = s3;
        if (c3 > cs2) cs2 = c3;
    }
    w.slab_elems = slab1 > slab2 ? slab1 : slab2;
    w.colsum_elems = cs1 > cs2 ? cs1 : cs2;
    w.tn256 = !ffn_fused(C) && plan_wg256(M, C, P, &w.wg, &w.conv3_images);
    if (w.tn256) {
        w.slab_elems = w.colsum_elems = 0;
        w.slab = w.colsum = nullptr;
        for (int i = 0; i < w.wg.n; ++i) {
            w.wg.p[i].slab = a.get<float>(gemm_tn_bf16_256_slab_floats(w.wg.p[i]));
            w.wg.p[i].colsum = a.get<float>(gemm_tn_bf16_256_colsum_floats(w.wg.p[i]));
        }
    } else {
        w.slab = a.get<float>(w.slab_elems);
        w.colsum = a.get<float>(w.colsum_elems);
    }
    w.ln_nblk = ln_bwd_bf16_num_blocks(M, C);
    w.lnpart = a.get<float>((size_t)w.ln_nblk * 2 * C);
    w.lnpart2 = a.get<float>((size_t)w.ln_nblk * 2 * C);
    w.ds_fused = sca_ds_fused_slices(P) > 0;
    w.ds_slices = w.ds_fused ? sca_ds_fused_slices(P) : sca_ds_num_blocks(P);
    w.ds_part = a.get<float>((size_t)B * w.ds_slices * C);
    w.ds = a.get<float>((size_t)B * C);
    w.dpool = a.get<float>((size_t)B * C);
    {
        const int n1 = dw_num_blocks_per_image_fused_bf16(g), n2 = dw_ring_bwd_usable(g, 2) ? dw_ring_bwd_num_blocks_per_image(g) : 0;
        w.nblk_b = n1 > n2 ? n1 : n2;
    }
    w.wpart = a.get<float>((size_t)B * w.nblk_b * 10 * 2 * C);
    w.u4 = a.get<float>((size_t)2 * C);
    w.c4 = a.get<float>((size_t)2 * C);
    w.u1 = a.get<float>((size_t)2 * C);
    w.c1 = a.get<float>((size_t)2 * C);
    {
        GemmNTB q{};
        q.N = C;
        w.rp_sg = gemm_nt_bf16_tiles_n(q, EB_SGBWD);
    }
    w.rp_dw = dw_fused_row_chunks_bf16(g);
    // OFF by default: exact and 4 tensor passes lighter, but measured slower at EVERY level also in bf16 (same box, A/B: level-0
    // backward 2.68 vs 2.47 ms, level 3 0.64 vs 0.60 ms, the NAFNet-64 step 53.2 vs 51.3 ms).  A GEMM whose epilogue moves more bytes
    // than its k-loop runs at ~3 TB/s -- its loads are not overlapped across tiles -- while the separate LayerNorm kernel streams
    // at 5.3 TB/s; the two passes it saves do not pay for that.
    static const int on = dcpt_tuning("DCPT_LN_ROWSUMS_BF16", 0);
    w.lrs = on && w.rp_sg <= 8 && w.rp_dw <= 8;
    w.rowpart = a.get<float>((size_t)M * (w.rp_sg > w.rp_dw ? w.rp_sg : w.rp_dw) * 2);
    {   // the GEMM-epilogue form of the LayerNorm column partials: [M / 128 tiles][2][C]
        const size_t need = (size_t)cdiv64(M, 128) * 2 * C;
        if (need > (size_t)w.ln_nblk * 2 * C) {
            w.lnpart = a.get<float>(need);
            w.lnpart2 = a.get<float>(need);
        }
    }
    w.mid = ffn_chain_use(C, M) && chain_mid_on() && w.ds_fused && w.tn256 && !w.lrs;
    w.ffn_part = ffn_fused(C) ? a.get<float>((size_t)ffn_bwd_bf16_waves(M) * 2 * C) : nullptr;
    w.ffn_part1 = ffn_fused(C) ? a.get<float>((size_t)ffn_bwd_bf16_waves(M) * 2 * C) : nullptr;
    w.ffn_g5 = w.ffn_g4 = w.ffn_cs5 = w.ffn_cs4 = nullptr;
    if (ffn_fused(C)) {
        const size_t nw = (size_t)ffn_bwd_bf16_waves(M);
        w.ffn_g5 = a.get<float>(nw * C * C);
        w.ffn_g4 = a.get<float>(nw * 2 * C * C);
        w.ffn_cs5 = a.get<float>(nw * C);
        w.ffn_cs4 = a.get<float>(nw * 2 * C);
    }
    if (out) *out = w;
    return a.off;
}

// G[n][k] = sum_m X[m][n] Y[m][k] into fp32 slabs, then dW = rowscale * sum(slabs) (+ gain / bias gradients) -- fp32 reducer of misc.hip
int wgrad_b(const bf16_t* X, int N, const bf16_t* Y, int K, int64_t M, float* slab, float* colsum, const float* rowscale, const float* Wfor_gain,
            const float* wbias, float* dW, float* dgain, float* dbias, hipStream_t s) {
    GemmTNB t{};
    t.X = X; t.ldx = N; t.N = N; t.Y = Y; t.ldy = K; t.K = K; t.M = M; t.slab = slab; t.colsum = colsum;
    gemm_tn_bf16_plan(M, N, K, &t.splits, &t.rows_per_split);
    DCPT_TRY(launch_gemm_tn_bf16(t, s));
    return launch_wgrad_reduce(slab, colsum, t.splits, t.splits * gemm_tn_bf16_tiles_k(N, K), N, K, rowscale, Wfor_gain, wbias, dW, dgain, dbias,
                               WR_PLAIN, s);
}

bool shape_ok(int B, int H, int W, int C) { return B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 1024; }

// The operand copies of a block's weights that depend on the PARAMETERS only (not on activations): bf16 [N][K] copies for the forward
// GEMMs, transposed (and beta / gamma-scaled) copies for the data-gradient GEMMs, the depthwise taps as [9][2C] fp32.  A caller that
// keeps this buffer per block and refreshes it when the parameters change (once per optimizer step) saves the per-call packs:
// 5 launches per block and step (dcpt_nafblock_wpack_bf16 / *_packed entry points).
struct PackB {
    bf16_t *W1, *W4, *W5, *wT5, *wT4, *wT3, *wT1, *W3, *Wf, *Wf1, *WfT3;
    float* w2p;
};
size_t pack_layout(int C, void* base, size_t bytes, PackB* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    PackB k{};
    k.W1 = a.get<bf16_t>((size_t)2 * C * C);
    k.W4 = a.get<bf16_t>((size_t)2 * C * C);
    k.W5 = a.get<bf16_t>((size_t)C * C);
    k.wT5 = a.get<bf16_t>((size_t)C * C);
    k.wT4 = a.get<bf16_t>((size_t)2 * C * C);
    k.wT3 = a.get<bf16_t>((size_t)C * C);
    k.wT1 = a.get<bf16_t>((size_t)2 * C * C);
    k.w2p = a.get<float>((size_t)18 * C);
    k.W3 = a.get<bf16_t>((size_t)C * C);
    k.Wf = ffn_chain(C) ? a.get<bf16_t>(chain_wstream_elems(C)) : nullptr;   // conv4 + conv5 in the chain kernel's fragment order
    k.Wf1 = ffn_chain(C) ? a.get<bf16_t>(chain_head_wstream_elems(C)) : nullptr;   // conv1 likewise
    k.WfT3 = ffn_chain(C) ? a.get<bf16_t>(chain_mid_wstream_elems(C)) : nullptr;   // (beta conv3)^T for the backward kernel
    if (out) *out = k;
    return a.off;
}
constexpr int PACK_JOBS = 12;   // jobs of one block at most
template <typename J>
int pack_jobs(const dcpt_nafblock_params* p, const PackB& k, int C, J& j, int b) {   // the block's nine or ten jobs at j[b ..]; returns their number
    const int C2 = 2 * C;
    if (k.Wf) {
        j.in[b + 9] = p->conv4_w; j.rs[b + 9] = p->conv5_w; j.out[b + 9] = k.Wf; j.N[b + 9] = 3 * C; j.K[b + 9] = C; j.transpose[b + 9] = 9;
        j.in[b + 10] = p->conv1_w; j.rs[b + 10] = nullptr; j.out[b + 10] = k.Wf1; j.N[b + 10] = 2 * C; j.K[b + 10] = C; j.transpose[b + 10] = 9;
        j.in[b + 11] = p->conv3_w; j.rs[b + 11] = p->beta; j.out[b + 11] = k.WfT3; j.N[b + 11] = C; j.K[b + 11] = C; j.transpose[b + 11] = 10;
    }
    j.in[b + 8] = p->conv3_w; j.out[b + 8] = k.W3; j.N[b + 8] = C; j.K[b + 8] = C;
    j.in[b + 0] = p->conv1_w; j.out[b + 0] = k.W1; j.N[b + 0] = C2; j.K[b + 0] = C;
    j.in[b + 1] = p->conv4_w; j.out[b + 1] = k.W4; j.N[b + 1] = C2; j.K[b + 1] = C;
    j.in[b + 2] = p->conv5_w; j.out[b + 2] = k.W5; j.N[b + 2] = C; j.K[b + 2] = C;
    j.in[b + 3] = p->conv5_w; j.out[b + 3] = k.wT5; j.rs[b + 3] = p->gamma; j.N[b + 3] = C;  j.K[b + 3] = C; j.transpose[b + 3] = 1;
    j.in[b + 4] = p->conv4_w; j.out[b + 4] = k.wT4; j.rs[b + 4] = nullptr;  j.N[b + 4] = C2; j.K[b + 4] = C; j.transpose[b + 4] = 1;
    j.in[b + 5] = p->conv3_w; j.out[b + 5] = k.wT3; j.rs[b + 5] = p->beta;  j.N[b + 5] = C;  j.K[b + 5] = C; j.transpose[b + 5] = 1;
    j.in[b + 6] = p->conv1_w; j.out[b + 6] = k.wT1; j.rs[b + 6] = nullptr;  j.N[b + 6] = C2; j.K[b + 6] = C; j.transpose[b + 6] = 1;
    j.in[b + 7] = p->conv2_w; j.out[b + 7] = reinterpret_cast<bf16_t*>(k.w2p); j.N[b + 7] = C2; j.K[b + 7] = 9; j.transpose[b + 7] = 8;
    return k.Wf ? 12 : 9;
}
int pack_all(const dcpt_nafblock_params* p, const PackB& k, int C, hipStream_t s) {
    WpackBJobs j{};
    j.n = pack_jobs(p, k, C, j, 0);
    return launch_wpack_bf16(j, s);
}

}  // namespace

extern "C" size_t dcpt_nafblock_fwd_bf16_ws_bytes(int B, int H, int W, int C) { return fwd_layout(B, H, W, C, nullptr, 0, nullptr); }
extern "C" size_t dcpt_nafblock_bwd_bf16_ws_bytes(int B, int H, int W, int C) { return bwd_layout(B, H, W, C, nullptr, 0, nullptr); }

extern "C" size_t dcpt_nafblock_wpack_bf16_bytes(int C) { return pack_layout(C, nullptr, 0, nullptr); }
// 1: LN2(y), SimpleGate(v) and LN2's statistics are never touched at this width (the fused backward recomputes them);
// 2: they are written for the backward pass but may be NULL (like v) when none follows -- the chain kernel of the wide levels
extern "C" int dcpt_nafblock_bf16_fused_ffn(int C) { return ffn_fused(C) ? 1 : ffn_chain(C) ? 2 : 0; }

// the packs of n blocks (any mix of widths) in ceil(n / 8) launches instead of n: what a network does once per optimizer step
extern "C" int dcpt_nafblock_wpack_bf16_multi(const dcpt_nafblock_params* ps, void* const* packed, const size_t* packed_bytes, const int* C,
                                              int n, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(ps && packed && packed_bytes && C && n >= 1, "nafblock_wpack_bf16_multi: null argument or n=%d", n);
    constexpr int PER = WPACKB_MAX_JOBS_L / PACK_JOBS;
    for (int b0 = 0; b0 < n; b0 += PER) {
        WpackBJobsL j{};
        const int nb = n - b0 < PER ? n - b0 : PER;
        int nj = 0;
        for (int b = 0; b < nb; ++b) {
            const int i = b0 + b;
            DCPT_CHECK_ARG(packed[i] && C[i] > 0 && C[i] % 8 == 0 && C[i] <= 1024, "nafblock_wpack_bf16_multi: block %d: null buffer or bad C=%d", i, C[i]);
            PackB k;
            DCPT_CHECK_ARG(pack_layout(C[i], packed[i], packed_bytes[i], &k) <= packed_bytes[i], "nafblock_wpack_bf16_multi: block %d: buffer too small", i);
            nj += pack_jobs(ps + i, k, C[i], j, nj);
        }
        j.n = nj;
        DCPT_TRY(launch_wpack_bf16(j, (hipStream_t)stream));
    }
    return DCPT_OK;
}

extern "C" int dcpt_nafblock_wpack_bf16(const dcpt_nafblock_params* p, void* packed, size_t packed_bytes, int C, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(p && packed && C > 0 && C % 8 == 0 && C <= 1024, "nafblock_wpack_bf16: null argument or bad C=%d", C);
    PackB k;
    DCPT_CHECK_ARG(pack_layout(C, packed, packed_bytes, &k) <= packed_bytes, "nafblock_wpack_bf16: buffer too small (%zu < %zu)", packed_bytes,
                   pack_layout(C, nullptr, 0, nullptr));
    return pack_all(p, k, C, (hipStream_t)stream);
}

static int nafblock_fwd_bf16_impl(const dcpt_nafblock_params* p, const uint16_t* inp, uint16_t* out, const dcpt_nafblock_saved_bf16* sv,
                                  void* ws, size_t ws_bytes, int B, int H, int W, int C, const void* packed, size_t packed_bytes,
                                  dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && inp && out && sv, "nafblock_fwd_bf16: null argument");
    DCPT_CHECK_ARG(shape_ok(B, H, W, C), "nafblock_fwd_bf16: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0, C <= 1024)", B, H, W, C);
    DCPT_CHECK_ARG(sv->t1 && sv->t2 && sv->y && sv->xn1 && sv->mu1 && sv->rstd1 && sv->pooled && sv->s, "nafblock_fwd_bf16: saved buffers missing");
    const bool infer = !sv->v && !sv->xn2 && !sv->g && !sv->mu2 && !sv->rstd2;   // inference with the fused second half: nothing of it is kept
    // (v alone may be null anywhere: a caller that runs no backward -- the bias+gate epilogue then writes SimpleGate(v) only)
    DCPT_CHECK_ARG((sv->xn2 && sv->g && sv->mu2 && sv->rstd2) || (infer && (ffn_fused(C) || ffn_chain(C))),
                   "nafblock_fwd_bf16: saved xn2 / g / mu2 / rstd2 missing (all of v / xn2 / g / mu2 / rstd2 may be null only where "
                   "dcpt_nafblock_bf16_fused_ffn(C) is 1; v alone may be null when no backward pass follows)");
    FwdWsB w;
    const size_t need = fwd_layout(B, H, W, C, ws, ws_bytes, &w);
    if (need > ws_bytes || ws == nullptr) {
        dcpt_set_error("nafblock_fwd_bf16: workspace too small (%zu < %zu)", ws_bytes, need);
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W;
    const float eps = 1e-6f;
    // operand copies of the weights that do not depend on SCA: the caller's per-block pack, or made here
    WpackBJobs j{};
    if (packed) {
        PackB k;
        DCPT_CHECK_ARG(pack_layout(C, const_cast<void*>(packed), packed_bytes, &k) <= packed_bytes, "nafblock_fwd_bf16: packed weights buffer too small");
        w.W1 = k.W1; w.W4 = k.W4; w.W5 = k.W5; w.w2p = k.w2p; w.Wf = k.Wf; w.Wf1 = k.Wf1;
        if (w.scale_act) w.W3 = k.W3;
    } else {
        j.n = 3;
        if (w.scale_act) {
            j.in[j.n] = p->conv3_w; j.out[j.n] = w.W3; j.N[j.n] = C; j.K[j.n] = C;
            ++j.n;
        }
        if (w.Wf) {
            j.in[j.n] = p->conv4_w; j.rs[j.n] = p->conv5_w; j.out[j.n] = w.Wf; j.N[j.n] = 3 * C; j.K[j.n] = C; j.transpose[j.n] = 9;
            ++j.n;
            j.in[j.n] = p->conv1_w; j.rs[j.n] = nullptr; j.out[j.n] = w.Wf1; j.N[j.n] = 2 * C; j.K[j.n] = C; j.transpose[j.n] = 9;
            ++j.n;
        }
        j.in[0] = p->conv1_w; j.out[0] = w.W1; j.N[0] = 2 * C; j.K[0] = C;
        j.in[1] = p->conv4_w; j.out[1] = w.W4; j.N[1] = 2 * C; j.K[1] = C;
        j.in[2] = p->conv5_w; j.out[2] = w.W5; j.N[2] = C; j.K[2] = C;
        DCPT_TRY(launch_wpack_bf16(j, s));
    }
    GemmNTB g{};
    if (ffn_fused(C)) {   // narrow levels: LayerNorm1 -> conv1 in one pass over the input (ffn_bf16.hip)
        FfnFwdB f{};
        f.y = inp; f.lnw = p->norm1_w; f.lnb = p->norm1_b; f.W4 = w.W1; f.b4 = p->conv1_b; f.v = sv->t1; f.xn2 = sv->xn1; f.mu = sv->mu1;
        f.rstd = sv->rstd1; f.M = M; f.eps = eps;
        DCPT_TRY(launch_ln_conv_bf16(f, C, s));
    } else if (ffn_chain_use(C, M) && chain_head_on()) {   // wide levels: the same pair on the chain kernel (chain_bf16.hip, HEAD form)
        ChainFwdB f{};
        f.y = inp; f.lnw = p->norm1_w; f.lnb = p->norm1_b; f.Wf = w.Wf1; f.b4 = p->conv1_b; f.v = sv->t1; f.xn2 = sv->xn1; f.mu = sv->mu1;
        f.rstd = sv->rstd1; f.M = M; f.eps = eps;
        DCPT_TRY(launch_chain_head_bf16(f, C, s));
    } else {
        DCPT_TRY(launch_ln_fwd_bf16(inp, p->norm1_w, p->norm1_b, sv->xn1, sv->mu1, sv->rstd1, M, C, eps, s));
        g.M = M; g.A = sv->xn1; g.lda = C; g.K = C; g.Bw = w.W1; g.N = 2 * C; g.C = sv->t1; g.ldc = 2 * C; g.bias = p->conv1_b;
        DCPT_TRY(launch_gemm_nt_bf16(g, EB_BIAS, s));
    }
    DwGeom dg{B, H, W, C};
    if (!packed) DCPT_TRY(launch_dw_pack_weights(p->conv2_w, w.w2p, 2 * C, s));
    if (dw_ring_usable(dg, 2)) DCPT_TRY(launch_dw_ring_fwd_bf16(sv->t1, w.w2p, p->conv2_b, sv->t2, w.pool_part, dg, s));
    else DCPT_TRY(launch_dw_fwd_bf16(sv->t1, w.w2p, p->conv2_b, sv->t2, w.pool_part, dg, s));
    DCPT_TRY(launch_sca_fwd(w.pool_part, w.nblk_pool, p->sca_w, p->sca_b, sv->pooled, sv->s, B, C, P, s));
    // wide levels, images of whole 128-pixel tiles: conv3 runs IN FRONT of the chain kernel below (its per-image weights as fragment streams)
    const bool conv3_in_chain = ffn_chain_use(C, M) && chain_conv3_on() && !w.scale_act && P % 128 == 0;
    g = GemmNTB{};
    if (conv3_in_chain) {
        j = WpackBJobs{};
        j.n = 1;
        j.in[0] = p->conv3_w; j.out[0] = w.W3s; j.N[0] = C; j.K[0] = C; j.kscale[0] = sv->s; j.nimg[0] = B; j.transpose[0] = 11;
        DCPT_TRY(launch_wpack_bf16(j, s));
    } else if (w.scale_act) {
        // y = inp + (conv3(t2 * s) + b3) * beta with the scale on the activations (small images): one pass over t2, one GEMM
        DCPT_TRY(launch_scale_rows_bf16(sv->t2, sv->s, w.t2s, M, C, P, s));
        g.M = M; g.A = w.t2s; g.lda = C; g.K = C; g.Bw = w.W3; g.N = C; g.C = sv->y; g.ldc = C; g.bias = p->conv3_b;
        g.res = inp; g.ldres = C; g.cscale = p->beta;
    } else {
        // ... with the per-image scale in the weights, one GEMM problem per image
        j = WpackBJobs{};
        j.n = 1;
        j.in[0] = p->conv3_w; j.out[0] = w.W3s; j.N[0] = C; j.K[0] = C; j.kscale[0] = sv->s; j.nimg[0] = B;
        DCPT_TRY(launch_wpack_bf16(j, s));
        g.M = P; g.A = sv->t2; g.lda = C; g.K = C; g.Bw = w.W3s; g.N = C; g.C = sv->y; g.ldc = C; g.bias = p->conv3_b;
        g.res = inp; g.ldres = C; g.cscale = p->beta;
        g.nb = B; g.sA = (int64_t)P * C; g.sB = (int64_t)C * C; g.sC = (int64_t)P * C; g.sR = (int64_t)P * C;
    }
    if (!conv3_in_chain) DCPT_TRY(launch_gemm_nt_bf16(g, EB_RESID, s));
    if (ffn_fused(C)) {   // narrow levels: LayerNorm2 -> conv4 -> SimpleGate -> conv5 -> residual in one pass over y (ffn_bf16.hip)
        FfnFwdB f{};
        f.y = sv->y; f.lnw = p->norm2_w; f.lnb = p->norm2_b; f.W4 = w.W4; f.W5 = w.W5; f.b4 = p->conv4_b; f.b5 = p->conv5_b; f.gamma = p->gamma;
        // (LN2(y), the gate and the statistics are recomputed by the backward kernels: sv->xn2 / g / mu2 / rstd2 stay unwritten)
        f.out = out; f.v = sv->v; f.M = M; f.eps = eps;
        return launch_ffn_fwd_bf16(f, C, s);
    }
    if (ffn_chain_use(C, M)) {   // wide levels: the same chain per 128-pixel tile with the weights streamed past it (chain_bf16.hip)
        ChainFwdB f{};
        f.y = sv->y; f.lnw = p->norm2_w; f.lnb = p->norm2_b; f.Wf = w.Wf; f.b4 = p->conv4_b; f.b5 = p->conv5_b; f.gamma = p->gamma;
        f.out = out; f.v = sv->v; f.xn2 = sv->xn2; f.g = sv->g; f.mu = sv->mu2; f.rstd = sv->rstd2; f.M = M; f.eps = eps;
        if (conv3_in_chain) {   // y = inp + beta (conv3(t2 s) + b3) is produced by the same kernel
            f.t2 = sv->t2; f.inp = inp; f.W3f = w.W3s; f.b3 = p->conv3_b; f.beta = p->beta; f.P = P;
        }
        return launch_chain_fwd_bf16(f, C, s);
    }
    // (inference at a chain width with a pixel count the chain kernel is not used for: LN2(y) / the gate / the statistics live in the workspace)
    bf16_t* const xn2 = sv->xn2 ? sv->xn2 : w.xn2;
    bf16_t* const gt = sv->g ? sv->g : w.g;
    DCPT_TRY(launch_ln_fwd_bf16(sv->y, p->norm2_w, p->norm2_b, xn2, sv->mu2 ? sv->mu2 : w.mu2, sv->rstd2 ? sv->rstd2 : w.rstd2, M, C, eps, s));
    g = GemmNTB{};
    g.M = M; g.A = xn2; g.lda = C; g.K = C; g.Bw = w.W4; g.N = 2 * C; g.C = sv->v; g.ldc = 2 * C; g.bias = p->conv4_b; g.gate = gt;
    DCPT_TRY(launch_gemm_nt_bf16(g, EB_BIASGATE, s));
    g = GemmNTB{};
    g.M = M; g.A = gt; g.lda = C; g.K = C; g.Bw = w.W5; g.N = C; g.C = out; g.ldc = C; g.bias = p->conv5_b; g.res = sv->y; g.ldres = C;
    g.cscale = p->gamma;
    return launch_gemm_nt_bf16(g, EB_RESID, s);
}

extern "C" int dcpt_nafblock_fwd_bf16(const dcpt_nafblock_params* p, const uint16_t* inp, uint16_t* out, const dcpt_nafblock_saved_bf16* sv,
                                      void* ws, size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    return nafblock_fwd_bf16_impl(p, inp, out, sv, ws, ws_bytes, B, H, W, C, nullptr, 0, stream);
}
extern "C" int dcpt_nafblock_fwd_bf16_packed(const dcpt_nafblock_params* p, const void* packed, size_t packed_bytes, const uint16_t* inp,
                                             uint16_t* out, const dcpt_nafblock_saved_bf16* sv, void* ws, size_t ws_bytes, int B, int H, int W,
                                             int C, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(packed, "nafblock_fwd_bf16_packed: null packed weights");
    return nafblock_fwd_bf16_impl(p, inp, out, sv, ws, ws_bytes, B, H, W, C, packed, packed_bytes, stream);
}

static int nafblock_bwd_bf16_impl(const dcpt_nafblock_params* p, const dcpt_nafblock_grads* gr, const uint16_t* inp,
                                  const dcpt_nafblock_saved_bf16* sv, const uint16_t* dout, uint16_t* dinp, void* ws, size_t ws_bytes, int B,
                                  int H, int W, int C, const void* packed, size_t packed_bytes, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && gr && inp && sv && dout && dinp, "nafblock_bwd_bf16: null argument");
    DCPT_CHECK_ARG(shape_ok(B, H, W, C), "nafblock_bwd_bf16: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    BwdWsB w;
    const size_t need = bwd_layout(B, H, W, C, ws, ws_bytes, &w);
    if (need > ws_bytes || ws == nullptr) {
        dcpt_set_error("nafblock_bwd_bf16: workspace too small (%zu < %zu)", ws_bytes, need);
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W;
    const int C2 = 2 * C;
    DwGeom dg{B, H, W, C};

    // transposed (and gain-scaled) bf16 weights of the four dgrad GEMMs + the depthwise [9][2C] pack: the caller's, or made here
    if (packed) {
        PackB k;
        DCPT_CHECK_ARG(pack_layout(C, const_cast<void*>(packed), packed_bytes, &k) <= packed_bytes, "nafblock_bwd_bf16: packed weights buffer too small");
        w.wT5 = k.wT5; w.wT4 = k.wT4; w.wT3 = k.wT3; w.wT1 = k.wT1; w.w2p = k.w2p; w.WfT3 = k.WfT3;
    } else {
        WpackBJobs j{};
        j.n = 4;
        if (w.mid) {
            j.in[4] = p->conv3_w; j.rs[4] = p->beta; j.out[4] = w.WfT3; j.N[4] = C; j.K[4] = C; j.transpose[4] = 10;
            j.n = 5;
        }
        j.in[0] = p->conv5_w; j.out[0] = w.wT5; j.rs[0] = p->gamma; j.N[0] = C;  j.K[0] = C; j.transpose[0] = 1;
        j.in[1] = p->conv4_w; j.out[1] = w.wT4; j.rs[1] = nullptr;  j.N[1] = C2; j.K[1] = C; j.transpose[1] = 1;
        j.in[2] = p->conv3_w; j.out[2] = w.wT3; j.rs[2] = p->beta;  j.N[2] = C;  j.K[2] = C; j.transpose[2] = 1;
        j.in[3] = p->conv1_w; j.out[3] = w.wT1; j.rs[3] = nullptr;  j.N[3] = C2; j.K[3] = C; j.transpose[3] = 1;
        DCPT_TRY(launch_wpack_bf16(j, s));
        DCPT_TRY(launch_dw_pack_weights(p->conv2_w, w.w2p, C2, s));
    }

    const int ln_tiles = (int)cdiv64(M, 128);
    if (w.lrs) {   // u = W w_ln, c = b_conv + W b_ln for conv4 o LN2 and conv1 o LN1 (fp32; gemm.h E_LNBWD2)
        LnVecJobs lj{};
        lj.n = 2; lj.N2 = C2; lj.C = C; lj.round_bf16 = 1;
        lj.W[0] = p->conv4_w; lj.bz[0] = p->conv4_b; lj.lnw[0] = p->norm2_w; lj.lnb[0] = p->norm2_b; lj.u[0] = w.u4; lj.cvec[0] = w.c4;
        lj.W[1] = p->conv1_w; lj.bz[1] = p->conv1_b; lj.lnw[1] = p->norm1_w; lj.lnb[1] = p->norm1_b; lj.u[1] = w.u1; lj.cvec[1] = w.c1;
        DCPT_TRY(launch_lnvec(lj, s));
    }
    // weight-gradient side stream: 0 never, 1 always, 2 only at the levels without the grouped 256-tile weight-gradient launch (whose
    // 128-KB blocks cannot share a CU with a main-stream GEMM block: with it the two streams only time-slice)
    static const int side_mode = dcpt_tuning("DCPT_BF16_SIDE", 2);
    Side* sd = (side_mode == 1 || (side_mode == 2 && !w.tn256)) ? side_for(s) : nullptr;
    hipStream_t sw = side_stream(sd, s);
    DCPT_TRY(side_fork(sd, 0, s));
    GemmNTB g{};
    const bool ffn = ffn_fused(C);
    if (ffn) {
        // B1 + B3 + B5 in one pass (ffn_bf16.hip): dv for conv4's weight gradient and dy = dout + LayerNorm2 backward
        FfnBwdB f{};
        f.dout = dout; f.v = sv->v; f.y = sv->y; f.wT5 = w.wT5; f.wT4 = w.wT4; f.lnw = p->norm2_w; f.dv = nullptr; f.dy = w.dy;
        f.lnpart = w.ffn_part; f.M = M; f.eps = 1e-6f;
        DCPT_TRY(launch_ffn_bwd_bf16(f, C, s));
        {   // B2 + B4: both weight gradients from (dout, v, y) on the side stream, operands recomputed per 32 pixels (ffn_wgrad_bf16)
            FfnWgB q{};
            q.dout = dout; q.v = sv->v; q.y = sv->y; q.wT5 = w.wT5; q.lnw = p->norm2_w; q.lnb = p->norm2_b; q.g5 = w.ffn_g5; q.g4 = w.ffn_g4;
            q.cs5 = w.ffn_cs5; q.cs4 = w.ffn_cs4; q.M = M; q.eps = 1e-6f;
            DCPT_TRY(launch_ffn_wgrad_bf16(q, C, sw));
            const int nw = ffn_bwd_bf16_waves(M);
            DCPT_TRY(launch_wgrad_reduce(w.ffn_g5, w.ffn_cs5, nw, nw, C, C, p->gamma, p->conv5_w, p->conv5_b, gr->conv5_w, gr->gamma, gr->conv5_b, WR_PLAIN, sw));
            DCPT_TRY(launch_wgrad_reduce(w.ffn_g4, w.ffn_cs4, nw, nw, C2, C, nullptr, nullptr, nullptr, gr->conv4_w, nullptr, gr->conv4_b, WR_PLAIN, sw));
        }
        DCPT_TRY(side_fork(sd, 2, s));
        DCPT_TRY(launch_colpart_reduce(w.ffn_part, ffn_bwd_bf16_waves(M), 2, C, gr->norm2_w, gr->norm2_b, nullptr, sw));
    } else {
        // B1: dv = SimpleGate'(dout * gamma * W5; v)   (+ the row sums of LN2's backward, linear in dv)
        g.M = M; g.A = dout; g.lda = C; g.K = C; g.Bw = w.wT5; g.N = C; g.C = w.dv; g.ldc = C2; g.aux = sv->v;
        if (w.lrs) {
            g.rowpart = w.rowpart; g.uvec = w.u4; g.cvec = w.c4;
        }
        DCPT_TRY(launch_gemm_nt_bf16(g, EB_SGBWD, s));
        // B2: conv5 / gamma gradients
        if (!w.tn256) DCPT_TRY(wgrad_b(dout, C, sv->g, C, M, w.slab, w.colsum, p->gamma, p->conv5_w, p->conv5_b, gr->conv5_w, gr->gamma, gr->conv5_b, sw));
        DCPT_TRY(side_fork(sd, 1, s));
        // B3: gradient of LN2's output
        g = GemmNTB{};
        g.M = M; g.A = w.dv; g.lda = C2; g.K = C2; g.Bw = w.wT4; g.N = C; g.C = w.gln; g.ldc = C;
        if (w.lrs) {   // B3 + B5 in one launch: dy = dout + LN2-backward(dv W4^T), the LayerNorm's incoming gradient is never written
            g.C = w.dy; g.res = sv->y; g.ldres = C; g.aux = dout; g.mu = sv->mu2; g.rstd = sv->rstd2; g.lnw = p->norm2_w; g.colpart = w.lnpart;
            g.rowpart = w.rowpart; g.rowparts = w.rp_sg;
            DCPT_TRY(launch_gemm_nt_bf16(g, EB_LNBWD2, s));
        } else {
            DCPT_TRY(launch_gemm_nt_bf16(g, EB_PLAIN, s));
        }
        // B4: conv4 gradients
        if (!w.tn256) DCPT_TRY(wgrad_b(w.dv, C2, sv->xn2, C, M, w.slab, w.colsum, nullptr, nullptr, nullptr, gr->conv4_w, nullptr, gr->conv4_b, sw));
        // B5: dy = dout + LN2-backward  (with the chain kernel: part of B6 below)
        if (!w.lrs && !w.mid) DCPT_TRY(launch_ln_bwd_bf16(w.gln, sv->y, sv->mu2, sv->rstd2, p->norm2_w, dout, w.dy, w.lnpart, w.ln_nblk, M, C, s));
        DCPT_TRY(side_fork(sd, 2, s));
        if (!w.tn256) DCPT_TRY(launch_colpart_reduce(w.lnpart, (w.lrs || w.mid) ? ln_tiles : w.ln_nblk, 2, C, gr->norm2_w, gr->norm2_b, nullptr, sw));
    }
    // B6: dts = d(t2 * s) (+ SCA's per-image channel sums out of the epilogue when an image is a whole number of 128-pixel tiles)
    g = GemmNTB{};
    g.M = M; g.A = w.dy; g.lda = C; g.K = C; g.Bw = w.wT3; g.N = C; g.C = w.dts; g.ldc = C;
    if (w.mid && !ffn) {   // B5 + B6 per 128-pixel tile, the LayerNorm backward as the GEMM's tile load (chain_bf16.hip)
        ChainMidB q{};
        q.gln = w.gln; q.y = sv->y; q.dout = dout; q.t2 = sv->t2; q.mu = sv->mu2; q.rstd = sv->rstd2; q.lnw = p->norm2_w; q.Wf = w.WfT3;
        q.dy = w.dy; q.dts = w.dts; q.lnpart = w.lnpart; q.dspart = w.ds_part; q.M = M;
        DCPT_TRY(launch_chain_bwd_mid_bf16(q, C, s));
    } else if (w.ds_fused) {
        g.res = sv->t2; g.ldres = C; g.colpart = w.ds_part;
        DCPT_TRY(launch_gemm_nt_bf16(g, EB_DOTCOL, s));
    } else {
        DCPT_TRY(launch_gemm_nt_bf16(g, EB_PLAIN, s));
        DCPT_TRY(launch_sca_ds_part_bf16(w.dts, sv->t2, w.ds_part, B, C, P, w.ds_slices, s));
    }
    // B7: conv3 / beta gradients: G = sum_m dy[m][n] t2[m][k] s[img(m)][k]
    if (w.tn256) {   // (joins conv1's weight gradient in the second grouped launch; only a scaled operand copy, if needed, is made here)
        if (!w.conv3_images) DCPT_TRY(launch_scale_rows_bf16(sv->t2, sv->s, w.t2s, M, C, P, sw));
    } else {
        GemmTNB t{};
        t.X = w.dy; t.ldx = C; t.N = C; t.Y = sv->t2; t.ldy = C; t.K = C; t.M = M; t.slab = w.slab; t.colsum = w.colsum;
        if (gemm_tn_bf16_plan_images(M, C, C, P, &t.splits, &t.rows_per_split) && (size_t)t.splits * C * C <= w.slab_elems &&
            (size_t)t.splits * gemm_tn_bf16_tiles_k(C, C) * C <= w.colsum_elems) {
            // every pixel chunk inside one image: plain operands, the scale weights the chunk slabs in the reducer
            DCPT_TRY(launch_gemm_tn_bf16(t, sw));
            DCPT_TRY(launch_wgrad_reduce_scaled(w.slab, w.colsum, t.splits, t.splits * gemm_tn_bf16_tiles_k(C, C), C, C, p->beta, p->conv3_w,
                                                p->conv3_b, gr->conv3_w, gr->beta, gr->conv3_b, WR_PLAIN, sv->s, (int)(P / t.rows_per_split), sw));
        } else {
            DCPT_TRY(launch_scale_rows_bf16(sv->t2, sv->s, w.t2s, M, C, P, sw));
            DCPT_TRY(wgrad_b(w.dy, C, w.t2s, C, M, w.slab, w.colsum, p->beta, p->conv3_w, p->conv3_b, gr->conv3_w, gr->beta, gr->conv3_b, sw));
        }
    }
    // B8: SCA backward
    DCPT_TRY(launch_sca_dpool(w.ds_part, w.ds_slices, p->sca_w, w.dpool, B, C, P, s, w.tn256 ? w.ds : nullptr));
    // B9 / B10: SimpleGate + depthwise backward, da on chip
    const bool ring_b = !w.lrs && dw_ring_bwd_usable(dg, 2);
    const int nblk_b = ring_b ? dw_ring_bwd_num_blocks_per_image(dg) : dw_num_blocks_per_image_fused_bf16(dg);
    if (ring_b) DCPT_TRY(launch_dw_ring_bwd_fused_bf16(w.dts, sv->t1, w.w2p, p->conv2_b, sv->s, w.dpool, w.dt1, w.wpart, dg, s));
    else
        DCPT_TRY(launch_dw_bwd_fused_bf16(w.dts, sv->t1, w.w2p, p->conv2_b, sv->s, w.dpool, w.dt1, w.wpart, dg, s, w.lrs ? w.rowpart : nullptr, w.u1,
                                          w.c1));
    DCPT_TRY(side_fork(sd, 3, s));
    if (w.tn256) {   // B2 + B4 + B7 + B12 as ONE grouped launch: G5 = dout^T g, G4 = dv^T LN2(y), G3 = dy^T t2 (per image, or t2 * s), G1 = dt1^T LN1(inp)
        w.wg.p[0].X = dout; w.wg.p[0].Y = sv->g;
        w.wg.p[1].X = w.dv; w.wg.p[1].Y = sv->xn2;
        w.wg.p[2].X = w.dy; w.wg.p[2].Y = w.conv3_images ? sv->t2 : w.t2s;
        w.wg.p[3].X = w.dt1; w.wg.p[3].Y = sv->xn1;
        DCPT_TRY(launch_gemm_tn_bf16_256(w.wg, sw));
    } else {
        DCPT_TRY(launch_sca_wgrad(w.ds_part, w.ds_slices, w.ds, sv->pooled, gr->sca_w, gr->sca_b, B, C, sw));
        DCPT_TRY(launch_dw_wgrad_reduce(w.wpart, B * nblk_b, C2, gr->conv2_w, gr->conv2_b, sw));
    }
    if (ffn) {
        // B11 + B13 in one pass (ffn_bf16.hip): dinp = dy + LayerNorm1 backward of dt1 W1^T
        FfnBwdB f{};
        f.dout = w.dy; f.v = w.dt1; f.y = inp; f.wT4 = w.wT1; f.lnw = p->norm1_w; f.dy = dinp; f.lnpart = w.ffn_part1; f.M = M; f.eps = 1e-6f;
        DCPT_TRY(launch_conv_ln_bwd_tail_bf16(f, C, s));
        DCPT_TRY(wgrad_b(w.dt1, C2, sv->xn1, C, M, w.slab, w.colsum, nullptr, nullptr, nullptr, gr->conv1_w, nullptr, gr->conv1_b, sw));
        DCPT_TRY(side_fork(sd, 5, s));
        DCPT_TRY(launch_colpart_reduce(w.ffn_part1, ffn_bwd_bf16_waves(M), 2, C, gr->norm1_w, gr->norm1_b, nullptr, sw));
    } else {
        // B11: gradient of LN1's output
        g = GemmNTB{};
        g.M = M; g.A = w.dt1; g.lda = C2; g.K = C2; g.Bw = w.wT1; g.N = C; g.C = w.gln; g.ldc = C;
        if (w.lrs) {   // B11 + B13: dinp = dy + LN1-backward(dt1 W1^T)
            g.C = dinp; g.res = inp; g.ldres = C; g.aux = w.dy; g.mu = sv->mu1; g.rstd = sv->rstd1; g.lnw = p->norm1_w; g.colpart = w.lnpart2;
            g.rowpart = w.rowpart; g.rowparts = w.rp_dw;
            DCPT_TRY(launch_gemm_nt_bf16(g, EB_LNBWD2, s));
        } else {
            DCPT_TRY(launch_gemm_nt_bf16(g, EB_PLAIN, s));
        }
        // B12: conv1 gradients
        if (!w.tn256) DCPT_TRY(wgrad_b(w.dt1, C2, sv->xn1, C, M, w.slab, w.colsum, nullptr, nullptr, nullptr, gr->conv1_w, nullptr, gr->conv1_b, sw));
        // B13: dinp = dy + LN1-backward
        if (!w.lrs) DCPT_TRY(launch_ln_bwd_bf16(w.gln, inp, sv->mu1, sv->rstd1, p->norm1_w, w.dy, dinp, w.lnpart2, w.ln_nblk, M, C, s));
        DCPT_TRY(side_fork(sd, 5, s));
        if (w.tn256) {
            // the finisher: the four weight gradients out of their partial sums (gain algebra of nafblock.hip: dW = gain G,
            // dgain = W . G + b sum dO, db = gain sum dO) and every other parameter-gradient reduction of the block, one launch
            FinJobs f{};
            f.nslab = 4;
            const TnProb* q[4] = {&w.wg.p[0], &w.wg.p[1], &w.wg.p[2], &w.wg.p[3]};
            for (int i = 0; i < 4; ++i) {
                f.slab[i].slab = q[i]->slab; f.slab[i].colsum = q[i]->colsum; f.slab[i].N = q[i]->N; f.slab[i].K = q[i]->K;
                f.slab[i].splits = q[i]->slots; f.slab[i].cs_rows = q[i]->splits; f.slab[i].tiles_k = q[i]->tiles_k; f.slab[i].ks_div = 1;
            }
            f.slab[0].rowscale = p->gamma; f.slab[0].W = p->conv5_w; f.slab[0].wbias = p->conv5_b;
            f.slab[0].dW = gr->conv5_w; f.slab[0].dgain = gr->gamma; f.slab[0].dbias = gr->conv5_b;
            f.slab[1].dW = gr->conv4_w; f.slab[1].dbias = gr->conv4_b;
            f.slab[2].rowscale = p->beta; f.slab[2].W = p->conv3_w; f.slab[2].wbias = p->conv3_b;
            f.slab[2].dW = gr->conv3_w; f.slab[2].dgain = gr->beta; f.slab[2].dbias = gr->conv3_b;
            if (w.conv3_images) {   // slots per image: 1 (whole images per block) or the number of ranges an image is cut into
                f.slab[2].kscale = sv->s;
                f.slab[2].ks_div = q[2]->seg_rows > 0 ? 1 : (int)(P / q[2]->rows_per_split);
            }
            f.slab[3].dW = gr->conv1_w; f.slab[3].dbias = gr->conv1_b;
            const int lnR = w.lrs ? ln_tiles : w.ln_nblk;
            f.ncols = 3;
            f.cols[0] = FinCols{w.lnpart, gr->norm2_w, gr->norm2_b, w.mid ? ln_tiles : lnR, 2, C, 0};
            f.cols[1] = FinCols{w.lnpart2, gr->norm1_w, gr->norm1_b, lnR, 2, C, 0};
            f.cols[2] = FinCols{w.wpart, gr->conv2_w, gr->conv2_b, B * nblk_b, 10, C2, 1};
            f.sca = FinSca{w.ds, sv->pooled, gr->sca_w, gr->sca_b, B, C};
            DCPT_TRY(launch_wgrad_finish(f, sw));
        } else {
            DCPT_TRY(launch_colpart_reduce(w.lnpart2, w.lrs ? ln_tiles : w.ln_nblk, 2, C, gr->norm1_w, gr->norm1_b, nullptr, sw));
        }
    }
    DCPT_TRY(side_join(sd, s));
    return DCPT_OK;
}

extern "C" int dcpt_nafblock_bwd_bf16(const dcpt_nafblock_params* p, const dcpt_nafblock_grads* gr, const uint16_t* inp,
                                      const dcpt_nafblock_saved_bf16* sv, const uint16_t* dout, uint16_t* dinp, void* ws, size_t ws_bytes, int B,
                                      int H, int W, int C, dcpt_stream_t stream) {
    return nafblock_bwd_bf16_impl(p, gr, inp, sv, dout, dinp, ws, ws_bytes, B, H, W, C, nullptr, 0, stream);
}
extern "C" int dcpt_nafblock_bwd_bf16_packed(const dcpt_nafblock_params* p, const void* packed, size_t packed_bytes, const dcpt_nafblock_grads* gr,
                                             const uint16_t* inp, const dcpt_nafblock_saved_bf16* sv, const uint16_t* dout, uint16_t* dinp, void* ws,
                                             size_t ws_bytes, int B, int H, int W, int C, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(packed, "nafblock_bwd_bf16_packed: null packed weights");
    return nafblock_bwd_bf16_impl(p, gr, inp, sv, dout, dinp, ws, ws_bytes, B, H, W, C, packed, packed_bytes, stream);
}

// ---- the weight gradient of a 1 x 1 convolution as an operator of its own ------------------------------------------------------------
namespace {
struct WgWs {
    bool tn256;
    GemmTNG g;
    GemmTNB t;
    float *slab, *colsum;
};
size_t wg_layout(int64_t M, int N, int K, void* base, size_t bytes, WgWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    WgWs w{};
    w.tn256 = gemm_tn_bf16_256_ok(N, K);
    if (w.tn256) {
        w.g.n = 1;
        w.g.p[0].M = M; w.g.p[0].N = w.g.p[0].ldx = N; w.g.p[0].K = w.g.p[0].ldy = K;
        gemm_tn_bf16_256_plan(w.g);
        w.g.p[0].slab = a.get<float>(gemm_tn_bf16_256_slab_floats(w.g.p[0]));
        w.g.p[0].colsum = a.get<float>(gemm_tn_bf16_256_colsum_floats(w.g.p[0]));
    } else {
        w.t.ldx = w.t.N = N; w.t.ldy = w.t.K = K; w.t.M = M;
        gemm_tn_bf16_plan(M, N, K, &w.t.splits, &w.t.rows_per_split);
        w.slab = a.get<float>((size_t)w.t.splits * N * K);
        w.colsum = a.get<float>((size_t)w.t.splits * gemm_tn_bf16_tiles_k(N, K) * N);
    }
    if (out) *out = w;
    return a.off;
}
}  // namespace

extern "C" size_t dcpt_conv1x1_wgrad_bf16_ws_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 8 || K % 8) return 0;
    return wg_layout(M, N, K, nullptr, 0, nullptr);
}
extern "C" int dcpt_conv1x1_wgrad_bf16(const uint16_t* dY, const uint16_t* X, float* dW, float* db, void* ws, size_t ws_bytes, int64_t M, int N,
                                       int K, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "conv1x1_wgrad_bf16: null argument or N=%d / K=%d not multiples of 8",
                   N, K);
    WgWs w;
    const size_t need = wg_layout(M, N, K, ws, ws_bytes, &w);
    if (need > ws_bytes || ws == nullptr) {
        dcpt_set_error("conv1x1_wgrad_bf16: workspace too small (%zu < %zu)", ws_bytes, need);
        return DCPT_ERR_WS;
    }
    if (w.tn256) {
        w.g.p[0].X = dY; w.g.p[0].Y = X;
        if (!db) w.g.p[0].colsum = nullptr;
        DCPT_TRY(launch_gemm_tn_bf16_256(w.g, s));
        FinJobs f{};
        f.nslab = 1;
        f.slab[0].slab = w.g.p[0].slab; f.slab[0].colsum = w.g.p[0].colsum; f.slab[0].N = N; f.slab[0].K = K; f.slab[0].splits = w.g.p[0].slots;
        f.slab[0].cs_rows = w.g.p[0].splits;
        f.slab[0].tiles_k = w.g.p[0].tiles_k; f.slab[0].ks_div = 1; f.slab[0].dW = dW; f.slab[0].dbias = db;
        return launch_wgrad_finish(f, s);
    }
    w.t.X = dY; w.t.Y = X; w.t.slab = w.slab; w.t.colsum = db ? w.colsum : nullptr;
    DCPT_TRY(launch_gemm_tn_bf16(w.t, s));
    return launch_wgrad_reduce(w.slab, w.t.colsum, w.t.splits, w.t.splits * gemm_tn_bf16_tiles_k(N, K), N, K, nullptr, nullptr, nullptr, dW, nullptr, db,
                               WR_PLAIN, s);
}

extern "C" int dcpt_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && y && n >= 0, "cast_f32_bf16: null argument");
    return n == 0 ? DCPT_OK : launch_cast_f32_bf16(x, y, n, (hipStream_t)stream);
}
extern "C" int dcpt_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && y && n >= 0, "cast_bf16_f32: null argument");
    return n == 0 ? DCPT_OK : launch_cast_bf16_f32(x, y, n, (hipStream_t)stream);
}
