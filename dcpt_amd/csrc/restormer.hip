// Restormer blocks (reference basicsr/archs/restormer_arch.py):
//   MDTA  x + project_out(attn(LN(x)))  (:103-145, :148-159)   dcpt_mdta_fwd/bwd
//   GDFN  x + project_out(gelu(x1)*x2)  (:75-100)              dcpt_gdfn_fwd/bwd
// plus the NHWC helpers between blocks: PixelShuffle/PixelUnshuffle(2) (:175-202) and channel concat (:390-400).
//
// MDTA in NHWC: qkv = dw3x3(conv1x1(LN(x))) is one [M][3c] tensor; q,k,v are column ranges of it and a head
// is a contiguous column group, so every per-(image, head) matrix product is a BATCHED MFMA GEMM that addresses
// the same buffer with pointer strides (no rearrange copies):
//   Gram   G[i][j]  = sum_p q[p][i] k[p][j]            gemm_tn, batch (b, head), reduction over the pixels
//   attn            = relu(temperature * G / (|q_i| |k_j|))   (F.normalize folds into a rank-1 scaling of G)
//   out[p][i]       = sum_j attn[i][j] v[p][j]          gemm_nt, batch (b, head)
// The L2 norms over the pixels come for free as per-block partial sums of squares from the depthwise kernel.
#include "gemm.h"
#include "kernels.h"
#include "side.h"
#include "../../include/dcpt_hip.h"

namespace {

// depthwise backward on the ring kernels (dwring.hip) when the channel count allows it (Ctot % 8 == 0); DCPT_DW_RING_BWD=0 disables
bool dwf_ring(int B, int H, int W, int Ctot) { return Ctot % 8 == 0 && dw_ring_usable(DwGeom{B, H, W, Ctot / 2}, 4); }
bool dwb_ring(int B, int H, int W, int Ctot) { return Ctot % 8 == 0 && dw_ring_bwd_usable(DwGeom{B, H, W, Ctot / 2}, 4); }


constexpr float NORM_EPS = 1e-12f;  // F.normalize eps

// nrm[b][c2] = max(sqrt(sum_blk part[b][blk][c2]), eps)
__global__ void sq_norm_kernel(const float* __restrict__ part, int nblk, float* __restrict__ nrm, int B, int C2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C2) return;
    const int b = i / C2, c = i % C2;
    float s = 0.f;
    for (int j = 0; j < nblk; ++j) s += part[((int64_t)b * nblk + j) * C2 + c];
    nrm[i] = fmaxf(sqrtf(s), NORM_EPS);
}

// one block per (b, head): G = sum_splits slab; Ghat = G/(nq_i nk_j); attn = relu(temp*Ghat); also attn^T
__global__ __launch_bounds__(256) void attn_finalize_kernel(const float* __restrict__ slab, int splits,
                                                            const float* __restrict__ nrm, const float* __restrict__ temp,
                                                            float* __restrict__ ghat, float* __restrict__ attn,
                                                            float* __restrict__ attnT, int heads, int ch, int c) {
    const int z = blockIdx.x, b = z / heads, h = z % heads;
    const float t = temp[h];
    const float* nq = nrm + (int64_t)b * 2 * c + h * ch;
    const float* nk = nrm + (int64_t)b * 2 * c + c + h * ch;
    const int64_t zo = (int64_t)z * ch * ch;
    for (int e = threadIdx.x; e < ch * ch; e += 256) {
        const int i = e / ch, j = e % ch;
        float g = 0.f;
        for (int s = 0; s < splits; ++s) g += slab[((int64_t)z * splits + s) * ch * ch + e];
        const float gh = g / (nq[i] * nk[j]);
        const float a = fmaxf(t * gh, 0.f);
        ghat[zo + e] = gh;
        attn[zo + e] = a;
        attnT[zo + (int64_t)j * ch + i] = a;
    }
}

// PromptIR's Attention (reference basicsr/archs/promptir_arch.py:129-141) applies softmax over j instead of ReLU.  Same
// inputs/outputs as attn_finalize_kernel; a wave owns rows i = wave, wave + 4, ..., its lanes the columns j = lane + 64 u.
constexpr int SM_MAXU = 4;   // ch <= 256
__global__ __launch_bounds__(256) void attn_finalize_softmax_kernel(const float* __restrict__ slab, int splits,
                                                                    const float* __restrict__ nrm, const float* __restrict__ temp,
                                                                    float* __restrict__ ghat, float* __restrict__ attn,
                                                                    float* __restrict__ attnT, int heads, int ch, int c) {
    const int z = blockIdx.x, b = z / heads, h = z % heads;
    const float t = temp[h];
    const float* nq = nrm + (int64_t)b * 2 * c + h * ch;
    const float* nk = nrm + (int64_t)b * 2 * c + c + h * ch;
    const int64_t zo = (int64_t)z * ch * ch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < ch; i += 4) {
        float pre[SM_MAXU], gh[SM_MAXU];
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < SM_MAXU; ++u) {
            const int j = lane + 64 * u;
            pre[u] = -INFINITY;
            gh[u] = 0.f;
            if (j < ch) {
                float g = 0.f;
                for (int sp = 0; sp < splits; ++sp) g += slab[((int64_t)z * splits + sp) * ch * ch + i * ch + j];
                gh[u] = g / (nq[i] * nk[j]);
                pre[u] = t * gh[u];
                m = fmaxf(m, pre[u]);
            }
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < SM_MAXU; ++u) {
            pre[u] = (lane + 64 * u < ch) ? expf(pre[u] - m) : 0.f;
            sum += pre[u];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int u = 0; u < SM_MAXU; ++u) {
            const int j = lane + 64 * u;
            if (j < ch) {
                const float a = pre[u] / sum;
                ghat[zo + i * ch + j] = gh[u];
                attn[zo + i * ch + j] = a;
                attnT[zo + (int64_t)j * ch + i] = a;
            }
        }
    }
}

// one block per (b, head).  dattn = sum_splits slab.  Outputs dG, dG^T, cq/ck (norm-path coefficients), dtemp_part[z]
// SOFTMAX (PromptIR): dpre_ij = attn_ij * (dattn_ij - sum_j' dattn_ij' attn_ij'), the row dots first (a wave per row) into LDS.
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ slab, int splits, const float* __restrict__ attn,
                                                       const float* __restrict__ ghat, const float* __restrict__ nrm,
                                                       const float* __restrict__ temp, float* __restrict__ dG,
                                                       float* __restrict__ dGT, float* __restrict__ cqk,
                                                       float* __restrict__ dtemp_part, float* __restrict__ scratch, int heads,
                                                       int ch, int c) {
    __shared__ float red[256];
    __shared__ float rowdot[256];
    const int z = blockIdx.x, b = z / heads, h = z % heads;
    const float t = temp[h];
    const float* nq = nrm + (int64_t)b * 2 * c + h * ch;
    const float* nk = nrm + (int64_t)b * 2 * c + c + h * ch;
    const int64_t zo = (int64_t)z * ch * ch;
    float* prod = scratch + zo;  // dGhat * Ghat
    if (SOFTMAX) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int i = wave; i < ch; i += 4) {
            float d = 0.f;
            for (int j = lane; j < ch; j += 64) {
                float da = 0.f;
                for (int s = 0; s < splits; ++s) da += slab[((int64_t)z * splits + s) * ch * ch + i * ch + j];
                d += da * attn[zo + i * ch + j];
            }
            d = wave_sum(d);
            if (lane == 0) rowdot[i] = d;
        }
        __syncthreads();
    }
    float acc = 0.f;
    for (int e = threadIdx.x; e < ch * ch; e += 256) {
        const int i = e / ch, j = e % ch;
        float da = 0.f;
        for (int s = 0; s < splits; ++s) da += slab[((int64_t)z * splits + s) * ch * ch + e];
        const float dpre = SOFTMAX ? attn[zo + e] * (da - rowdot[i]) : (attn[zo + e] > 0.f ? da : 0.f);
        const float gh = ghat[zo + e];
        acc += dpre * gh;
        const float dgh = dpre * t;
        const float g = dgh / (nq[i] * nk[j]);
        dG[zo + e] = g;
        dGT[zo + (int64_t)j * ch + i] = g;
        prod[e] = dgh * gh;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dtemp_part[z] = red[0];
    // cq_i = -(sum_j prod[i][j]) / nq_i^2 (unless the norm was clamped);  ck_j = -(sum_i prod[i][j]) / nk_j^2
    for (int r = threadIdx.x; r < 2 * ch; r += 256) {
        float s = 0.f;
        if (r < ch) {
            for (int j = 0; j < ch; ++j) s += prod[r * ch + j];
            const float n = nq[r];
            cqk[(int64_t)b * 2 * c + h * ch + r] = (n > NORM_EPS) ? -s / (n * n) : 0.f;
        } else {
            const int j = r - ch;
            for (int i = 0; i < ch; ++i) s += prod[i * ch + j];
            const float n = nk[j];
            cqk[(int64_t)b * 2 * c + c + h * ch + j] = (n > NORM_EPS) ? -s / (n * n) : 0.f;
        }
    }
}

__global__ void dtemp_reduce_kernel(const float* __restrict__ part, float* __restrict__ dtemp, int B, int heads) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= heads) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[b * heads + h];
    dtemp[h] = s;
}

// ---- weight packing with channel padding (GDFN hidden = int(dim*2.66) is not a multiple of 4) ----
// mode 0: project_in  [2h][c]      -> [2hp][c]   (rows h..hp-1 and hp+h..2hp-1 are zero)
// mode 1: dwconv      [2h][9]      -> [9][2hp]
// mode 2: project_out [c][h]       -> [c][hp]    (zero columns)
// mode 3: project_out^T            -> [hp][c]
// mode 4: project_in^T             -> [c][2hp]
__global__ void gdfn_pack_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h, int hp, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) {
        if (i >= (int64_t)2 * hp * c) return;
        const int r = (int)(i / c), k = (int)(i % c);
        const int half = r / hp, rr = r % hp;
        out[i] = (rr < h) ? in[((int64_t)half * h + rr) * c + k] : 0.f;
    } else if (mode == 1) {
        if (i >= (int64_t)9 * 2 * hp) return;
        const int t = (int)(i / (2 * hp)), r = (int)(i % (2 * hp));
        const int half = r / hp, rr = r % hp;
        out[i] = (rr < h) ? in[((int64_t)half * h + rr) * 9 + t] : 0.f;
    } else if (mode == 2) {
        if (i >= (int64_t)c * hp) return;
        const int n = (int)(i / hp), k = (int)(i % hp);
        out[i] = (k < h) ? in[(int64_t)n * h + k] : 0.f;
    } else if (mode == 3) {
        if (i >= (int64_t)hp * c) return;
        const int k = (int)(i / c), n = (int)(i % c);
        out[i] = (k < h) ? in[(int64_t)n * h + k] : 0.f;
    } else {
        if (i >= (int64_t)c * 2 * hp) return;
        const int k = (int)(i / (2 * hp)), r = (int)(i % (2 * hp));
        const int half = r / hp, rr = r % hp;
        out[i] = (rr < h) ? in[((int64_t)half * h + rr) * c + k] : 0.f;
    }
}

// gradient un-padding: mode 0: [2hp][c] -> [2h][c]; mode 1: dw2 [2hp*9] (ch*9+tap) -> [2h*9]; mode 2: [c][hp] -> [c][h]
__global__ void gdfn_unpack_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h, int hp, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) {
        if (i >= (int64_t)2 * h * c) return;
        const int r = (int)(i / c), k = (int)(i % c);
        out[i] = in[((int64_t)(r / h) * hp + r % h) * c + k];
    } else if (mode == 1) {
        if (i >= (int64_t)2 * h * 9) return;
        const int r = (int)(i / 9), t = (int)(i % 9);
        out[i] = in[((int64_t)(r / h) * hp + r % h) * 9 + t];
    } else {
        if (i >= (int64_t)c * h) return;
        const int n = (int)(i / h), k = (int)(i % h);
        out[i] = in[(int64_t)n * hp + k];
    }
}

// ---- NHWC pixel shuffle / unshuffle and channel concat -------------------------------------------
// unshuffle: in [B][H][W][C] -> out [B][H/2][W/2][4C], out ch = 4k + 2i + j  <- in pixel (2h+i, 2w+j) ch k
// shuffle is the inverse (dir = 1): in [B][H][W][4C] -> out [B][2H][2W][C]
__global__ void pixel_shuffle_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Hc, int Wc, int C, int dir) {
    // (Hc, Wc) = coarse grid, C = fine channels
    const int64_t total = (int64_t)B * Hc * Wc * 4 * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(e % (4 * C));  // coarse channel 4k + 2i + j
        int64_t t = e / (4 * C);
        const int w = (int)(t % Wc);
        t /= Wc;
        const int h = (int)(t % Hc);
        const int64_t b = t / Hc;
        const int k = cc >> 2, ij = cc & 3;
        const int64_t fine = ((b * (2 * Hc) + 2 * h + (ij >> 1)) * (int64_t)(2 * Wc) + 2 * w + (ij & 1)) * C + k;
        if (dir == 0) out[e] = in[fine];
        else out[fine] = in[e];
    }
}

// out[m][0:Ca] = a[m], out[m][Ca:Ca+Cb] = b[m]   (dir 0);  split back (dir 1: a,b are outputs)
__global__ void concat_kernel(float* __restrict__ a, float* __restrict__ b, float* __restrict__ cat, int64_t M, int Ca, int Cb,
                              int dir) {
    const int qa = Ca / 4, qb = Cb / 4, qt = qa + qb;
    const int64_t total = M * qt;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % qt);
        const int64_t m = e / qt;
        float* src = (q < qa) ? a + m * Ca + 4 * q : b + m * Cb + 4 * (q - qa);
        float* dst = cat + m * (int64_t)(Ca + Cb) + 4 * q;
        if (dir == 0) stg4(dst, ldg4(src));
        else stg4(src, ldg4(dst));
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t nb = cdiv64(n, 256);
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

inline int attn_splits(int P, int nbatch) {
    int s = 512 / (nbatch > 0 ? nbatch : 1);
    const int cap = P / 256 > 0 ? P / 256 : 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return s;
}

// ---- MDTA workspace --------------------------------------------------------------------------------
struct MdtaWs {
    float* w2p;      // [9][3c]
    float* sqpart;   // [B][nblk][2c]
    float* gslab;    // [B*heads][splits][ch][ch]
    // backward
    float *wT_proj, *wT_qkv;          // [c][c], [c][3c]
    float *d_att;                     // [M][c]
    float *dqkv, *dqkv1;              // [M][3c]
    float *dxn;                       // [M][c]
    float *dG, *dGT, *scr;            // [B*heads][ch][ch]
    float *scr2;                      // [3c] sink of the (non-existent) depthwise bias gradient
    float *scr3;                      // [2M] sink of the statistics of a recomputed LayerNorm
    float *cqk;                       // [B][2c]
    float *dtpart;                    // [B*heads]
    float *slab;                      // weight-gradient slabs
    float *wpart;                     // dw partials
    float *lnpart;
    float *r_xn, *r_qkv1, *r_out;     // lean mode: LN(x), qkv conv output, attn @ v recomputed here when the caller did not keep them
    int nblk_dw, splits_a, ln_nblk, nblk_dwb;
};

size_t mdta_layout(int B, int H, int W, int c, int heads, int backward, void* base, size_t bytes, MdtaWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    MdtaWs w{};
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W, ch = c / heads;
    w.nblk_dw = dw_num_blocks_generic(B, H, W, 3 * c);
    w.splits_a = attn_splits(P, B * heads);
    w.w2p = a.get<float>((size_t)27 * c);
    w.gslab = a.get<float>((size_t)B * heads * w.splits_a * ch * ch);
    w.r_xn = a.get<float>((size_t)M * c);
    w.r_qkv1 = a.get<float>((size_t)M * 3 * c);
    w.r_out = a.get<float>((size_t)M * c);
    if (!backward) {
        w.sqpart = a.get<float>((size_t)B * w.nblk_dw * 2 * c);
    } else {
        w.wT_proj = a.get<float>((size_t)c * c);
        w.wT_qkv = a.get<float>((size_t)3 * c * c);
        w.d_att = a.get<float>((size_t)M * c);
        w.dqkv = a.get<float>((size_t)M * 3 * c);
        w.dqkv1 = a.get<float>((size_t)M * 3 * c);
        w.dxn = a.get<float>((size_t)M * c);
        w.dG = a.get<float>((size_t)B * heads * ch * ch);
        w.dGT = a.get<float>((size_t)B * heads * ch * ch);
        w.scr = a.get<float>((size_t)B * heads * ch * ch);
        w.scr2 = a.get<float>((size_t)3 * c);
        w.scr3 = a.get<float>((size_t)2 * M);
        w.cqk = a.get<float>((size_t)B * 2 * c);
        w.dtpart = a.get<float>((size_t)B * heads);
        int sp1, sp2;
        int64_t r1, r2;
        gemm_tn_plan(M, 3 * c, c, &sp1, &r1);
        gemm_tn_plan(M, c, c, &sp2, &r2);
        const size_t s1 = (size_t)sp1 * 3 * c * c, s2 = (size_t)sp2 * c * c;
        w.slab = a.get<float>(s1 > s2 ? s1 : s2);
        w.nblk_dwb = dw_num_blocks_generic(B, H, W, 3 * c);
        if (dwb_ring(B, H, W, 3 * c)) {
            const int nr = dw_ring_bwd_num_blocks_per_image(DwGeom{B, H, W, 3 * c / 2});
            if (nr > w.nblk_dwb) w.nblk_dwb = nr;
        }
        w.wpart = a.get<float>((size_t)B * w.nblk_dwb * 10 * 3 * c);
        w.ln_nblk = ln_bwd_num_blocks(M, c);
        w.lnpart = a.get<float>((size_t)w.ln_nblk * 3 * c);
    }
    if (out) *out = w;
    return a.off;
}

void set_attn_batch_nt(GemmNT& g, int P, int c, int heads, int ch) {
    g.nb1 = 0;  // filled by caller (B)
    g.nb2 = heads;
    (void)P; (void)c; (void)ch;
}

struct GdfnWs {
    float *wp_in, *w2p, *wp_out;   // packed/padded weights
    float *wT_out, *wT_in;         // backward: [hp][c], [c][2hp]
    float *dt, *da, *du, *dxn;     // [M][hp], [M][2hp], [M][2hp], [M][c]
    float *slab, *wpart, *lnpart, *gpad;
    float *r_xn, *r_t;             // lean mode: LN(x) and gelu(x1) * x2 recomputed here when the caller did not keep them
    int ln_nblk, nblk_dwb;
};

size_t gdfn_layout(int B, int H, int W, int c, int hp, int backward, void* base, size_t bytes, GdfnWs* out) {
    WsAlloc a(base, base ? bytes : (size_t)-1);
    GdfnWs w{};
    const int64_t M = (int64_t)B * H * W;
    w.wp_in = a.get<float>((size_t)2 * hp * c);
    w.w2p = a.get<float>((size_t)18 * hp);
    w.wp_out = a.get<float>((size_t)c * hp);
    w.r_xn = a.get<float>((size_t)M * c);
    w.r_t = a.get<float>((size_t)M * hp);
    if (backward) {
        w.wT_out = a.get<float>((size_t)hp * c);
        w.wT_in = a.get<float>((size_t)c * 2 * hp);
        w.dt = a.get<float>((size_t)M * hp);
        w.da = a.get<float>((size_t)M * 2 * hp);
        w.du = a.get<float>((size_t)M * 2 * hp);
        w.dxn = a.get<float>((size_t)M * c);
        int sp1, sp2;
        int64_t r1, r2;
        gemm_tn_plan(M, 2 * hp, c, &sp1, &r1);
        gemm_tn_plan(M, c, hp, &sp2, &r2);
        const size_t s1 = (size_t)sp1 * 2 * hp * c, s2 = (size_t)sp2 * c * hp;
        w.slab = a.get<float>(s1 > s2 ? s1 : s2);
        w.nblk_dwb = dw_num_blocks_generic(B, H, W, 2 * hp);
        if (dwb_ring(B, H, W, 2 * hp)) {
            const int nr = dw_ring_bwd_num_blocks_per_image(DwGeom{B, H, W, hp});
            if (nr > w.nblk_dwb) w.nblk_dwb = nr;
        }
        w.wpart = a.get<float>((size_t)B * w.nblk_dwb * 10 * 2 * hp);
        w.ln_nblk = ln_bwd_num_blocks(M, c);
        w.lnpart = a.get<float>((size_t)w.ln_nblk * 3 * c);
        w.gpad = a.get<float>((size_t)2 * hp * (c > 9 ? c : 9) + (size_t)c * hp);
    }
    if (out) *out = w;
    return a.off;
}

int tn_reduce(const float* X, int ldx, int N, const float* Y, int ldy, int K, int yload, const GemmTN& proto, int64_t M,
              float* slab, float* dW, hipStream_t s) {
    GemmTN t = proto;
    t.X = X; t.ldx = ldx; t.N = N; t.Y = Y; t.ldy = ldy; t.K = K; t.M = M; t.slab = slab; t.colsum = nullptr;
    gemm_tn_plan(M, N, K, &t.splits, &t.rows_per_split);
    DCPT_TRY(launch_gemm_tn(t, A_PLAIN, yload, s));
    return launch_wgrad_reduce(slab, nullptr, t.splits, 0, N, K, nullptr, nullptr, nullptr, dW, nullptr, nullptr, WR_PLAIN, s);
}

}  // namespace

// =====================================================================================================
extern "C" size_t dcpt_mdta_ws_bytes(int B, int H, int W, int C, int heads, int backward) {
    return mdta_layout(B, H, W, C, heads, backward, nullptr, 0, nullptr);
}

extern "C" int dcpt_mdta_fwd(const dcpt_mdta_params* p, const float* x, float* y, const dcpt_mdta_saved* sv, void* ws,
                             size_t ws_bytes, int B, int H, int W, int C, int heads, int flags, dcpt_stream_t stream) {
    const int biasfree = flags & DCPT_LN_BIASFREE;
    const float ln_eps = (flags & DCPT_LN_EPS_1E5) ? 1e-5f : 1e-6f;
    const bool softmax_attn = (flags & DCPT_ATTN_SOFTMAX) != 0;
    (void)ln_eps; (void)softmax_attn;
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && x && y && sv, "mdta_fwd: null argument");
    DCPT_CHECK_ARG(heads > 0 && C % heads == 0 && (C / heads) % 4 == 0, "mdta_fwd: C=%d heads=%d (C/heads must be a multiple of 4)", C, heads);
    MdtaWs w;
    const size_t need = mdta_layout(B, H, W, C, heads, 0, ws, ws_bytes, &w);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("mdta_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W, ch = C / heads, C3 = 3 * C;
    // the normalised activations are materialised once (restormer_arch.py:40,59); the qkv conv and, in backward, its weight
    // gradient then take them as plain operands (global -> LDS by DMA).  Lean mode (saved.xn / qkv1 / out_att null): they live in
    // the workspace and the backward pass recomputes them (dcpt_hip.h).
    float* xn = sv->xn ? sv->xn : w.r_xn;
    float* qkv1 = sv->qkv1 ? sv->qkv1 : w.r_qkv1;
    float* out_att = sv->out_att ? sv->out_att : w.r_out;
    DCPT_TRY(launch_ln_fwd(x, p->norm_w, biasfree ? nullptr : p->norm_b, xn, sv->mu, sv->rstd, M, C, ln_eps, s));
    GemmNT g{};
    g.M = M; g.A = xn; g.lda = C; g.K = C; g.Bw = p->qkv_w; g.N = C3; g.C = qkv1; g.ldc = C3;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    DCPT_TRY(launch_dw_pack_weights(p->dw_w, w.w2p, C3, s));
    // (the register kernel: the ring form of this plain conv measured slower, 221 vs 191 us per launch -- with 3c / 2 = 72..576
    // channels per half its 128-byte runs are not line-aligned)
    DCPT_TRY(launch_dw_plain_fwd(qkv1, w.w2p, sv->qkv, w.sqpart, 2 * C, B, H, W, C3, s));
    sq_norm_kernel<<<dim3(cdiv(B * 2 * C, 256)), dim3(256), 0, s>>>(w.sqpart, w.nblk_dw, sv->nrm, B, 2 * C);
    DCPT_CHECK_LAUNCH("sq_norm");
    // Gram matrices per (image, head)
    GemmTN t{};
    t.M = P; t.X = sv->qkv; t.ldx = C3; t.N = ch; t.Y = sv->qkv + C; t.ldy = C3; t.K = ch;
    t.nb1 = B; t.nb2 = heads; t.sX1 = (int64_t)P * C3; t.sX2 = ch; t.sY1 = (int64_t)P * C3; t.sY2 = ch;
    t.slab = w.gslab; t.splits = w.splits_a; t.rows_per_split = cdiv64(cdiv64(P, w.splits_a), 32) * 32;
    t.splits = (int)cdiv64(P, t.rows_per_split);
    DCPT_TRY(launch_gemm_tn(t, A_PLAIN, A_PLAIN, s));
    if (softmax_attn) {
        DCPT_CHECK_ARG(ch <= 64 * SM_MAXU, "mdta: softmax attention supports up to %d channels per head", 64 * SM_MAXU);
        attn_finalize_softmax_kernel<<<dim3(B * heads), dim3(256), 0, s>>>(w.gslab, t.splits, sv->nrm, p->temperature, sv->ghat,
                                                                           sv->attn, sv->attnT, heads, ch, C);
    } else {
        attn_finalize_kernel<<<dim3(B * heads), dim3(256), 0, s>>>(w.gslab, t.splits, sv->nrm, p->temperature, sv->ghat, sv->attn,
                                                                   sv->attnT, heads, ch, C);
    }
    DCPT_CHECK_LAUNCH("attn_finalize");
    // out_att[p][i] = sum_j attn[i][j] v[p][j]
    g = GemmNT{};
    g.M = P; g.A = sv->qkv + 2 * C; g.lda = C3; g.K = ch; g.Bw = sv->attn; g.N = ch; g.C = out_att; g.ldc = C;
    g.nb1 = B; g.nb2 = heads; g.sA1 = (int64_t)P * C3; g.sA2 = ch; g.sB1 = (int64_t)heads * ch * ch; g.sB2 = (int64_t)ch * ch;
    g.sC1 = (int64_t)P * C; g.sC2 = ch;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    // y = x + project_out(out_att)
    g = GemmNT{};
    g.M = M; g.A = out_att; g.lda = C; g.K = C; g.Bw = p->proj_w; g.N = C; g.C = y; g.ldc = C; g.res = x;
    return launch_gemm_nt(g, A_PLAIN, E_RESID, s);
}

extern "C" int dcpt_mdta_bwd(const dcpt_mdta_params* p, const dcpt_mdta_params_grads* gr, const float* x,
                             const dcpt_mdta_saved* sv, const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int H,
                             int W, int C, int heads, int flags, dcpt_stream_t stream) {
    const int biasfree = flags & DCPT_LN_BIASFREE;
    const float ln_eps = (flags & DCPT_LN_EPS_1E5) ? 1e-5f : 1e-6f;
    const bool softmax_attn = (flags & DCPT_ATTN_SOFTMAX) != 0;
    (void)ln_eps; (void)softmax_attn;
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && gr && x && sv && dy && dx, "mdta_bwd: null argument");
    DCPT_CHECK_ARG(heads > 0 && C % heads == 0 && (C / heads) % 4 == 0, "mdta_bwd: bad shape");
    MdtaWs w;
    const size_t need = mdta_layout(B, H, W, C, heads, 1, ws, ws_bytes, &w);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("mdta_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    const int P = H * W, ch = C / heads, C3 = 3 * C;
    GemmNT g{};
    GemmTN tp{};
    // (the weight-gradient side stream of side.hip measured 1.7 % SLOWER on Restormer -- its depthwise kernels lose more to
    // the co-running GEMMs than the GEMMs gain -- so these blocks keep everything on the caller's stream)
    Side* sd = nullptr;
    hipStream_t sw = side_stream(sd, s);
    DCPT_TRY(side_fork(sd, 0, s));          // dy and the saved activations are ready
    // lean mode: rebuild what the forward pass did not keep -- LN(x) (one bandwidth pass), the qkv conv output (the depthwise
    // conv's tap gradients need it: one GEMM, +8 % of the block's flops) and attn @ v (a small batched GEMM)
    const float* xn = sv->xn;
    const float* qkv1 = sv->qkv1;
    const float* out_att = sv->out_att;
    if (!xn) {
        DCPT_TRY(launch_ln_fwd(x, p->norm_w, biasfree ? nullptr : p->norm_b, w.r_xn, w.scr3, w.scr3 + M, M, C, ln_eps, s));
        xn = w.r_xn;
    }
    if (!qkv1) {
        g.M = M; g.A = xn; g.lda = C; g.K = C; g.Bw = p->qkv_w; g.N = C3; g.C = w.r_qkv1; g.ldc = C3;
        DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
        qkv1 = w.r_qkv1;
        g = GemmNT{};
    }
    if (!out_att) {
        g.M = P; g.A = sv->qkv + 2 * C; g.lda = C3; g.K = ch; g.Bw = sv->attn; g.N = ch; g.C = w.r_out; g.ldc = C;
        g.nb1 = B; g.nb2 = heads; g.sA1 = (int64_t)P * C3; g.sA2 = ch; g.sB1 = (int64_t)heads * ch * ch; g.sB2 = (int64_t)ch * ch;
        g.sC1 = (int64_t)P * C; g.sC2 = ch;
        DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
        out_att = w.r_out;
        g = GemmNT{};
    }
    // B1: d_att = dy * Wproj ; dWproj = dy^T out_att
    DCPT_TRY(launch_wpack(p->proj_w, w.wT_proj, nullptr, C, C, WP_TRANSPOSE, s));
    g.M = M; g.A = dy; g.lda = C; g.K = C; g.Bw = w.wT_proj; g.N = C; g.C = w.d_att; g.ldc = C;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    DCPT_TRY(tn_reduce(dy, C, C, out_att, C, C, A_PLAIN, tp, M, w.slab, gr->proj_w, sw));
    // B2: dattn[i][j] = sum_p d_att[p][i] v[p][j]   (batched TN)
    GemmTN t{};
    t.M = P; t.X = w.d_att; t.ldx = C; t.N = ch; t.Y = sv->qkv + 2 * C; t.ldy = C3; t.K = ch;
    t.nb1 = B; t.nb2 = heads; t.sX1 = (int64_t)P * C; t.sX2 = ch; t.sY1 = (int64_t)P * C3; t.sY2 = ch;
    t.slab = w.gslab; t.rows_per_split = cdiv64(cdiv64(P, w.splits_a), 32) * 32;
    t.splits = (int)cdiv64(P, t.rows_per_split);
    DCPT_TRY(launch_gemm_tn(t, A_PLAIN, A_PLAIN, s));
    //     dv[p][j] = sum_i d_att[p][i] attn[i][j]        (batched NT with attn^T)
    g = GemmNT{};
    g.M = P; g.A = w.d_att; g.lda = C; g.K = ch; g.Bw = sv->attnT; g.N = ch; g.C = w.dqkv + 2 * C; g.ldc = C3;
    g.nb1 = B; g.nb2 = heads; g.sA1 = (int64_t)P * C; g.sA2 = ch; g.sB1 = (int64_t)heads * ch * ch; g.sB2 = (int64_t)ch * ch;
    g.sC1 = (int64_t)P * C3; g.sC2 = ch;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    // B3: through relu / temperature / normalisation
    if (softmax_attn) {
        DCPT_CHECK_ARG(ch <= 256, "mdta: softmax attention supports up to 256 channels per head");
        attn_bwd_kernel<true><<<dim3(B * heads), dim3(256), 0, s>>>(w.gslab, t.splits, sv->attn, sv->ghat, sv->nrm, p->temperature,
                                                                    w.dG, w.dGT, w.cqk, w.dtpart, w.scr, heads, ch, C);
    } else {
        attn_bwd_kernel<false><<<dim3(B * heads), dim3(256), 0, s>>>(w.gslab, t.splits, sv->attn, sv->ghat, sv->nrm, p->temperature,
                                                                     w.dG, w.dGT, w.cqk, w.dtpart, w.scr, heads, ch, C);
    }
    DCPT_CHECK_LAUNCH("attn_bwd");
    dtemp_reduce_kernel<<<dim3(cdiv(heads, 64)), dim3(64), 0, s>>>(w.dtpart, gr->temperature, B, heads);
    DCPT_CHECK_LAUNCH("dtemp_reduce");
    // B4: dq[p][i] = sum_j dG[i][j] k[p][j] + cq_i q[p][i] ;  dk[p][j] = sum_i dG[i][j] q[p][i] + ck_j k[p][j]
    g = GemmNT{};
    g.M = P; g.A = sv->qkv + C; g.lda = C3; g.K = ch; g.Bw = w.dG; g.N = ch; g.C = w.dqkv; g.ldc = C3;
    g.res = sv->qkv; g.ldres = C3; g.cscale = w.cqk;
    g.nb1 = B; g.nb2 = heads; g.sA1 = (int64_t)P * C3; g.sA2 = ch; g.sB1 = (int64_t)heads * ch * ch; g.sB2 = (int64_t)ch * ch;
    g.sC1 = (int64_t)P * C3; g.sC2 = ch; g.sR1 = (int64_t)P * C3; g.sR2 = ch; g.sS1 = 2 * C; g.sS2 = ch;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_ADDSCALED, s));
    g.A = sv->qkv; g.Bw = w.dGT; g.C = w.dqkv + C; g.res = sv->qkv + C; g.cscale = w.cqk + C;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_ADDSCALED, s));
    // B5: depthwise backward
    DCPT_TRY(launch_dw_pack_weights(p->dw_w, w.w2p, C3, s));
    int nblk_dwb = dw_num_blocks_generic(B, H, W, C3);
    if (dwb_ring(B, H, W, C3)) {   // transposed conv + tap gradients on the LDS-DMA row ring (dwring.hip)
        nblk_dwb = dw_ring_bwd_num_blocks_per_image(DwGeom{B, H, W, C3 / 2});
        DCPT_TRY(launch_dw_ring_bwd_plain_f32(w.dqkv, qkv1, w.w2p, w.dqkv1, w.wpart, B, H, W, C3, s));
    } else {
        DCPT_TRY(launch_dw_generic_bwd(w.dqkv, qkv1, w.w2p, w.dqkv1, w.wpart, B, H, W, C3, s));
    }
    DCPT_TRY(side_fork(sd, 1, s));          // dqkv1, depthwise partial sums
    DCPT_TRY(launch_dw_wgrad_reduce(w.wpart, B * nblk_dwb, C3, gr->dw_w, w.scr2 /*unused bias grad*/, sw));
    // B6: qkv 1x1
    DCPT_TRY(launch_wpack(p->qkv_w, w.wT_qkv, nullptr, C3, C, WP_TRANSPOSE, s));
    g = GemmNT{};
    g.M = M; g.A = w.dqkv1; g.lda = C3; g.K = C3; g.Bw = w.wT_qkv; g.N = C; g.C = w.dxn; g.ldc = C;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    tp = GemmTN{};
    DCPT_TRY(tn_reduce(w.dqkv1, C3, C3, xn, C, C, A_PLAIN, tp, M, w.slab, gr->qkv_w, sw));
    // B7: dx = dy + LN-backward
    DCPT_TRY(launch_ln_bwd_ex(w.dxn, x, sv->mu, sv->rstd, p->norm_w, dy, nullptr, nullptr, biasfree, dx, w.lnpart, w.ln_nblk, M, C, s));
    DCPT_TRY(side_fork(sd, 2, s));          // LayerNorm partial sums
    DCPT_TRY(launch_colpart_reduce(w.lnpart, w.ln_nblk, 3, C, gr->norm_w, biasfree ? nullptr : gr->norm_b, nullptr, sw));
    return side_join(sd, s);                // the caller's stream continues only after every weight gradient is written
}

// =====================================================================================================
extern "C" size_t dcpt_gdfn_ws_bytes(int B, int H, int W, int C, int hidden, int backward) {
    return gdfn_layout(B, H, W, C, (hidden + 3) / 4 * 4, backward, nullptr, 0, nullptr);
}

extern "C" int dcpt_gdfn_fwd(const dcpt_gdfn_params* p, const float* x, float* y, const dcpt_gdfn_saved* sv, void* ws,
                             size_t ws_bytes, int B, int H, int W, int C, int hidden, int flags, dcpt_stream_t stream) {
    const int biasfree = flags & DCPT_LN_BIASFREE;
    const float ln_eps = (flags & DCPT_LN_EPS_1E5) ? 1e-5f : 1e-6f;
    const bool softmax_attn = (flags & DCPT_ATTN_SOFTMAX) != 0;
    (void)ln_eps; (void)softmax_attn;
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && x && y && sv && hidden > 0 && C % 4 == 0, "gdfn_fwd: bad argument");
    const int hp = (hidden + 3) / 4 * 4;
    GdfnWs w;
    const size_t need = gdfn_layout(B, H, W, C, hp, 0, ws, ws_bytes, &w);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("gdfn_fwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)2 * hp * C)), dim3(256), 0, s>>>(p->in_w, w.wp_in, C, hidden, hp, 0);
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)18 * hp)), dim3(256), 0, s>>>(p->dw_w, w.w2p, C, hidden, hp, 1);
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)C * hp)), dim3(256), 0, s>>>(p->out_w, w.wp_out, C, hidden, hp, 2);
    DCPT_CHECK_LAUNCH("gdfn_pack");
    float* xn = sv->xn ? sv->xn : w.r_xn;   // lean mode: LN(x) and the gated product live in the workspace (recomputed in backward)
    float* tg = sv->t ? sv->t : w.r_t;
    DCPT_TRY(launch_ln_fwd(x, p->norm_w, biasfree ? nullptr : p->norm_b, xn, sv->mu, sv->rstd, M, C, ln_eps, s));
    GemmNT g{};
    g.M = M; g.A = xn; g.lda = C; g.K = C; g.Bw = w.wp_in; g.N = 2 * hp; g.C = sv->u; g.ldc = 2 * hp;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    if (dwf_ring(B, H, W, 2 * hp)) DCPT_TRY(launch_dw_ring_gelu_fwd_f32(sv->u, w.w2p, tg, B, H, W, hp, s));
    else DCPT_TRY(launch_dw_gelu_fwd(sv->u, w.w2p, tg, B, H, W, hp, s));
    g = GemmNT{};
    g.M = M; g.A = tg; g.lda = hp; g.K = hp; g.Bw = w.wp_out; g.N = C; g.C = y; g.ldc = C; g.res = x;
    return launch_gemm_nt(g, A_PLAIN, E_RESID, s);
}

extern "C" int dcpt_gdfn_bwd(const dcpt_gdfn_params* p, const dcpt_gdfn_params_grads* gr, const float* x,
                             const dcpt_gdfn_saved* sv, const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int H,
                             int W, int C, int hidden, int flags, dcpt_stream_t stream) {
    const int biasfree = flags & DCPT_LN_BIASFREE;
    const float ln_eps = (flags & DCPT_LN_EPS_1E5) ? 1e-5f : 1e-6f;
    const bool softmax_attn = (flags & DCPT_ATTN_SOFTMAX) != 0;
    (void)ln_eps; (void)softmax_attn;
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(p && gr && x && sv && dy && dx && hidden > 0 && C % 4 == 0, "gdfn_bwd: bad argument");
    const int hp = (hidden + 3) / 4 * 4;
    GdfnWs w;
    const size_t need = gdfn_layout(B, H, W, C, hp, 1, ws, ws_bytes, &w);
    if (ws == nullptr || need > ws_bytes) {
        dcpt_set_error("gdfn_bwd: workspace too small");
        return DCPT_ERR_WS;
    }
    const int64_t M = (int64_t)B * H * W;
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)18 * hp)), dim3(256), 0, s>>>(p->dw_w, w.w2p, C, hidden, hp, 1);
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)hp * C)), dim3(256), 0, s>>>(p->out_w, w.wT_out, C, hidden, hp, 3);
    gdfn_pack_kernel<<<dim3(grid_for((int64_t)C * 2 * hp)), dim3(256), 0, s>>>(p->in_w, w.wT_in, C, hidden, hp, 4);
    DCPT_CHECK_LAUNCH("gdfn_pack");
    float* g_in = w.gpad;                       // [2hp][c] (also reused as [2hp*9])
    float* g_out = w.gpad + (size_t)2 * hp * (C > 9 ? C : 9);  // [c][hp]
    GemmNT g{};
    GemmTN tp{};
    Side* sd = nullptr;                     // see dcpt_mdta_bwd: no side stream for the Restormer blocks
    hipStream_t sw = side_stream(sd, s);
    DCPT_TRY(side_fork(sd, 0, s));          // dy and the saved activations are ready
    const float* xn = sv->xn;
    const float* tg = sv->t;
    if (!xn) {   // lean mode: rebuild LN(x) (statistics discarded into du, which is written later) and gelu(x1) * x2
        DCPT_TRY(launch_ln_fwd(x, p->norm_w, biasfree ? nullptr : p->norm_b, w.r_xn, w.du, w.du + M, M, C, ln_eps, s));
        xn = w.r_xn;
    }
    // the gate product the caller did not keep: the fused backward on the ring recomputes the conv output anyway and writes
    // gelu(x1) * x2 along the way (no extra pass; the project_out weight gradient then runs after it); otherwise the forward kernel
    const bool t_from_bwd = !tg && dwb_ring(B, H, W, 2 * hp);
    if (!tg && !t_from_bwd) {
        if (dwf_ring(B, H, W, 2 * hp)) DCPT_TRY(launch_dw_ring_gelu_fwd_f32(sv->u, w.w2p, w.r_t, B, H, W, hp, s));
        else DCPT_TRY(launch_dw_gelu_fwd(sv->u, w.w2p, w.r_t, B, H, W, hp, s));
    }
    if (!tg) tg = w.r_t;
    // dt = dy * Wout ; dWout = dy^T t
    g.M = M; g.A = dy; g.lda = C; g.K = C; g.Bw = w.wT_out; g.N = hp; g.C = w.dt; g.ldc = hp;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    if (!t_from_bwd) {
        DCPT_TRY(tn_reduce(dy, C, C, tg, hp, hp, A_PLAIN, tp, M, w.slab, g_out, sw));
        gdfn_unpack_kernel<<<dim3(grid_for((int64_t)C * hidden)), dim3(256), 0, sw>>>(g_out, gr->out_w, C, hidden, hp, 2);
        DCPT_CHECK_LAUNCH("gdfn_unpack_out");
    }
    // gate backward, depthwise backward
    int nblk_dwb = dw_num_blocks_generic(B, H, W, 2 * hp);
    if (dwb_ring(B, H, W, 2 * hp)) {   // one pass on the row ring: the gate's da never goes to memory
        nblk_dwb = dw_ring_bwd_num_blocks_per_image(DwGeom{B, H, W, hp});
        DCPT_TRY(launch_dw_ring_bwd_gelu_f32(w.dt, sv->u, w.w2p, w.du, w.wpart, B, H, W, hp, s, t_from_bwd ? w.r_t : nullptr));
        if (t_from_bwd) {
            DCPT_TRY(tn_reduce(dy, C, C, tg, hp, hp, A_PLAIN, tp, M, w.slab, g_out, sw));
            gdfn_unpack_kernel<<<dim3(grid_for((int64_t)C * hidden)), dim3(256), 0, sw>>>(g_out, gr->out_w, C, hidden, hp, 2);
            DCPT_CHECK_LAUNCH("gdfn_unpack_out");
        }
    } else {
        DCPT_TRY(launch_dw_gelu_bwd_a(w.dt, sv->u, w.w2p, w.da, B, H, W, hp, s));
        DCPT_TRY(launch_dw_generic_bwd(w.da, sv->u, w.w2p, w.du, w.wpart, B, H, W, 2 * hp, s));
    }
    DCPT_TRY(side_fork(sd, 1, s));          // du, depthwise partial sums (dt is dead from here on)
    DCPT_TRY(launch_dw_wgrad_reduce(w.wpart, B * nblk_dwb, 2 * hp, g_in, w.dt /*unused bias grad*/, sw));
    gdfn_unpack_kernel<<<dim3(grid_for((int64_t)2 * hidden * 9)), dim3(256), 0, sw>>>(g_in, gr->dw_w, C, hidden, hp, 1);
    DCPT_CHECK_LAUNCH("gdfn_unpack_dw");
    // project_in
    g = GemmNT{};
    g.M = M; g.A = w.du; g.lda = 2 * hp; g.K = 2 * hp; g.Bw = w.wT_in; g.N = C; g.C = w.dxn; g.ldc = C;
    DCPT_TRY(launch_gemm_nt(g, A_PLAIN, E_PLAIN, s));
    tp = GemmTN{};
    DCPT_TRY(tn_reduce(w.du, 2 * hp, 2 * hp, xn, C, C, A_PLAIN, tp, M, w.slab, g_in, sw));
    gdfn_unpack_kernel<<<dim3(grid_for((int64_t)2 * hidden * C)), dim3(256), 0, sw>>>(g_in, gr->in_w, C, hidden, hp, 0);
    DCPT_CHECK_LAUNCH("gdfn_unpack_in");
    DCPT_TRY(launch_ln_bwd_ex(w.dxn, x, sv->mu, sv->rstd, p->norm_w, dy, nullptr, nullptr, biasfree, dx, w.lnpart, w.ln_nblk, M, C, s));
    DCPT_TRY(side_fork(sd, 2, s));          // LayerNorm partial sums
    DCPT_TRY(launch_colpart_reduce(w.lnpart, w.ln_nblk, 3, C, gr->norm_w, biasfree ? nullptr : gr->norm_b, nullptr, sw));
    return side_join(sd, s);                // the caller's stream continues only after every weight gradient is written
}

// =====================================================================================================
extern "C" int dcpt_pixel_unshuffle(const float* x, float* y, int B, int H, int W, int C, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && y && H % 2 == 0 && W % 2 == 0, "pixel_unshuffle: bad argument");
    pixel_shuffle_kernel<<<dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream>>>(x, y, B, H / 2, W / 2, C, 0);
    DCPT_CHECK_LAUNCH("pixel_unshuffle");
    return DCPT_OK;
}

extern "C" int dcpt_pixel_shuffle(const float* x, float* y, int B, int H, int W, int C4, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(x && y && C4 % 4 == 0, "pixel_shuffle: bad argument");
    pixel_shuffle_kernel<<<dim3(grid_for((int64_t)B * H * W * C4)), dim3(256), 0, (hipStream_t)stream>>>(x, y, B, H, W, C4 / 4, 1);
    DCPT_CHECK_LAUNCH("pixel_shuffle");
    return DCPT_OK;
}

extern "C" int dcpt_concat_channels(const float* a, const float* b, float* out, int64_t M, int Ca, int Cb, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(a && b && out && Ca % 4 == 0 && Cb % 4 == 0, "concat_channels: bad argument");
    concat_kernel<<<dim3(grid_for(M * ((Ca + Cb) / 4))), dim3(256), 0, (hipStream_t)stream>>>((float*)a, (float*)b, out, M, Ca, Cb, 0);
    DCPT_CHECK_LAUNCH("concat_channels");
    return DCPT_OK;
}

extern "C" int dcpt_split_channels(const float* cat, float* a, float* b, int64_t M, int Ca, int Cb, dcpt_stream_t stream) {
    DCPT_CHECK_ARG(a && b && cat && Ca % 4 == 0 && Cb % 4 == 0, "split_channels: bad argument");
    concat_kernel<<<dim3(grid_for(M * ((Ca + Cb) / 4))), dim3(256), 0, (hipStream_t)stream>>>(a, b, (float*)cat, M, Ca, Cb, 1);
    DCPT_CHECK_LAUNCH("split_channels");
    return DCPT_OK;
}
