// Small kernels around the GEMMs: SCA (simplified channel attention) forward/backward pieces,
// weight packing/transposition, and the deterministic split-slab reduction of weight gradients.
// Reference: basicsr/archs/nafnet_arch.py:116-127,173 (SCA), :162-163,178,186 (beta/gamma).
#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// SCA forward. grid (B, row-chunks of SCA_OPB outputs); every block first rebuilds pooled[b][:] in LDS.  Latency-bound:
// many small blocks, all loads of a wave's outputs issued together.
constexpr int SCA_OPB = 16;   // outputs per block: 4 per wave
__global__ __launch_bounds__(256) void sca_fwd_kernel(const float* __restrict__ pool_part, int nblk,
                                                      const float* __restrict__ Wsca, const float* __restrict__ bsca,
                                                      float* __restrict__ pooled, float* __restrict__ simg, int C, float invP) {
    extern __shared__ __attribute__((aligned(16))) float pl[];  // [C]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < C; k += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < nblk; ++j) s += pool_part[((int64_t)b * nblk + j) * C + k];
        s *= invP;
        pl[k] = s;
        if (blockIdx.y == 0) pooled[(int64_t)b * C + k] = s;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int nbeg = blockIdx.y * SCA_OPB + wave * (SCA_OPB / 4);
    float acc[SCA_OPB / 4];
#pragma unroll
    for (int i = 0; i < SCA_OPB / 4; ++i) acc[i] = 0.f;
    for (int k = 4 * lane; k < C; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(&pl[k]);
#pragma unroll
        for (int i = 0; i < SCA_OPB / 4; ++i) {
            const int n = nbeg + i;
            if (n < C) acc[i] += f4_sum(f4_mul(ldg4(Wsca + (int64_t)n * C + k), v));
        }
    }
#pragma unroll
    for (int i = 0; i < SCA_OPB / 4; ++i) {
        const float s = group_sum(acc[i], 64);
        const int n = nbeg + i;
        if (lane == 0 && n < C) simg[(int64_t)b * C + n] = s + bsca[n];
    }
}

// ds stage 1: part[b][j][k] = sum over the j-th pixel slice of image b of dts*t2
__global__ __launch_bounds__(256) void sca_ds_part_kernel(const float* __restrict__ dts, const float* __restrict__ t2,
                                                          float* __restrict__ part, int C, int P, int nslices) {
    __shared__ float4 red[256];
    const int b = blockIdx.z, j = blockIdx.y;
    const int nq = C / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    const int pb = 256 / qb;
    const int tid = threadIdx.x, ql = tid % qb, pl = tid / qb;
    const int q = blockIdx.x * qb + ql;
    const bool qok = q < nq;
    const int per = (P + nslices - 1) / nslices;
    const int pbeg = j * per;
    int pend = pbeg + per;
    if (pend > P) pend = P;
    float4 acc = f4_zero();
    if (qok)
        for (int px = pbeg + pl; px < pend; px += pb) {
            const int64_t o = ((int64_t)b * P + px) * C + 4 * q;
            acc = f4_fma(ldg4(dts + o), ldg4(t2 + o), acc);
        }
    red[tid] = acc;
    __syncthreads();
    if (pl == 0 && qok) {
        float4 s = red[ql];
        for (int i = 1; i < pb; ++i) s = f4_add(s, red[i * qb + ql]);
        stg4(part + ((int64_t)b * nslices + j) * C + 4 * q, s);
    }
}

__global__ void sca_ds_final_kernel(const float* __restrict__ part, float* __restrict__ ds, int BC, int C, int nslices) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, k = i % C;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < nslices; ++j) s += part[((int64_t)b * nslices + j) * C + k];
    ds[i] = s;
}

// SCA backward, critical-path part:  dpool[b][k] = invP * sum_n Wsca[n][k] * ds[b][n],  ds[b][n] = sum_j part[b][j][n].
// grid (C/32, B): 32 k-columns x 8 n-groups; the block first sums the pixel slices of its image into LDS (so there is no
// separate ds pass on this path), then every thread runs an 8-way unrolled dot over its n values.
__global__ __launch_bounds__(256) void sca_dpool_kernel(const float* __restrict__ part, int nslices, const float* __restrict__ Wsca,
                                                        float* __restrict__ dpool, int C, float invP, float* __restrict__ ds_out) {
    extern __shared__ __attribute__((aligned(16))) float dsl[];   // ds[C]
    __shared__ float red[8][32];
    __shared__ float psum[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (C < 256 && (256 % C) == 0) {
        // narrow levels have many slices (up to 512 per image) and few columns: 256 / C threads share a column, each sums
        // every (256 / C)-th slice, and the partial sums are combined in a fixed order
        const int parts = 256 / C, n = tid % C, pt = tid / C;
        float v = 0.f;
#pragma unroll 8
        for (int j = pt; j < nslices; j += parts) v += part[((int64_t)b * nslices + j) * C + n];
        psum[tid] = v;
        __syncthreads();
        if (pt == 0) {
            float t = psum[n];
            for (int q = 1; q < parts; ++q) t += psum[q * C + n];
            dsl[n] = t;
        }
    } else {
        for (int n = tid; n < C; n += 256) {
            float v = 0.f;
#pragma unroll 8
            for (int j = 0; j < nslices; ++j) v += part[((int64_t)b * nslices + j) * C + n];
            dsl[n] = v;
        }
    }
    __syncthreads();
    if (ds_out != nullptr && blockIdx.x == 0)   // (the parameter-gradient side reads ds instead of summing the slices again)
        for (int n = tid; n < C; n += 256) ds_out[(int64_t)b * C + n] = dsl[n];
    const int kl = tid & 31, ng = tid >> 5;
    const int k = blockIdx.x * 32 + kl;
    float s = 0.f;
    if (k < C) {
#pragma unroll 8
        for (int n = ng; n < C; n += 8) s = fmaf(Wsca[(int64_t)n * C + k], dsl[n], s);
    }
    red[ng][kl] = s;
    __syncthreads();
    if (ng == 0 && k < C) {
        float t = red[0][kl];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i][kl];
        dpool[(int64_t)b * C + k] = t * invP;
    }
}
//   role 0: dWsca[n][k] = sum_b ds[b][n] * pooled[b][k]     role 1: dbsca[n] = sum_b ds[b][n]
__global__ __launch_bounds__(256) void sca_bwd_w_kernel(const float* __restrict__ ds, const float* __restrict__ pooled,
                                                        float* __restrict__ dWsca, float* __restrict__ dbsca, int B, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {
        if (i >= (int64_t)C * C) return;
        const int n = (int)(i / C), k = (int)(i % C);
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s = fmaf(ds[(int64_t)b * C + n], pooled[(int64_t)b * C + k], s);
        dWsca[i] = s;
    } else {
        if (i >= C) return;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += ds[(int64_t)b * C + i];
        dbsca[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void wpack_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ rs, int N,
                             int K, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    // i indexes the OUTPUT (coalesced writes)
    if (mode == WP_TRANSPOSE) {
        const int k = (int)(i / N), n = (int)(i % N);
        const float v = in[(int64_t)n * K + k];
        out[i] = rs ? v * rs[n] : v;
    } else if (mode == WP_DOWN) {  // out[oc][ij*C + ic], K = 4C
        const int C = K / 4;
        const int oc = (int)(i / K), r = (int)(i % K);
        const int ij = r / C, ic = r % C;
        out[i] = in[((int64_t)oc * C + ic) * 4 + ij];
    } else if (mode == WP_DOWN_T) {  // out[ij*C + ic][oc]
        const int C = K / 4;
        const int r = (int)(i / N), oc = (int)(i % N);
        const int ij = r / C, ic = r % C;
        out[i] = in[((int64_t)oc * C + ic) * 4 + ij];
    } else if (mode == WP_UP) {  // out[ij*G + kk][ic] = in[4kk+ij][ic]
        const int G = N / 4;
        const int r = (int)(i / K), ic = (int)(i % K);
        const int ij = r / G, kk = r % G;
        out[i] = in[(int64_t)(4 * kk + ij) * K + ic];
    } else if (mode == WP_UP_T) {  // out[ic][ij*G + kk] = in[4kk+ij][ic]
        const int G = N / 4;
        const int ic = (int)(i / N), r = (int)(i % N);
        const int ij = r / G, kk = r % G;
        out[i] = in[(int64_t)(4 * kk + ij) * K + ic];
    } else if (mode == WP_CONV3) {  // N = Co, K = 9*Ci: out[oc][tap*Ci + ic] = in[(oc*Ci + ic)*9 + tap]
        const int Ci = K / 9;
        const int oc = (int)(i / K), r = (int)(i % K);
        const int tap = r / Ci, ic = r % Ci;
        out[i] = in[((int64_t)oc * Ci + ic) * 9 + tap];
    } else {  // WP_CONV3_T: N = Co, K = 9*Ci (of the forward conv); out[ic][tap*Co + oc] = in[(oc*Ci+ic)*9 + 8-tap]
        const int Ci = K / 9, Co = N;
        const int ic = (int)(i / (9 * Co)), r = (int)(i % (9 * Co));
        const int tap = r / Co, oc = r % Co;
        out[i] = in[((int64_t)oc * Ci + ic) * 9 + (8 - tap)];
    }
}

// one block per output row n; threads = (split group, k quad)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ colsum,
                                                           int splits, int N, int K, const float* __restrict__ rowscale,
                                                           const float* __restrict__ W, const float* __restrict__ wbias,
                                                           float* __restrict__ dW, float* __restrict__ dgain,
                                                           float* __restrict__ dbias, int mode, int tpr, int cs_rows,
                                                           const float* __restrict__ kscale, int ks_div) {
    __shared__ float4 red4[256];
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int sg = 256 / tpr;             // split groups
    const int ql = tid % tpr, sgi = tid / tpr;
    const float rsc = rowscale ? rowscale[n] : 1.f;
    float dot = 0.f;
    for (int kb = 0; kb < K; kb += 4 * tpr) {
        const int k = kb + 4 * ql;
        float4 g = f4_zero();
        if (k < K) {
            // 4 independent partial sums keep 4 loads in flight per thread (fixed order -> still deterministic)
            float4 g1 = f4_zero(), g2 = f4_zero(), g3 = f4_zero();
            const float* base = slab + (int64_t)n * K + k;
            const int64_t sstride = (int64_t)N * K;
            int s = sgi;
            if (kscale == nullptr) {
                for (; s + 3 * sg < splits; s += 4 * sg) {
                    g = f4_add(g, ldg4(base + (int64_t)s * sstride));
                    g1 = f4_add(g1, ldg4(base + (int64_t)(s + sg) * sstride));
                    g2 = f4_add(g2, ldg4(base + (int64_t)(s + 2 * sg) * sstride));
                    g3 = f4_add(g3, ldg4(base + (int64_t)(s + 3 * sg) * sstride));
                }
                for (; s < splits; s += sg) g = f4_add(g, ldg4(base + (int64_t)s * sstride));
            } else {
                // every split lies inside one image: its slab is weighted by that image's per-column scale (SCA)
                const float* ks = kscale + k;
                for (; s + 3 * sg < splits; s += 4 * sg) {
                    g = f4_fma(ldg4(base + (int64_t)s * sstride), ldg4(ks + (int64_t)(s / ks_div) * K), g);
                    g1 = f4_fma(ldg4(base + (int64_t)(s + sg) * sstride), ldg4(ks + (int64_t)((s + sg) / ks_div) * K), g1);
                    g2 = f4_fma(ldg4(base + (int64_t)(s + 2 * sg) * sstride), ldg4(ks + (int64_t)((s + 2 * sg) / ks_div) * K), g2);
                    g3 = f4_fma(ldg4(base + (int64_t)(s + 3 * sg) * sstride), ldg4(ks + (int64_t)((s + 3 * sg) / ks_div) * K), g3);
                }
                for (; s < splits; s += sg) g = f4_fma(ldg4(base + (int64_t)s * sstride), ldg4(ks + (int64_t)(s / ks_div) * K), g);
            }
            g = f4_add(f4_add(g, g1), f4_add(g2, g3));
        }
        if (sg > 1) {
            __syncthreads();
            red4[tid] = g;
            __syncthreads();
            if (sgi == 0) {
                for (int i = 1; i < sg; ++i) g = f4_add(g, red4[i * tpr + ql]);
            }
        }
        if (sgi != 0 || k >= K) continue;
        if (dgain) dot += f4_sum(f4_mul(g, ldg4(W + (int64_t)n * K + k)));
        const float4 o = f4_scale(g, rsc);
        if (mode == WR_PLAIN) {
            stg4(dW + (int64_t)n * K + k, o);
        } else if (mode == WR_DOWN) {  // packed k = ij*C + ic  ->  dW[n][ic][ij]
            const int C = K / 4;
            const int ij = k / C, ic = k % C;
            float* d = dW + ((int64_t)n * C + ic) * 4 + ij;
            d[0] = o.x; d[4] = o.y; d[8] = o.z; d[12] = o.w;
        } else if (mode == WR_UP) {  // packed row n = ij*G + kk -> dW[4kk+ij][k]
            const int G = N / 4;
            const int ij = n / G, kk = n % G;
            stg4(dW + (int64_t)(4 * kk + ij) * K + k, o);
        } else {  // WR_CONV3: packed k = tap*Ci + ic -> dW[n][ic][tap]
            const int Ci = K / 9;
            const int tap = k / Ci, ic = k % Ci;
            float* d = dW + ((int64_t)n * Ci + ic) * 9 + tap;
            d[0] = o.x; d[9] = o.y; d[18] = o.z; d[27] = o.w;
        }
    }
    if (dgain == nullptr && dbias == nullptr) return;
    // column sum of this row: cs_rows partial rows, one load per thread + a fixed-order block reduction
    float cs = 0.f;
    if (colsum)
        for (int r = tid; r < cs_rows; r += 256) cs += colsum[(int64_t)r * N + n];
    cs = group_sum(cs, 64);
    dot = group_sum(dot, 64);
    __syncthreads();
    if ((tid & 63) == 0) {
        red[tid >> 6] = cs;
        red4[tid >> 6].x = dot;
    }
    __syncthreads();
    if (tid == 0) {
        cs = (red[0] + red[1]) + (red[2] + red[3]);
        dot = (red4[0].x + red4[1].x) + (red4[2].x + red4[3].x);
        if (dgain) dgain[n] = dot + (wbias ? wbias[n] * cs : 0.f);
        if (dbias) dbias[n] = rsc * cs;
    }
}

// ---------------------------------------------------------------------------------------------
// TLSC local average pooling (reference basicsr/archs/arch_util.py:378-396): box mean with a k1 x k2 window, the
// (H-k1+1) x (W-k2+1) result replicate-padded back to H x W.  Separable running sums, exact window sums (no prefix-sum
// cancellation): pass 1 along W, pass 2 along H (+ divide).
// pass 1: rs[b][h][j][c] = sum_{x=j}^{j+k2-1} in[b][h][x][c], j in [0, W-k2]   thread = (b, h, channel quad)
__global__ __launch_bounds__(256) void box_rows_kernel(const float* __restrict__ in, float* __restrict__ rs, int B, int H, int W,
                                                       int C, int k2) {
    const int nq = C / 4, Wo = W - k2 + 1;
    const int64_t total = (int64_t)B * H * nq;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % nq);
    const int64_t bh = i / nq;
    const float* src = in + bh * W * (int64_t)C + 4 * q;
    float* dst = rs + bh * Wo * (int64_t)C + 4 * q;
    float4 acc = f4_zero();
    for (int x = 0; x < k2; ++x) acc = f4_add(acc, ldg4(src + (int64_t)x * C));
    stg4(dst, acc);
    for (int j = 1; j < Wo; ++j) {
        acc = f4_add(acc, ldg4(src + (int64_t)(j + k2 - 1) * C));
        const float4 o = ldg4(src + (int64_t)(j - 1) * C);
        acc = make_float4(acc.x - o.x, acc.y - o.y, acc.z - o.z, acc.w - o.w);
        stg4(dst + (int64_t)j * C, acc);
    }
}
// pass 2: out[b][h][w][c] = (sum_{y=r}^{r+k1-1} rs[b][y][cw][c]) / (k1*k2), r = clamp(h - pt, 0, H-k1), cw = clamp(w - pl, 0, W-k2)
// thread = (b, output column w, channel quad): walks h with a running column sum over the distinct r values
__global__ __launch_bounds__(256) void box_cols_kernel(const float* __restrict__ rs, float* __restrict__ out, int B, int H, int W,
                                                       int C, int k1, int k2) {
    const int nq = C / 4, Wo = W - k2 + 1, Ho = H - k1 + 1;
    const int pl = (W - Wo) / 2, pt = (H - Ho) / 2;
    const int64_t total = (int64_t)B * W * nq;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % nq);
    const int64_t bw = i / nq;
    const int w = (int)(bw % W);
    const int64_t b = bw / W;
    int cw = w - pl;
    cw = cw < 0 ? 0 : (cw > Wo - 1 ? Wo - 1 : cw);
    const float* src = rs + (b * H * (int64_t)Wo + cw) * C + 4 * q;   // row y at src + y*Wo*C
    const int64_t rstride = (int64_t)Wo * C;
    const float inv = 1.0f / (float)(k1 * k2);
    float4 acc = f4_zero();
    for (int y = 0; y < k1; ++y) acc = f4_add(acc, ldg4(src + y * rstride));
    int r = 0;
    for (int h = 0; h < H; ++h) {
        int want = h - pt;
        want = want < 0 ? 0 : (want > Ho - 1 ? Ho - 1 : want);
        while (r < want) {
            acc = f4_add(acc, ldg4(src + (int64_t)(r + k1) * rstride));
            const float4 o = ldg4(src + (int64_t)r * rstride);
            acc = make_float4(acc.x - o.x, acc.y - o.y, acc.z - o.z, acc.w - o.w);
            ++r;
        }
        stg4(out + ((b * H + h) * (int64_t)W + w) * C + 4 * q, f4_scale(acc, inv));
    }
}

}  // namespace

int launch_box_mean(const float* in, float* rowsum, float* out, int B, int H, int W, int C, int k1, int k2, hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && k1 >= 1 && k2 >= 1 && k1 <= H && k2 <= W, "box_mean: bad window %dx%d for %dx%d", k1, k2, H, W);
    box_rows_kernel<<<dim3((unsigned)cdiv64((int64_t)B * H * (C / 4), 256)), dim3(256), 0, s>>>(in, rowsum, B, H, W, C, k2);
    DCPT_CHECK_LAUNCH("box_rows");
    box_cols_kernel<<<dim3((unsigned)cdiv64((int64_t)B * W * (C / 4), 256)), dim3(256), 0, s>>>(rowsum, out, B, H, W, C, k1, k2);
    DCPT_CHECK_LAUNCH("box_cols");
    return DCPT_OK;
}

int launch_sca_fwd(const float* pool_part, int nblk, const float* Wsca, const float* bsca, float* pooled, float* simg,
                   int B, int C, int P, hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && C * 4 <= 65536, "sca_fwd: C=%d unsupported", C);
    sca_fwd_kernel<<<dim3(B, cdiv(C, SCA_OPB)), dim3(256), C * sizeof(float), s>>>(pool_part, nblk, Wsca, bsca, pooled, simg, C,
                                                                                1.0f / (float)P);
    DCPT_CHECK_LAUNCH("sca_fwd");
    return DCPT_OK;
}

int sca_ds_num_blocks(int P) {
    int n = P / 64;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

// slices when the sums come out of the dts GEMM's E_DOTCOL epilogue (one per 128-pixel tile), 0 if that form does not apply
int sca_ds_fused_slices(int P) { return (P % 128 == 0 && P / 128 <= 1024) ? P / 128 : 0; }

int launch_sca_ds_part(const float* dts, const float* t2, float* ds_part, int B, int C, int P, hipStream_t s) {
    DCPT_CHECK_ARG(C % 4 == 0 && B <= 65535, "sca_ds: C=%d", C);
    const int nsl = sca_ds_num_blocks(P);
    const int nq = C / 4;
    int qb = 1;
    while (qb < nq && qb < 256) qb <<= 1;
    sca_ds_part_kernel<<<dim3(cdiv(nq, qb), nsl, B), dim3(256), 0, s>>>(dts, t2, ds_part, C, P, nsl);
    DCPT_CHECK_LAUNCH("sca_ds_part");
    return DCPT_OK;
}

int launch_sca_dpool(const float* ds_part, int nslices, const float* Wsca, float* dpool, int B, int C, int P, hipStream_t s, float* ds_out) {
    DCPT_CHECK_ARG(B <= 65535 && C * 4 <= 65536 && nslices >= 1, "sca_dpool: B=%d C=%d", B, C);
    sca_dpool_kernel<<<dim3(cdiv(C, 32), B), dim3(256), C * sizeof(float), s>>>(ds_part, nslices, Wsca, dpool, C, 1.0f / (float)P, ds_out);
    DCPT_CHECK_LAUNCH("sca_dpool");
    return DCPT_OK;
}

int launch_sca_wgrad(const float* ds_part, int nslices, float* ds, const float* pooled, float* dWsca, float* dbsca, int B, int C,
                     hipStream_t s) {
    sca_ds_final_kernel<<<dim3(cdiv(B * C, 256)), dim3(256), 0, s>>>(ds_part, ds, B * C, C, nslices);
    DCPT_CHECK_LAUNCH("sca_ds_final");
    sca_bwd_w_kernel<<<dim3((unsigned)cdiv64((int64_t)C * C, 256), 2), dim3(256), 0, s>>>(ds, pooled, dWsca, dbsca, B, C);
    DCPT_CHECK_LAUNCH("sca_bwd_w");
    return DCPT_OK;
}

// several row-scaled transposes in one launch: out_j[k][n] = in_j[n][k] * (rs_j ? rs_j[n] : 1)
// (32 x 32 tiles through LDS: reads coalesced along k, writes along n -- the element-per-thread form read with stride K)
__global__ __launch_bounds__(256) void wpack_multi_kernel(const WpackJobs jobs) {
    __shared__ float tile[32][33];
    const int y = blockIdx.y;
    const int N = jobs.N[y], K = jobs.K[y];
    const float* __restrict__ in = jobs.in[y];
    float* __restrict__ out = jobs.out[y];
    const float* __restrict__ rs = jobs.rs[y];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tn = (N + 31) / 32, tk = (K + 31) / 32;
    for (int t = blockIdx.x; t < tn * tk; t += gridDim.x) {
        const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + ty + 8 * r, k = k0 + tx;
            float v = 0.f;
            if (n < N && k < K) {
                v = in[(int64_t)n * K + k];
                if (rs) v = v * rs[n];
            }
            tile[ty + 8 * r][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + ty + 8 * r, n = n0 + tx;
            if (k < K && n < N) out[(int64_t)k * N + n] = tile[tx][ty + 8 * r];
        }
        __syncthreads();
    }
}

// one wave per output j: u[j] = W[j][:] . lnw,  cvec[j] = bz[j] + W[j][:] . lnb
__global__ __launch_bounds__(256) void lnvec_kernel(const LnVecJobs jobs) {
    const int y = blockIdx.y, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= jobs.N2) return;
    const float* __restrict__ w = jobs.W[y] + (int64_t)j * jobs.C;
    float a = 0.f, b = 0.f;
    for (int c = 4 * lane; c < jobs.C; c += 256) {
        float4 wv = ldg4(w + c);
        if (jobs.round_bf16) {   // round-to-nearest-even to bf16, as the operand pack does
            auto r = [](float x) {
                uint32_t u = __builtin_bit_cast(uint32_t, x);
                u += 0x7fffu + ((u >> 16) & 1u);
                return __builtin_bit_cast(float, u & 0xffff0000u);
            };
            wv = make_float4(r(wv.x), r(wv.y), r(wv.z), r(wv.w));
        }
        a += f4_sum(f4_mul(wv, ldg4(jobs.lnw[y] + c)));
        b += f4_sum(f4_mul(wv, ldg4(jobs.lnb[y] + c)));
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        jobs.u[y][j] = a;
        jobs.cvec[y][j] = b + (jobs.bz[y] ? jobs.bz[y][j] : 0.f);
    }
}

int launch_lnvec(const LnVecJobs& jobs, hipStream_t s) {
    DCPT_CHECK_ARG(jobs.n >= 1 && jobs.n <= 2 && jobs.C % 4 == 0, "lnvec: bad job");
    lnvec_kernel<<<dim3(cdiv(jobs.N2, 4), jobs.n), dim3(256), 0, s>>>(jobs);
    DCPT_CHECK_LAUNCH("lnvec");
    return DCPT_OK;
}

int launch_wpack_multi(const WpackJobs& jobs, hipStream_t s) {
    DCPT_CHECK_ARG(jobs.n >= 1 && jobs.n <= WPACK_MAX_JOBS, "wpack_multi: %d jobs", jobs.n);
    int64_t mx = 0;
    for (int j = 0; j < jobs.n; ++j) mx = (int64_t)jobs.N[j] * jobs.K[j] > mx ? (int64_t)jobs.N[j] * jobs.K[j] : mx;
    int64_t g = cdiv64(mx, 1024);   // one 32 x 32 tile per block and iteration
    if (g > 256) g = 256;
    wpack_multi_kernel<<<dim3((unsigned)g, jobs.n), dim3(256), 0, s>>>(jobs);
    DCPT_CHECK_LAUNCH("wpack_multi");
    return DCPT_OK;
}

int launch_wpack(const float* in, float* out, const float* rs, int N, int K, int mode, hipStream_t s) {
    wpack_kernel<<<dim3((unsigned)cdiv64((int64_t)N * K, 256)), dim3(256), 0, s>>>(in, out, rs, N, K, mode);
    DCPT_CHECK_LAUNCH("wpack");
    return DCPT_OK;
}

int launch_wgrad_reduce(const float* slab, const float* colsum, int splits, int cs_rows, int N, int K, const float* rowscale,
                        const float* W, const float* wbias, float* dW, float* dgain, float* dbias, int mode, hipStream_t s) {
    return launch_wgrad_reduce_scaled(slab, colsum, splits, cs_rows, N, K, rowscale, W, wbias, dW, dgain, dbias, mode, nullptr, 1, s);
}

int launch_wgrad_reduce_scaled(const float* slab, const float* colsum, int splits, int cs_rows, int N, int K, const float* rowscale,
                               const float* W, const float* wbias, float* dW, float* dgain, float* dbias, int mode,
                               const float* kscale, int splits_per_image, hipStream_t s) {
    DCPT_CHECK_ARG(K % 4 == 0, "wgrad_reduce: K=%d", K);
    DCPT_CHECK_ARG(!(dgain || dbias) || colsum, "wgrad_reduce: gain/bias gradients need column sums");
    int tpr = 1;
    while (tpr < K / 4 && tpr < 256) tpr <<= 1;
    DCPT_CHECK_ARG(splits_per_image >= 1, "wgrad_reduce: splits_per_image=%d", splits_per_image);
    wgrad_reduce_kernel<<<dim3(N), dim3(256), 0, s>>>(slab, colsum, splits, N, K, rowscale, W, wbias, dW, dgain, dbias, mode,
                                                      tpr, cs_rows, kscale, splits_per_image);
    DCPT_CHECK_LAUNCH("wgrad_reduce");
    return DCPT_OK;
}
