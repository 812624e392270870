// Depthwise 3x3 + bias + SimpleGate forward on an LDS-DMA ROW RING (reference basicsr/archs/nafnet_arch.py:96-104,77-80,171-172).
//
// Why a second implementation next to dwconv.hip: the register version loads, per thread and input row, its pixel and both
// x-neighbours from global memory, needs ~140 VGPRs and keeps ONE row of loads in flight per wave.  It is bound by memory-level
// parallelism -- a bf16 row takes almost as long as an fp32 row (278 vs 402 us at level 0) although it moves half the bytes, and
// holding the next row's loads in registers only lowers the occupancy.  Here the rows travel HBM -> LDS by DMA
// (buffer_load_dwordx4 ... lds: no VGPRs for data in flight) into a ring of R row slots, R - 2 rows ahead of the row being
// consumed, every element exactly once per block (plus one halo pixel on each side); neighbours are LDS reads.
//
// Block = 256 threads, one image-row segment of PB = 2 * PP pixels x one channel chunk.  A thread owns an 8-byte piece (2 fp32 or
// 4 bf16 channels) of BOTH gate halves at TWO adjacent pixels, so the nine taps' weights are loaded once for both pixels and
// fp32 and bf16 share one thread map.  Ring slot = (PB + 2) pixels x 2 halves x LP pieces (<= 12 KiB = twelve 1-KiB wave DMAs,
// three per wave and row, always issued -- out-of-range pixels / rows / channels get a sentinel offset and land as zeros, which is
// also the conv's zero padding).  One raw s_barrier per row: the counted s_waitcnt vmcnt(N) before it retires this wave's DMAs of
// the row about to be read (N = the VMEM operations issued after them: 3 per prefetched row + the 2 stores per iteration), the
// barrier publishes all four waves' pieces and retires the slot read in the previous iteration, which the next DMA then refills.
#include "bf16.h"
#include "prof.h"
#include "bf16_ops.h"
#include "kernels.h"

namespace {

constexpr int RING = 5;            // row slots: the row in use + 3 rows in flight + the one being refilled
constexpr int SLOT_BYTES = 12288;  // 12 wave-DMAs of 1 KiB

struct DwrP {
    const void* in;     // t1 [M][2C]
    const float* w2p;   // [9][2C]
    const float* b2;    // [2C]
    void* out;          // t2 [M][C]
    float* part;        // pool partials [B][gridDim.y][C]
    int B, H, W, C;
    int LP;             // pieces (8 bytes) per pixel-half handled by a block: power of two, 2..64
    int nwc;            // column chunks
};

// Block -> (channel chunk, column chunk + nwc * row part, image), optionally XCD-aware: the hardware deals the blocks of the 3-D grid to the
// XCDs in the linear order x + gridDim.x * (y + gridDim.y * z), i.e. the channel chunks of one image -- neighbouring 128-byte runs of the
// same pixels -- to different XCDs.  With the chunks of an image as neighbours on ONE XCD the backward kernel of the narrow images (one tile
// per image row, levels 2 and 3) reads half as much over the fabric in bf16 (201 -> 101 MB at level 3) and runs 9-24 % faster in a same-box
// A/B (profiles/r5/dwring_xcd/); the forward kernel and the column-tiled backward (levels 0 and 1) read less too but are no faster (+-5 %),
// they keep the plain grid order.
struct DwrBlk {
    int qc, by, b;
};
__device__ __forceinline__ DwrBlk dwr_block(int nwc, int xcd_order) {
    if (!xcd_order) return DwrBlk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    const int L = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
    int t = xcd_remap(L, gx * gy * (int)gridDim.z);
    DwrBlk k;
    const int wc = t % nwc;   // column neighbours first, then the channel chunks, then the row parts, then the images
    t /= nwc;
    k.qc = t % gx;
    t /= gx;
    const int nrp = gy / nwc;
    const int rp = t % nrp;
    k.b = t / nrp;
    k.by = wc + nwc * rp;
    return k;
}

template <int VW>
struct pv {
    float v[VW];
};

template <typename ST>
__device__ __forceinline__ pv<8 / sizeof(ST)> lds_piece(const unsigned char* p) {
    pv<8 / sizeof(ST)> r;
    const u32x2 w = *reinterpret_cast<const u32x2*>(p);
    if constexpr (sizeof(ST) == 4) {
        const floatx2 f = __builtin_bit_cast(floatx2, w);   // (whole vector: a bit_cast of ONE ext-vector element reads element 0 on this hipcc)
        r.v[0] = f.x;
        r.v[1] = f.y;
    } else {
        r.v[0] = bf_lo(w.x); r.v[1] = bf_hi(w.x); r.v[2] = bf_lo(w.y); r.v[3] = bf_hi(w.y);
    }
    return r;
}

// gelu(a) = a Phi(a) and its derivative Phi(a) + a phi(a) from ONE exponential, branch-free: erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, i.e. below fp32 resolution of Phi; ocml's erff is ~50 instructions with divergent branches and made the
// GELU-gate loop 2.3x the SimpleGate one), e = exp(-a^2 / 2) is shared by the erf tail and the density term.
__device__ __forceinline__ void dwr_gelu_both(float a, float& g, float& gd) {
    const float ax = fabsf(a) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float e = __expf(-ax * ax);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.0f);
    const float cdf = 0.5f * (1.0f + copysignf(erf_abs, a));
    g = a * cdf;
    gd = fmaf(a * 0.39894228040143268f, e, cdf);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// GATE: 0 SimpleGate (a_1 + b_1)(a_2 + b_2) + pooling partials (NAFNet), 1 gelu(a_1) a_2 (Restormer GDFN, restormer_arch.py:75-92).
// (A gate-less mode for Restormer's qkv depthwise conv was measured slower than the register kernel, 221 vs 191 us per launch, and
// is not kept.)
template <typename ST, int GATE = 0>
__global__ __launch_bounds__(256) void dwr_gate_fwd_kernel(const DwrP p) {
    constexpr int ES = sizeof(ST), VW = 8 / ES, R = RING;
    constexpr int NST = 2;   // output stores per thread and row
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * SLOT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, LP = p.LP, U = LP / 2;               // U: 16-byte units per pixel-half in this chunk
    const int PH = C * ES / 8;                              // pieces per half over all channels
    const int PP = 256 / LP, PB = 2 * PP;
    // block -> (channel chunk, column chunk, row part, image)
    const int nrp = gridDim.y / p.nwc, rpp = (p.H + nrp - 1) / nrp;
    const DwrBlk blk = dwr_block(p.nwc, 0);   // (the XCD-aware order gains nothing here: same-box A/B in profiles/r5/dwring_xcd/)
    const int wc = blk.by % p.nwc, rp = blk.by / p.nwc, b = blk.b;
    const int x0 = wc * PB;
    const int h0 = rp * rpp, h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
    const int piece0 = blk.qc * LP;
    const int g = tid % LP, pp = tid / LP;
    const bool gok = piece0 + g < PH;
    const int c1 = (piece0 + g) * VW, c2 = C + c1;

    // ---- per-thread constants (loaded and consumed before any DMA is issued: the compiler's own vmcnt bookkeeping does not see
    //      the DMAs, a wait it places later would drain the ring)
    pv<VW> w1[9], w2[9], bias1, bias2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            w1[t].v[i] = gok ? p.w2p[t * 2 * C + c1 + i] : 0.f;
            w2[t].v[i] = gok ? p.w2p[t * 2 * C + c2 + i] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        bias1.v[i] = (gok && p.b2) ? p.b2[c1 + i] : 0.f;
        bias2.v[i] = (gok && p.b2) ? p.b2[c2 + i] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < VW; ++i) asm volatile("" ::"v"(w1[t].v[i]), "v"(w2[t].v[i]));
#pragma unroll
    for (int i = 0; i < VW; ++i) asm volatile("" ::"v"(bias1.v[i]), "v"(bias2.v[i]));
    wait_vm<0>();

    // ---- DMA map of this lane: unit e of a row slot, three per wave (j = 0..2)
    const int rb = h0 - 1 > 0 ? h0 - 1 : 0;   // first image row the window reaches
    const ST* img_in = (const ST*)p.in + ((int64_t)b * p.H + rb) * p.W * 2 * C;
    const i32x4 rs_in = make_rsrc_dma(img_in);
    uint32_t colo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int e = (j * 4 + wave) * 64 + lane;
        const int pixel = e / (2 * U), within = e % (2 * U);
        const int half = within / U, unit = within % U;
        const int x = x0 - 1 + pixel;
        const bool ok = pixel < PB + 2 && x >= 0 && x < p.W && (piece0 / 2 + unit) * 2 < PH;
        colo[j] = ok ? (uint32_t)((x * 2 * C + half * C) * ES + (piece0 / 2 + unit) * 16) : COL_SENT;
    }
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(ring));
    const uint32_t rowbytes = (uint32_t)p.W * 2u * (uint32_t)C * ES;
    auto issue = [&](int r, int slot) {   // r: image row (may lie outside [0, H): zeros)
        const bool rok = r >= 0 && r < p.H;
        const uint32_t ro = rok ? (uint32_t)(r - rb) * rowbytes : 0u;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            dma16(rs_in, lds0 + slot * SLOT_BYTES + ((j * 4 + wave) * 64) * 16, (rok && colo[j] != COL_SENT) ? ro + colo[j] : ROW_SENT, 0);
    };

    ST* img_out = (ST*)p.out + ((int64_t)b * p.H + rb) * p.W * C;
    const rsrc_t rs_out = make_rsrc(img_out);
    const int xa = x0 + 2 * pp;                                   // this thread's first output pixel
    const bool ok0 = gok && xa < p.W, ok1 = gok && xa + 1 < p.W;
    pv<VW> a0[2][2], a1[2][2], pool;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < VW; ++i) a0[o][h].v[i] = a1[o][h].v[i] = 0.f;
#pragma unroll
    for (int i = 0; i < VW; ++i) pool.v[i] = 0.f;

    // prologue: rows h0-1 .. h0-1+R-2 in flight
#pragma unroll
    for (int k = 0; k < R - 1; ++k) issue(h0 - 1 + k, k);
    const int niter = h1 > h0 ? h1 - h0 + 2 : 0;   // input rows h0-1 .. h1 (a row part past the image still writes its zero partials)
    for (int it = 0; it < niter; ++it) {
        const int r = h0 - 1 + it;
#ifdef DWR_SAFE
        wait_vm<0>();
#else
        if (it < R - 1) wait_vm<3 * (R - 2)>();                 // no / fewer stores issued yet: conservative count
        else wait_vm<3 * (R - 2) + NST * (R - 1)>();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(r + R - 1, (it + R - 1) % R);                      // refills the slot read in the previous iteration
        const unsigned char* sl = ring + (it % R) * SLOT_BYTES + g * 8;
        pv<VW> v[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[h][j] = lds_piece<ST>(sl + (((2 * pp + j) * 2 + h) * LP) * 8);
        const int y = r - 1;
        const bool yok = y >= h0;   // (y < h1 always: r <= h1)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            pv<VW> t;
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                // kernel row 2 completes output row r-1, row 1 feeds output row r, row 0 starts output row r+1
                const float l1 = v[0][o].v[i], m1 = v[0][o + 1].v[i], r1 = v[0][o + 2].v[i];
                const float l2 = v[1][o].v[i], m2 = v[1][o + 1].v[i], r2 = v[1][o + 2].v[i];
                const float f1 = fmaf(w1[6].v[i], l1, fmaf(w1[7].v[i], m1, fmaf(w1[8].v[i], r1, a0[o][0].v[i])));
                const float f2 = fmaf(w2[6].v[i], l2, fmaf(w2[7].v[i], m2, fmaf(w2[8].v[i], r2, a0[o][1].v[i])));
                a0[o][0].v[i] = fmaf(w1[3].v[i], l1, fmaf(w1[4].v[i], m1, fmaf(w1[5].v[i], r1, a1[o][0].v[i])));
                a0[o][1].v[i] = fmaf(w2[3].v[i], l2, fmaf(w2[4].v[i], m2, fmaf(w2[5].v[i], r2, a1[o][1].v[i])));
                a1[o][0].v[i] = fmaf(w1[0].v[i], l1, fmaf(w1[1].v[i], m1, w1[2].v[i] * r1));
                a1[o][1].v[i] = fmaf(w2[0].v[i], l2, fmaf(w2[1].v[i], m2, w2[2].v[i] * r2));
                if constexpr (GATE == 0) {
                    t.v[i] = (f1 + bias1.v[i]) * (f2 + bias2.v[i]);
                } else {
                    float gv, gd;
                    dwr_gelu_both(f1 + bias1.v[i], gv, gd);
                    t.v[i] = gv * (f2 + bias2.v[i]);
                }
            }
            const bool ok = yok && (o == 0 ? ok0 : ok1);
            const uint32_t off = ok ? (uint32_t)(((y - rb) * p.W + xa + o) * C + c1) * ES : ROW_SENT;
            u32x2 wv;
            if constexpr (ES == 4) {
                wv.x = __builtin_bit_cast(uint32_t, t.v[0]);
                wv.y = __builtin_bit_cast(uint32_t, t.v[1]);
            } else {
                wv.x = bf_pack(t.v[0], t.v[1]);
                wv.y = bf_pack(t.v[2], t.v[3]);
            }
            __builtin_amdgcn_raw_buffer_store_b64(wv, rs_out, off, 0, 0);
            if (ok) {
#pragma unroll
                for (int i = 0; i < VW; ++i) pool.v[i] += t.v[i];
            }
        }
    }
    wait_vm<0>();   // the ring's tail DMAs (rows past h1: zeros) before the LDS is reused
    __syncthreads();
    // pooling partials of this block: the PP pixel-pair threads of a piece through LDS, fixed order
    if (p.part != nullptr) {
        float* red = reinterpret_cast<float*>(ring);
#pragma unroll
        for (int i = 0; i < VW; ++i) red[tid * VW + i] = pool.v[i];
        __syncthreads();
        if (pp == 0 && gok) {
            float sres[VW];
#pragma unroll
            for (int i = 0; i < VW; ++i) sres[i] = red[g * VW + i];
            for (int k = 1; k < PP; ++k)
#pragma unroll
                for (int i = 0; i < VW; ++i) sres[i] += red[(k * LP + g) * VW + i];
            float* dst = p.part + ((int64_t)b * gridDim.y + blk.by) * C + c1;
#pragma unroll
            for (int i = 0; i < VW; ++i) dst[i] = sres[i];
        }
    }
}

struct DwrGeom {
    int LP, nqc, nwc, nrp;
};

DwrGeom dwr_geom(const DwGeom& g, int es) {
    DwrGeom d;
    const int PH = g.C * es / 8;
    int lp = 2;
    while (lp < PH && lp < 64) lp <<= 1;
    static const int force_lp = dcpt_tuning("DCPT_DWR_FWD_LP", 0);   // experiments
    static const int force_nrp = dcpt_tuning("DCPT_DWR_FWD_NRP", 0);
    if (force_lp >= 2 && force_lp <= lp) lp = force_lp;
    d.LP = lp;
    d.nqc = cdiv(PH, lp);
    const int PB = 2 * (256 / lp);
    d.nwc = cdiv(g.W, PB);
    // row parts: enough blocks to fill the chip (>= 1024), at least 16 rows each so that the two halo rows stay a small fraction (8-row parts
    // at level 3 -- 32 x 32 images -- measured 41.8 us against 30.6 us with 16-row parts: profiles/r4/dwring_fwd_geometry.txt)
    int64_t nrp = cdiv64(cdiv64(1024, (int64_t)g.B * d.nqc), d.nwc);
    const int64_t maxp = g.H / 16 > 0 ? g.H / 16 : 1;
    if (nrp > maxp) nrp = maxp;
    if (nrp < 1) nrp = 1;
    if (force_nrp > 0) nrp = force_nrp < g.H ? force_nrp : g.H;
    d.nrp = (int)nrp;
    return d;
}

bool dwr_enabled() {
    static const int on = dcpt_tuning("DCPT_DW_RING", 1);
    return on != 0;
}

template <typename ST, int GATE = 0>
int launch_fwd(const void* t1, const float* w2p, const float* b2, void* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    const DwrGeom d = dwr_geom(g, (int)sizeof(ST));
    DwrP p{};
    p.in = t1; p.w2p = w2p; p.b2 = b2; p.out = t2; p.part = pool_part;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C; p.LP = d.LP; p.nwc = d.nwc;
    DCPT_CHECK_ARG(((double)cdiv(g.H, d.nrp) + 4.0) * g.W * 2.0 * g.C * sizeof(ST) < 1.0e9, "depthwise ring: a row range exceeds the 32-bit window");
    dwr_gate_fwd_kernel<ST, GATE><<<dim3(d.nqc, d.nwc * d.nrp, g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dwr_gate_fwd");
    return DCPT_OK;
}


// ---- fused SimpleGate + depthwise BACKWARD on the row ring ------------------------------------------------------------------
// Same math as dw_bwd_fused_kernel (dwconv.hip):  a = dw3x3(t1) + b2 (recomputed),  dt2 = dts * s + dpool,  da_1 = dt2 * a_2,
// da_2 = dt2 * a_1,  dt1 = dw3x3^T(da),  per-block partial sums of the tap / bias gradients.  The register version recomputes a
// at three columns per thread (5 overlapping global loads of t1 and 3 of dts per pixel: it takes 826 / 790 us at level 0 in
// fp32 / bf16 -- the same time for half the bytes, i.e. it is instruction-bound).  Here every quantity is computed ONCE:
//   * t1 row r and dts row r-1 arrive by LDS-DMA in ring slot (it % R), R - 2 rows ahead;
//   * a thread owns TWO channels of BOTH gate halves at TWO adjacent pixels: it completes a[r-1] from the running forward-conv
//     accumulators, forms da[r-1] at its two pixels (both halves are in the thread, no lane exchange), accumulates the 2 x 9 tap
//     gradients against the three t1 rows it holds (two in registers, one just read), and parks da in an LDS exchange row;
//   * one iteration later (after the iteration's single barrier) it reads its left / right neighbours' da from the exchange row
//     and advances the transposed-conv accumulators; dt1 row r-3 is complete and stored.
// Column tiles: with one tile per image row (MULTI = 0) the neighbours outside the image are zeros; with several (MULTI = 1) a
// tile computes da at PB columns and owns the PB - 2 interior ones (the edge columns are recomputed by the neighbouring tile), so
// nothing crosses blocks.  LDS planes are split by column parity so that the PP pixel-pair groups of a wave read consecutive
// 8-byte (fp32) / 4-byte (bf16) runs: conflict-free for every LP.
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 v2z() {
    v2 z;
    z.x = 0.f;
    z.y = 0.f;
    return z;
}

template <typename ST>
__device__ __forceinline__ v2 lds_pair(const unsigned char* p) {   // two consecutive channels
    if constexpr (sizeof(ST) == 4) {
        return *reinterpret_cast<const v2*>(p);
    } else {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
        v2 r;
        r.x = bf_lo(w);
        r.y = bf_hi(w);
        return r;
    }
}

#ifndef DWRB_R32
#define DWRB_R32 5
#endif
#ifndef DWRB_R16
#define DWRB_R16 7
#endif
// ring slots of the backward kernel (fp32 / bf16): the row in use + R - 2 in flight + the one being refilled.  vmcnt also counts the
// dt1 stores, so the ring depth has to cover the store acknowledgement latency as well; two blocks per CU must fit the 160 KiB.

struct DwrBP {
    const void* t1;     // [M][2C]
    const void* dts;    // [M][C]
    const float* w2p;   // [9][2C]
    const float* b2;    // [2C] or null
    const float* simg;  // [B][C] or null (= 1)
    const float* dpool; // [B][C] (with simg)
    void* dt1;          // [M][2C]
    float* part;        // [B][gridDim.y][10][2C]
    int B, H, W, C;
    int nwc;            // column tiles
    float* tout;        // GATE 1 only, may be null: [M][C] the gate product gelu(a_1) a_2 itself, recomputed on the way (bit-identical
                        // to the forward kernel's: same conv recurrences, same GELU) for a caller that did not keep it
};

// GATE: 0 SimpleGate a_1 a_2 (+ bias, SCA scale / pooled gradient; NAFNet), 1 gelu(a_1) a_2 with the exact erf GELU (Restormer GDFN,
// reference restormer_arch.py:75-92), 2 no gate at all: `dts` is the gradient of the conv output itself over all 2C channels and
// nothing is recomputed (the qkv depthwise conv of Restormer's MDTA, :108-145) -- the kernel is then the transposed conv + the tap
// gradients on the ring.
template <typename ST, int LP, int MULTI, int GATE = 0>
__global__ __launch_bounds__(256) void dwr_bwd_fused_kernel(const DwrBP p) {
    constexpr int ES = sizeof(ST), PBY = 2 * ES;
    constexpr int R = GATE == 2 ? 3 : (ES == 4 ? DWRB_R32 : DWRB_R16);   // (GATE 2: 16-KB slots, three keep two blocks per CU)
    constexpr int DD = GATE == 2 ? 2 : 1;          // gate halves carried by the dts rows
    constexpr int NSTO = GATE == 1 ? 6 : 4;        // stores per thread and row (vmcnt counts them): dt1 x 4 (+ the gate product x 2)
    constexpr int NDD = (DD * ES + 3) / 4;         // dts DMAs per wave and row (DD x 2 planes x 256 runs x PBY bytes)
    constexpr int NDT = ES / 2;                    // t1 DMAs per wave and row (4 planes x 256 runs x PBY bytes = NDT x 4 KiB)
    constexpr int NCH = 256 / LP;                  // pixel pairs per plane
    constexpr int PP = NCH - MULTI, PB = 2 * PP;   // pixel-pair groups that work / da columns of the tile
    constexpr int NC = PB + 2 * MULTI;             // t1 columns in a slot
    constexpr int UPH = LP * PBY / 16;             // 16-byte DMA units per (pixel, half) run
    constexpr int BLK = LP * PBY, PLANE = NCH * BLK;
    constexpr int TBYTES = 4 * PLANE, SLOT = TBYTES + NDD * 4096;
    constexpr int XBLK = LP * 8, XPLANE = NCH * XBLK, XSLOT = 4 * XPLANE;
    constexpr int XOFF = R * SLOT, ZOFF = XOFF + 2 * XSLOT;
    static_assert(TBYTES == NDT * 4096 && UPH >= 1 && DD * 2 * NCH * BLK <= NDD * 4096, "slot geometry");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ZOFF + 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, PH = C / 2;
    const int nrp = gridDim.y / p.nwc, rpp = (p.H + nrp - 1) / nrp;
    const DwrBlk blk = dwr_block(p.nwc, !MULTI);
    const int wc = blk.by % p.nwc, rp = blk.by / p.nwc, b = blk.b;
    const int h0 = rp * rpp, h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
    const int x0 = MULTI ? wc * (PB - 2) - 1 : 0;   // image column of da column 0
    const int xs = x0 - MULTI;                      // image column of t1 slot column 0
    const int piece0 = blk.qc * LP;
    const int g = tid % LP, pp = tid / LP;
    const bool act = pp < PP;
    const bool gok = act && piece0 + g < PH;
    const int c1 = (piece0 + g) * 2, c2 = C + c1;

    // ---- per-thread constants (consumed before the first DMA: the compiler's vmcnt bookkeeping does not see the DMAs)
    v2 w[2][9], bias[2], sv, dpv;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w[0][t] = gok ? *reinterpret_cast<const v2*>(p.w2p + t * 2 * C + c1) : v2z();
        w[1][t] = gok ? *reinterpret_cast<const v2*>(p.w2p + t * 2 * C + c2) : v2z();
    }
    bias[0] = (gok && p.b2) ? *reinterpret_cast<const v2*>(p.b2 + c1) : v2z();
    bias[1] = (gok && p.b2) ? *reinterpret_cast<const v2*>(p.b2 + c2) : v2z();
    sv.x = sv.y = 1.f;
    dpv = v2z();
    if (gok && p.simg) {
        sv = *reinterpret_cast<const v2*>(p.simg + (int64_t)b * C + c1);
        dpv = *reinterpret_cast<const v2*>(p.dpool + (int64_t)b * C + c1);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(w[0][t]), "v"(w[1][t]));
    asm volatile("" ::"v"(bias[0]), "v"(bias[1]), "v"(sv), "v"(dpv));
    wait_vm<0>();

    // ---- DMA map of this lane
    const int rb = h0 - 2 > 0 ? h0 - 2 : 0;   // first image row the windows reach
    const i32x4 rs_t = make_rsrc_dma((const ST*)p.t1 + ((int64_t)b * p.H + rb) * p.W * 2 * C);
    const i32x4 rs_d = make_rsrc_dma((const ST*)p.dts + ((int64_t)b * p.H + rb) * p.W * (DD * C));
    uint32_t colo[NDT], cold[NDD];
#pragma unroll
    for (int j = 0; j < NDT; ++j) {
        const int e = (j * 4 + wave) * 64 + lane;
        const int blk = e / UPH, unit = e % UPH;
        const int plane = blk / NCH, idx = blk % NCH;
        const int lc = idx * 2 + (plane >> 1), h = plane & 1;
        const int x = xs + lc, cb = piece0 * PBY + unit * 16;
        const bool ok = lc < NC && x >= 0 && x < p.W && cb < C * ES;
        colo[j] = ok ? (uint32_t)((x * 2 * C + h * C) * ES + cb) : COL_SENT;
    }
#pragma unroll
    for (int j = 0; j < NDD; ++j) {   // dts planes: (half, pixel parity) -> (hd * 2 + o) * NCH + pixel pair
        const int e = (j * 4 + wave) * 64 + lane;
        const int blk = e / UPH, unit = e % UPH;
        const int plane = blk / NCH, ppx = blk % NCH;
        const int hd = plane >> 1, o = plane & 1;
        const int px = 2 * ppx + o, x = x0 + px, cb = piece0 * PBY + unit * 16;
        const bool ok = plane < 2 * DD && px < PB && x >= 0 && x < p.W && cb < C * ES;
        cold[j] = ok ? (uint32_t)((x * DD * C + hd * C) * ES + cb) : COL_SENT;
    }
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(smem));
    const uint32_t rowT = (uint32_t)p.W * 2u * (uint32_t)C * ES, rowD = (uint32_t)p.W * (uint32_t)(DD * C) * ES;
    auto issue = [&](int r, int slot) {   // t1 row r and dts row r - 1 (rows that are not needed or do not exist: zeros)
        const bool tok = r >= 0 && r < p.H && r <= h1 + 1;
        const int rd = r - 1;
        const bool dok = rd >= 0 && rd >= h0 - 1 && rd < p.H && rd <= h1;
        const uint32_t ro = tok ? (uint32_t)(r - rb) * rowT : 0u, rod = dok ? (uint32_t)(rd - rb) * rowD : 0u;
#pragma unroll
        for (int j = 0; j < NDT; ++j)
            dma16(rs_t, lds0 + slot * SLOT + ((j * 4 + wave) * 64) * 16, (tok && colo[j] != COL_SENT) ? ro + colo[j] : ROW_SENT, 0);
#pragma unroll
        for (int j = 0; j < NDD; ++j)
            dma16(rs_d, lds0 + slot * SLOT + TBYTES + ((j * 4 + wave) * 64) * 16, (dok && cold[j] != COL_SENT) ? rod + cold[j] : ROW_SENT, 0);
    };

    // ---- this thread's LDS addresses
    //  t1 column j (image column xa - 1 + j, xa = x0 + 2 pp): slot column lc = MULTI + 2 pp - 1 + j -> plane (lc & 1) * 2 + h, index lc >> 1
    const int tb = pp * BLK + g * PBY;
    const bool z0 = !MULTI && pp == 0, z3 = !MULTI && pp == PP - 1;
    constexpr int TJ0 = MULTI ? 0 : 2 * PLANE - BLK, TJ1 = MULTI ? 2 * PLANE : 0, TJ2 = MULTI ? BLK : 2 * PLANE,
                  TJ3 = MULTI ? 2 * PLANE + BLK : BLK;
    const int db = TBYTES + pp * BLK + g * PBY;                                  // dts (half hd, pixel o): + (hd * 2 + o) * PLANE
    const int xb = pp * XBLK + g * 8;                                            // own da (o, h): + (o * 2 + h) * XPLANE
    const bool zl = pp == 0, zr = pp + 1 >= NCH;
    const unsigned char* zero = smem + ZOFF;
    // zero the exchange rows (every column has an owner among the 256 threads) and the zero cell
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v2*>(smem + XOFF + k * XSLOT + q * XPLANE + xb) = v2z();
    if (tid < 4) reinterpret_cast<float*>(smem + ZOFF)[tid] = 0.f;

    // ---- output columns
    const int xa = x0 + 2 * pp;
    bool oval[2];
    uint32_t ocol[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int x = xa + o, lcol = 2 * pp + o;
        oval[o] = gok && x >= 0 && x < p.W && (!MULTI || (lcol >= 1 && lcol <= PB - 2));
        ocol[o] = oval[o] ? (uint32_t)((x * 2 * C + c1) * ES) : COL_SENT;
    }
    bool pval[2];   // the pixel exists (da is defined)
    v2 svo[2], dpo[2];   // SCA scale / pooled gradient of this thread's two pixels, zero where the pixel does not exist
    v2 gmask[2];         // 1 where the pixel's tap gradients are this tile's to count (MULTI: not the recomputed edge columns)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        pval[o] = gok && xa + o >= 0 && xa + o < p.W;
        svo[o] = pval[o] ? sv : v2z();
        dpo[o] = pval[o] ? dpv : v2z();
        gmask[o].x = gmask[o].y = oval[o] ? 1.f : 0.f;
    }
    const rsrc_t rs_o = make_rsrc((ST*)p.dt1 + ((int64_t)b * p.H + rb) * p.W * 2 * C);
    rsrc_t rs_to = rs_o;
    if constexpr (GATE == 1) rs_to = make_rsrc((p.tout ? p.tout : (float*)p.dt1) + ((int64_t)b * p.H + rb) * p.W * C);

    v2 a0[2][2], a1[2][2], B0[2][2], B1[2][2], daP[2][2], T1[2][4], T2[2][4], gr[2][10];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int h = 0; h < 2; ++h) a0[o][h] = a1[o][h] = B0[o][h] = B1[o][h] = daP[o][h] = v2z();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) T1[h][j] = T2[h][j] = v2z();
#pragma unroll
        for (int t = 0; t < 10; ++t) gr[h][t] = v2z();
    }

#pragma unroll
    for (int k = 0; k < R - 1; ++k) issue(h0 - 2 + k, k);
    const int niter = h1 > h0 ? h1 - h0 + 5 : 0;
    for (int it = 0; it < niter; ++it) {
        const int r = h0 - 2 + it;
#ifdef DWR_SAFE
        wait_vm<0>();
#else
        if (it < R - 1) wait_vm<(R - 2) * (NDT + NDD)>();
        else wait_vm<NSTO + (R - 2) * (NDT + NDD + NSTO)>();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(r + R - 1, (it + R - 1) % R);
        const unsigned char* sl = smem + (it % R) * SLOT;
        // ---- t1 row r at columns xa-1 .. xa+2, dts row r-1 at xa, xa+1
        v2 T[2][4], D[2][DD];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            T[h][0] = lds_pair<ST>(z0 ? zero : sl + tb + TJ0 + h * PLANE);
            T[h][1] = lds_pair<ST>(sl + tb + TJ1 + h * PLANE);
            T[h][2] = lds_pair<ST>(sl + tb + TJ2 + h * PLANE);
            T[h][3] = lds_pair<ST>(z3 ? zero : sl + tb + TJ3 + h * PLANE);
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int hd = 0; hd < DD; ++hd) D[o][hd] = lds_pair<ST>(sl + db + (hd * 2 + o) * PLANE);
        // ---- neighbours' da of row rho' = r - 2 (written in the previous iteration)
        const unsigned char* xr = smem + XOFF + ((it + 1) & 1) * XSLOT + xb;
        v2 X[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            X[h][0] = *reinterpret_cast<const v2*>(zl ? zero : xr + (2 + h) * XPLANE - XBLK);
            X[h][1] = daP[0][h];
            X[h][2] = daP[1][h];
            X[h][3] = *reinterpret_cast<const v2*>(zr ? zero : xr + h * XPLANE + XBLK);
        }
        // ---- forward conv: row r completes a[r-1] (kernel row 2), feeds a[r] (row 1), starts a[r+1] (row 0)
        const int rho = r - 1;
        const bool rho_ok = rho >= 0 && rho < p.H && rho >= h0 - 1 && rho <= h1;
        v2 rmask;   // (uniform) 1 on the rows whose da exists
        rmask.x = rmask.y = rho_ok ? 1.f : 0.f;
        v2 da[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const bool pin = rho_ok && pval[o];
            if constexpr (GATE == 2) {
                da[o][0] = pin ? D[o][0] : v2z();
                da[o][1] = pin ? D[o][1] : v2z();
            } else {
                v2 f[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f[h] = fma2(w[h][6], T[h][o], fma2(w[h][7], T[h][o + 1], fma2(w[h][8], T[h][o + 2], a0[o][h]))) + bias[h];
                    a0[o][h] = fma2(w[h][3], T[h][o], fma2(w[h][4], T[h][o + 1], fma2(w[h][5], T[h][o + 2], a1[o][h])));
                    a1[o][h] = fma2(w[h][0], T[h][o], fma2(w[h][1], T[h][o + 1], w[h][2] * T[h][o + 2]));
                }
                if constexpr (GATE == 0) {
                    // (pixel / row validity is folded into svo / dpo / rmask: dt2 is exactly 0 where da must be, and f is finite)
                    const v2 dt2 = fma2(D[o][0], svo[o], dpo[o] * rmask);
                    da[o][0] = dt2 * f[1];
                    da[o][1] = dt2 * f[0];
                } else {   // t = gelu(a_1) a_2:  da_1 = dt a_2 gelu'(a_1),  da_2 = dt gelu(a_1)
                    float g0, d0, g1, d1;
                    dwr_gelu_both(f[0].x, g0, d0);
                    dwr_gelu_both(f[0].y, g1, d1);
                    v2 gd, gv;
                    gv.x = g0; gv.y = g1;
                    gd.x = d0; gd.y = d1;
                    da[o][0] = pin ? D[o][0] * f[1] * gd : v2z();
                    da[o][1] = pin ? D[o][0] * gv : v2z();
                    {   // the gate product of this pixel for a caller that recomputes it here (rows of this block, owned columns)
                        const bool tok = p.tout != nullptr && rho >= h0 && rho < h1 && oval[o];
                        const uint32_t toff = tok ? (uint32_t)((((rho - rb) * p.W + xa + o) * C + c1) * 4) : ROW_SENT;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, gv * f[1]), rs_to, toff, 0, 0);
                    }
                }
            }
        }
        unsigned char* xw = smem + XOFF + (it & 1) * XSLOT + xb;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<v2*>(xw + (o * 2 + h) * XPLANE) = da[o][h];
        // ---- tap gradients of row rho: da[rho][x] with t1 rows rho-1 (T2), rho (T1), rho+1 (T) at columns x-1..x+1
        if (rho >= h0 && rho < h1) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2 d = MULTI ? da[o][h] * gmask[o] : da[o][h];   // (one tile per row: da is already 0 wherever it must not count)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        gr[h][0 * 3 + kx] = fma2(d, T2[h][o + kx], gr[h][0 * 3 + kx]);
                        gr[h][1 * 3 + kx] = fma2(d, T1[h][o + kx], gr[h][1 * 3 + kx]);
                        gr[h][2 * 3 + kx] = fma2(d, T[h][o + kx], gr[h][2 * 3 + kx]);
                    }
                    gr[h][9] += d;
                }
        }
        // ---- transposed conv with da row rho' = r - 2: completes dt1 row r - 3;  da[.][x - 1 + j] <-> kx = 2 - j
        const int y = r - 3;
        const bool yok = y >= h0 && y < h1;
        const uint32_t yo = yok ? (uint32_t)(y - rb) * rowT : ROW_SENT;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2 out = fma2(w[h][2], X[h][o], fma2(w[h][1], X[h][o + 1], fma2(w[h][0], X[h][o + 2], B0[o][h])));
                B0[o][h] = fma2(w[h][5], X[h][o], fma2(w[h][4], X[h][o + 1], fma2(w[h][3], X[h][o + 2], B1[o][h])));
                B1[o][h] = fma2(w[h][8], X[h][o], fma2(w[h][7], X[h][o + 1], w[h][6] * X[h][o + 2]));
                const uint32_t off = (yok && ocol[o] != COL_SENT) ? yo + ocol[o] + (uint32_t)(h * C * ES) : ROW_SENT;
                if constexpr (ES == 4) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, out), rs_o, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(bf_pack(out.x, out.y), rs_o, off, 0, 0);
            }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int h = 0; h < 2; ++h) daP[o][h] = da[o][h];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T2[h][j] = T1[h][j];
                T1[h][j] = T[h][j];
            }
    }
    wait_vm<0>();
    __syncthreads();
    // ---- per-block partial sums of the tap / bias gradients: the PP pixel-pair groups of a channel pair through LDS, fixed order
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 10; ++t) *reinterpret_cast<v2*>(red + tid * 40 + (h * 10 + t) * 2) = gr[h][t];
    __syncthreads();
    float* dst = p.part + ((int64_t)b * gridDim.y + blk.by) * 10 * 2 * C;
    for (int idx = tid; idx < LP * 40; idx += 256) {
        const int gg = idx / 40, k = idx % 40;
        if (piece0 + gg >= PH) continue;
        float s = 0.f;
        for (int q = 0; q < PP; ++q) s += red[(q * LP + gg) * 40 + k];
        const int h = k / 20, t = (k % 20) / 2, i = k & 1;
        dst[t * 2 * C + h * C + (piece0 + gg) * 2 + i] = s;
    }
}


struct DwrBGeom {
    int LP, multi, nqc, nwc, nrp;
};

DwrBGeom dwr_bwd_geom(const DwGeom& g) {
    DwrBGeom d;
    const int PH = g.C / 2;
    int cap = 8;
    while (cap < PH && cap < 32) cap <<= 1;
    static const int force_lp = dcpt_tuning("DCPT_DWR_BWD_LP", 0);   // experiments
    static const int force_multi = dcpt_tuning("DCPT_DWR_BWD_MULTI", -1);
    int lp = g.W <= 16 ? 32 : g.W <= 32 ? 16 : g.W <= 64 ? 8 : 0;
    d.multi = lp == 0;
    if (lp == 0) lp = 16;
    if (lp > cap) lp = cap;
    if (force_lp == 8 || force_lp == 16 || force_lp == 32) lp = force_lp;
    if (force_multi == 1) d.multi = 1;
    if (!d.multi && 2 * (256 / lp) < g.W) d.multi = 1;
    d.LP = lp;
    d.nqc = cdiv(PH, lp);
    const int PB = 2 * (256 / lp - d.multi);
    d.nwc = d.multi ? cdiv(g.W, PB - 2) : 1;
    // row parts: two blocks are resident per CU (512 slots); a grid of 1.25 rounds costs two, so aim at >= 1024 blocks while a part
    // keeps >= 32 rows (it runs 5 extra iterations).  Measured at B = 32: level 0 647 -> 554 us (fp32), 614 -> 518 (bf16) with 2 parts.
    int64_t nrp = cdiv64(1024, (int64_t)g.B * d.nqc * d.nwc);
    const int64_t maxp = g.H / 32 > 0 ? g.H / 32 : 1;
    if (nrp > maxp) nrp = maxp;
    if (nrp < 1) nrp = 1;
    static const int force_nrp = dcpt_tuning("DCPT_DWR_BWD_NRP", 0);
    if (force_nrp > 0) nrp = force_nrp < g.H ? force_nrp : g.H;
    d.nrp = (int)nrp;
    return d;
}

bool dwr_bwd_enabled() {
    static const int on = dcpt_tuning("DCPT_DW_RING_BWD", 1);
    return on != 0;
}

template <typename ST, int GATE = 0>
int launch_bwd(const void* dts, const void* t1, const float* w2p, const float* b2, const float* simg, const float* dpool, void* dt1,
               float* wpart, const DwGeom& g, hipStream_t s, float* tout = nullptr) {
    const DwrBGeom d = dwr_bwd_geom(g);
    DwrBP p{};
    p.tout = tout;
    p.t1 = t1; p.dts = dts; p.w2p = w2p; p.b2 = b2; p.simg = simg; p.dpool = dpool; p.dt1 = dt1; p.part = wpart;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C; p.nwc = d.nwc;
    DCPT_CHECK_ARG(((double)cdiv(g.H, d.nrp) + 6.0) * g.W * 2.0 * g.C * sizeof(ST) < 1.0e9, "depthwise ring: a row range exceeds the 32-bit window");
    const dim3 grid(d.nqc, d.nwc * d.nrp, g.B), blk(256);
#define DWRB(LP_, MU_) dwr_bwd_fused_kernel<ST, LP_, MU_, GATE><<<grid, blk, 0, s>>>(p)
    if (d.multi) {
        if (d.LP == 8) DWRB(8, 1);
        else if (d.LP == 16) DWRB(16, 1);
        else DWRB(32, 1);
    } else {
        if (d.LP == 8) DWRB(8, 0);
        else if (d.LP == 16) DWRB(16, 0);
        else DWRB(32, 0);
    }
#undef DWRB
    DCPT_CHECK_LAUNCH("dwr_bwd_fused");
    static const bool dbg = dcpt_tuning("DCPT_DWR_DEBUG", 0) != 0;
    if (dbg) {
        int n16 = -1, n8 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n16, dwr_bwd_fused_kernel<ST, 16, 1, GATE>, 256, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n8, dwr_bwd_fused_kernel<ST, 16, 0, GATE>, 256, 0);
        fprintf(stderr, "dwr_bwd gate=%d es=%d LP=%d multi=%d grid=(%d,%d,%d) occupancy(blocks/CU) multi %d single %d\n", GATE, (int)sizeof(ST), d.LP, d.multi, d.nqc,
                d.nwc * d.nrp, g.B, n16, n8);
    }
    return DCPT_OK;
}

}  // namespace

bool dw_ring_usable(const DwGeom& g, int es) { return dwr_enabled() && g.B <= 65535 && (g.C * es) % 16 == 0 && g.W >= 1; }
int dw_ring_num_blocks_per_image(const DwGeom& g, int es) {
    const DwrGeom d = dwr_geom(g, es);
    return d.nwc * d.nrp;
}
int launch_dw_ring_fwd_f32(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    trace_tag("dw.ring_fwd_f32");
    return launch_fwd<float>(t1, w2p, b2, t2, pool_part, g, s);
}
int launch_dw_ring_fwd_bf16(const bf16_t* t1, const float* w2p, const float* b2, bf16_t* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    trace_tag("dw.ring_fwd_bf16");
    return launch_fwd<bf16_t>(t1, w2p, b2, t2, pool_part, g, s);
}

// fused SimpleGate + depthwise backward on the ring; wpart[B][dw_ring_bwd_num_blocks_per_image][10][2C]
bool dw_ring_bwd_usable(const DwGeom& g, int es) { return dwr_bwd_enabled() && g.B <= 65535 && (g.C * es) % 16 == 0 && g.C % 2 == 0 && g.W >= 1; }
int dw_ring_bwd_num_blocks_per_image(const DwGeom& g) {
    const DwrBGeom d = dwr_bwd_geom(g);
    return d.nwc * d.nrp;
}
int launch_dw_ring_bwd_fused_f32(const float* dts, const float* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                                 float* dt1, float* wpart, const DwGeom& g, hipStream_t s) {
    trace_tag("dw.ring_bwd_f32");
    return launch_bwd<float>(dts, t1, w2p, b2, simg, dpool, dt1, wpart, g, s);
}
int launch_dw_ring_bwd_fused_bf16(const bf16_t* dts, const bf16_t* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                                  bf16_t* dt1, float* wpart, const DwGeom& g, hipStream_t s) {
    trace_tag("dw.ring_bwd_bf16");
    return launch_bwd<bf16_t>(dts, t1, w2p, b2, simg, dpool, dt1, wpart, g, s);
}

// Restormer (fp32): GDFN gate backward gelu(a_1) a_2 fused with the transposed conv and the tap gradients (u [M][2 Ch], dt [M][Ch]),
// and the plain depthwise backward dx = dw^T(dy) + tap gradients over Ctot channels (Ctot % 8 == 0); wpart as above with C = Ch / Ctot / 2
int launch_dw_ring_bwd_gelu_f32(const float* dt, const float* u, const float* w2p, float* du, float* wpart, int B, int H, int W, int Ch, hipStream_t s,
                                float* tout) {
    const DwGeom g{B, H, W, Ch};
    return launch_bwd<float, 1>(dt, u, w2p, nullptr, nullptr, nullptr, du, wpart, g, s, tout);
}
int launch_dw_ring_bwd_plain_f32(const float* dy, const float* x, const float* w2p, float* dx, float* wpart, int B, int H, int W, int Ctot, hipStream_t s) {
    DCPT_CHECK_ARG(Ctot % 8 == 0, "depthwise ring: Ctot=%d must be a multiple of 8", Ctot);
    const DwGeom g{B, H, W, Ctot / 2};
    return launch_bwd<float, 2>(dy, x, w2p, nullptr, nullptr, nullptr, dx, wpart, g, s);
}

// Restormer GDFN forward on the ring (fp32): t [M][Ch] = gelu(dw(u)[:, :Ch]) * dw(u)[:, Ch:]
int launch_dw_ring_gelu_fwd_f32(const float* u, const float* w2p, float* t, int B, int H, int W, int Ch, hipStream_t s) {
    const DwGeom g{B, H, W, Ch};
    return launch_fwd<float, 1>(u, w2p, nullptr, t, nullptr, g, s);
}
