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

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename ST>
__global__ __launch_bounds__(256) void dwr_gate_fwd_kernel(const DwrP p) {
    constexpr int ES = sizeof(ST), VW = 8 / ES, R = RING;
    __shared__ __attribute__((aligned(16))) unsigned char ring[R * SLOT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, LP = p.LP, U = LP / 2;               // U: 16-byte units per pixel-half in this chunk
    const int PH = C * ES / 8;                              // pieces per half over all channels
    const int PP = 256 / LP, PB = 2 * PP;
    // block -> (channel chunk, column chunk, row part, image)
    const int nrp = gridDim.y / p.nwc, rpp = (p.H + nrp - 1) / nrp;
    const int wc = blockIdx.y % p.nwc, rp = blockIdx.y / p.nwc, b = blockIdx.z;
    const int x0 = wc * PB;
    const int h0 = rp * rpp, h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
    const int piece0 = blockIdx.x * LP;
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
        else wait_vm<3 * (R - 2) + 2 * (R - 1)>();
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
                t.v[i] = (f1 + bias1.v[i]) * (f2 + bias2.v[i]);
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
            float s[VW];
#pragma unroll
            for (int i = 0; i < VW; ++i) s[i] = red[g * VW + i];
            for (int k = 1; k < PP; ++k)
#pragma unroll
                for (int i = 0; i < VW; ++i) s[i] += red[(k * LP + g) * VW + i];
            float* dst = p.part + ((int64_t)b * gridDim.y + blockIdx.y) * C + c1;
#pragma unroll
            for (int i = 0; i < VW; ++i) dst[i] = s[i];
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
    d.LP = lp;
    d.nqc = cdiv(PH, lp);
    const int PB = 2 * (256 / lp);
    d.nwc = cdiv(g.W, PB);
    // row parts: enough blocks to fill the chip (>= 1024), at least 8 rows each so that the two halo rows stay a small fraction
    int64_t nrp = cdiv64(cdiv64(1024, (int64_t)g.B * d.nqc), d.nwc);
    const int64_t maxp = g.H / 8 > 0 ? g.H / 8 : 1;
    if (nrp > maxp) nrp = maxp;
    if (nrp < 1) nrp = 1;
    d.nrp = (int)nrp;
    return d;
}

bool dwr_enabled() {
    static const int on = getenv("DCPT_DW_RING") ? atoi(getenv("DCPT_DW_RING")) : 1;
    return on != 0;
}

template <typename ST>
int launch_fwd(const void* t1, const float* w2p, const float* b2, void* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    const DwrGeom d = dwr_geom(g, (int)sizeof(ST));
    DwrP p{};
    p.in = t1; p.w2p = w2p; p.b2 = b2; p.out = t2; p.part = pool_part;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C; p.LP = d.LP; p.nwc = d.nwc;
    DCPT_CHECK_ARG(((double)cdiv(g.H, d.nrp) + 4.0) * g.W * 2.0 * g.C * sizeof(ST) < 1.0e9, "depthwise ring: a row range exceeds the 32-bit window");
    dwr_gate_fwd_kernel<ST><<<dim3(d.nqc, d.nwc * d.nrp, g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dwr_gate_fwd");
    return DCPT_OK;
}

}  // namespace

bool dw_ring_usable(const DwGeom& g, int es) { return dwr_enabled() && g.B <= 65535 && (g.C * es) % 16 == 0 && g.W >= 1; }
int dw_ring_num_blocks_per_image(const DwGeom& g, int es) {
    const DwrGeom d = dwr_geom(g, es);
    return d.nwc * d.nrp;
}
int launch_dw_ring_fwd_f32(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    return launch_fwd<float>(t1, w2p, b2, t2, pool_part, g, s);
}
int launch_dw_ring_fwd_bf16(const bf16_t* t1, const float* w2p, const float* b2, bf16_t* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    return launch_fwd<bf16_t>(t1, w2p, b2, t2, pool_part, g, s);
}
