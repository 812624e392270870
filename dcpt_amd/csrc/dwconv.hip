// Depthwise 3x3 (zero pad 1) + bias + SimpleGate on NHWC, forward and backward
// (reference basicsr/archs/nafnet_arch.py:96-104 conv2, :77-80 SimpleGate, :171-172).
//
// Thread = one channel quad (float4, coalesced along C) x one pixel column; it walks a band of RH
// rows, streaming each input row once and scattering it into three running output-row accumulators
// (kernel rows 2/1/0), so the only re-reads are the left/right neighbours (L1/L2 hits) and the two
// halo rows per band.  A block is QB quads x PB columns (QB*PB = 256) and loops over several
// (band, column-chunk) items of one image so that per-channel sums (pooling for SCA, depthwise
// weight gradients) are reduced in registers, then once per block in LDS in a fixed order, and
// written as per-block partials: deterministic, no float atomics.
#include <stdlib.h>
#include "prof.h"

#include "bf16.h"
#include "bufops.h"
#include "kernels.h"

namespace {

constexpr int RH = 8;  // rows per band

// A thread owns VW channels (4 by default; VW = 2 keeps the 9 (x2) taps, the running rows and the loads at ~80 VGPRs
// -> 5-7 waves per SIMD instead of 2-3, which measured no faster on MI355X), and every global access is a range-checked
// buffer access relative to the image, so halo rows / columns are ordinary loads that return 0 (no exec-mask
// branches: the loads of a row are all in flight together).
template <int VW>
struct vf {
    float v[VW];
};
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define VFOR for (int i_ = 0; i_ < VW; ++i_)
template <int VW> __device__ __forceinline__ vf<VW> vz() { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = 0.f; return r; }
template <int VW> __device__ __forceinline__ vf<VW> vfma(vf<VW> a, vf<VW> b, vf<VW> c) { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = fmaf(a.v[i_], b.v[i_], c.v[i_]); return r; }
template <int VW> __device__ __forceinline__ vf<VW> vmul(vf<VW> a, vf<VW> b) { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = a.v[i_] * b.v[i_]; return r; }
template <int VW> __device__ __forceinline__ vf<VW> vadd(vf<VW> a, vf<VW> b) { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = a.v[i_] + b.v[i_]; return r; }
template <int VW> __device__ __forceinline__ vf<VW> bld(rsrc_t rs, uint32_t off) {
    vf<VW> r;
    if constexpr (VW == 4) {
        const floatx4 t = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        const floatx2 t = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
        r.v[0] = t.x; r.v[1] = t.y;
    }
    return r;
}
template <int VW> __device__ __forceinline__ void bst(rsrc_t rs, uint32_t off, vf<VW> a) {
    if constexpr (VW == 4) {
        floatx4 t; t.x = a.v[0]; t.y = a.v[1]; t.z = a.v[2]; t.w = a.v[3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs, off, 0, 0);
    } else {
        floatx2 t; t.x = a.v[0]; t.y = a.v[1];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, t), rs, off, 0, 0);
    }
}
// storage-typed variants: T = float (as above) or bf16_t (VW bf16 values = VW * 2 bytes per access, converted to / from fp32)
template <int VW, typename T> __device__ __forceinline__ vf<VW> bldT(rsrc_t rs, uint32_t off) {
    if constexpr (sizeof(T) == 4) {
        return bld<VW>(rs, off);
    } else {
        vf<VW> r;
        if constexpr (VW == 4) {
            const float4 t = bbuf_ld4(rs, off);
            r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
        } else {
            const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
            r.v[0] = bf_lo(w); r.v[1] = bf_hi(w);
        }
        return r;
    }
}
template <int VW, typename T> __device__ __forceinline__ void bstT(rsrc_t rs, uint32_t off, vf<VW> a) {
    if constexpr (sizeof(T) == 4) {
        bst<VW>(rs, off, a);
    } else if constexpr (VW == 4) {
        bbuf_st4(rs, off, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(bf_pack(a.v[0], a.v[1]), rs, off, 0, 0);
    }
}
template <int VW> __device__ __forceinline__ vf<VW> gld(const float* p, bool ok) {   // small per-channel vectors
    vf<VW> r = vz<VW>();
    if (ok) { _Pragma("unroll") VFOR r.v[i_] = p[i_]; }
    return r;
}
template <int VW> __device__ __forceinline__ void gst(float* p, vf<VW> a) { _Pragma("unroll") VFOR p[i_] = a.v[i_]; }

struct DwMap {
    int QW;   // channel groups (of VW channels) handled per pixel
    int QB;   // groups per block (power of two)
    int PB;   // columns per block
    int nqc;  // group chunks
    int nwc;  // column chunks
};

__host__ __device__ inline DwMap dw_map(int H, int W, int groups) {
    DwMap m;
    m.QW = groups;
    // at most 32 groups per block row, so a block spans >= 8 pixel columns and the left/right halo columns are
    // mostly served by the same CU's L1 instead of another XCD's HBM fetch
    int qb = 1;
    while (qb < groups && qb < 32) qb <<= 1;
    m.QB = qb;
    m.PB = 256 / qb;
    m.nqc = (groups + qb - 1) / qb;
    m.nwc = (W + m.PB - 1) / m.PB;
    return m;
}

// Block -> (channel chunk, y, image), XCD-aware: the blocks of one (image, channel chunk) get consecutive logical ids, i.e.
// run on ONE XCD.  Block y owns column chunk y % nwc and the y / nwc-th of gridDim.y / nwc equal row ranges, which it walks
// top to bottom in one pass: only the two rows around a range and the two columns around a chunk are read twice.
struct DwBlk {
    int x, y, z;      // channel chunk, (column chunk, row part), image
};
__device__ __forceinline__ DwBlk dw_block(const DwMap& mp) {
    const int nx = gridDim.x, ny = gridDim.y;
    const int total = nx * ny * gridDim.z;
    const int lin = xcd_remap(blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), total);
    DwBlk k;
    k.y = lin % ny;
    k.x = (lin / ny) % nx;
    k.z = lin / (ny * nx);
    return k;
}

// first image row a block's windows must reach: two rows above its row range (the t1 window of the weight gradient)
__device__ __forceinline__ int dw_row_base(const DwMap& mp, const DwBlk& bk, int H) {
    const int nrp = gridDim.y / mp.nwc, rpp = (H + nrp - 1) / nrp;
    const int rb = (bk.y / mp.nwc) * rpp - 2;
    return rb > 0 ? rb : 0;
}

struct DwP {
    const float* in0;   // fwd/bwd_a: t1 [M][2C];  bwd_b: da [M][2C]
    const float* in1;   // bwd_a: dts [M][C];      bwd_b: t1 [M][2C]
    const float* w2p;   // [9][2C]
    const float* b2;    // [2C]
    const float* simg;  // bwd_a: [B][C]
    const float* dpool; // bwd_a: [B][C]
    float* out;         // fwd: t2 [M][C]; bwd_a: da [M][2C]; bwd_b: dt1 [M][2C]
    float* part;        // fwd: pool_part [B][NBLK][C]; bwd_b: wpart [B*NBLK][10][2C]
    int B, H, W, C;
    int Ctot;  // bwd_b / plain: total channel count of the depthwise conv
    // fused backward, optional: per-pixel partials of  dt1 . u  and  dt1 . (t1 - cvec)  over this block's channel chunk,
    // rowpart[pixel][gridDim.x][2] -- the two row sums of the LayerNorm backward downstream (gemm.h, E_LNBWD2)
    float* rowpart;
    const float* uvec;   // [2C]
    const float* cvec;   // [2C]
};

__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.39894228040143268f * expf(-0.5f * a * a);
}
template <int VW> __device__ __forceinline__ vf<VW> vgelu(vf<VW> a) { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = gelu_f(a.v[i_]); return r; }
template <int VW> __device__ __forceinline__ vf<VW> vgelu_d(vf<VW> a) { vf<VW> r; _Pragma("unroll") VFOR r.v[i_] = gelu_df(a.v[i_]); return r; }

// per-block reduction over the PB pixel columns of the per-thread partial `val` (fixed order), result to dst[group]
template <int VW>
__device__ __forceinline__ void dw_block_reduce(float* red, vf<VW> val, int tid, int ql, int pl, int QB, int PB, bool qok, float* dst) {
    __syncthreads();
#pragma unroll
    VFOR red[tid * VW + i_] = val.v[i_];
    __syncthreads();
    if (pl == 0 && qok) {
        vf<VW> s;
#pragma unroll
        VFOR s.v[i_] = red[ql * VW + i_];
        for (int j = 1; j < PB; ++j) {
#pragma unroll
            VFOR s.v[i_] += red[(j * QB + ql) * VW + i_];
        }
        gst<VW>(dst, s);
    }
}

// MODE 0: forward (t2 + pool partials);  MODE 1: backward-a (da)
// GATE 0: SimpleGate g1*g2 (NAFNet);  GATE 1: gelu(g1)*g2 (Restormer GDFN, exact erf GELU)
template <int VW, int MODE, int GATE, typename ST = float>
__global__ __launch_bounds__(256) void dw_gate_kernel(const DwP p) {
    constexpr uint32_t ES = sizeof(ST);   // bytes per stored activation element (fp32 or bf16); arithmetic is fp32 either way
    __shared__ float red[256 * VW];
    const DwMap mp = dw_map(p.H, p.W, p.C / VW);
    const DwBlk bk = dw_block(mp);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = bk.x * mp.QB + ql;
    const int b = bk.z;
    const bool qok = q < mp.QW;
    const int C = p.C, C2 = 2 * p.C;
    const int c1 = VW * q, c2 = C + VW * q;

    vf<VW> w1[9], w2[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w1[t] = gld<VW>(p.w2p + t * C2 + c1, qok);
        w2[t] = gld<VW>(p.w2p + t * C2 + c2, qok);
    }
    const vf<VW> bias1 = gld<VW>(p.b2 + c1, qok && p.b2), bias2 = gld<VW>(p.b2 + c2, qok && p.b2);
    vf<VW> sv, dpv = vz<VW>();
#pragma unroll
    VFOR sv.v[i_] = 1.f;
    if (MODE == 1 && qok && p.simg) {
        sv = gld<VW>(p.simg + (int64_t)b * C + c1, true);
        dpv = gld<VW>(p.dpool + (int64_t)b * C + c1, true);
    }
    vf<VW> pool = vz<VW>();
    // windows start two rows above this block's row range of image b (offsets of the range fit 32 bits: checked by the
    // launchers), rows are addressed relative to rb
    const int rb = dw_row_base(mp, bk, p.H);
    const int64_t img = ((int64_t)b * p.H + rb) * p.W;
    const rsrc_t rs_in = make_rsrc((const ST*)p.in0 + img * C2);
    const rsrc_t rs_out = make_rsrc((ST*)p.out + img * (MODE == 0 ? C : C2));
    const rsrc_t rs_d = make_rsrc(MODE == 1 ? (const ST*)p.in1 + img * C : (const ST*)p.in0);
    const uint32_t st = ES * (uint32_t)C2;

    {
        const int wc = bk.y % mp.nwc, nrp = gridDim.y / mp.nwc, rpp = (p.H + nrp - 1) / nrp;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = (bk.y / mp.nwc) * rpp;
        const int h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
        const uint32_t cl = (ok && x > 0) ? 0u : COL_SENT, cc = ok ? 0u : COL_SENT, cr = (ok && x + 1 < p.W) ? 0u : COL_SENT;
        vf<VW> a0_1 = vz<VW>(), a0_2 = vz<VW>(), a1_1 = vz<VW>(), a1_2 = vz<VW>();
        for (int r = h0 - 1; r <= h1; ++r) {
            const uint32_t ro = (r >= 0 && r < p.H) ? (uint32_t)(((r - rb) * p.W + x) * C2) * ES : ROW_SENT;
            const uint32_t o1 = ro + ES * (uint32_t)c1, o2 = ro + ES * (uint32_t)c2;
            const vf<VW> xl1 = bldT<VW, ST>(rs_in, (o1 - st) | cl), xl2 = bldT<VW, ST>(rs_in, (o2 - st) | cl);
            const vf<VW> xc1 = bldT<VW, ST>(rs_in, o1 | cc), xc2 = bldT<VW, ST>(rs_in, o2 | cc);
            const vf<VW> xr1 = bldT<VW, ST>(rs_in, (o1 + st) | cr), xr2 = bldT<VW, ST>(rs_in, (o2 + st) | cr);
            const int y = r - 1;
            const bool yok = ok && y >= h0;
            vf<VW> dts = vz<VW>();
            if (MODE == 1) dts = bldT<VW, ST>(rs_d, yok ? (uint32_t)(((y - rb) * p.W + x) * C + c1) * ES : ROW_SENT);
            // kernel row 2 completes output row r-1
            a0_1 = vfma(w1[6], xl1, vfma(w1[7], xc1, vfma(w1[8], xr1, a0_1)));
            a0_2 = vfma(w2[6], xl2, vfma(w2[7], xc2, vfma(w2[8], xr2, a0_2)));
            // kernel row 1 -> output row r
            a1_1 = vfma(w1[3], xl1, vfma(w1[4], xc1, vfma(w1[5], xr1, a1_1)));
            a1_2 = vfma(w2[3], xl2, vfma(w2[4], xc2, vfma(w2[5], xr2, a1_2)));
            // kernel row 0 starts output row r+1
            const vf<VW> a2_1 = vfma(w1[0], xl1, vfma(w1[1], xc1, vmul(w1[2], xr1)));
            const vf<VW> a2_2 = vfma(w2[0], xl2, vfma(w2[1], xc2, vmul(w2[2], xr2)));
            const vf<VW> g1 = vadd(a0_1, bias1), g2 = vadd(a0_2, bias2);
            if (MODE == 0) {
                const vf<VW> t = vmul(GATE == 0 ? g1 : vgelu(g1), g2);
                bstT<VW, ST>(rs_out, yok ? (uint32_t)(((y - rb) * p.W + x) * C + c1) * ES : ROW_SENT, t);
                if (yok) pool = vadd(pool, t);
            } else {
                const vf<VW> dt2 = vfma(dts, sv, dpv);
                const uint32_t oo = yok ? (uint32_t)(((y - rb) * p.W + x) * C2) * ES : ROW_SENT;
                if (GATE == 0) {
                    bstT<VW, ST>(rs_out, oo + ES * (uint32_t)c1, vmul(dt2, g2));
                    bstT<VW, ST>(rs_out, oo + ES * (uint32_t)c2, vmul(dt2, g1));
                } else {
                    bstT<VW, ST>(rs_out, oo + ES * (uint32_t)c1, vmul(vmul(dt2, g2), vgelu_d(g1)));
                    bstT<VW, ST>(rs_out, oo + ES * (uint32_t)c2, vmul(dt2, vgelu(g1)));
                }
            }
            a0_1 = a1_1; a0_2 = a1_2;
            a1_1 = a2_1; a1_2 = a2_2;
        }
    }
    if (MODE == 0 && p.part != nullptr)
        dw_block_reduce<VW>(red, pool, tid, ql, pl, mp.QB, mp.PB, qok, p.part + ((int64_t)b * gridDim.y + bk.y) * C + c1);
}

// backward-b: dt1 = dw3x3^T(da), plus per-block partial sums of dw2[ch][tap] and db2[ch].
// Thread = VW of the Ctot channels.
template <int VW>
__global__ __launch_bounds__(256) void dw_bwd_b_kernel(const DwP p) {
    __shared__ float red[256 * VW];
    const int C2 = p.Ctot;
    const DwMap mp = dw_map(p.H, p.W, C2 / VW);
    const DwBlk bk = dw_block(mp);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = bk.x * mp.QB + ql;
    const int b = bk.z;
    const bool qok = q < mp.QW;
    const int c0 = VW * q;
    vf<VW> w[9], wa[10];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = gld<VW>(p.w2p + t * C2 + c0, qok);
#pragma unroll
    for (int t = 0; t < 10; ++t) wa[t] = vz<VW>();
    const int rb = dw_row_base(mp, bk, p.H);
    const int64_t img = ((int64_t)b * p.H + rb) * p.W;
    const rsrc_t rs_d = make_rsrc(p.in0 + img * C2);
    const rsrc_t rs_t = make_rsrc(p.in1 + img * C2);
    const rsrc_t rs_o = make_rsrc(p.out + img * C2);
    const uint32_t st = 4u * (uint32_t)C2;

    {
        const int wc = bk.y % mp.nwc, nrp = gridDim.y / mp.nwc, rpp = (p.H + nrp - 1) / nrp;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = (bk.y / mp.nwc) * rpp;
        const int h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
        const uint32_t cl = (ok && x > 0) ? 0u : COL_SENT, cc = ok ? 0u : COL_SENT, cr = (ok && x + 1 < p.W) ? 0u : COL_SENT;
        auto rowoff = [&](int r) -> uint32_t { return (r >= 0 && r < p.H) ? ((uint32_t)(((r - rb) * p.W + x) * C2) + (uint32_t)c0) * 4u : ROW_SENT; };
        vf<VW> a0 = vz<VW>(), a1 = vz<VW>();
        // t1 rows r-1, r at column x (for the weight gradient); row r+1 is loaded in the loop
        vf<VW> tm = bld<VW>(rs_t, rowoff(h0 - 2) | cc), tc = bld<VW>(rs_t, rowoff(h0 - 1) | cc);
        for (int r = h0 - 1; r <= h1; ++r) {
            const uint32_t o = rowoff(r);
            const vf<VW> dl = bld<VW>(rs_d, (o - st) | cl), dc = bld<VW>(rs_d, o | cc), dr = bld<VW>(rs_d, (o + st) | cr);
            const vf<VW> tp = bld<VW>(rs_t, rowoff(r + 1) | cc);
            // dt1[y][x] = sum_{ky,kx} da[y+1-ky][x+1-kx] * w[ky][kx];  da row r feeds y = r-1+ky
            // d{l,c,r} = da[r][x-1+j], j=0,1,2  <->  kx = 2-j
            a0 = vfma(w[0 * 3 + 2], dl, vfma(w[0 * 3 + 1], dc, vfma(w[0 * 3 + 0], dr, a0)));  // ky=0 -> y=r-1
            a1 = vfma(w[1 * 3 + 2], dl, vfma(w[1 * 3 + 1], dc, vfma(w[1 * 3 + 0], dr, a1)));  // ky=1 -> y=r
            const vf<VW> a2 = vfma(w[2 * 3 + 2], dl, vfma(w[2 * 3 + 1], dc, vmul(w[2 * 3 + 0], dr)));  // ky=2 -> y=r+1
            const int y = r - 1;
            bst<VW>(rs_o, (ok && y >= h0) ? rowoff(y) : ROW_SENT, a0);
            a0 = a1;
            a1 = a2;
            // weight gradient: each da row counted once (rows of this band only; dl/dc/dr are 0 outside the image)
            if (r >= h0 && r < h1) {
                // dw[ky][kx=2-j] += da[r][x-1+j] * t1[r-1+ky][x]
                wa[0 * 3 + 2] = vfma(dl, tm, wa[0 * 3 + 2]);
                wa[0 * 3 + 1] = vfma(dc, tm, wa[0 * 3 + 1]);
                wa[0 * 3 + 0] = vfma(dr, tm, wa[0 * 3 + 0]);
                wa[1 * 3 + 2] = vfma(dl, tc, wa[1 * 3 + 2]);
                wa[1 * 3 + 1] = vfma(dc, tc, wa[1 * 3 + 1]);
                wa[1 * 3 + 0] = vfma(dr, tc, wa[1 * 3 + 0]);
                wa[2 * 3 + 2] = vfma(dl, tp, wa[2 * 3 + 2]);
                wa[2 * 3 + 1] = vfma(dc, tp, wa[2 * 3 + 1]);
                wa[2 * 3 + 0] = vfma(dr, tp, wa[2 * 3 + 0]);
                wa[9] = vadd(wa[9], dc);
            }
            tm = tc;
            tc = tp;
        }
    }
    float* part = p.part + ((int64_t)b * gridDim.y + bk.y) * 10 * C2;
#pragma unroll
    for (int t = 0; t < 10; ++t) dw_block_reduce<VW>(red, wa[t], tid, ql, pl, mp.QB, mp.PB, qok, part + t * C2 + c0);
}

// Fused SimpleGate + depthwise backward (NAFNet):  dt1 = dw3x3^T(da),  da_1 = dt2 * a_2,  da_2 = dt2 * a_1,
// a = dw3x3(t1) + b2 (recomputed),  dt2 = dts * s + dpool,  plus the per-block partial sums of dw2[ch][tap] and db2[ch].
// `da` never goes to memory.  A thread owns VW channels of ONE half at one pixel column and streams down its row range; the
// thread of the other half sits in the neighbouring lane (lane ^ 1), and the only thing the two exchange is the recomputed
// conv output: da_own = dt2 * a_other, one DPP quad-permute per value.  For every t1 row a thread loads columns x-2..x+2,
// advances the forward-conv accumulators of columns x-1, x, x+1, turns the completed row of a into da at those three
// columns, and feeds it to the transposed-conv accumulators and the tap gradients (which pair da[row][x] with the t1 rows
// it still holds).  Splitting the halves across lanes halves the per-thread state of the earlier one-thread-both-halves form,
// which makes VW = 4 (16-byte accesses) fit: measured on the training step, split + VW 4 is +0.5 % over the unsplit VW 2
// kernel, while split + VW 2 (106 VGPRs, 4 waves/SIMD) is -1 % -- the kernel is limited by memory instructions issued per
// byte, not by occupancy.
// HBM traffic: t1 and dts once (+2-row halos), dt1 once -- 5 tensor units instead of 11 for the two-kernel form.
template <int VW, typename ST = float>
__global__ __launch_bounds__(256) void dw_bwd_fused_kernel(const DwP p) {
    constexpr uint32_t ES = sizeof(ST);
    __shared__ float red[256 * VW];
    const int C = p.C, C2 = 2 * p.C;
    const DwMap mp = dw_map(p.H, p.W, C2 / VW);   // "groups" = (channel group, half) pairs
    const DwBlk bk = dw_block(mp);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int hq = bk.x * mp.QB + ql;             // even: first half, odd: second half of channel group hq / 2
    const int b = bk.z;
    const bool qok = hq < mp.QW;
    const int cg = VW * (hq >> 1);                // channels inside a half (dts / s / dpool index)
    const int co = (hq & 1) * C + cg;             // this thread's channels of t1 / dt1 / the depthwise weights
    vf<VW> w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = gld<VW>(p.w2p + t * C2 + co, qok);
    const vf<VW> bias = gld<VW>(p.b2 + co, qok && p.b2);
    vf<VW> sv, dpv = vz<VW>();
#pragma unroll
    VFOR sv.v[i_] = 1.f;
    if (qok && p.simg) {
        sv = gld<VW>(p.simg + (int64_t)b * C + cg, true);
        dpv = gld<VW>(p.dpool + (int64_t)b * C + cg, true);
    }
    vf<VW> uv = vz<VW>(), cv = vz<VW>();
    if (p.rowpart && qok) {
        uv = gld<VW>(p.uvec + co, true);
        cv = gld<VW>(p.cvec + co, true);
    }
    vf<VW> g[10];   // tap gradients (0..8) and bias gradient (9)
#pragma unroll
    for (int t = 0; t < 10; ++t) g[t] = vz<VW>();
    const int wc = bk.y % mp.nwc, nrp = gridDim.y / mp.nwc, rpp = (p.H + nrp - 1) / nrp;
    const int x = wc * mp.PB + pl;
    const bool ok = qok && x < p.W;
    const int h0 = (bk.y / mp.nwc) * rpp;
    const int h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
    const int rb = h0 - 2 > 0 ? h0 - 2 : 0;
    const int64_t img = ((int64_t)b * p.H + rb) * p.W;
    const rsrc_t rs_t = make_rsrc((const ST*)p.in0 + img * C2);
    const rsrc_t rs_d = make_rsrc((const ST*)p.in1 + img * C);
    const rsrc_t rs_o = make_rsrc((ST*)p.out + img * C2);
    // column validity of x-2 .. x+2 (a column outside the image: its t1 is zero padding, its da does not exist)
    uint32_t cs[5];
    bool cin[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        cin[j] = ok && x + j - 2 >= 0 && x + j - 2 < p.W;
        cs[j] = cin[j] ? 0u : COL_SENT;
    }
    // forward-conv running accumulators of columns x-1, x, x+1 (index 0..2): A0 = row r-1 (complete after this step), A1 = row r
    vf<VW> A0[3], A1[3];
    // transposed-conv running accumulators at column x: B0 = output row rho-1, B1 = output row rho
    vf<VW> B0 = vz<VW>(), B1 = vz<VW>();
    // t1 at columns x-1..x+1 of the two previous rows (for the tap gradients)
    vf<VW> Tm[3], Tc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) A0[j] = A1[j] = Tm[j] = Tc[j] = vz<VW>();
    for (int r = h0 - 2; r <= h1 + 1; ++r) {
        // ---- t1 row r, columns x-2..x+2
        const bool rin = r >= 0 && r < p.H;
        const uint32_t ro = rin ? (uint32_t)(((r - rb) * p.W + x) * C2 + co) * ES : ROW_SENT;
        vf<VW> T[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) T[j] = bldT<VW, ST>(rs_t, (ro + (uint32_t)((j - 2) * C2 * (int)ES)) | cs[j]);
        // ---- dts of row rho = r-1 at columns x-1..x+1
        const int rho = r - 1;
        const bool rho_in = rho >= 0 && rho < p.H;
        vf<VW> D[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            D[j] = bldT<VW, ST>(rs_d, rho_in ? ((uint32_t)(((rho - rb) * p.W + x + j - 1) * C + cg) * ES) | cs[j + 1] : ROW_SENT);
        // ---- forward conv: row r contributes kernel row 2 to a[r-1], row 1 to a[r], row 0 to a[r+1]
        vf<VW> A2[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A0[j] = vfma(w[6], T[j], vfma(w[7], T[j + 1], vfma(w[8], T[j + 2], A0[j])));
            A1[j] = vfma(w[3], T[j], vfma(w[4], T[j + 1], vfma(w[5], T[j + 2], A1[j])));
            A2[j] = vfma(w[0], T[j], vfma(w[1], T[j + 1], vmul(w[2], T[j + 2])));
        }
        // ---- da of row rho at columns x-1..x+1 (zero where the pixel does not exist): dt2 * (a of the OTHER half, lane ^ 1)
        vf<VW> da[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool pin = rho_in && cin[j + 1];
            const vf<VW> dt2 = vfma(D[j], sv, dpv);
            const vf<VW> a_own = vadd(A0[j], bias);
            vf<VW> a_other;
#pragma unroll
            VFOR a_other.v[i_] = dpp_perm<0xB1>(a_own.v[i_]);
            da[j] = pin ? vmul(dt2, a_other) : vz<VW>();
        }
        // ---- transposed conv: da row rho feeds dt1 rows rho-1 (ky = 0), rho (ky = 1), rho+1 (ky = 2); da[.][x-1+j] <-> kx = 2-j
        B0 = vfma(w[2], da[0], vfma(w[1], da[1], vfma(w[0], da[2], B0)));
        B1 = vfma(w[5], da[0], vfma(w[4], da[1], vfma(w[3], da[2], B1)));
        const vf<VW> B2 = vfma(w[8], da[0], vfma(w[7], da[1], vmul(w[6], da[2])));
        {
            const int y = rho - 1;   // complete now
            bstT<VW, ST>(rs_o, (ok && y >= h0 && y < h1) ? (uint32_t)(((y - rb) * p.W + x) * C2 + co) * ES : ROW_SENT, B0);
            if (p.rowpart) {   // t1[y][x] is Tm[1] here (rows r-2 = y, r-1, r are in Tm, Tc, T)
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                VFOR {
                    a1 = fmaf(B0.v[i_], uv.v[i_], a1);
                    a2 = fmaf(B0.v[i_], Tm[1].v[i_] - cv.v[i_], a2);
                }
                if (!qok) a1 = a2 = 0.f;
                a1 = group_sum(a1, mp.QB);   // the QB lanes of this pixel (one channel chunk)
                a2 = group_sum(a2, mp.QB);
                if (ql == 0 && x < p.W && y >= h0 && y < h1)
                    *reinterpret_cast<float2*>(p.rowpart + ((((int64_t)b * p.H + y) * p.W + x) * gridDim.x + bk.x) * 2) = make_float2(a1, a2);
            }
        }
        // ---- tap gradients: da[rho][x] with t1 rows rho-1 (Tm), rho (Tc), rho+1 (= row r, T) at columns x-1..x+1; rows of this
        //      block's range only, so that every pixel is counted once
        if (rho >= h0 && rho < h1) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                g[0 * 3 + kx] = vfma(da[1], Tm[kx], g[0 * 3 + kx]);
                g[1 * 3 + kx] = vfma(da[1], Tc[kx], g[1 * 3 + kx]);
                g[2 * 3 + kx] = vfma(da[1], T[kx + 1], g[2 * 3 + kx]);
            }
            g[9] = vadd(g[9], da[1]);
        }
        // ---- shift the pipelines
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A0[j] = A1[j];
            A1[j] = A2[j];
            Tm[j] = Tc[j];
            Tc[j] = T[j + 1];
        }
        B0 = B1;
        B1 = B2;
    }
    float* part = p.part + ((int64_t)b * gridDim.y + bk.y) * 10 * C2;
#pragma unroll
    for (int t = 0; t < 10; ++t) dw_block_reduce<VW>(red, g[t], tid, ql, pl, mp.QB, mp.PB, qok, part + t * C2 + co);
}

// Plain depthwise 3x3 (no bias, no gate) over Ctot channels (Restormer MDTA qkv_dwconv), plus per-block partial
// sums of out^2 for the first nsq channels (the L2 norms of q and k over the pixels).
template <int VW>
__global__ __launch_bounds__(256) void dw_plain_kernel(const DwP p, int nsq) {
    __shared__ float red[256 * VW];
    const int C = p.Ctot;
    const DwMap mp = dw_map(p.H, p.W, C / VW);
    const DwBlk bk = dw_block(mp);
    const int tid = threadIdx.x;
    const int ql = tid % mp.QB, pl = tid / mp.QB;
    const int q = bk.x * mp.QB + ql;
    const int b = bk.z;
    const bool qok = q < mp.QW;
    const int c0 = VW * q;
    vf<VW> w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = gld<VW>(p.w2p + t * C + c0, qok);
    vf<VW> sq = vz<VW>();
    const int rb = dw_row_base(mp, bk, p.H);
    const int64_t img = ((int64_t)b * p.H + rb) * p.W;
    const rsrc_t rs_in = make_rsrc(p.in0 + img * C);
    const rsrc_t rs_o = make_rsrc(p.out + img * C);
    const uint32_t st = 4u * (uint32_t)C;
    {
        const int wc = bk.y % mp.nwc, nrp = gridDim.y / mp.nwc, rpp = (p.H + nrp - 1) / nrp;
        const int x = wc * mp.PB + pl;
        const bool ok = qok && x < p.W;
        const int h0 = (bk.y / mp.nwc) * rpp;
        const int h1 = (h0 + rpp < p.H) ? h0 + rpp : p.H;
        const uint32_t cl = (ok && x > 0) ? 0u : COL_SENT, cc = ok ? 0u : COL_SENT, cr = (ok && x + 1 < p.W) ? 0u : COL_SENT;
        vf<VW> a0 = vz<VW>(), a1 = vz<VW>();
        for (int r = h0 - 1; r <= h1; ++r) {
            const uint32_t o = (r >= 0 && r < p.H) ? ((uint32_t)(((r - rb) * p.W + x) * C) + (uint32_t)c0) * 4u : ROW_SENT;
            const vf<VW> xl = bld<VW>(rs_in, (o - st) | cl), xc = bld<VW>(rs_in, o | cc), xr = bld<VW>(rs_in, (o + st) | cr);
            a0 = vfma(w[6], xl, vfma(w[7], xc, vfma(w[8], xr, a0)));
            a1 = vfma(w[3], xl, vfma(w[4], xc, vfma(w[5], xr, a1)));
            const vf<VW> a2 = vfma(w[0], xl, vfma(w[1], xc, vmul(w[2], xr)));
            const int y = r - 1;
            const bool yok = ok && y >= h0;
            bst<VW>(rs_o, yok ? ((uint32_t)(((y - rb) * p.W + x) * C) + (uint32_t)c0) * 4u : ROW_SENT, a0);
            if (yok) sq = vfma(a0, a0, sq);
            a0 = a1;
            a1 = a2;
        }
    }
    if (p.part != nullptr)
        dw_block_reduce<VW>(red, sq, tid, ql, pl, mp.QB, mp.PB, qok && c0 < nsq, p.part + ((int64_t)b * gridDim.y + bk.y) * nsq + c0);
}

__global__ void dw_pack_kernel(const float* __restrict__ w2, float* __restrict__ w2p, int C2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C2 * 9) {
        const int ch = i / 9, t = i % 9;
        w2p[t * C2 + ch] = w2[i];
    }
}

// dw2[ch*9+tap] = sum_r wpart[r][tap][ch];  db2[ch] = sum_r wpart[r][9][ch]
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ wpart, int R, int C2,
                                                              float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float red[8][32];
    const int t = blockIdx.y;
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C2) {
#pragma unroll 8
        for (int r = rg; r < R; r += 8) s += wpart[((int64_t)r * 10 + t) * C2 + c];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C2) {
        float v = red[0][cl];
#pragma unroll
        for (int i = 1; i < 8; ++i) v += red[i][cl];
        if (t < 9) dw2[c * 9 + t] = v;
        else db2[c] = v;
    }
}

// channels per thread (2 or 4) and grid target; DCPT_DW_VW / DCPT_DW_BLOCKS override them for experiments
int dw_vw() {
    static int v = 0;
    if (v == 0) {
        v = dcpt_tuning("DCPT_DW_VW", 4) == 2 ? 2 : 4;
    }
    return v;
}
int dw_target_blocks() {
    static int v = 0;
    if (v == 0) {
        v = dcpt_tuning("DCPT_DW_BLOCKS", 1024);
        if (v <= 0) v = 1024;
    }
    return v;
}

int nblk_for(const DwGeom& g, int quads) {
    const DwMap mp = dw_map(g.H, g.W, quads);
    // column chunks x row parts; a part is at least RH rows so that the two halo rows stay a small fraction
    int64_t nrp = cdiv64(cdiv64(dw_target_blocks(), (int64_t)g.B * mp.nqc), mp.nwc);
    const int64_t maxp = g.H / RH > 0 ? g.H / RH : 1;
    if (nrp > maxp) nrp = maxp;
    if (nrp < 1) nrp = 1;
    return (int)(nrp * mp.nwc);
}
}  // namespace

int dw_fused_vw() {   // channels per thread of the fused backward kernel (DCPT_DW_FUSED_VW overrides for experiments)
    static int v = 0;
    if (v == 0) {
        v = dcpt_tuning("DCPT_DW_FUSED_VW", 4) == 2 ? 2 : 4;
    }
    return v;
}
int dw_num_blocks_per_image_fused(const DwGeom& g) { return nblk_for(g, 2 * g.C / dw_fused_vw()); }
int dw_num_blocks_per_image(const DwGeom& g) { return nblk_for(g, g.C / dw_vw()); }

// a block's windows span its row range + halo: (rows/part + 5) * W * Ct floats must fit the 32-bit offsets
#define DW_CHECK_RANGE(H, W, Ct, NBLK, NWC)                                                                             \
    DCPT_CHECK_ARG(((double)cdiv((H), (NBLK) / (NWC) > 0 ? (NBLK) / (NWC) : 1) + 5.0) * (W) * (Ct) * 4.0 < 1.0e9,        \
                   "depthwise conv: a row range of a %d x %d x %d image exceeds the 32-bit window", H, W, Ct)
#define DW_LAUNCH(KERNEL, ...)                                  \
    do {                                                        \
        if (dw_vw() == 2) KERNEL<2 __VA_ARGS__;                 \
        else KERNEL<4 __VA_ARGS__;                              \
    } while (0)

int launch_dw_pack_weights(const float* w2, float* w2p, int C2, hipStream_t s) {
    dw_pack_kernel<<<dim3(cdiv(C2 * 9, 256)), dim3(256), 0, s>>>(w2, w2p, C2);
    DCPT_CHECK_LAUNCH("dw_pack");
    return DCPT_OK;
}

int launch_dw_fwd(const float* t1, const float* w2p, const float* b2, float* t2, float* pool_part, const DwGeom& g,
                  hipStream_t s) {
    trace_tag("dw.reg_fwd_f32");
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_fwd: C=%d must be a multiple of 4, B<=65535", g.C);
    DwP p{};
    p.in0 = t1; p.w2p = w2p; p.b2 = b2; p.out = t2; p.part = pool_part;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const DwMap mp = dw_map(g.H, g.W, g.C / dw_vw());
    DW_CHECK_RANGE(g.H, g.W, 2 * g.C, dw_num_blocks_per_image(g), mp.nwc);
    DW_LAUNCH(dw_gate_kernel, , 0, 0><<<dim3(mp.nqc, dw_num_blocks_per_image(g), g.B), dim3(256), 0, s>>>(p));
    DCPT_CHECK_LAUNCH("dw_fwd");
    return DCPT_OK;
}

int dw_fused_row_chunks(const DwGeom& g) { return dw_map(g.H, g.W, 2 * g.C / dw_fused_vw()).nqc; }

int launch_dw_bwd_fused(const float* dts, const float* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                        float* dt1, float* wpart, const DwGeom& g, hipStream_t s, float* rowpart, const float* uvec, const float* cvec) {
    trace_tag("dw.reg_bwd_f32");
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_bwd: C=%d must be a multiple of 4", g.C);
    DwP p{};
    p.in0 = t1; p.in1 = dts; p.w2p = w2p; p.b2 = b2; p.simg = simg; p.dpool = dpool; p.out = dt1; p.part = wpart;
    p.rowpart = rowpart; p.uvec = uvec; p.cvec = cvec;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const int vw = dw_fused_vw();
    const DwMap mp = dw_map(g.H, g.W, 2 * g.C / vw);
    const int nblk = dw_num_blocks_per_image_fused(g);
    DW_CHECK_RANGE(g.H, g.W, 2 * g.C, nblk, mp.nwc);
    if (vw == 2) dw_bwd_fused_kernel<2><<<dim3(mp.nqc, nblk, g.B), dim3(256), 0, s>>>(p);
    else dw_bwd_fused_kernel<4><<<dim3(mp.nqc, nblk, g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_bwd_fused");
    return DCPT_OK;
}

// ---- bf16-storage variants (bf16.h): same kernels, activations read / written as bf16, partial sums and parameters fp32
int launch_dw_fwd_bf16(const bf16_t* t1, const float* w2p, const float* b2, bf16_t* t2, float* pool_part, const DwGeom& g, hipStream_t s) {
    trace_tag("dw.reg_fwd_bf16");
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_fwd_bf16: C=%d must be a multiple of 4, B<=65535", g.C);
    DwP p{};
    p.in0 = reinterpret_cast<const float*>(t1); p.w2p = w2p; p.b2 = b2; p.out = reinterpret_cast<float*>(t2); p.part = pool_part;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const DwMap mp = dw_map(g.H, g.W, g.C / 4);
    const int nblk = nblk_for(g, g.C / 4);
    DW_CHECK_RANGE(g.H, g.W, 2 * g.C, nblk, mp.nwc);
    dw_gate_kernel<4, 0, 0, bf16_t><<<dim3(mp.nqc, nblk, g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_fwd_bf16");
    return DCPT_OK;
}
int dw_num_blocks_per_image_bf16(const DwGeom& g) { return nblk_for(g, g.C / 4); }
int dw_num_blocks_per_image_fused_bf16(const DwGeom& g) { return nblk_for(g, 2 * g.C / 4); }

int dw_fused_row_chunks_bf16(const DwGeom& g) { return dw_map(g.H, g.W, 2 * g.C / 4).nqc; }

int launch_dw_bwd_fused_bf16(const bf16_t* dts, const bf16_t* t1, const float* w2p, const float* b2, const float* simg, const float* dpool,
                             bf16_t* dt1, float* wpart, const DwGeom& g, hipStream_t s, float* rowpart, const float* uvec, const float* cvec) {
    trace_tag("dw.reg_bwd_bf16");
    DCPT_CHECK_ARG(g.C % 4 == 0 && g.B <= 65535, "dw_bwd_bf16: C=%d must be a multiple of 4", g.C);
    DwP p{};
    p.in0 = reinterpret_cast<const float*>(t1); p.in1 = reinterpret_cast<const float*>(dts); p.w2p = w2p; p.b2 = b2; p.simg = simg;
    p.dpool = dpool; p.out = reinterpret_cast<float*>(dt1); p.part = wpart;
    p.rowpart = rowpart; p.uvec = uvec; p.cvec = cvec;
    p.B = g.B; p.H = g.H; p.W = g.W; p.C = g.C;
    const DwMap mp = dw_map(g.H, g.W, 2 * g.C / 4);
    const int nblk = nblk_for(g, 2 * g.C / 4);
    DW_CHECK_RANGE(g.H, g.W, 2 * g.C, nblk, mp.nwc);
    dw_bwd_fused_kernel<4, bf16_t><<<dim3(mp.nqc, nblk, g.B), dim3(256), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("dw_bwd_fused_bf16");
    return DCPT_OK;
}

// ---- generic entry points used by the Restormer blocks ---------------------------------------------
int dw_num_blocks_generic(int B, int H, int W, int Ctot) {
    DwGeom g{B, H, W, Ctot};
    return nblk_for(g, Ctot / dw_vw());
}

int launch_dw_gelu_fwd(const float* u, const float* w2p, float* t, int B, int H, int W, int Ch, hipStream_t s) {
    DCPT_CHECK_ARG(Ch % 4 == 0 && B <= 65535, "dw_gelu_fwd: Ch=%d must be a multiple of 4", Ch);
    DwP p{};
    p.in0 = u; p.w2p = w2p; p.out = t; p.B = B; p.H = H; p.W = W; p.C = Ch;
    DwGeom g{B, H, W, Ch};
    const DwMap mp = dw_map(H, W, Ch / dw_vw());
    DW_CHECK_RANGE(H, W, 2 * Ch, dw_num_blocks_per_image(g), mp.nwc);
    DW_LAUNCH(dw_gate_kernel, , 0, 1><<<dim3(mp.nqc, dw_num_blocks_per_image(g), B), dim3(256), 0, s>>>(p));
    DCPT_CHECK_LAUNCH("dw_gelu_fwd");
    return DCPT_OK;
}

int launch_dw_gelu_bwd_a(const float* dt, const float* u, const float* w2p, float* da, int B, int H, int W, int Ch, hipStream_t s) {
    DCPT_CHECK_ARG(Ch % 4 == 0 && B <= 65535, "dw_gelu_bwd_a: Ch=%d must be a multiple of 4", Ch);
    DwP p{};
    p.in0 = u; p.in1 = dt; p.w2p = w2p; p.out = da; p.B = B; p.H = H; p.W = W; p.C = Ch;
    DwGeom g{B, H, W, Ch};
    const DwMap mp = dw_map(H, W, Ch / dw_vw());
    DW_CHECK_RANGE(H, W, 2 * Ch, dw_num_blocks_per_image(g), mp.nwc);
    DW_LAUNCH(dw_gate_kernel, , 1, 1><<<dim3(mp.nqc, dw_num_blocks_per_image(g), B), dim3(256), 0, s>>>(p));
    DCPT_CHECK_LAUNCH("dw_gelu_bwd_a");
    return DCPT_OK;
}

int launch_dw_plain_fwd(const float* x, const float* w2p, float* y, float* sq_part, int nsq, int B, int H, int W, int Ctot,
                        hipStream_t s) {
    DCPT_CHECK_ARG(Ctot % 4 == 0 && nsq % 4 == 0 && B <= 65535, "dw_plain_fwd: Ctot=%d", Ctot);
    DwP p{};
    p.in0 = x; p.w2p = w2p; p.out = y; p.part = sq_part; p.B = B; p.H = H; p.W = W; p.Ctot = Ctot;
    const DwMap mp = dw_map(H, W, Ctot / dw_vw());
    DW_CHECK_RANGE(H, W, Ctot, dw_num_blocks_generic(B, H, W, Ctot), mp.nwc);
    DW_LAUNCH(dw_plain_kernel, ><<<dim3(mp.nqc, dw_num_blocks_generic(B, H, W, Ctot), B), dim3(256), 0, s>>>(p, nsq));
    DCPT_CHECK_LAUNCH("dw_plain_fwd");
    return DCPT_OK;
}

// dx = dw^T(dy) over Ctot channels, wpart[B*nblk][10][Ctot] (nblk = dw_num_blocks_generic)
int launch_dw_generic_bwd(const float* dy, const float* x, const float* w2p, float* dx, float* wpart, int B, int H, int W, int Ctot,
                          hipStream_t s) {
    DCPT_CHECK_ARG(Ctot % 4 == 0 && B <= 65535, "dw_generic_bwd: Ctot=%d", Ctot);
    DwP p{};
    p.in0 = dy; p.in1 = x; p.w2p = w2p; p.out = dx; p.part = wpart; p.B = B; p.H = H; p.W = W; p.Ctot = Ctot;
    const DwMap mp = dw_map(H, W, Ctot / dw_vw());
    DW_CHECK_RANGE(H, W, Ctot, dw_num_blocks_generic(B, H, W, Ctot), mp.nwc);
    DW_LAUNCH(dw_bwd_b_kernel, ><<<dim3(mp.nqc, dw_num_blocks_generic(B, H, W, Ctot), B), dim3(256), 0, s>>>(p));
    DCPT_CHECK_LAUNCH("dw_generic_bwd");
    return DCPT_OK;
}

int launch_dw_wgrad_reduce(const float* wpart, int R, int C2, float* dw2, float* db2, hipStream_t s) {
    dw_wgrad_reduce_kernel<<<dim3(cdiv(C2, 32), 10), dim3(256), 0, s>>>(wpart, R, C2, dw2, db2);
    DCPT_CHECK_LAUNCH("dw_wgrad_reduce");
    return DCPT_OK;
}
