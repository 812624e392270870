// Second half of a NAFBlock at the WIDE levels (C = 256 / 512) as ONE kernel in bf16 storage:
//     out = y + gamma * (conv5(SimpleGate(conv4(LayerNorm2(y)) + b4)) + b5)          reference basicsr/archs/nafnet_arch.py:180-186
// It replaces three launches of the unfused schedule (ln_fwd_bf16, conv4 with the bias + gate epilogue, conv5 with the residual epilogue:
// 19 + 52 + 44 us per level-3 block at B = 32, 256^2) and the HBM round trips of LN2(y), v and the gate between them.
//
// How the work is laid out (SURVEY.md 8d's canonical schedule for C >= 128: the chain per pixel tile, the weights streamed):
//   * A block owns a tile of TM = 128 pixels.  The tile's CURRENT operand -- LN2(y), later the gate -- lives in LDS as [128][C] bf16
//     (128 KB at C = 512), 16-byte chunks XOR-swizzled by row so that MFMA fragment reads, row reads and fragment-layout writes are all
//     conflict-free.  It is rewritten IN PLACE between the two GEMMs (a chain is sequential anyway: conv5 cannot start before the gate is
//     complete), so one buffer serves the whole chain and a CU holds a whole tile of rows.
//   * The products are computed TRANSPOSED, D[n][m] = sum_k W[n][k] x[m][k]: weights are the A operand, pixels the B operand.  Wave w of
//     the 8 owns the output channels [w C/8, (w+1) C/8) of every GEMM, so a weight element is used by exactly ONE wave: weights never
//     touch LDS.  They come L2 -> registers as ready-made MFMA fragments from a per-wave stream packed in consumption order
//     (bf16_ops.hip, pack mode 9; 1 KB per fragment, fully coalesced), eight fragments ahead of their use; the k-loops have no barriers
//     and no LDS writes.  Per k-step a wave issues 2 fragment loads + 4 ds_read_b128 for 8 MFMAs (32x32x16, 4 pixel tiles x 2 channel
//     tiles); a 128-pixel tile needs every weight once, i.e. 1.5 MB of L2 reads per 268 MFLOP -- 32 B/clk/CU at the MFMA peak.
//   * In the accumulator layout a lane holds ONE pixel and (the pack permutes the MFMA rows) 16 CONSECUTIVE channels per 32 x 32 tile:
//     bias, gate, gamma and the residual are per-lane arithmetic, a lane's results are 32 contiguous bytes of a bf16 row.
//   * conv4's 2C outputs do not fit the registers at once (128 x 1024 fp32 = the whole register file): a wave takes its gate channels in
//     passes of 32, each pass the v1 tile and its v2 partner (8 accumulator tiles = 128 registers), applies bias + gate at the end of the
//     pass, stores v, and keeps the gate as packed bf16 (32 registers per pass) until every wave is done reading LN2(y).
//   * LayerNorm2: a wave normalises 16 rows.  Rows come in coalesced (16 B per lane), go to LDS raw, and are re-read 16 rows x 4 lanes
//     (a lane holds a quarter row: lane-local sums + two shuffles), normalised in place; two-pass statistics in fp32 over the bf16
//     inputs exactly as ln_fwd_bf16.
// Rounding points are the unfused path's (nafblock_bf16.hip / oracle nafblock_bf16): LN2(y) -> bf16, v -> bf16 on store, the gate =
// product of the UNROUNDED halves rounded once, out rounded once.  Only the summation order inside a dot product differs.
#include "bf16_ops.h"
#include "chain_bf16.h"
#include "prof.h"

namespace {

constexpr int NW = CHAIN_NW;

template <int C, int TM>
struct Geo {
    static_assert(TM == 128 && (C == 256 || C == 512), "chain_bf16: TM = 128, C = 256 / 512");
    static constexpr int MT = TM / 32;       // pixel tiles
    static constexpr int CW = C / NW;        // channels of a wave
    static constexpr int NT = CW / 32;       // its 32-channel tiles: conv4 passes (a v1 and a v2 tile each) and conv5 tiles
    static constexpr int KS = C / 16;        // k-steps of a GEMM
    static constexpr int F4 = 2 * NT * KS;   // conv4 fragments of a wave
    static constexpr int FRAGS = F4 + NT * KS;
    static constexpr uint32_t WTOT = FRAGS * 1024u;   // bytes of a wave's stream (a multiple of 8 KB)
    static constexpr int PITCH = C * 2;      // bytes of a tile row
    static constexpr int XB = TM * PITCH;
    static constexpr int CPR = C / 8;        // 16-byte chunks of a row
    static constexpr int RPI = 64 / CPR;     // rows one coalesced wave access covers
    static constexpr int RW = TM / NW;       // rows a wave normalises (16)
    static constexpr int NI = RW / RPI;      // coalesced accesses per wave for its rows
    static constexpr int NJ = CPR / 4;       // chunks per lane in the (row, quarter) layout
    // LDS: the fp32 parameter tables first (ds offsets below 64 KB are immediates), then the tile
    static constexpr int T_LW = 0, T_LB = C, T_B4 = 2 * C, T_B5 = 4 * C, T_GM = 5 * C, T_B3 = 6 * C, T_BT = 7 * C, T_N = 8 * C;
    static constexpr int XOFF = T_N * 4;
    static constexpr int SMEM = XOFF + XB;
    static_assert(XOFF % 1024 == 0, "tile base alignment (the fragment addresses XOR bits 5..7)");
    static_assert(WTOT % 8192 == 0 && F4 % 8 == 0, "ring alignment");
};

// Compile-time ablations for tools/build_variant.sh (never in the product): CHAIN_ABL_NOW no weight loads in the k-loops, CHAIN_ABL_NOMFMA
// no MFMAs, CHAIN_ABL_NOLDS no pixel-fragment reads, CHAIN_ABL_NOST every global store dropped by the range check.
__device__ __forceinline__ bf16x8 ldfrag(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void st16(u32x4 v, rsrc_t r, uint32_t voff) {
#ifdef CHAIN_ABL_NOST
    voff |= ROW_SENT;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 o;
    o.x = bf_pack(f[0], f[1]);
    o.y = bf_pack(f[2], f[3]);
    o.z = bf_pack(f[4], f[5]);
    o.w = bf_pack(f[6], f[7]);
    return o;
}
__device__ __forceinline__ void unpack8(u32x4 w, float* f) {
    f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); f[2] = bf_lo(w.y); f[3] = bf_hi(w.y);
    f[4] = bf_lo(w.z); f[5] = bf_hi(w.z); f[6] = bf_lo(w.w); f[7] = bf_hi(w.w);
}
#ifdef CHAIN_TIMELINE   // diagnostic builds: shader-clock stamps of wave 0 / wave 4 of every block at the phase boundaries (tools/chain_timeline.py)
__device__ unsigned long long g_chain_tl[512 * 2 * 24];
#define TL(i) do { if (lane == 0 && (wave & 3) == 0) g_chain_tl[(blockIdx.x * 2 + (wave >> 2)) * 24 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define TL(i) do { } while (0)
#endif
// LDS is shared by the eight waves: their LDS traffic must have landed before the barrier, the weight stream (vmcnt) stays in flight
__device__ __forceinline__ void block_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One GEMM of the chain for this wave: acc[f][mt] += W-tile f (32 channels) x pixel tile mt over all k.  The fragment ring holds the next
// eight 1-KB fragments of the wave's stream; a slot is refilled right after its last use (the stream wraps: after the tile's last
// fragment come the first ones of the next tile).
// Rows of the tile that leave for HBM WHILE a GEMM runs (LN2(y) during conv4, the gate during conv5: both sit in LDS as that GEMM's
// operand).  vmcnt retires in issue order, so a wave that has just issued a burst of stores cannot take its next weight fragment until
// the burst has drained -- with every CU storing at once that is the HBM write time of the whole tensor, exposed (measured: 40 of 97 us).
// One coalesced row access every PER k-steps instead: ds_read in one step, the store in the next, never more than one in flight.
// A wave's weight stream: fragments of 1 KB at r + next, next + 1024, ...; after `tot` bytes the stream that FOLLOWS takes over (the same
// one again for the next tile, or -- with conv3 in front of the chain -- conv3's per-image stream after conv5 and the shared conv4 / conv5
// stream after conv3): the ring never runs dry across the seams.
struct Stream {
    rsrc_t r;
    uint32_t next, tot;
};
struct Trickle {
    rsrc_t dst;
    uint32_t lds0, glb0;   // lane constants: (RW wave + crow) PITCH [+ chunk 16 for glb0]
    int cchunk, rbase;     // rbase = (RW wave + crow) & 15
    int j0;                // first access of this GEMM
    bool on;
};
template <int NF, int PER, class G>
__device__ __forceinline__ void kloop(floatx16 (&acc)[NF][G::MT], bf16x8 (&ring)[8], Stream& st, const Stream& after, uint32_t l16,
                                      const unsigned char* X, uint32_t xlane, const Trickle& tr) {
    static_assert(8 % NF == 0, "fragments per k-step must divide the ring");
    static_assert(PER >= 2 && 8 % PER == 0, "trickle period");
    constexpr int MT = G::MT, NO = G::KS / 8;
    u32x4 pend = {0, 0, 0, 0};
    asm volatile("" : "+v"(xlane));   // (keeps the fragment addresses of a GEMM from being hoisted out of the tile loop and spilled)
    // The schedule is pinned per k-step (sched_barrier): pixel fragments of step k + 1 are requested BEFORE the MFMAs of step k, a ring
    // slot is refilled right after its last use -- left alone, hipcc reads fragments just in time and sinks the refills until the ring
    // is drained (s_waitcnt vmcnt(0) inside the loop).
    bf16x8 b[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[0][mt] = *reinterpret_cast<const bf16x8*>(X + xlane + mt * 32 * G::PITCH);
#pragma unroll 1
    for (int o = 0; o < NO; ++o) {
        const uint32_t xo = xlane + (uint32_t)o * 256u;
        const uint32_t xwrap = o + 1 < NO ? xo + 256u : xlane;   // (the last step's look-ahead re-reads step 0: harmless, in range)
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const uint32_t xn = k8 < 7 ? (xo ^ (uint32_t)((k8 + 1) << 5)) : xwrap;
#pragma unroll
#ifndef CHAIN_ABL_NOLDS
            for (int mt = 0; mt < MT; ++mt) b[(k8 + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(X + xn + mt * 32 * G::PITCH);
#else
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(b[(k8 + 1) & 1][mt]));
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int slot = (k8 * NF + f) & 7;
                const bf16x8 a = ring[slot];
#pragma unroll
#ifndef CHAIN_ABL_NOMFMA
                for (int mt = 0; mt < MT; ++mt) acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[k8 & 1][mt], acc[f][mt], 0, 0, 0);
#else
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[f][mt]) : "v"(a), "v"(b[k8 & 1][mt]));
#endif
#ifndef CHAIN_ABL_NOW
                ring[slot] = ldfrag(st.r, l16, st.next + (uint32_t)slot * 1024u);
#else
                asm volatile("" : "+v"(ring[slot]));
#endif
                if (slot == 7) {
                    st.next += 8192u;
                    if (st.next >= st.tot) st = after;   // (after.next = 0; wave-uniform: four scalar selects)
                }
            }
            {   // (unconditional: with nothing to store the window is empty and the range check drops the store -- no branch in the loop)
                const int j = tr.j0 + (o * 8 + k8) / PER;   // coalesced access j of the wave's rows: row (RW wave + crow) + j RPI
                if (k8 % PER == 0)
                    pend = *reinterpret_cast<const u32x4*>(X + tr.lds0 + (uint32_t)(j * G::RPI) * G::PITCH +
                                                           (uint32_t)(((tr.cchunk & ~15) | ((tr.cchunk ^ (tr.rbase + j * G::RPI)) & 15)) << 4));
                if (k8 % PER == 1) st16(pend, tr.dst, tr.glb0 + (uint32_t)(j * G::RPI) * G::PITCH);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// HEAD = 1: the first two links of the block instead, t1 = conv1(LayerNorm1(inp)) + b1 (reference nafnet_arch.py:169-170): `y` is the block
// input, the weight stream holds conv1 only (its 2C outputs in the same passes of a lower and an upper 32-channel tile; b4 = conv1's
// bias), `v` receives t1, `xn2` / `mu` / `rstd` LayerNorm1's output and statistics (conv1's weight-gradient operand; null in inference);
// b5 / gamma / out / g are not touched.
template <int C, int TM, int HEAD>
__global__ __launch_bounds__(512) void chain_fwd_bf16_kernel(const ChainFwdB p) {
    using G = Geo<C, TM>;
    constexpr int MT = G::MT, NT = G::NT, PITCH = G::PITCH, CW = G::CW;
    constexpr bool HD = HEAD == 1, C3 = HEAD == 2;   // HEAD = 2: the second half WITH conv3 in front (below)
    constexpr uint32_t WTOT = HD ? G::F4 * 1024u : G::WTOT, W3TOT = NT * G::KS * 1024u;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
    unsigned char* const X = smem + G::XOFF;
    float* const tab = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < G::T_N; i += 512)
        tab[i] = i < G::T_LB ? p.lnw[i] : i < G::T_B4 ? p.lnb[i - G::T_LB] : i < G::T_B5 ? p.b4[i - G::T_B4] : HD ? 0.f : i < G::T_GM ? p.b5[i - G::T_B5]
               : i < G::T_B3 ? p.gamma[i - G::T_GM] : !C3 ? 0.f : i < G::T_BT ? p.b3[i - G::T_B3] : p.beta[i - G::T_BT];

    TL(0);
    block_sync();   // (the parameter tables)
    TL(1);

    // this wave's weight stream; the first eight fragments are on their way while the tile is normalised
    const int64_t ntiles = (p.M + TM - 1) / TM;
    const Stream smain{make_rsrc(p.Wf + (size_t)wave * (WTOT / 2)), 0u, WTOT};
    // conv3's weights carry SCA's per-image scale (W3[n][k] s[img][k], wpack mode 11): one stream per (image, wave); a tile lies inside an image
    auto stream3 = [&](int64_t tile) {
        const int64_t img = tile * TM / (p.P > 0 ? p.P : 1);
        return Stream{make_rsrc(p.W3f + ((size_t)img * NW + wave) * (W3TOT / 2)), 0u, W3TOT};
    };
    const uint32_t l16 = (uint32_t)lane * 16u;
    Stream st = C3 ? stream3(blockIdx.x < ntiles ? blockIdx.x : 0) : smain;
    bf16x8 ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ring[i] = ldfrag(st.r, l16, (uint32_t)i * 1024u);
    st.next = 8192u % st.tot;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // Every per-lane address of the tile derives from this copy of the lane id, opaque to the optimiser: otherwise the ~150 loop-invariant
        // addresses of the phases below are hoisted out of the tile loop and live in scratch.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // accumulator layout: pixel m of a pixel tile, channel half h of a channel tile
        const int m = ln & 31, h = ln >> 5;
        const uint32_t xlane = (uint32_t)m * PITCH + (uint32_t)((((m & 14) | (h ^ (m & 1)))) << 4);   // fragment reads: chunk (2 ks + h) ^ (m & 15)
        // coalesced row layout: RPI rows per access, lane -> (row, chunk)
        const int crow = ln / G::CPR, cchunk = ln % G::CPR;
        // LayerNorm layout: 16 rows x 4 lanes
        const int rl = ln & 15, q = ln >> 4;
        const int64_t row0 = tile * TM;
        const uint32_t nrows = (uint32_t)((p.M - row0) < TM ? (p.M - row0) : TM);
        // windows at the tile's first row, as long as its valid rows: rows past M read 0 and their stores are dropped
        const rsrc_t yr = make_rsrc(p.y + row0 * C, nrows * PITCH);
        const rsrc_t t2r = make_rsrc(C3 ? p.t2 + row0 * C : p.y, C3 ? nrows * PITCH : 0);
        const rsrc_t outr = make_rsrc(HD ? p.y : p.out + row0 * C, HD ? 0 : nrows * PITCH);
        // what follows the shared conv4 / conv5 stream: itself (next tile), or the next tile's conv3 stream
        const Stream after = C3 ? stream3(tile + gridDim.x < ntiles ? tile + gridDim.x : tile) : smain;

        // ---- the wave's 16 rows: HBM -> registers -> LDS (raw), coalesced ----
        {
            u32x4 raw[G::NI];
#pragma unroll
            for (int j = 0; j < G::NI; ++j) {
                const int R = G::RW * wave + j * G::RPI + crow;
                raw[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(C3 ? t2r : yr, (uint32_t)R * PITCH + (uint32_t)cchunk * 16u, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < G::NI; ++j) {
                const int R = G::RW * wave + j * G::RPI + crow;
                *reinterpret_cast<u32x4*>(X + R * PITCH + (((cchunk & ~15) | ((cchunk ^ R) & 15)) << 4)) = raw[j];
            }
        }
        if constexpr (C3) {
            // ---- conv3 in front (reference nafnet_arch.py:174-178): y = inp + beta (conv3(t2 s) + b3).  The tile in LDS is t2; the product
            // comes back INTO the tile (accumulator layout -> LDS), leaves for HBM as whole rows, and is normalised where it lies ----
            TL(16);
            block_sync();   // t2 of all 128 rows is in LDS
            TL(17);
            floatx16 acc[NT][MT];
#pragma unroll
            for (int f = 0; f < NT; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][mt][r] = 0.f;
            Trickle tn;
            tn.lds0 = tn.glb0 = 0;
            tn.cchunk = tn.rbase = tn.j0 = 0;
            tn.on = false;
            tn.dst = make_rsrc(p.y, 0);   // (nothing leaves during this GEMM: an empty window)
            const rsrc_t inr = make_rsrc(p.inp + row0 * C, nrows * PITCH);
            u32x4 iv[NT][MT][2];   // the residual in the accumulator layout: the first channel tile's share is requested before the GEMM, the
            {                      // rest before the barrier behind it (it lands while the block waits)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint32_t io = (uint32_t)(32 * mt + m) * PITCH + (uint32_t)(CW * wave + 16 * h) * 2u;
                    iv[0][mt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(inr, io, 0, 0));
                    iv[0][mt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(inr, io + 16, 0, 0));
                }
            }
            kloop<NT, G::KS / G::NI, G>(acc, ring, st, smain, l16, X, xlane, tn);
            TL(18);
#pragma unroll
            for (int t = 1; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint32_t io = (uint32_t)(32 * mt + m) * PITCH + (uint32_t)(CW * wave + 32 * t + 16 * h) * 2u;
                    iv[t][mt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(inr, io, 0, 0));
                    iv[t][mt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(inr, io + 16, 0, 0));
                }
            block_sync();   // every wave has read t2 for the last time
            TL(19);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int cb = CW * wave + 32 * t + 16 * h;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int R = 32 * mt + m, ch0 = cb >> 3;
                    unsigned char* const xr = X + R * PITCH + ((ch0 & ~15) << 4);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float4 ba = *reinterpret_cast<const float4*>(tab + G::T_B3 + cb + 8 * hf), bb = *reinterpret_cast<const float4*>(tab + G::T_B3 + cb + 8 * hf + 4);
                        const float4 ga = *reinterpret_cast<const float4*>(tab + G::T_BT + cb + 8 * hf), gb = *reinterpret_cast<const float4*>(tab + G::T_BT + cb + 8 * hf + 4);
                        const float b3[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w}, bt[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                        float rf[8], o[8];
                        unpack8(iv[t][mt][hf], rf);
#pragma unroll
                        for (int r = 0; r < 8; ++r) o[r] = rf[r] + (acc[t][mt][8 * hf + r] + b3[r]) * bt[r];
                        *reinterpret_cast<u32x4*>(xr + ((((ch0 + hf) ^ R) & 15) << 4)) = pack8(o);
                    }
                }
            }
            TL(20);
            block_sync();   // y of all 128 rows is in LDS
            TL(21);
        }
        TL(2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // ---- LayerNorm2 in place: lane (rl, q) holds chunks 4 j + q of row RW wave + rl ----
        {
            const int R = G::RW * wave + rl;
            unsigned char* const xr = X + R * PITCH;
            // (the row quarter is unpacked ONCE into fp32 registers: the three passes over it are VALU-bound -- 128 elements per lane)
            float xf[G::NJ * 8];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const u32x4 rawy = *reinterpret_cast<const u32x4*>(xr + (j >> 2) * 256 + (((4 * (j & 3) + q) ^ rl) << 4));
                // (conv3 in front: y exists only in this tile -- it leaves for HBM from the LayerNorm's own reads, 16 rows x 64 bytes per store,
                // spread over the statistics pass so that the burst drains behind the arithmetic)
                if constexpr (C3) st16(rawy, yr, (uint32_t)R * PITCH + (uint32_t)(4 * j + q) * 16u);
                unpack8(rawy, xf + 8 * j);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += xf[8 * j + e];
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < G::NJ * 8; ++e) {
                xf[e] -= mean;
                sq += xf[e] * xf[e];
            }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            const float rs = 1.0f / sqrtf(sq * (1.0f / C) + p.eps);
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const int ch = 8 * (4 * j + q);
                const float4 w0 = *reinterpret_cast<const float4*>(tab + G::T_LW + ch), w1 = *reinterpret_cast<const float4*>(tab + G::T_LW + ch + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(tab + G::T_LB + ch), b1 = *reinterpret_cast<const float4*>(tab + G::T_LB + ch + 4);
                const float* f = xf + 8 * j;
                float o[8];
                o[0] = f[0] * rs * w0.x + b0.x; o[1] = f[1] * rs * w0.y + b0.y; o[2] = f[2] * rs * w0.z + b0.z; o[3] = f[3] * rs * w0.w + b0.w;
                o[4] = f[4] * rs * w1.x + b1.x; o[5] = f[5] * rs * w1.y + b1.y; o[6] = f[6] * rs * w1.z + b1.z; o[7] = f[7] * rs * w1.w + b1.w;
                *reinterpret_cast<u32x4*>(xr + (j >> 2) * 256 + (((4 * (j & 3) + q) ^ rl) << 4)) = pack8(o);
            }
            if (p.mu && q == 0 && (uint32_t)R < nrows) {
                p.mu[row0 + R] = mean;
                p.rstd[row0 + R] = rs;
            }
        }
        // LN2(y) is conv4's weight-gradient operand, the gate conv5's: the wave's 16 rows of each leave coalesced during the GEMM that reads them
        Trickle tr;
        tr.lds0 = (uint32_t)(G::RW * wave + crow) * PITCH;
        tr.glb0 = tr.lds0 + (uint32_t)cchunk * 16u;
        tr.cchunk = cchunk;
        tr.rbase = (G::RW * wave + crow) & 15;
        tr.on = p.xn2 != nullptr;
        tr.dst = make_rsrc(tr.on ? p.xn2 + row0 * C : p.y, tr.on ? nrows * PITCH : 0);
        TL(3);
        // (conv3 in front: the rows of y this wave stored are re-read at the end of the tile by OTHER waves, in the accumulator layout -- they
        // must have left this wave's queue before any wave passes the barrier; the LayerNorm above gave them the time)
        if constexpr (C3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_sync();   // LN2(y) of all 128 rows is in LDS
        TL(4);

        // ---- conv4 + bias + SimpleGate, a pass per 32 gate channels ----
        u32x4 greg[NT][MT][2];
#pragma unroll
        for (int ps = 0; ps < NT; ++ps) {
            floatx16 acc[2][MT];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][mt][r] = 0.f;
            tr.j0 = ps * (G::NI / NT);
            kloop<2, G::KS / (G::NI / NT), G>(acc, ring, st, after, l16, X, xlane, tr);
            TL(5 + 2 * ps);
            const int cb = CW * wave + 32 * ps + 16 * h;   // the lane's 16 consecutive channels (v1; the v2 partner is C + cb)
            const rsrc_t vr = make_rsrc(p.v ? p.v + row0 * 2 * C : p.y, p.v ? nrows * 2 * PITCH : 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {   // 8 channels at a time: the bias values are transient (LDS broadcast reads)
                    const float4 ba = *reinterpret_cast<const float4*>(tab + G::T_B4 + cb + 8 * hf), bb = *reinterpret_cast<const float4*>(tab + G::T_B4 + cb + 8 * hf + 4);
                    const float4 ca = *reinterpret_cast<const float4*>(tab + G::T_B4 + C + cb + 8 * hf), cc = *reinterpret_cast<const float4*>(tab + G::T_B4 + C + cb + 8 * hf + 4);
                    const float b1[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w}, b2[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                    float u1[8], u2[8], gv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        u1[r] = acc[0][mt][8 * hf + r] + b1[r];
                        u2[r] = acc[1][mt][8 * hf + r] + b2[r];
                        gv[r] = u1[r] * u2[r];
                    }
                    if (p.v) {
                        const uint32_t vo = (uint32_t)(32 * mt + m) * (2 * PITCH) + (uint32_t)cb * 2u + 16u * hf;
                        st16(pack8(u1), vr, vo);
                        st16(pack8(u2), vr, vo + PITCH);
                    }
                    if constexpr (!HD) {
                        greg[ps][mt][hf] = pack8(gv);
                        asm volatile("" : "+v"(greg[ps][mt][hf]));   // (pack NOW: otherwise the fp32 products stay live across the next pass and spill)
                    }
                }
            TL(6 + 2 * ps);
        }
        if constexpr (!HD) {
        // the residual in the accumulator layout: the tile's rows of y again (read once already: L2 / Infinity Cache).  The first channel
        // tile's share is requested HERE -- it arrives while the block waits at the two barriers below (a load issued right before the
        // GEMM would hold up its first weight fragment: vmcnt retires in order) --, the rest while the first is finished
        u32x4 yv[MT][2];
        {
            const int cb0 = CW * wave + 16 * h;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint32_t yo = (uint32_t)(32 * mt + m) * PITCH + (uint32_t)cb0 * 2u;
                yv[mt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, yo, 0, 0));
                yv[mt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, yo + 16, 0, 0));
            }
        }
        block_sync();   // every wave has read LN2(y) for the last time: the gate takes its place
        TL(9);
        {
#pragma unroll
            for (int ps = 0; ps < NT; ++ps)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int R = 32 * mt + m, ch0 = (CW * wave + 32 * ps + 16 * h) >> 3;   // chunks ch0, ch0 + 1 (ch0 even)
                    unsigned char* const xr = X + R * PITCH + ((ch0 & ~15) << 4);
                    *reinterpret_cast<u32x4*>(xr + (((ch0 ^ R) & 15) << 4)) = greg[ps][mt][0];
                    *reinterpret_cast<u32x4*>(xr + ((((ch0 + 1) ^ R) & 15) << 4)) = greg[ps][mt][1];
                }
        }
        TL(10);
        block_sync();   // the gate of all 128 rows is in LDS
        TL(11);

        // ---- conv5 + bias, gamma, residual ----
        {
            floatx16 acc[NT][MT];
#pragma unroll
            for (int f = 0; f < NT; ++f)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][mt][r] = 0.f;
            tr.j0 = 0;
            tr.on = p.g != nullptr;
            tr.dst = make_rsrc(tr.on ? p.g + row0 * C : p.y, tr.on ? nrows * PITCH : 0);
            kloop<NT, G::KS / G::NI, G>(acc, ring, st, after, l16, X, xlane, tr);
            TL(12);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int cb = CW * wave + 32 * t + 16 * h;
                if (t > 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const uint32_t yo = (uint32_t)(32 * mt + m) * PITCH + (uint32_t)cb * 2u;
                        yv[mt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, yo, 0, 0));
                        yv[mt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, yo + 16, 0, 0));
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float4 ba = *reinterpret_cast<const float4*>(tab + G::T_B5 + cb + 8 * hf), bb = *reinterpret_cast<const float4*>(tab + G::T_B5 + cb + 8 * hf + 4);
                        const float4 ga = *reinterpret_cast<const float4*>(tab + G::T_GM + cb + 8 * hf), gb = *reinterpret_cast<const float4*>(tab + G::T_GM + cb + 8 * hf + 4);
                        const float b5[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w}, gm[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                        float yf[8], o[8];
                        unpack8(yv[mt][hf], yf);
#pragma unroll
                        for (int r = 0; r < 8; ++r) o[r] = yf[r] + (acc[t][mt][8 * hf + r] + b5[r]) * gm[r];
                        st16(pack8(o), outr, (uint32_t)(32 * mt + m) * PITCH + (uint32_t)cb * 2u + 16u * hf);
                    }
            }
        }
        }
        TL(13);
#ifdef CHAIN_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TL(14);
        if (tile + gridDim.x < ntiles) block_sync();   // the next tile's rows overwrite the gate
    }
}

// ---- backward, the middle of the block: dy = dout + LayerNorm2'(gln; y), dts = dy (beta W3)^T, SCA's channel sums -------------------------
// (reference nafnet_arch.py:178-180 under autograd.)  The unfused schedule runs ln_bwd_bf16 (reads gln, y, dout, writes dy: 25 us at level 3,
// 88 us once B = 64 stacked batches leave the Infinity Cache) and the conv3^T data-gradient GEMM with the column-dot epilogue (36 us).  Here the
// LayerNorm backward IS the tile load of the GEMM: a wave computes dy for its 16 rows in the coalesced layout (a lane = 8 channels of a row;
// row sums by shuffles, the column partials of dLN2.weight / bias accumulate per lane over the rows), writes them to the LDS tile and lets
// them trickle out to HBM during the GEMM; the product leaves THROUGH the tile (accumulator layout -> LDS -> whole rows): coalesced 1-KB
// stores instead of 16-byte pieces, and SCA's sums  ds[img][k] = sum_px dts t2  are taken on the way out in the same row layout, t2 read
// coalesced.  (The unfused epilogue multiplies t2 with the fp32 accumulator, this one with the bf16-rounded dts it stores: 1e-4 relative
// on a sum over >= 128 pixels.)
template <int LANES>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

template <int C, int TM>
__global__ __launch_bounds__(512) void chain_bwd_mid_bf16_kernel(const ChainMidB p) {
    using G = Geo<C, TM>;
    constexpr int MT = G::MT, NT = G::NT, PITCH = G::PITCH, CW = G::CW;
    constexpr uint32_t WTOT = NT * G::KS * 1024u;
    constexpr int RED = 2 * NW * C * 4;   // bytes: [plane][wave][C] fp32 partial column sums
    static_assert(RED % 1024 == 0 && WTOT % 8192 == 0, "layout");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RED + G::XB];
    float* const red = reinterpret_cast<float*>(smem);
    unsigned char* const X = smem + RED;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Stream smain{make_rsrc(p.Wf + (size_t)wave * (WTOT / 2)), 0u, WTOT};
    Stream st = smain;
    const uint32_t l16 = (uint32_t)lane * 16u;
    bf16x8 ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ring[i] = ldfrag(st.r, l16, (uint32_t)i * 1024u);
    st.next = 8192u % WTOT;
    const float invC = 1.0f / (float)C;

    const int64_t ntiles = (p.M + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int m = ln & 31, h = ln >> 5;
        const uint32_t xlane = (uint32_t)m * PITCH + (uint32_t)((((m & 14) | (h ^ (m & 1)))) << 4);
        const int crow = ln / G::CPR, cchunk = ln % G::CPR;
        const int64_t row0 = tile * TM;
        const uint32_t nrows = (uint32_t)((p.M - row0) < TM ? (p.M - row0) : TM);
        const rsrc_t gr = make_rsrc(p.gln + row0 * C, nrows * PITCH), yr = make_rsrc(p.y + row0 * C, nrows * PITCH);
        const rsrc_t dr = make_rsrc(p.dout + row0 * C, nrows * PITCH), tr2 = make_rsrc(p.t2 + row0 * C, nrows * PITCH);
        const rsrc_t mur = make_rsrc(p.mu + row0, nrows * 4), rsr = make_rsrc(p.rstd + row0, nrows * 4);
        const rsrc_t dtsr = make_rsrc(p.dts + row0 * C, nrows * PITCH);

        // ---- LayerNorm2 backward of the wave's rows, coalesced: dy -> LDS tile; column partials per lane ----
        float ww[8], aw[8], ab[8];
        {
            const float4 w0 = ldg4(p.lnw + 8 * cchunk), w1 = ldg4(p.lnw + 8 * cchunk + 4);
            ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[e] = ab[e] = 0.f;
        constexpr int GRP = 4;   // accesses in flight together (3 tensors each)
#pragma unroll 1
        for (int j0 = 0; j0 < G::NI; j0 += GRP) {
            u32x4 rg[GRP], rx[GRP], rd[GRP];
            float rmean[GRP], rrs[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int R = G::RW * wave + (j0 + u) * G::RPI + crow;
                const uint32_t off = (uint32_t)R * PITCH + (uint32_t)cchunk * 16u;
                rg[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, off, 0, 0));
                rx[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0));
                rd[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, off, 0, 0));
                rmean[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mur, (uint32_t)R * 4u, 0, 0));
                rrs[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, (uint32_t)R * 4u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int R = G::RW * wave + (j0 + u) * G::RPI + crow;
                float g[8], x[8], d[8], gw[8];
                unpack8(rg[u], g);
                unpack8(rx[u], x);
                unpack8(rd[u], d);
                const float mean = rmean[u], rs = rrs[u];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = (x[e] - mean) * rs;   // xhat
                    gw[e] = g[e] * ww[e];
                    s1 += gw[e];
                    s2 += gw[e] * x[e];
                    aw[e] = fmaf(g[e], x[e], aw[e]);
                    ab[e] += g[e];
                }
                s1 = row_sum<G::CPR>(s1) * invC;
                s2 = row_sum<G::CPR>(s2) * invC;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (gw[e] - x[e] * s2 - s1) + d[e];
                *reinterpret_cast<u32x4*>(X + R * PITCH + (((cchunk & ~15) | ((cchunk ^ R) & 15)) << 4)) = pack8(o);
            }
        }
        if constexpr (G::RPI == 2) {   // two rows per access: lanes l and l ^ 32 hold the same channels
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                aw[e] += __shfl_xor(aw[e], 32);
                ab[e] += __shfl_xor(ab[e], 32);
            }
        }
        if (ln < G::CPR) {
            float* r0 = red + wave * C + 8 * cchunk;
            float* r1 = red + NW * C + wave * C + 8 * cchunk;
            *reinterpret_cast<float4*>(r0) = make_float4(aw[0], aw[1], aw[2], aw[3]);
            *reinterpret_cast<float4*>(r0 + 4) = make_float4(aw[4], aw[5], aw[6], aw[7]);
            *reinterpret_cast<float4*>(r1) = make_float4(ab[0], ab[1], ab[2], ab[3]);
            *reinterpret_cast<float4*>(r1 + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
        }
        block_sync();   // dy of all 128 rows and the eight waves' column partials are in LDS
        for (int c = tid; c < 2 * C; c += 512) {   // sum_rows gln xhat (-> dLN2.weight), sum_rows gln (-> dLN2.bias) of this tile, fixed order
            const int pl = c / C, ch = c - pl * C;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[(pl * NW + w) * C + ch];
            p.lnpart[(tile * 2 + pl) * C + ch] = a;
        }

        // ---- dts^T = (beta W3)^T-stream x dy^T; dy leaves for HBM row by row meanwhile ----
        Trickle tr;
        tr.lds0 = (uint32_t)(G::RW * wave + crow) * PITCH;
        tr.glb0 = tr.lds0 + (uint32_t)cchunk * 16u;
        tr.cchunk = cchunk;
        tr.rbase = (G::RW * wave + crow) & 15;
        tr.on = true;
        tr.dst = make_rsrc(p.dy + row0 * C, nrows * PITCH);
        tr.j0 = 0;
        floatx16 acc[NT][MT];
#pragma unroll
        for (int f = 0; f < NT; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][mt][r] = 0.f;
        kloop<NT, G::KS / G::NI, G>(acc, ring, st, smain, l16, X, xlane, tr);
        block_sync();   // every wave is done with dy (fragment reads and the trickle): the product takes its place
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int R = 32 * mt + m, ch0 = (CW * wave + 32 * t + 16 * h) >> 3;
                unsigned char* const xr = X + R * PITCH + ((ch0 & ~15) << 4);
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = acc[t][mt][r];
                *reinterpret_cast<u32x4*>(xr + (((ch0 ^ R) & 15) << 4)) = pack8(o);
                *reinterpret_cast<u32x4*>(xr + ((((ch0 + 1) ^ R) & 15) << 4)) = pack8(o + 8);
            }
        block_sync();   // dts of all 128 rows is in LDS
        // ---- whole rows out, SCA's channel sums on the way ----
        float ds[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ds[e] = 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 < G::NI; j0 += GRP) {
            u32x4 rt[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int R = G::RW * wave + (j0 + u) * G::RPI + crow;
                rt[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(tr2, (uint32_t)R * PITCH + (uint32_t)cchunk * 16u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int R = G::RW * wave + (j0 + u) * G::RPI + crow;
                const u32x4 v = *reinterpret_cast<const u32x4*>(X + R * PITCH + (((cchunk & ~15) | ((cchunk ^ R) & 15)) << 4));
                st16(v, dtsr, (uint32_t)R * PITCH + (uint32_t)cchunk * 16u);
                float a[8], b[8];
                unpack8(v, a);
                unpack8(rt[u], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) ds[e] = fmaf(a[e], b[e], ds[e]);
            }
        }
        if constexpr (G::RPI == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ds[e] += __shfl_xor(ds[e], 32);
        }
        if (ln < G::CPR) {
            float* r0 = red + wave * C + 8 * cchunk;
            *reinterpret_cast<float4*>(r0) = make_float4(ds[0], ds[1], ds[2], ds[3]);
            *reinterpret_cast<float4*>(r0 + 4) = make_float4(ds[4], ds[5], ds[6], ds[7]);
        }
        block_sync();
        for (int ch = tid; ch < C; ch += 512) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w * C + ch];
            p.dspart[tile * C + ch] = a;
        }
        if (tile + gridDim.x < ntiles) block_sync();   // the next tile's rows overwrite the tile and the partials
    }
}

int num_cus() {   // of the CURRENT device, cached per device index (a process may drive devices with different CU counts)
    static int n[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        hipDeviceProp_t pr;
        n[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n[dev];
}

template <int C, int HEAD>
int launch_t(const ChainFwdB& p, hipStream_t s) {
    trace_tag(HEAD == 1 ? "chain.head" : HEAD == 2 ? "chain.conv3+ffn" : (p.xn2 ? "chain.ffn_train" : "chain.ffn_infer"));
    const int64_t ntiles = (p.M + 127) / 128;
    const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
    // (live timing for bench.py: flops of the GEMMs, algorithmic bytes of what the launch reads and writes)
    const double units = HEAD == 1 ? 1.0 + 2.0 + (p.xn2 ? 1.0 : 0.0) : (HEAD == 2 ? 3.0 : 0.0) + 2.0 + 1.0 + (p.v ? 2.0 : 0.0) + (p.xn2 ? 2.0 : 0.0);
    ProfScope prof(s, PROF_OTHER + (HEAD == 2 ? 3 : HEAD), p.M, HEAD == 1 ? 2 * C : HEAD == 2 ? 4 * C : 3 * C, C,
                   (HEAD == 1 ? 4.0 : HEAD == 2 ? 8.0 : 6.0) * (double)p.M * C * C, units * (double)p.M * C * 2.0);
    chain_fwd_bf16_kernel<C, 128, HEAD><<<dim3(grid), dim3(512), 0, s>>>(p);
    DCPT_CHECK_LAUNCH("chain_fwd_bf16");
    return DCPT_OK;
}

}  // namespace

#ifdef CHAIN_TIMELINE
extern "C" int dcpt_chain_timeline_read(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chain_tl), bytes < sizeof(g_chain_tl) ? bytes : sizeof(g_chain_tl)) == hipSuccess ? 0 : 3;
}
#endif
bool chain_fwd_bf16_ok(int C, int64_t M) {
    static const int on = dcpt_tuning("DCPT_FFN_CHAIN", 1);
    // a block = one CU for a 128-pixel tile: taken when the last round of tiles fills at least 3/4 of the chip (nafblock_bf16.hip)
    const int64_t nt = (M + 127) / 128, cus = num_cus(), rounds = (nt + cus - 1) / cus;
    return on && (C == 256 || C == 512) && nt * 4 >= rounds * cus * 3;
}
size_t chain_wstream_elems(int C) { return (C == 256 || C == 512) ? (size_t)3 * C * C : 0; }

int launch_chain_fwd_bf16(const ChainFwdB& p, int C, hipStream_t s) {
    DCPT_CHECK_ARG(p.y && p.Wf && p.out && p.lnw && p.lnb && p.b4 && p.b5 && p.gamma && p.M > 0, "chain_fwd_bf16: null argument");
    DCPT_CHECK_ARG((p.xn2 == nullptr) == (p.g == nullptr) && (p.mu == nullptr) == (p.rstd == nullptr), "chain_fwd_bf16: xn2 / g and mu / rstd come in pairs");
    if (p.t2) {   // conv3 in front: y is an OUTPUT
        DCPT_CHECK_ARG(p.inp && p.W3f && p.b3 && p.beta && p.P >= 128 && p.P % 128 == 0 && p.M % p.P == 0,
                       "chain_fwd_bf16 (conv3 in front): null argument or image size %d not a multiple of 128 pixels", p.P);
        if (C == 512) return launch_t<512, 2>(p, s);
        if (C == 256) return launch_t<256, 2>(p, s);
    }
    if (C == 512) return launch_t<512, 0>(p, s);
    if (C == 256) return launch_t<256, 0>(p, s);
    dcpt_set_error("chain_fwd_bf16: no kernel for C=%d", C);
    return DCPT_ERR_ARG;
}

size_t chain_mid_wstream_elems(int C) { return (C == 256 || C == 512) ? (size_t)C * C : 0; }

int launch_chain_bwd_mid_bf16(const ChainMidB& p, int C, hipStream_t s) {
    trace_tag("chain.mid");
    DCPT_CHECK_ARG(p.gln && p.y && p.dout && p.t2 && p.mu && p.rstd && p.lnw && p.Wf && p.dy && p.dts && p.lnpart && p.dspart && p.M > 0,
                   "chain_bwd_mid_bf16: null argument");
    const int64_t ntiles = (p.M + 127) / 128;
    const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
    ProfScope prof(s, PROF_OTHER + 2, p.M, C, C, 2.0 * (double)p.M * C * C, 6.0 * (double)p.M * C * 2.0);
    if (C == 512) chain_bwd_mid_bf16_kernel<512, 128><<<dim3(grid), dim3(512), 0, s>>>(p);
    else if (C == 256) chain_bwd_mid_bf16_kernel<256, 128><<<dim3(grid), dim3(512), 0, s>>>(p);
    else {
        dcpt_set_error("chain_bwd_mid_bf16: no kernel for C=%d", C);
        return DCPT_ERR_ARG;
    }
    DCPT_CHECK_LAUNCH("chain_bwd_mid_bf16");
    return DCPT_OK;
}

size_t chain_head_wstream_elems(int C) { return (C == 256 || C == 512) ? (size_t)2 * C * C : 0; }

int launch_chain_head_bf16(const ChainFwdB& p, int C, hipStream_t s) {
    DCPT_CHECK_ARG(p.y && p.Wf && p.v && p.lnw && p.lnb && p.b4 && p.M > 0, "chain_head_bf16: null argument");
    DCPT_CHECK_ARG((p.mu == nullptr) == (p.rstd == nullptr) && !p.g, "chain_head_bf16: mu / rstd come in pairs, g is not written");
    if (C == 512) return launch_t<512, 1>(p, s);
    if (C == 256) return launch_t<256, 1>(p, s);
    dcpt_set_error("chain_head_bf16: no kernel for C=%d", C);
    return DCPT_ERR_ARG;
}
