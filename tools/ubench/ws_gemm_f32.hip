// Prototype for the round-4 verdict's item 2 ("fp32 GEMMs: spend fewer instructions per MFMA, then measure whether the clock really drops"):
// an exact-fp32 NT GEMM  C[m][n] = sum_k A[m][k] W[n][k] + b[n]  in the chain kernel's layout (dcpt_amd/csrc/chain_bf16.hip) -- a tile of
// 64 pixel rows resident in LDS, the weights streamed L2 -> registers as ready-made MFMA fragments, every wave its own output channels,
// NO barrier and NO LDS write inside the k-loop.  Per 16 v_mfma_f32_32x32x2_f32 (1024 cycles of the matrix pipe) a wave issues 2 buffer
// loads + 2 ds_read_b128 + a handful of scalar instructions; the product kernel (gemm_nt.hip: 128 x 128 x 32 tiles through LDS-DMA,
// one barrier per k-tile) needs 16 ds_read + its DMA issue + a barrier for the same MFMAs and pays a prologue / epilogue per 128 x 128 tile.
//   hipcc --offload-arch=gfx950 -O3 -I dcpt_amd/csrc tools/ubench/ws_gemm_f32.hip -o tools/ubench/ws_gemm_f32 && tools/ubench/ws_gemm_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "bufops.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int NW = 8, TM = 64, MT = TM / 32;

__device__ __forceinline__ void block_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wf: per wave FR = (K / 8) * NF fragments of 1 KB: q-group q (k = kk K/2 + 4q .. 4q + 3), then channel tile f
template <int K, int N>
__global__ __launch_bounds__(512) void ws_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Wf, const float* __restrict__ bias,
                                                     float* __restrict__ Cout, int64_t M, int reps) {
    constexpr int CW = N / NW, NF = CW / 32, PITCH = K * 4, NQ = K / 8, FRAGS = NQ * NF;
    constexpr uint32_t WTOT = FRAGS * 1024u;
    constexpr int RING = 8, QPR = RING / NF;   // q-groups per turn of the ring
    static_assert(NQ % 16 == 0 && 16 % QPR == 0, "shape");
    __shared__ __attribute__((aligned(1024))) unsigned char X[TM * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const rsrc_t wrs = make_rsrc(Wf + (size_t)wave * FRAGS * 256);
    const uint32_t l16 = (uint32_t)lane * 16u;
    floatx4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wrs, l16, (uint32_t)i * 1024u, 0));
    uint32_t wnext = RING * 1024u;
    const int64_t ntiles = (M + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int m = ln & 31, kk = ln >> 5;
        const int64_t row0 = tile * TM;
        const uint32_t nrows = (uint32_t)((M - row0) < TM ? (M - row0) : TM);
        const rsrc_t ar = make_rsrc(A + row0 * K, nrows * PITCH);
        const rsrc_t cr = make_rsrc(Cout + row0 * N, nrows * N * 4);
        // the wave's 8 rows, coalesced: a row = K / 4 chunks of 16 B = 2 accesses (K = 512)
        constexpr int APR = K / 256;   // accesses per row
        {
            floatx4 raw[8 * APR];
#pragma unroll
            for (int j = 0; j < 8 * APR; ++j) {
                const int R = 8 * wave + j / APR, c = (j % APR) * 64 + ln;
                raw[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(ar, (uint32_t)R * PITCH + (uint32_t)c * 16u, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 8 * APR; ++j) {
                const int R = 8 * wave + j / APR, c = (j % APR) * 64 + ln;
                *reinterpret_cast<floatx4*>(X + R * PITCH + (((c & ~15) | ((c ^ R) & 15)) << 4)) = raw[j];
            }
        }
        block_sync();
        // fragment reads: lane (pixel m, half kk) reads chunk kk * K/8 + q of row 32 mt + m, stored at chunk ^ (row & 15)
        uint32_t xlane = (uint32_t)m * PITCH + (uint32_t)kk * (K / 8) * 16u + (uint32_t)((m & 15) << 4);
        asm volatile("" : "+v"(xlane));
        floatx16 acc[NF][MT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][mt][r] = 0.f;
        floatx4 b[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[0][mt] = *reinterpret_cast<const floatx4*>(X + xlane + mt * 32 * PITCH);
        // reps > 1: the k-loop over and over on the same tile (the result is reps x the product) -- the loop alone, without the tile load and
        // the epilogue: what the matrix pipe sustains when NOTHING else is in the way (MfmaUtil and clock by tools/pmc_cmd.sh)
#pragma unroll 1
        for (int rp = 0; rp < reps; ++rp)
#pragma unroll 1
        for (int o = 0; o < NQ / 16; ++o) {
            const uint32_t xo = xlane + (uint32_t)o * 256u;
            const uint32_t xwrap = o + 1 < NQ / 16 ? xo + 256u : xlane;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t xn = q < 15 ? (xo ^ (uint32_t)((q + 1) << 4)) : xwrap;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) b[(q + 1) & 1][mt] = *reinterpret_cast<const floatx4*>(X + xn + mt * 32 * PITCH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int slot = (q * NF + f) % RING;
                    const floatx4 a = ring[slot];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q & 1][mt][s], acc[f][mt], 0, 0, 0);
                    ring[slot] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wrs, l16, wnext + (uint32_t)slot * 1024u, 0));
                    if (slot == RING - 1) {
                        wnext += RING * 1024u;
                        if (wnext >= WTOT) wnext = 0;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: lane (m, h = kk) holds channels cb .. cb + 15 of tile f for pixel 32 mt + m
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int cb = CW * wave + 32 * f + 16 * kk;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + cb + 4 * r4);
                    floatx4 o;
                    o.x = acc[f][mt][4 * r4] + bv.x; o.y = acc[f][mt][4 * r4 + 1] + bv.y; o.z = acc[f][mt][4 * r4 + 2] + bv.z; o.w = acc[f][mt][4 * r4 + 3] + bv.w;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), cr, (uint32_t)(32 * mt + m) * N * 4 + (uint32_t)(cb + 4 * r4) * 4u, 0, 0);
                }
        }
        if (tile + gridDim.x < ntiles) block_sync();
    }
}

template <int K, int N>
void run(int64_t M, int grid, int kreps = 1) {
    constexpr int CW = N / NW, NF = CW / 32, NQ = K / 8, FRAGS = NQ * NF;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hWf((size_t)NW * FRAGS * 256);
    srand(1);
    for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hW) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
    for (int w = 0; w < NW; ++w)
        for (int q = 0; q < NQ; ++q)
            for (int f = 0; f < NF; ++f)
                for (int l = 0; l < 64; ++l) {
                    const int rho = l & 31, kk = l >> 5, cc = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
                    const int n = w * CW + 32 * f + cc, k0 = kk * (K / 2) + 4 * q;
                    for (int e = 0; e < 4; ++e) hWf[(((size_t)w * FRAGS + (size_t)q * NF + f) * 64 + l) * 4 + e] = hW[(size_t)n * K + k0 + e];
                }
    float *A, *Wf, *b, *C;
    (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&Wf, hWf.size() * 4); (void)hipMalloc(&b, N * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(Wf, hWf.data(), hWf.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) ws_gemm_kernel<K, N><<<grid, 512>>>(A, Wf, b, C, M, kreps);
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) ws_gemm_kernel<K, N><<<grid, 512>>>(A, Wf, b, C, M, kreps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 4000; ++t) {
        const int64_t mi = (int64_t)(rand() % M);
        const int n = rand() % N;
        double ref = hb[n];
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)mi * K + k] * hW[(size_t)n * K + k];
        if (kreps == 1) worst = fmax(worst, fabs(ref - hC[(size_t)mi * N + n]));
    }
    const double gf = 2.0 * M * N * K / 1e9 * kreps;
    printf("M %lld K %d N %d grid %d k-loop x%d: %8.1f us  %6.1f TF/s  (%.3f of 157.3)   max |err| vs fp64 %.2e\n", (long long)M, K, N, grid, kreps, ms * 1e3, gf / ms, gf / ms / 157.3, worst);
    fflush(stdout);
    (void)hipFree(A); (void)hipFree(Wf); (void)hipFree(b); (void)hipFree(C);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;   // one configuration (for the PMC passes: tools/pmc_cmd.sh groups launches by kernel name and grid)
    if (only < 0 || only == 0) run<512, 512>(32768, 256);
    if (only < 0) run<512, 512>(32768, 512);
    if (only < 0 || only == 1) run<512, 1024>(32768, 256);
    if (only < 0) run<512, 1024>(32768, 512);
    if (only < 0) run<512, 512>(65536, 256);
    if (only < 0 || only == 2) run<512, 512>(16384, 256, 16);   // one tile per CU, its k-loop 16 times: the loop alone
    if (only < 0 || only == 3) run<512, 1024>(16384, 256, 8);
    return 0;
}
