// LDS-DMA fill-rate probe (gfx950): how fast can a CU pull L2-resident tiles into LDS with buffer_load_dwordx4 ... lds, as a function of
// the number of 1-KiB wave-DMAs in flight per wave (DEPTH) and blocks per CU?  Each block streams a private 256-KiB window (L2-resident
// after the first pass) through a ring of LDS slots; nothing reads the LDS (pure fill).
//   hipcc --offload-arch=gfx950 -O3 -I dcpt_amd/csrc tools/ubench/dma_fill.hip -o tools/ubench/dma_fill && tools/ubench/dma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bufops.h"

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int LDS_KB, int WIN_KB>
__global__ __launch_bounds__(256) void fill_kernel(const float* __restrict__ src, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[LDS_KB * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WIN = WIN_KB * 1024;   // bytes per block window
    const i32x4 rs = make_rsrc_dma(reinterpret_cast<const unsigned char*>(src) + (size_t)(blockIdx.x % 1024) * 256 * 1024, WIN);
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(ring)) + wave * 1024;
    constexpr int SLOTS = LDS_KB / 4;   // 4 KiB (one DMA per wave) per slot
    uint32_t off = (uint32_t)(wave * 1024 + lane * 16);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        dma16(rs, lds0 + (d % SLOTS) * 4096, off, 0);
        off = (off + 4096) & (WIN - 1);
    }
    for (int it = 0; it < iters; ++it) {
        wait_vm<DEPTH - 1>();
        dma16(rs, lds0 + ((it + DEPTH) % SLOTS) * 4096, off, 0);
        off = (off + 4096) & (WIN - 1);
    }
    wait_vm<0>();
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(ring)[iters & 1023];
}

template <int DEPTH, int LDS_KB, int WIN_KB = 256>
void run(const float* src, float* out, int blocks) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fill_kernel<DEPTH, LDS_KB, WIN_KB><<<blocks, 256>>>(src, 64, out);
    (void)hipEventRecord(e0);
    fill_kernel<DEPTH, LDS_KB, WIN_KB><<<blocks, 256>>>(src, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * (iters + DEPTH) * 4096.0;
    printf("window %3d KiB  depth %2d  LDS %3d KiB/block  blocks %4d (%.0f per CU): %7.3f ms  %6.2f TB/s  = %5.1f B/ns/CU  (per block: one 4-KiB slot every %.0f ns)\n", WIN_KB, DEPTH, LDS_KB,
           blocks, blocks / 256.0, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256.0, ms * 1e6 / (iters + DEPTH));
    fflush(stdout);
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, (size_t)1024 * 256 * 1024);
    (void)hipMemset(src, 0, (size_t)1024 * 256 * 1024);
    (void)hipMalloc(&out, 4096 * 4);
    for (int blocks : {256, 512, 1024}) {
        run<1, 64>(src, out, blocks);
        run<2, 64>(src, out, blocks);
        run<4, 64>(src, out, blocks);
        run<8, 64>(src, out, blocks);
        run<16, 64>(src, out, blocks);
    }
    run<8, 32>(src, out, 1024);
    run<16, 32>(src, out, 2048);
    for (int blocks : {256, 512}) {   // 64-KiB windows: 16 / 32 MB in total, resident in the 8 x 4 MB of L2
        run<1, 64, 64>(src, out, blocks);
        run<2, 64, 64>(src, out, blocks);
        run<4, 64, 64>(src, out, blocks);
        run<8, 64, 64>(src, out, blocks);
    }
    for (int blocks : {256, 512}) {   // 16-KiB windows
        run<2, 64, 16>(src, out, blocks);
        run<4, 64, 16>(src, out, blocks);
        run<8, 64, 16>(src, out, blocks);
    }
    return 0;
}
