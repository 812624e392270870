// HBM fill rate of the LDS-DMA path for two access patterns over a 1-GiB buffer (nothing is L2 / MALL resident):
//   contiguous: a block streams 16-KiB contiguous chunks (what a tile would be in a tile-blocked layout)
//   gemm:       a block reads [128 rows][128 B] k-tiles of a row-major [M][K] panel (row stride KB bytes), walking k then the next panel
//               -- the A operand of the bf16 NT GEMM (K = 512 bf16 -> 1024-byte rows)
// Each block keeps DEPTH k-tiles (16 KiB) in flight; 2 blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -I dcpt_amd/csrc tools/ubench/dma_pattern.hip -o tools/ubench/dma_pattern && tools/ubench/dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bufops.h"

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int GEMM>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned char* __restrict__ src, size_t total_bytes, int row_bytes, int tiles_per_block, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[(DEPTH + 1) * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(ring)) + wave * 1024;
    const int kt_per_panel = row_bytes / 128;
    const size_t panel_bytes = (size_t)128 * row_bytes;
    const size_t npanels = total_bytes / panel_bytes;
    auto issue = [&](int t, int slot) {
        // tile t of this block: panel = (t / kt_per_panel) * gridDim.x + blockIdx.x, k-tile = t % kt_per_panel
        size_t base;
        uint32_t off[4];
        if (GEMM) {
            const size_t panel = ((size_t)(t / kt_per_panel) * gridDim.x + blockIdx.x) % npanels;
            base = panel * panel_bytes + (size_t)(t % kt_per_panel) * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) off[i] = (uint32_t)(((i * 4 + wave) * 8 + (lane >> 3)) * row_bytes + (lane & 7) * 16);
        } else {
            const size_t chunk = ((size_t)t * gridDim.x + blockIdx.x) % (total_bytes / 16384);
            base = chunk * 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i) off[i] = (uint32_t)((i * 4 + wave) * 1024 + lane * 16);
        }
        const i32x4 rs = make_rsrc_dma(src + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rs, lds0 + slot * 16384 + i * 4096, off[i], 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    for (int t = 0; t < tiles_per_block; ++t) {
        wait_vm<(DEPTH - 1) * 4>();
        issue(t + DEPTH, (t + DEPTH) % (DEPTH + 1));
    }
    wait_vm<0>();
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = reinterpret_cast<float*>(ring)[tiles_per_block & 1023];
}

template <int DEPTH, int GEMM>
void run(const unsigned char* src, size_t total, int row_bytes, float* out) {
    const int blocks = 512, tiles = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fill_kernel<DEPTH, GEMM><<<blocks, 256>>>(src, total, row_bytes, 64, out);
    (void)hipEventRecord(e0);
    fill_kernel<DEPTH, GEMM><<<blocks, 256>>>(src, total, row_bytes, tiles, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * (tiles + DEPTH) * 16384.0;
    printf("%-10s row %5d B  depth %d (x2 blocks/CU = %3d KiB in flight per CU): %7.3f ms  %5.2f TB/s\n", GEMM ? "gemm" : "contiguous", row_bytes, DEPTH,
           2 * DEPTH * 16, ms, bytes / ms / 1e9);
    fflush(stdout);
}

int main() {
    unsigned char* src; float* out;
    const size_t total = (size_t)1 << 30;
    (void)hipMalloc(&src, total + (1 << 20));
    (void)hipMemset(src, 0, total);
    (void)hipMalloc(&out, 4096 * 4);
    run<1, 0>(src, total, 1024, out); run<2, 0>(src, total, 1024, out); run<3, 0>(src, total, 1024, out);
    for (int rb : {1024, 2048, 512}) { run<1, 1>(src, total, rb, out); run<2, 1>(src, total, rb, out); run<3, 1>(src, total, rb, out); }
    return 0;
}
