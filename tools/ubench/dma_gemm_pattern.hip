// LDS-DMA fill rate in the access patterns of the 256 x 256-tile bf16 GEMMs (gfx950): every block (512 threads, one per CU) streams
// k-tiles of [ROWS rows][128 B] out of a row-major panel (row stride = STRIDE bytes) into an LDS ring, DEPTH k-tiles in flight, one
// barrier per k-tile like the GEMM loop.  Knobs: how many blocks share a panel (the A operand of an NT GEMM is shared by the N / 256
// column tiles of a row panel, the weights by every block), the row stride, contiguous (pre-tiled) k-tiles.
//   hipcc --offload-arch=gfx950 -O3 -I dcpt_amd/csrc tools/ubench/dma_gemm_pattern.hip -o tools/ubench/dma_gemm_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bufops.h"

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0: strided rows (row r of k-tile kt at r * stride + kt * 128); mode 1: contiguous k-tiles (tile kt at kt * rows * 128)
template <int DEPTH, int ROWS>
__global__ __launch_bounds__(512) void fill_kernel(const unsigned char* __restrict__ src, int nkt, int stride, int share, size_t panel_bytes, int mode,
                                                   int swz, float* out) {
    constexpr int TILE = ROWS * 128;
    constexpr int PER_WAVE = TILE / 8192;   // 1-KiB DMAs per wave and k-tile
    constexpr int SLOTS = 131072 / TILE;
    __shared__ __attribute__((aligned(16))) unsigned char ring[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int lin = ((bid & 7) * (gridDim.x >> 3)) + (bid >> 3);   // XCD-contiguous logical id (gridDim.x % 8 == 0)
    const i32x4 rs = make_rsrc_dma(src + (size_t)(lin / share) * panel_bytes);
    const uint32_t lds0 = lds_addr(reinterpret_cast<const float*>(ring)) + wave * 1024;
    uint32_t voff[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int row = 8 * wave + (lane >> 3) + 64 * i;
        const int ch = swz ? ((lane & 7) ^ ((row >> 1) & 7)) : (lane & 7);
        voff[i] = mode == 0 ? (uint32_t)row * (uint32_t)stride + ch * 16 : (uint32_t)row * 128u + ch * 16;
    }
    const uint32_t kstep = mode == 0 ? 128u : (uint32_t)TILE;
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) dma16(rs, lds0 + (kt % SLOTS) * TILE + i * 8192, voff[i], (uint32_t)kt * kstep);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d);
    for (int kt = 0; kt < nkt; ++kt) {
        wait_vm<PER_WAVE*(DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        issue(kt + DEPTH < nkt ? kt + DEPTH : kt);   // (tail: re-read the last tiles, keeps the counts uniform)
    }
    wait_vm<0>();
    __syncthreads();
    if (tid == 0) out[bid] = reinterpret_cast<float*>(ring)[nkt & 1023];
}

template <int DEPTH, int ROWS>
void run(const char* what, const unsigned char* src, float* out, int nkt, int stride, int share, size_t panel_bytes, int mode, int swz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256, reps = 20;
    fill_kernel<DEPTH, ROWS><<<blocks, 512>>>(src, nkt, stride, share, panel_bytes, mode, swz, out);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) fill_kernel<DEPTH, ROWS><<<blocks, 512>>>(src, nkt, stride, share, panel_bytes, mode, swz, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)blocks * (nkt + DEPTH) * ROWS * 128.0;
    printf("%-44s rows %3d stride %5d share %3d depth %d swz %d nkt %3d: %7.2f us/launch  %6.2f TB/s = %5.1f B/ns/CU\n", what, ROWS, stride, share, DEPTH, swz,
           nkt, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256.0);
    fflush(stdout);
}

int main() {
    unsigned char* src;
    float* out;
    const size_t SZ = (size_t)256 << 20;
    (void)hipMalloc(&src, SZ);
    (void)hipMemset(src, 1, SZ);
    (void)hipMalloc(&out, 4096 * 4);
    // A-like: 256-row panels of an [M][K] bf16 matrix, K = 512 / 1024 (row stride 1024 / 2048 B), nkt = K / 64, shared by 1 / 2 / 4 blocks
    for (int K : {512, 1024}) {
        const int stride = K * 2, nkt = K / 64;
        const size_t panel = (size_t)256 * stride;
        for (int share : {1, 2, 4}) {
            run<2, 256>("A panel (HBM/L2), strided rows", src, out, nkt, stride, share, panel, 0, 1);
        }
        run<3, 256>("A panel, strided rows, depth 3", src, out, nkt, stride, 2, panel, 0, 1);
        run<2, 256>("A panel, strided rows, no swizzle", src, out, nkt, stride, 2, panel, 0, 0);
        run<2, 256>("A panel, PRE-TILED contiguous k-tiles", src, out, nkt, stride, 2, panel, 1, 1);
        // odd stride (pad 128 B per row)
        run<2, 256>("A panel, rows padded by 128 B", src, out, nkt, stride + 128, 2, (size_t)256 * (stride + 128), 0, 1);
        // B-like: every block reads the same [256][K] weight panel (or one of 2 / 4)
        run<2, 256>("B panel shared by all blocks, strided", src, out, nkt, stride, 256, panel, 0, 1);
        run<2, 256>("B panel shared by 128 blocks, strided", src, out, nkt, stride, 128, panel, 0, 1);
        run<2, 256>("B panel shared by all, PRE-TILED", src, out, nkt, stride, 256, panel, 1, 1);
        run<2, 256>("B panel shared by all, rows padded 128 B", src, out, nkt, stride + 128, 256, (size_t)256 * (stride + 128), 0, 1);
    }
    // long streams (steady state): 64 k-tiles
    run<2, 256>("A-like long stream, stride 8192", src, out, 64, 8192, 2, (size_t)256 * 8192, 0, 1);
    run<2, 256>("A-like long stream, pre-tiled", src, out, 64, 8192, 2, (size_t)256 * 8192, 1, 1);
    run<2, 256>("shared long stream, stride 8192", src, out, 64, 8192, 256, (size_t)256 * 8192, 0, 1);
    run<2, 256>("shared long stream, pre-tiled", src, out, 64, 8192, 256, (size_t)256 * 8192, 1, 1);
    run<2, 512>("A+B-like: 512 rows per k-tile, stride 1024, share 2", src, out, 8, 1024, 2, (size_t)512 * 1024, 0, 1);
    return 0;
}
