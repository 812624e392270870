// Do blocks of ONE XCD that stream the same bytes share them through that XCD's L2 when they run in lockstep, or does every block of a
// sibling set fetch its own copy (a miss that is already outstanding is not merged)?  The grouped weight-gradient GEMM
// (gemm_tn_bf16_256.hip) reads 491 MB over the fabric for 320 MB of operands although the tiles of one pixel range sit on one XCD.
//   G sibling blocks per XCD read the same region (one region per XCD) with 16-byte loads, UNROLL loads in flight per thread; sibling j
//   starts `delay` x j microseconds late.  Run under rocprofv3 --pmc FETCH_SIZE (x 2 on gfx950): one region's bytes x 8 if the L2 shares.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_share.hip -o tools/ubench/l2_share && tools/ubench/l2_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int UNROLL>
__global__ __launch_bounds__(256) void share_kernel(const uint4* __restrict__ src, uint4* __restrict__ sink, size_t region16, int G, int delay_us,
                                                    int sets) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;   // (observed dispatch: block b runs on XCD b % 8)
    const int set = slot / G, j = slot % G;                    // `sets` sibling sets per XCD, each with its own region
    if (delay_us > 0 && j > 0) {
        const long long t0 = wall_clock64();                   // 100 MHz
        while (wall_clock64() - t0 < (long long)delay_us * j * 100) __builtin_amdgcn_s_sleep(8);
    }
    const uint4* p = src + (size_t)(xcd * sets + set) * region16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i < region16; i += 256 * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc.x ^= v[u].x;
            acc.y ^= v[u].y;
            acc.z ^= v[u].z;
            acc.w ^= v[u].w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const size_t region_mb = 8;
    const int max_sets = 8;
    const size_t region16 = region_mb * 1048576 / 16;
    uint4 *src, *sink;
    hipMalloc(&src, region_mb * 1048576 * 8 * max_sets);
    hipMalloc(&sink, 8 * 64 * 256 * 16);
    hipMemset(src, 1, region_mb * 1048576 * 8 * max_sets);
    struct Cfg { int G, sets, delay; } cfgs[] = {{1, 1, 0}, {4, 1, 0}, {4, 1, 2}, {4, 1, 10}, {8, 1, 0}, {8, 1, 2}, {4, 7, 0}, {4, 7, 2}, {4, 7, 10}, {8, 4, 0}, {8, 4, 2}};
    for (const Cfg& c : cfgs) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int blocks = 8 * c.G * c.sets;
        hipEventRecord(e0);
        share_kernel<8><<<blocks, 256>>>(src, sink, region16, c.G, c.delay, c.sets);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("G=%d sets/XCD=%d delay=%2d us: grid %4d  %8.1f us   unique bytes %zu MB, requested %zu MB\n", c.G, c.sets, c.delay, blocks, ms * 1e3,
               region_mb * 8 * c.sets, region_mb * 8 * c.sets * c.G);
    }
    return 0;
}
