// What does a streaming kernel get from HBM on this part as a function of (a) read streams per written stream, (b) bytes per stream (inside /
// beyond the 256 MB Infinity Cache), (c) blocks and loads in flight?  The bandwidth kernels of the bf16 path (LayerNorm backward: 3 reads + 1
// write) run at 5 TB/s at B = 32 and 3 TB/s at B = 64: which of the three is it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_mix.hip -o tools/ubench/stream_mix && tools/ubench/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int UNROLL>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                  float4* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + u * stride;
            float4 x = j < n ? a[j] : make_float4(0, 0, 0, 0);
            if (R > 1 && j < n) { const float4 y = b[j]; x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; }
            if (R > 2 && j < n) { const float4 y = c[j]; x.x *= y.x; x.y *= y.y; x.z *= y.z; x.w *= y.w; }
            v[u] = x;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + u * stride;
            if (j < n) o[j] = v[u];
        }
    }
}

template <int R, int UNROLL>
void run(size_t mb, int blocks, const std::vector<float4*>& bufs, int sets) {
    const size_t n = mb * 1024 * 1024 / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 12;
    for (int w = 0; w < 2; ++w) mix_kernel<R, UNROLL><<<blocks, 256>>>(bufs[0], bufs[1], bufs[2], bufs[3], n);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
        const int s = (r % sets) * 4;
        mix_kernel<R, UNROLL><<<blocks, 256>>>(bufs[s], bufs[s + 1], bufs[s + 2], bufs[s + 3], n);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("reads %d  %4zu MB/stream  blocks %6d  unroll %d  sets %d: %7.1f us  %5.2f TB/s\n", R, mb, blocks, UNROLL, sets, us,
           (double)(R + 1) * mb * 1048576.0 / us / 1e6);
}

int main() {
    const int SETS = 3;
    std::vector<float4*> bufs;
    for (int i = 0; i < 4 * SETS; ++i) {
        float4* p; hipMalloc(&p, (size_t)256 << 20); hipMemset(p, 0, (size_t)256 << 20); bufs.push_back(p);
    }
    for (size_t mb : {32, 64, 128, 256}) {
        for (int sets : {1, 3}) {
            run<1, 1>(mb, (int)(mb * 1024 * 1024 / 16 / 256), bufs, sets);
            run<3, 1>(mb, (int)(mb * 1024 * 1024 / 16 / 256), bufs, sets);
            run<3, 1>(mb, 2048, bufs, sets);
            run<3, 4>(mb, 2048, bufs, sets);
            run<3, 4>(mb, 4096, bufs, sets);
            run<1, 4>(mb, 2048, bufs, sets);
        }
    }
    return 0;
}
