// Register canary: every lane keeps NR VGPRs with known values alive for a while and then checks them.  A mismatch can only come
// from outside the wave.  out[0] = number of mismatching (lane, register) pairs; out[1 + k] = first few (block, lane, reg, got) records.
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int NR = 66;
__device__ __forceinline__ uint32_t val(uint32_t t, uint32_t i) { return (t * 2654435761u) ^ (i * 40503u + 0x9e3779b9u); }
__global__ __launch_bounds__(256) void canary_kernel(uint32_t* out, int iters) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        r[i] = val(t, i);
        asm volatile("" : "+v"(r[i]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (r[i] != val(t, i)) {
            const uint32_t k = atomicAdd(out, 1u);
            if (k < 64) {
                out[1 + 4 * k] = blockIdx.x;
                out[2 + 4 * k] = threadIdx.x;
                out[3 + 4 * k] = i;
                out[4 + 4 * k] = r[i] ^ val(t, i);
            }
        }
    }
}
extern "C" int canary_launch(uint32_t* out, int blocks, int iters, void* stream) {
    canary_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(out, iters);
    return (int)hipGetLastError();
}
