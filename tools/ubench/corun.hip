// standalone: stream B loops an MFMA-heavy kernel (no memory traffic), stream A times simple streaming kernels
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters, int lds_kb) {
    extern __shared__ float sm[];
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    if (lds_kb < 0) sm[threadIdx.x] = 1.f;
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void valu_spin(float* out, int iters) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); }
    out[blockIdx.x * 256 + threadIdx.x] = a + c + d;
}
template <int PRIO>
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ x, float4* __restrict__ y, size_t n) {
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ x, float* __restrict__ y, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) y[0] = s;
}
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
template <class F> float time_us(hipStream_t s, F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipEventRecord(e0, s); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    hipStream_t sA, sB; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    size_t n4 = 8u << 20;   // 128 MB
    float4 *x, *y; float* o; CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&o, 64 << 20));
    CK(hipMemset(x, 0, n4 * 16));
    auto cp = [&]() { copy4<0><<<8192, 256, 0, sA>>>(x, y, n4); };
    auto rd = [&]() { copy4<3><<<8192, 256, 0, sA>>>(x, y, n4); };     // same copy at wave priority 3
    auto vs = [&]() { valu_spin<<<2048, 256, 0, sA>>>(o, 2000); };
    printf("alone: copy %.1f us  copy(prio 3) %.1f us  valu %.1f us\n", time_us(sA, cp, 20), time_us(sA, rd, 20), time_us(sA, vs, 20));
    struct Cfg { const char* name; int blocks; int lds; } cfgs[] = {{"mfma 512 blk, 64 KB LDS (2/CU)", 512, 65536}, {"mfma 512 blk, no LDS", 512, 0},
                                                                  {"mfma 256 blk, 64 KB LDS (1/CU)", 256, 65536}, {"mfma 1024 blk no LDS (4/CU)", 1024, 0}};
    for (auto& c : cfgs) {
        float t[3];
        for (int k = 0; k < 3; ++k) {
            for (int i = 0; i < 60; ++i) mfma_spin<<<c.blocks, 256, c.lds, sB>>>(o + (1 << 20), 2000, 1);   // ~60 x 0.43 ms queued
            t[k] = (k == 0) ? time_us(sA, cp, 20) : (k == 1) ? time_us(sA, rd, 20) : time_us(sA, vs, 20);
            CK(hipDeviceSynchronize());
        }
        printf("next to %-34s: copy %.1f us  copy(prio 3) %.1f us  valu %.1f us\n", c.name, t[0], t[1], t[2]);
    }
    float tm = time_us(sB, [&]() { mfma_spin<<<512, 256, 65536, sB>>>(o + (1 << 20), 2000, 1); }, 5);
    printf("mfma_spin alone %.1f us\n", tm);
    return 0;
}
