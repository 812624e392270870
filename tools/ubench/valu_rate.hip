// VALU issue-rate probe (gfx950): cycles per wave-instruction of v_fma_f32 / v_pk_fma_f32 / v_pk_mul_f32 with 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = i * 0.5f; }
    v2 m; m.x = 1.0001f; m.y = 0.9999f;
    unsigned long long msk = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters * 0x9E3779B97F4A7C15ull;
    unsigned ai[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ai[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(m.x), "v"(m.y));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(m));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(m.x));
                if (MODE == 5) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (MODE == 6) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i].x) : "v"(m.x), "v"(a[i].y), "s"(msk));
                if (MODE == 7) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(a[i].x) : "v"(a[i].y), "s"(msk));
                if (MODE == 8) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ai[i]) : "v"(ai[(i + 3) & 15]));
                if (MODE == 9) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
            }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y + (float)ai[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0) * 1e-30f;
}
template <int MODE>
void run(const char* name, float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 64;   // per wave
    printf("%-14s blocks=%4d: %.3f ms -> %.2f ns per wave-instruction (x2.4 GHz = %.2f cycles)\n", name, blocks, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int blocks : {256, 512}) {
        run<0>("v_fma_f32", d, blocks); run<1>("v_pk_fma_f32", d, blocks); run<2>("v_pk_mul_f32", d, blocks); run<3>("v_pk_add_f32", d, blocks);
        run<4>("v_cndmask_b32", d, blocks); run<5>("v_mov_b64", d, blocks); run<6>("cndmask_e64 vv", d, blocks); run<7>("cndmask_e64 0v", d, blocks);
        run<8>("v_add_u32", d, blocks); run<9>("v_mul_f32", d, blocks);
    }
    return 0;
}
