// Workgroup dispatch cost (gfx950): duration of a launch of N trivially short 256-thread workgroups that declare 64 KB / 32 KB / 0 KB of
// LDS (2 / 4 / 8 resident per CU), measured with events around 20 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_rate.hip -o tools/ubench/dispatch_rate && tools/ubench/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_KB>
__global__ __launch_bounds__(256) void k(float* out, int spin) {
    __shared__ float buf[LDS_KB > 0 ? LDS_KB * 256 : 1];
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (LDS_KB > 0) buf[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0 && a == 12345.f) out[blockIdx.x] = (LDS_KB > 0 ? buf[1] : a);
}
template <int LDS_KB>
void run(float* d, int blocks, int spin) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<LDS_KB><<<blocks, 256>>>(d, spin);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<LDS_KB><<<blocks, 256>>>(d, spin);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %2d KB  blocks %5d  spin %5d: %7.2f us per launch\n", LDS_KB, blocks, spin, ms * 1e3 / 20); fflush(stdout);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 20);
    for (int spin : {0, 2000}) for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) { run<64>(d, blocks, spin); run<32>(d, blocks, spin); run<0>(d, blocks, spin); }
    return 0;
}
