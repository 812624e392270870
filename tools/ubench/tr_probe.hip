// What does ds_read_b64_tr_b16 deliver?  LDS holds raw u16 = element index of a row-major [R][STRIDE] image; every lane passes the
// address of 4 contiguous elements; prints, per lane, the 4 values it receives.
//   hipcc --offload-arch=gfx950 -O2 -o tr_probe tools/ubench/tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 128;
__global__ void k(uint16_t* out, int mode) {
    __shared__ uint16_t sm[64 * STRIDE];
    for (int i = threadIdx.x; i < 64 * STRIDE; i += 64) sm[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    int row, col;
    if (mode == 0) { row = t >> 2; col = 16 * (g & 1) + 4 * (t & 3); row += 8 * (g >> 1); }   // lane t = 4r+q: row r, cols 4q..
    else { row = t & 3; col = 16 * (g & 1) + 4 * (t >> 2); row += 8 * (g >> 1); }              // lane t = r+4q
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sm + row * STRIDE + col));
    u16x4 u = __builtin_bit_cast(u16x4, v);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = u[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (value = row*%d+col -> printed as row:col)\n", mode, STRIDE);
        for (int l = 0; l < 64; ++l) {
            printf(" lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %2d:%-3d", h[l * 4 + j] / STRIDE, h[l * 4 + j] % STRIDE);
            if (l % 2) printf("\n");
        }
    }
    return 0;
}
