// Shared device/host helpers for the dcpt_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define DCPT_OK 0
#define DCPT_ERR_ARG 1
#define DCPT_ERR_WS 2
#define DCPT_ERR_HIP 3

void dcpt_set_error(const char* fmt, ...);

#define DCPT_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            dcpt_set_error(__VA_ARGS__);     \
            return DCPT_ERR_ARG;             \
        }                                    \
    } while (0)

#define DCPT_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            dcpt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return DCPT_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

#define DCPT_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != DCPT_OK) return rc__; \
    } while (0)

// Tuning switches (A/B experiments on the GPU box) exist only in diagnostic builds: tools/build_variant.sh compiles with -DDCPT_TUNING and
// then DCPT_* environment variables override the defaults below; the product library reads NO environment variables -- every switch is
// its measured default (DESIGN.md lists what each one was for).
#include <stdlib.h>
static inline int dcpt_tuning(const char* name, int dflt) {
#ifdef DCPT_TUNING
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-owned workspace (the caller -- PyTorch -- owns every buffer).
struct WsAlloc {
    char* base;
    size_t cap;
    size_t off;
    bool ok;
    WsAlloc(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0), ok(true) {}
    template <typename T>
    T* get(size_t n) {
        size_t bytes = align_up(n * sizeof(T), 256);
        if (off + bytes > cap) {
            ok = false;
            off += bytes;
            return nullptr;
        }
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};

#ifdef __HIPCC__
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4_sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Sum over `width` consecutive lanes (width a power of two <= 64), result in every lane of the group, bit-identical
// across the group.  Done with DPP lane permutes and v_readlane only: the usual __shfl_xor lowers to ds_bpermute /
// ds_swizzle, i.e. goes through the LDS pipeline -- where it queues behind the ds_reads of a co-resident MFMA GEMM
// (the LayerNorm kernels ran 4-5x slower next to the weight-gradient GEMMs of the side stream because of that).
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum(float v, int width) {
    if (width >= 2) v += dpp_perm<0xB1>(v);    // quad_perm [1,0,3,2]   (lane ^ 1)
    if (width >= 4) v += dpp_perm<0x4E>(v);    // quad_perm [2,3,0,1]   (lane ^ 2)
    if (width >= 8) v += dpp_perm<0x141>(v);   // row_half_mirror: the other quad of the 8-lane half row
    if (width >= 16) v += dpp_perm<0x140>(v);  // row_mirror: the other half of the 16-lane row
    if (width >= 32) {
        const int vi = __builtin_bit_cast(int, v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
        if (width == 32) v = ((threadIdx.x & 32) == 0) ? r0 + r1 : r2 + r3;
        else v = (r0 + r1) + (r2 + r3);
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, 64); }
// maximum over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_perm<0xB1>(v));
    v = fmaxf(v, dpp_perm<0x4E>(v));
    v = fmaxf(v, dpp_perm<0x141>(v));
    v = fmaxf(v, dpp_perm<0x140>(v));
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// XCD-aware, bijective remap of a 1-D block id: blocks that land on the same XCD (bid % 8 by
// observed dispatch) get a contiguous range of logical ids, so neighbouring tiles share that
// XCD's private L2.  Only a speed choice; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
// The same for a BATCHED launch (grid.y = independent problems): the hardware deals the blocks of a 2-D grid to the XCDs in the linear order
// x + y * gridDim.x, so a remap of blockIdx.x alone leaves the blocks of one problem on gridDim.x % 8 ... different XCDs -- with 8 blocks per
// image (conv3 with per-image weights at level 3) every block of an image on its own XCD, the activations read 2 x and the image's weights
// 4 x over the fabric (168 MB for 84: profiles/r5/nt_batched_xcd/).  Over the whole grid, a problem's blocks are neighbours on one XCD.
__device__ __forceinline__ void xcd_remap_batched(int& lin, int& batch) {
    const int gx = (int)gridDim.x;
    if (gridDim.y == 1) {
        lin = xcd_remap((int)blockIdx.x, gx);
        batch = 0;
        return;
    }
    const int lg = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), gx * (int)gridDim.y);
    batch = lg / gx;
    lin = lg - batch * gx;
}
#endif
