// Buffer-resource memory access helpers (gfx950): hardware range-checked loads/stores and LDS-DMA.
//
// A block opens a WINDOW at the first element it can touch (64-bit, wave-uniform base in SGPRs) and addresses
// everything with 32-bit byte offsets relative to it.  Elements that do not exist (rows past M, columns past
// the width, zero-padding taps) get an offset above the window size (ROW_SENT / COL_SENT), which the hardware
// range check turns into "load returns 0 / store is dropped": no exec-mask branches around memory
// instructions, so the compiler is free to schedule them.
#pragma once
#include "dcpt_common.h"

#ifndef DCPT_ST_AUX
#define DCPT_ST_AUX 0   // cache policy of the GEMM output stores (experiments: 2 = nt, 16 = sc1 write-through)
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t WIN_BYTES = 0x3FFFFFFFu;  // every real offset of a tile is far below this (checked at launch)
constexpr uint32_t ROW_SENT = 0x80000000u;   // row past M
constexpr uint32_t COL_SENT = 0x40000000u;   // column past ncols / padding tap

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes = WIN_BYTES) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    // (bit-cast the whole vector: __builtin_bit_cast of a single ext-vector element reads element 0 on this hipcc)
    const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_st4(rsrc_t r, uint32_t voff, float4 f) {
    floatx4 v;
    v.x = f.x;
    v.y = f.y;
    v.z = f.z;
    v.w = f.w;
#ifdef DCPT_ABL_NOSTORE   // ablation builds (tools/build_variant.sh): the store is issued but dropped by the range check
    voff |= ROW_SENT;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, DCPT_ST_AUX);
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void buf_st1(rsrc_t r, uint32_t voff, float f) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), r, voff, 0, 0);
}
// LDS-DMA: 16 B per lane straight into LDS at (wave-uniform LDS byte address) + lane * 16, no VGPR staging.
// Issued as inline asm on purpose: hipcc would otherwise order every later ds_read behind the pending LDS
// write with an s_waitcnt vmcnt(0) right after the issue, exposing the whole HBM/L2 latency.  hipcc does not
// count these loads, so the kernels wait for them themselves (dma_wait_all) before the barrier that
// publishes the tile.  The resource is passed as four SGPR words (same content as rsrc_t).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc_dma(const void* base, uint32_t bytes = WIN_BYTES) {
    const uint64_t a = (uint64_t)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));  // stride 0, 48-bit address
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)((__attribute__((address_space(3))) const void*)p));
}
__device__ __forceinline__ void dma16(i32x4 rs, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds_byte), "s"(soff)
                 : "memory");
}
// 4 B per lane: (wave-uniform LDS byte address) + lane * 4
__device__ __forceinline__ void dma4(i32x4 rs, uint32_t lds_byte, uint32_t voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds_byte)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

