// Shared by the 32-point-step kernels (mmd_q32.hip, mmd_q32s.hip).
#pragma once
#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 q_f16x8;

// chunk swizzle of a step's LDS image (mmd_q32.hip, "LDS image of a step")
__device__ __forceinline__ int q32_swz(int q) { return (2 * (q & 1)) ^ ((q >> 1) & 1) ^ (3 * ((q >> 2) & 1)); }


// LDS-DMA of the 32-point-step kernels: M0 is set and LEFT (a reserved register hipcc sets itself right before any use of
// its own, and these kernels have none: no LDS-direct, no relative moves, no GWS; mmd_common.h's helpers save and restore
// it -- two more instructions per use on a wave that issues one instruction per four cycles), and up to four 1 KiB pieces
// go out under one setting: the instruction offset moves the LDS address AND the global address, so the caller's per-lane
// offsets carry the difference.
__device__ __forceinline__ void q32_dma_quad(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048\n\tglobal_load_lds_dwordx4 %3, %4 offset:3072"
                 : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void q32_dma_pair(const void* sbase, unsigned v0, unsigned v1, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                 : : "v"(v0), "v"(v1), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void q32_dma_dword(const void* sbase, unsigned v0, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(v0), "s"(sbase), "s"(lds_byte) : "memory");
}

}  // namespace
}  // namespace tvae

#define Q32_LDS(T, addr) (*reinterpret_cast<__attribute__((address_space(3))) T*>((unsigned)(addr)))
#define Q32_PIN() __builtin_amdgcn_sched_barrier(0)
