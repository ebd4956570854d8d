// Shared by the 32-point-step kernels (mmd_q32.hip, mmd_q32s.hip).
#pragma once
#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 q_f16x8;

// chunk swizzle of a step's LDS image (mmd_q32.hip, "LDS image of a step")
__device__ __forceinline__ int q32_swz(int q) { return (2 * (q & 1)) ^ ((q >> 1) & 1) ^ (3 * ((q >> 2) & 1)); }

}  // namespace
}  // namespace tvae

#define Q32_LDS(T, addr) (*reinterpret_cast<__attribute__((address_space(3))) T*>((unsigned)(addr)))
#define Q32_PIN() __builtin_amdgcn_sched_barrier(0)
