// LDS-DMA and 32x32 MFMA helpers shared by the prefill kernels (prefill_attention.hip, prefill_score.hip).
#pragma once

#include "svk_common.hpp"

namespace svk {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int pa_u32x4_t;

// the value of lane ^ 32 (the other half of the 32x32 accumulator's row split): v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second
__device__ __forceinline__ float lane_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// Four 16-byte-per-lane LDS-DMA loads (HBM/L2 -> LDS, wave-uniform base + 16 * lane) to the consecutive KiBs at
// `lds_addr` behind ONE M0 write.  Issued from inline asm: with the builtin, hipcc drains vmcnt(0) in front of every
// ds_read of the written region, so the caller counts `s_waitcnt vmcnt` itself.  M0 is saved and restored inside the
// statement (it is compiler-reserved).  The instruction offset moves BOTH the LDS destination and the global source
// (tools/probe_dma_offset.hip), so the caller's addresses carry 3072 - 1024 i and, in the 32-bit form, `base` is the
// tensor base - 3072.
__device__ __forceinline__ void pa_dma4x16_off32(const uint32_t (&voff)[4], const char* base, uint32_t lds_addr) {
  uint32_t keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void pa_dma4x16(const char* const (&src)[4], uint32_t lds_addr) {
  uint32_t keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
               "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "s"(lds_addr) : "memory");
}

}  // namespace svk
