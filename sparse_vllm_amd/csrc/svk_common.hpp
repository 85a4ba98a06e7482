// Shared helpers for the gfx950 kernels behind include/svk.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "svk.h"

namespace svk {

void set_error(const char* fmt, ...);

#define SVK_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      ::svk::set_error(__VA_ARGS__);      \
      return (code);                      \
    }                                     \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return SVK_ERR_LAUNCH;
  }
  return SVK_OK;
}

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 bit pattern (finite inputs; NaN quieted)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_round(float f) {
  return __builtin_bit_cast(float, f32_to_bf16_bits(f) << 16);
}

// x*y rounded on its own (never contracted into an fma with a following add): ROCm's __fmul_rn is a
// plain `x * y` and still fuses under -ffp-contract=fast, so the contraction is switched off locally.
__device__ __forceinline__ float mul_rn(float x, float y) {
#pragma clang fp contract(off)
  return x * y;
}
__device__ __forceinline__ float add_rn(float x, float y) {
#pragma clang fp contract(off)
  return x + y;
}

// Lane id recomputed on the spot (2 VALU ops).  `volatile` keeps the compiler from hoisting it
// out of a loop and then spilling it when the loop body is at its VGPR budget.
__device__ __forceinline__ int lane_id_fresh() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}

// DPP row rotate inside each 16-lane row: returns x from lane (l - n) mod 16 of the row.
template <int N>
__device__ __forceinline__ float row_ror(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}
// max(x, x rotated by N inside the 16-lane row) as ONE instruction.  The compiler's form of fmaxf(x, row_ror(x)) is
// v_mov 0 / v_mov_dpp / v_max (canonicalise) / v_max: the DPP move cannot be folded because the `old` operand (0) is
// not the identity of max, and fmaxf quiets NaNs.  `s_nop 1` covers the 2 wait states a DPP read needs after the VALU
// write of its source (inline asm is opaque to the hazard recogniser).
#define SVK_ROW_ROR_MAX(N_)                                                                                       \
  __device__ __forceinline__ float row_ror_max##N_(float x) {                                                     \
    float y;                                                                                                      \
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:" #N_ " row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));   \
    return y;                                                                                                     \
  }
SVK_ROW_ROR_MAX(8) SVK_ROW_ROR_MAX(4) SVK_ROW_ROR_MAX(2) SVK_ROW_ROR_MAX(1)
#undef SVK_ROW_ROR_MAX
// plain v_max_f32 (no NaN-quieting canonicalisation of the operands; callers pass finite values or -inf)
__device__ __forceinline__ float vmax(float a, float b) {
  float y;
  asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
  return y;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float y;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c));
  return y;
}
__device__ __forceinline__ float row16_allmax(float x) {
  return row_ror_max1(row_ror_max2(row_ror_max4(row_ror_max8(x))));
}
__device__ __forceinline__ float row16_allsum(float x) {
  x += row_ror<8>(x);
  x += row_ror<4>(x);
  x += row_ror<2>(x);
  x += row_ror<1>(x);
  return x;
}

__device__ __forceinline__ float wave_allmax(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float wave_allsum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}


// Workgroup -> (row lane, range) with the expensive ranges first.  The first range of a row holds the raw sink tokens
// and the last one the raw tail and the ragged end: their per-token tiles cost 50-100 us of dependent loads and used to
// start LAST (x-major dispatch), alone on the critical path (B = 4 x 256k: 357 us against 241 us without raw tokens).
// Dispatched first they run under the bulk of the grid.  Any bijection is valid: the mapping only changes start order.
__device__ __forceinline__ void kivi_wg_to_range(int& b, int& blk) {
  const int nblk = gridDim.x, B = gridDim.y;
  const int id = blockIdx.x + nblk * blockIdx.y;
  if (nblk < 3) { b = blockIdx.y; blk = blockIdx.x; return; }
  if (id < B) { b = id; blk = nblk - 1; }
  else if (id < 2 * B) { b = id - B; blk = 0; }
  else { const int r = id - 2 * B; b = r / (nblk - 2); blk = 1 + r % (nblk - 2); }
}

// the same with `n_extra` trailing block indices per row that are dispatched before everything else (the wide KIVI
// kernel's latency-bound raw / ragged pieces)
__device__ __forceinline__ void kivi_wg_to_range_extra(int& b, int& blk, int n_extra) {
  if (n_extra <= 0) { kivi_wg_to_range(b, blk); return; }
  const int nblk = gridDim.x, B = gridDim.y, nreg = nblk - n_extra;
  const int id = blockIdx.x + nblk * blockIdx.y;
  if (id < B * n_extra) { b = id / n_extra; blk = nreg + id % n_extra; }
  else { const int r = id - B * n_extra; b = r / nreg; blk = r % nreg; }
}

}  // namespace svk
