// One (token, kv head) key row of a DeltaKV sparse layer on its way into the attention view: optional RMS k-norm, then
// rotate-half RoPE, in the arithmetic of deltakv_materialize_sparse_view (kernels/triton/deltakv_kernels.py:3588-3693).
// D/16 consecutive lanes share a row: lane j of the group owns elements p = 8j .. 8j+7 and their rotate-half partners
// p + D/2 ..; the k-norm sum of squares is reduced over those lanes.  Shared by the view launch (deltakv_view.hip) and by
// the rotated store of the newest row inside the attention launch (decode_attention.hip) so that both write the same bits.
#pragma once

#include "svk_common.hpp"

namespace svk {

__device__ __forceinline__ void rope_unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
  f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}

__device__ __forceinline__ uint4 rope_pack8(const float (&f)[8]) {
  return make_uint4(f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16), f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16),
                    f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16), f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16));
}

__device__ __forceinline__ void rope_load8(const void* base, int64_t off, int dtype, float (&f)[8]) {
  if (dtype == SVK_DTYPE_F32) {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else if (dtype == SVK_DTYPE_BF16) {
    rope_unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off), f);
  } else {
    const _Float16* h = reinterpret_cast<const _Float16*>(base) + off;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
  }
}

// k-norm of the row's two pieces held by this lane (n1 / n2 receive the normalised values, or the raw ones without a
// weight).  Every lane of the D/16-lane group must call it (cross-lane sum).
template <int D>
__device__ __forceinline__ void rope_row_norm(const uint4& rk1, const uint4& rk2, const float* k_norm_weight, float k_norm_eps, int p,
                                              float (&n1)[8], float (&n2)[8]) {
  constexpr int HD2 = D / 2, LPH = HD2 / 8;
  float k1[8], k2[8];
  rope_unpack8(rk1, k1);
  rope_unpack8(rk2, k2);
#pragma unroll
  for (int e = 0; e < 8; ++e) { n1[e] = k1[e]; n2[e] = k2[e]; }
  if (k_norm_weight != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += k1[e] * k1[e] + k2[e] * k2[e];
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) ss += __shfl_xor(ss, off, 64);
    const float rstd = rsqrtf(ss / (float)D + k_norm_eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      n1[e] = k1[e] * rstd * k_norm_weight[p + e];
      n2[e] = k2[e] * rstd * k_norm_weight[p + HD2 + e];
    }
  }
}

// rotate-half at row `cos_row` of the [max_pos, D] cos | sin table: o1 = elements p.., o2 = elements p + D/2 ..
template <int D>
__device__ __forceinline__ void rope_row_rotate(const float (&n1)[8], const float (&n2)[8], const void* cos_sin, int64_t cos_row,
                                                int cos_dtype, int p, uint4& o1, uint4& o2) {
  constexpr int HD2 = D / 2;
  float c[8], s[8], r1[8], r2[8];
  rope_load8(cos_sin, cos_row + p, cos_dtype, c);
  rope_load8(cos_sin, cos_row + p + HD2, cos_dtype, s);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r1[e] = n1[e] * c[e] - n2[e] * s[e];
    r2[e] = n2[e] * c[e] + n1[e] * s[e];
  }
  o1 = rope_pack8(r1);
  o2 = rope_pack8(r2);
}

}  // namespace svk
