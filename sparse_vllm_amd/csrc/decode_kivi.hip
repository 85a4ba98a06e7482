// Decode stage 1 over KIVI-int4 blocks + raw tail (DeltaKV full-attention layers), gfx950.
// Same wave-per-KV-head / MFMA Q.K^T / VALU P.V structure as decode_attention.hip (un-pipelined form);
// K fragments of quantised tokens are dequantised per lane straight into the MFMA B-operand registers
// (per-channel scale/min: one 16-byte load per 8 channels; codes: one 4-byte word per channel, shared by
// the 8 tokens of a word), V words (8 codes = 8 head dims of one token) match the P.V lane layout 1:1.

#include <stdlib.h>

#include <type_traits>

#include "svk_common.hpp"
#include "decode_kivi_tile.hpp"


namespace svk {
namespace {

constexpr int kTile = 32;

template <int D, int G, bool KF32>
__global__ void __launch_bounds__(512) kivi_stage1_kernel(const SvkKiviDecodeStage1Args a) {
  constexpr int NC = D / 32, JQ = (G + 3) / 4, PH = JQ * 4, DC = D / 8, TQ = 64 / DC, NV = kTile / TQ;
  constexpr int WAVE_FLOATS = kTile * PH + 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads, GS = a.group_size;
  int b, blk;
  kivi_wg_to_range(b, blk);
  const int n = lane & 15, jq = lane >> 4, dc = lane % DC, tq = lane / DC;
  const int len = a.context_lens[b];
  const int start = blk * a.block_seq;
  const int end = min(len, start + a.block_seq);
  float* Pw = lds + w * WAVE_FLOATS;
  float* bc = Pw + kTile * PH;
  float* mid_o = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)blk * a.mid_o_stride_s;
  float* mid_lse = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + blk;
  if (end <= start) {
    for (int h = 0; h < G; ++h) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h;
      for (int d = lane; d < D; d += 64) o[d] = 0.f;
      if (lane == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = -INFINITY;
    }
    return;
  }
  bf16x8_t qa[NC];
  {
    const uint16_t* qp = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h + jq * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (n < G) t = *reinterpret_cast<const uint4*>(qp + c * 32);
      qa[c] = __builtin_bit_cast(bf16x8_t, t);
    }
  }
  const int row = a.req_indices[b];
  const int32_t* raw_map = a.raw_slots_map + (int64_t)row * a.map_stride;
  const int32_t* blk_map = a.kivi_block_slots_map + (int64_t)row * a.map_stride;
  const float sm_scale = rsqrtf((float)D);
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  float acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;

  // K fragment of token t for k-chunk c (8 channels c*32 + jq*8 ..), raw or dequantised; returns validity
  auto load_k = [&](int t, uint4 (&kr)[NC]) -> bool {
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[c] = make_uint4(0, 0, 0, 0);
    if (t >= end) return false;
    const int rs = raw_map[t];
    if (rs >= 0) {
      const uint16_t* kp = a.raw_k + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + jq * 8;
#pragma unroll
      for (int c = 0; c < NC; ++c) kr[c] = *reinterpret_cast<const uint4*>(kp + c * 32);
      return true;
    }
    const int bs = blk_map[t];
    if (bs < 0) return false;
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= GS) return false;
    const int64_t hb = (int64_t)bs * Hkv + w;
    const int shift = (lt & 7) * 4, widx = lt >> 3, wpd = GS / 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int d0 = c * 32 + jq * 8;
      float scf[8], mnf[8];
      if constexpr (KF32) {     // the reference manager keeps per-channel key scale/min in fp32 (deltakv_less_memory.py:1122-1130)
        const float* sp = reinterpret_cast<const float*>(a.key_scales) + hb * D + d0;
        const float* mp = reinterpret_cast<const float*>(a.key_mins) + hb * D + d0;
        const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
        const float4 m0 = *reinterpret_cast<const float4*>(mp), m1 = *reinterpret_cast<const float4*>(mp + 4);
        scf[0] = s0.x; scf[1] = s0.y; scf[2] = s0.z; scf[3] = s0.w; scf[4] = s1.x; scf[5] = s1.y; scf[6] = s1.z; scf[7] = s1.w;
        mnf[0] = m0.x; mnf[1] = m0.y; mnf[2] = m0.z; mnf[3] = m0.w; mnf[4] = m1.x; mnf[5] = m1.y; mnf[6] = m1.z; mnf[7] = m1.w;
      } else {
        const uint4 sc = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.key_scales) + hb * D + d0);
        const uint4 mn = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.key_mins) + hb * D + d0);
        const uint32_t scw[4] = {sc.x, sc.y, sc.z, sc.w}, mnw[4] = {mn.x, mn.y, mn.z, mn.w};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          scf[2 * e2] = bf16_lo(scw[e2]); scf[2 * e2 + 1] = bf16_hi(scw[e2]);
          mnf[2 * e2] = bf16_lo(mnw[e2]); mnf[2 * e2 + 1] = bf16_hi(mnw[e2]);
        }
      }
      uint32_t outw[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int32_t* cw = a.key_packed + (hb * D + d0 + e2 * 2) * wpd + widx;
        const float q0 = (float)(((uint32_t)cw[0] >> shift) & 15u), q1 = (float)(((uint32_t)cw[wpd] >> shift) & 15u);
        const float v0 = add_rn(mul_rn(q0, scf[2 * e2]), mnf[2 * e2]);
        const float v1 = add_rn(mul_rn(q1, scf[2 * e2 + 1]), mnf[2 * e2 + 1]);
        outw[e2] = f32_to_bf16_bits(v0) | (f32_to_bf16_bits(v1) << 16);
      }
      kr[c] = make_uint4(outw[0], outw[1], outw[2], outw[3]);
    }
    return true;
  };
  // V: 8 head dims dc*8.. of token t as floats (bf16-rounded); zero when invalid
  auto load_v = [&](int t, float (&vf)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) vf[e] = 0.f;
    if (t >= end) return;
    const int rs = raw_map[t];
    if (rs >= 0) {
      const uint4 vv = *reinterpret_cast<const uint4*>(a.raw_v + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + dc * 8);
      vf[0] = bf16_lo(vv.x); vf[1] = bf16_hi(vv.x); vf[2] = bf16_lo(vv.y); vf[3] = bf16_hi(vv.y);
      vf[4] = bf16_lo(vv.z); vf[5] = bf16_hi(vv.z); vf[6] = bf16_lo(vv.w); vf[7] = bf16_hi(vv.w);
      return;
    }
    const int bs = blk_map[t];
    if (bs < 0) return;
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= GS) return;
    const int64_t tb = ((int64_t)bs * Hkv + w) * GS + lt;
    const uint32_t word = (uint32_t)a.value_packed[tb * (D / 8) + dc];
    const int g = (dc * 8) / GS;
    const float sc = __builtin_bit_cast(float, (uint32_t)a.value_scales[tb * (D / GS) + g] << 16);
    const float mn = __builtin_bit_cast(float, (uint32_t)a.value_mins[tb * (D / GS) + g] << 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) vf[e] = bf16_round(add_rn(mul_rn((float)((word >> (e * 4)) & 15u), sc), mn));
  };

  for (int t0 = start; t0 < end; t0 += kTile) {
    uint4 kr[2][NC];
    bool tv[2];
    tv[0] = load_k(t0 + n, kr[0]);
    tv[1] = load_k(t0 + 16 + n, kr[1]);
    f32x4_t s[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      s[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NC; ++c)
        s[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[g][c]), s[g], 0, 0, 0);
    }
    if (a.attn_score != nullptr) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int h = jq * 4 + r;
          if (h < G && tv[g])
            a.attn_score[(int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + t0 + g * 16 + n] = s[g][r];
        }
    }
    float p[2][4], alpha[4];
    bool rescale = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool hv = (jq * 4 + r < G);
      const float x0 = (hv && tv[0]) ? s[0][r] * sm_scale : -INFINITY;
      const float x1 = (hv && tv[1]) ? s[1][r] * sm_scale : -INFINITY;
      const float tmax = row16_allmax(fmaxf(x0, x1));
      const float nm = fmaxf(m[r], tmax);
      if (hv && nm > -INFINITY) {          // a tile without any valid slot leaves the state untouched (:868-874)
        alpha[r] = __expf(m[r] - nm);
        p[0][r] = __expf(x0 - nm);
        p[1][r] = __expf(x1 - nm);
        rescale |= (nm != m[r]);
        m[r] = nm;
      } else {
        alpha[r] = 1.f; p[0][r] = 0.f; p[1][r] = 0.f;
      }
      l[r] = l[r] * alpha[r] + row16_allsum(p[0][r] + p[1][r]);
    }
    if (jq < JQ) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
        *reinterpret_cast<float4*>(Pw + (g * 16 + n) * PH + jq * 4) =
            make_float4(bf16_round(p[g][0]), bf16_round(p[g][1]), bf16_round(p[g][2]), bf16_round(p[g][3]));
    }
    const bool any_rescale = __any(rescale);
    if (any_rescale && n == 0 && jq < JQ) *reinterpret_cast<float4*>(bc + jq * 4) = make_float4(alpha[0], alpha[1], alpha[2], alpha[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (any_rescale) {
#pragma unroll
      for (int h = 0; h < G; ++h) {
        const float al = bc[h];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] *= al;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float vf[8];
      load_v(t0 + i * TQ + tq, vf);
      float ph[PH];
#pragma unroll
      for (int q4 = 0; q4 < JQ; ++q4) {
        const float4 t = *reinterpret_cast<const float4*>(Pw + (i * TQ + tq) * PH + q4 * 4);
        ph[q4 * 4 + 0] = t.x; ph[q4 * 4 + 1] = t.y; ph[q4 * 4 + 2] = t.z; ph[q4 * 4 + 3] = t.w;
      }
#pragma unroll
      for (int h = 0; h < G; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(ph[h], vf[e], acc[h][e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (n == 0 && jq < JQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = jq * 4 + r;
      if (h < G) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = m[r] + __logf(l[r]);
    }
    *reinterpret_cast<float4*>(bc + jq * 4) = make_float4(l[0], l[1], l[2], l[3]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int h = 0; h < G; ++h) {
    const float lh = bc[h];
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[h][e];
#pragma unroll
      for (int off = DC; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
      o8[e] = x / lh;
    }
    if (tq == 0) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + dc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
  }
}

template <int D>
int dispatch(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  const int G = a.num_q_heads / a.num_kv_heads;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  dim3 grid(nblk, a.batch), block(64 * a.num_kv_heads);
  const size_t shm = sizeof(float) * a.num_kv_heads * (kTile * (((G + 3) / 4) * 4) + 16);
  // head_dim 128, <= 4 KV heads, fp32 key parameters, block_seq a multiple of 128: 128-token tiles whose whole-block tiles
  // take 16-byte loads through a per-wave LDS buffer; other group-32 launches: register-staged 128-token tiles with 4-byte
  // code loads; other group sizes: the 32-token kernel
  if (kivi_wide_launch(a)) return launch_kivi_wide(a, s);
  if (a.group_size == 32) return launch_kivi_narrow(a, s);
  switch (G) {
#define SVK_CASE(G_)                                                                                     \
  case G_:                                                                                               \
    if (a.key_param_dtype == SVK_DTYPE_F32) hipLaunchKernelGGL((kivi_stage1_kernel<D, G_, true>), grid, block, shm, s, a); \
    else hipLaunchKernelGGL((kivi_stage1_kernel<D, G_, false>), grid, block, shm, s, a);                  \
    break;
    SVK_ALL_G_CASES
#undef SVK_CASE
    default:
      set_error("svk_kivi_decode_stage1: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
  return check_launch("svk_kivi_decode_stage1");
}

}  // namespace
}  // namespace svk

extern "C" int32_t svk_kivi_decode_stage1_extra_partials(const SvkKiviDecodeStage1Args* a) {
  return (a != nullptr && svk::kivi_wide_launch(*a)) ? 3 : 0;
}

extern "C" int svk_kivi_decode_stage1(const SvkKiviDecodeStage1Args* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_kivi_decode_stage1: null args");
  SVK_REQUIRE(a->new_k == nullptr || (a->new_v != nullptr && a->new_slots != nullptr && svk_kivi_decode_stage1_extra_partials(a) > 0 &&
                                      (a->new_stride_b % 8) == 0 && (a->new_stride_h % 8) == 0 &&
                                      (reinterpret_cast<uintptr_t>(a->new_k) % 16) == 0 && (reinterpret_cast<uintptr_t>(a->new_v) % 16) == 0 &&
                                      (a->raw_slot_stride % 8) == 0 && (a->raw_head_stride % 8) == 0),
              SVK_ERR_VALUE, "svk_kivi_decode_stage1: the fused raw store needs new_k, new_v, new_slots with 16-byte aligned rows and a "
              "launch the wide kernel serves (svk_kivi_decode_stage1_extra_partials() > 0)");
  SVK_REQUIRE(a->extra_partials == 0 || a->extra_partials == svk_kivi_decode_stage1_extra_partials(a), SVK_ERR_VALUE,
              "svk_kivi_decode_stage1: extra_partials %d, this launch supports 0 or %d (svk_kivi_decode_stage1_extra_partials)",
              a->extra_partials, svk_kivi_decode_stage1_extra_partials(a));
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_VALUE, "Unsupported decode head_dim=%d.", a->head_dim);
  SVK_REQUIRE(a->group_size > 0 && a->head_dim % a->group_size == 0, SVK_ERR_VALUE,
              "Invalid KIVI group_size=%d for head_dim=%d.", a->group_size, a->head_dim);
  SVK_REQUIRE(a->group_size % 8 == 0, SVK_ERR_VALUE, "int4 KIVI requires group_size/head_dim divisible by 8, got %d/%d.",
              a->group_size, a->head_dim);
  SVK_REQUIRE(a->key_param_dtype == SVK_DTYPE_F32 || a->key_param_dtype == SVK_DTYPE_BF16, SVK_ERR_VALUE,
              "svk_kivi_decode_stage1: key scale/min dtype must be f32 or bf16, got %d", a->key_param_dtype);
  SVK_REQUIRE(a->block_seq > 0 && a->block_seq % 16 == 0, SVK_ERR_VALUE,
              "block_seq must be a positive multiple of 16, got %d.", a->block_seq);
  SVK_REQUIRE(a->num_kv_heads >= 1 && a->num_kv_heads <= 8 && a->num_q_heads % a->num_kv_heads == 0, SVK_ERR_LAYOUT,
              "svk_kivi_decode_stage1: unsupported head configuration %d/%d", a->num_q_heads, a->num_kv_heads);
  if (a->max_len_in_batch <= 0 || a->batch <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->head_dim == 128 ? dispatch<128>(*a, s) : dispatch<64>(*a, s);
}
