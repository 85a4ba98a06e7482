// KIVI-int4 full-layer decode stage 1, LDS-DMA staged pipeline (gfx950).
//
// Same arithmetic as kivi_stage1_tile128_kernel (decode_kivi.hip) - both products on MFMA, int4 codes dequantised
// straight into operand registers - but the codes and scales of a 128-token tile travel HBM -> LDS with
// `global_load_lds_dwordx4` (16 B per lane, no VGPR in flight) one phase ahead of their use:
//     Q.K^T phase of tile t   : V(t) codes/scales are in flight      (V buffer is free: P.V(t-1) is done)
//     softmax + P.V of tile t : K(t+1) codes/scales are in flight    (K buffer is free: Q.K^T(t) is done)
// A (block, head) is 5.5 KiB in five contiguous runs (K codes 2 KiB, K scale / min 512 B each, V codes 2 KiB, V scale /
// min 256 B each), so every DMA instruction copies a contiguous 1 KiB (or two halves) - fully coalesced.
//
// Workgroup ranges are aligned to KIVI blocks: workgroup i covers [i*BS + f(i*BS), (i+1)*BS + f((i+1)*BS)) with
// f(p) = distance from p to the next block boundary (0 for raw tokens), so after at most one short head tile every tile
// is four whole blocks.  Tiles that are not four aligned blocks (sink, raw tail, ragged ends) take the per-token path.
//
// Index conventions (chosen so every LDS read of the fast path is bank-conflict free):
//   k index (kc, e) of Q.K^T  <-> channel c*32 + e*4 + kc      (Q fragments are loaded in the same order)
//   k index (kc, e) of P.V    <-> token   e*4 + kc of the 32-token block (the P tile is written in that order)
//   K codes of block j sit at j*(16*D + 64) bytes: the 64-byte skew spreads the four blocks over all 64 banks.

#include "svk_common.hpp"

namespace svk {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 kl_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float kl_f32x2_t;

__device__ __forceinline__ uint32_t kl_pack(float lo, float hi) {
  const kl_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kl_bf16x2_t));
}

template <int N>
__device__ __forceinline__ float kl_ubyte(uint32_t x) {
  float f;
  if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(x));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(x));
  return f;
}
template <int I>
__device__ __forceinline__ float kl_nibble(uint32_t lo, uint32_t hi) {
  return (I & 1) ? kl_ubyte<I / 2>(hi) : kl_ubyte<I / 2>(lo);     // odd nibbles come out as 16*q (callers use scale/16)
}

// 16 bytes per lane global -> LDS; `lds_base` is wave-uniform, lane l lands at lds_base + 16*l
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_base, 16, 0, 0);
}

constexpr int kT = 128, kGS = 32;

template <int D, bool KF32>
struct KlLayout {
  static constexpr int NG = D / 32;
  static constexpr int P_BYTES = 8 * kT * 2;                     // P tile rows 0..7 (rows 8..15 of the A operand read on into KC)
  static constexpr int KC_STRIDE = 16 * D + 64;                  // K codes of one block + bank skew
  static constexpr int KP_HALF = KF32 ? 4 * D : 2 * D;           // scale bytes (= min bytes) of one block
  static constexpr int KP_STRIDE = 2 * KP_HALF + 16;
  static constexpr int VC_STRIDE = 16 * D;
  static constexpr int VP_HALF = 64 * NG;                        // [32 tokens][NG] bf16
  static constexpr int VP_STRIDE = 2 * VP_HALF;
  static constexpr int KC = P_BYTES;
  static constexpr int KP = KC + 4 * KC_STRIDE;
  static constexpr int VC = KP + 4 * KP_STRIDE;
  static constexpr int VP = VC + 4 * VC_STRIDE;
  static constexpr int WAVE_BYTES = ((VP + 4 * VP_STRIDE + 255) / 256) * 256;
};

template <int D, int G, bool KF32>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) kivi_stage1_lds_kernel(const SvkKiviDecodeStage1Args a, int waves_per_wg) {
  using L = KlLayout<D, KF32>;
  constexpr int NC = D / 32, JQ = (G + 3) / 4, DW = D / 8, NG = D / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = blockIdx.z * waves_per_wg + wv;      // KV head of this wave
  const int Hkv = a.num_kv_heads;
  int b, blk;
  kivi_wg_to_range(b, blk);
  const int n = lane & 15, kc = lane >> 4;
  const int dg = n % DW;
  unsigned char* wl = lds_raw + wv * L::WAVE_BYTES;
  uint16_t* Pl = reinterpret_cast<uint16_t*>(wl);
  const int len = a.context_lens[b];
  const int row = a.req_indices[b];
  const int32_t* raw_map = a.raw_slots_map + (int64_t)row * a.map_stride;
  const int32_t* blk_map = a.kivi_block_slots_map + (int64_t)row * a.map_stride;
  // distance from position p to the next KIVI block boundary (0 when p is raw / invalid / already on a boundary)
  auto align_shift = [&](int p) -> int {
    if (p <= 0 || p >= len) return 0;
    if (raw_map[p] >= 0) return 0;
    const int bs = blk_map[p];
    if (bs < 0) return 0;
    const int lt = p - a.kivi_block_start_pos[bs];
    return (lt > 0 && lt < kGS) ? kGS - lt : 0;
  };
  const int start = blk == 0 ? 0 : blk * a.block_seq + align_shift(blk * a.block_seq);
  const int end = min(len, (blk + 1) * a.block_seq + align_shift((blk + 1) * a.block_seq));
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
  for (int i = lane; i < L::P_BYTES / 4; i += 64) reinterpret_cast<uint32_t*>(Pl)[i] = 0u;
  // Q fragments in the permuted channel order: lane (m = n, kc), element e <-> channel c*32 + e*4 + kc
  bf16x8_t qa[NC];
  {
    const uint16_t* qp = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint32_t qw[4] = {0u, 0u, 0u, 0u};
      if (n < G) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          qw[e2] = (uint32_t)qp[c * 32 + (2 * e2) * 4 + kc] | ((uint32_t)qp[c * 32 + (2 * e2 + 1) * 4 + kc] << 16);
      }
      qa[c] = __builtin_bit_cast(bf16x8_t, make_uint4(qw[0], qw[1], qw[2], qw[3]));
    }
  }
  const float sm_scale = rsqrtf((float)D);
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- per-token path (raw tokens, misaligned / ragged tiles), same index conventions as the fast path
  auto token_k = [&](int t, int t1, uint4 (&kr)[NC]) -> bool {
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[c] = make_uint4(0, 0, 0, 0);
    if (t >= t1) return false;
    const int rs = raw_map[t];
    if (rs >= 0) {
      const uint16_t* kp = a.raw_k + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        uint32_t ow[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          ow[e2] = (uint32_t)kp[c * 32 + (2 * e2) * 4 + kc] | ((uint32_t)kp[c * 32 + (2 * e2 + 1) * 4 + kc] << 16);
        kr[c] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      }
      return true;
    }
    const int bs = blk_map[t];
    if (bs < 0) return false;
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= kGS) return false;
    const int64_t hb = (int64_t)bs * Hkv + w;
    const int shift = (lt & 7) * 4, widx = lt >> 3;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint32_t ow[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float xv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ch = c * 32 + (2 * e2 + q) * 4 + kc;
          float sc, mn;
          if constexpr (KF32) {
            sc = reinterpret_cast<const float*>(a.key_scales)[hb * D + ch];
            mn = reinterpret_cast<const float*>(a.key_mins)[hb * D + ch];
          } else {
            sc = __builtin_bit_cast(float, (uint32_t) reinterpret_cast<const uint16_t*>(a.key_scales)[hb * D + ch] << 16);
            mn = __builtin_bit_cast(float, (uint32_t) reinterpret_cast<const uint16_t*>(a.key_mins)[hb * D + ch] << 16);
          }
          const uint32_t cw = (uint32_t)a.key_packed[(hb * D + ch) * (kGS / 8) + widx];
          xv[q] = add_rn(mul_rn((float)((cw >> shift) & 15u), sc), mn);
        }
        ow[e2] = kl_pack(xv[0], xv[1]);
      }
      kr[c] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    return true;
  };
  auto token_v = [&](int t, int t1) -> uint4 {
    if (t >= t1) return make_uint4(0, 0, 0, 0);
    const int rs = raw_map[t];
    if (rs >= 0) return *reinterpret_cast<const uint4*>(a.raw_v + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + dg * 8);
    const int bs = blk_map[t];
    if (bs < 0) return make_uint4(0, 0, 0, 0);
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= kGS) return make_uint4(0, 0, 0, 0);
    const int64_t tb = ((int64_t)bs * Hkv + w) * kGS + lt;
    const uint32_t word = (uint32_t)a.value_packed[tb * DW + dg];
    const float sc = __builtin_bit_cast(float, (uint32_t)a.value_scales[tb * NG + dg / 4] << 16);
    const float mn = __builtin_bit_cast(float, (uint32_t)a.value_mins[tb * NG + dg / 4] << 16);
    uint32_t o[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2)
      o[e2] = kl_pack(add_rn(mul_rn((float)((word >> (e2 * 8)) & 15u), sc), mn),
                      add_rn(mul_rn((float)((word >> (e2 * 8 + 4)) & 15u), sc), mn));
    return make_uint4(o[0], o[1], o[2], o[3]);
  };

  // ---- tile classification in three steps, so both dependent round trips (slot maps, then the blocks' start
  //      positions) travel under the previous tile's arithmetic.  All 64 lanes read two map entries each (tokens
  //      tn + lane and tn + 64 + lane); a group of 8 tokens is 8 consecutive lanes of one half.
  //      A fast tile is four whole, aligned blocks: group g sits at local position (g & 3) * 8 of block g >> 2.
  struct Cls { int gb; bool fast; int len; };
  int c_raw0 = 0, c_raw1 = 0, c_blk0 = -1, c_blk1 = -1, c_sp = 0, c_gb = -1;
  bool c_pre = false;
  auto cls_issue = [&](int tn) {
    const int p0 = min(tn + lane, len - 1), p1 = min(tn + 64 + lane, len - 1);
    c_raw0 = raw_map[p0]; c_blk0 = blk_map[p0];
    c_raw1 = raw_map[p1]; c_blk1 = blk_map[p1];
  };
  auto cls_mid = [&](int tn) {
    // every token of every group quantised and in the same block as the group's first token
    const bool in0 = tn + lane < end, in1 = tn + 64 + lane < end;
    const int h0 = __shfl(c_blk0, lane & ~7, 64), h1 = __shfl(c_blk1, lane & ~7, 64);
    const bool ok0 = in0 && c_raw0 < 0 && c_blk0 >= 0 && c_blk0 == h0;
    const bool ok1 = in1 && c_raw1 < 0 && c_blk1 >= 0 && c_blk1 == h1;
    c_pre = (tn + kT <= end) && __all(ok0 && ok1);
    // group g's block id -> lane g (g < 16), then its start position (dependent read)
    const int g0 = __shfl(c_blk0, (lane & 7) * 8, 64), g1 = __shfl(c_blk1, (lane & 7) * 8, 64);
    c_gb = lane < 8 ? g0 : g1;
    c_sp = 0;
    if (c_pre && lane < 16) c_sp = a.kivi_block_start_pos[c_gb];
  };
  auto cls_finish = [&](int tn) -> Cls {
    Cls c;
    c.gb = c_gb;
    c.fast = false;
    c.len = 0;
    if (tn >= end) return c;
    bool gok = true;
    const int gh = __shfl(c_gb, lane & ~3, 64);
    if (c_pre && lane < 16) gok = (tn + lane * 8 - c_sp) == (lane & 3) * 8 && c_gb == gh;
    c.fast = c_pre && __all(gok);
    if (c.fast) {
      c.len = kT;
    } else {
      // per-token tile: up to the next block boundary when tn sits inside a block, else one group of 8 tokens
      const int raw_t = __shfl(c_raw0, 0, 64), blk_t = __shfl(c_blk0, 0, 64);
      int ln = 8;
      if (raw_t < 0 && blk_t >= 0) {
        const int lt = tn - a.kivi_block_start_pos[blk_t];
        if (lt > 0 && lt < kGS) ln = kGS - lt;
        else if (lt == 0) ln = kGS;
      }
      c.len = min(ln, end - tn);
    }
    return c;
  };
  // DMA of a fast tile's K side: 4 blocks x (codes + scale|min)
  auto issue_k_dma = [&](int gb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bs = __shfl(gb, 4 * j, 64);
      const int64_t hb = (int64_t)bs * Hkv + w;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.key_packed) + hb * (16 * D);
#pragma unroll
      for (int hh = 0; hh < 16 * D / 1024; ++hh)
        dma16(src + hh * 1024 + lane * 16, wl + L::KC + j * L::KC_STRIDE + hh * 1024);
      // scale | min: lanes [0, KP_HALF/16) fetch the scales, the next KP_HALF/16 lanes the mins
      constexpr int HL = L::KP_HALF / 16;
      if (lane < 2 * HL) {
        const unsigned char* ps = reinterpret_cast<const unsigned char*>(lane < HL ? a.key_scales : a.key_mins) +
                                  hb * L::KP_HALF + (lane % HL) * 16;
        dma16(ps, wl + L::KP + j * L::KP_STRIDE);
      }
    }
  };
  auto issue_v_dma = [&](int gb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bs = __shfl(gb, 4 * j, 64);
      const int64_t hb = (int64_t)bs * Hkv + w;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.value_packed) + hb * (16 * D);
#pragma unroll
      for (int hh = 0; hh < 16 * D / 1024; ++hh)
        dma16(src + hh * 1024 + lane * 16, wl + L::VC + j * L::VC_STRIDE + hh * 1024);
      constexpr int HL = L::VP_HALF / 16;
      if (lane < 2 * HL) {
        const unsigned char* ps = reinterpret_cast<const unsigned char*>(lane < HL ? a.value_scales : a.value_mins) +
                                  hb * L::VP_HALF + (lane % HL) * 16;
        dma16(ps, wl + L::VP + j * L::VP_STRIDE);
      }
    }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  int t0 = start;
  cls_issue(t0);
  cls_mid(t0);
  Cls cur = cls_finish(t0);
  if (cur.fast) issue_k_dma(cur.gb);
  while (t0 < end) {
    const int t1 = t0 + cur.len;
    const bool fast = cur.fast;
    f32x4_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned tvmask = 0xffu;
    if (fast) {
      // K(t) has had the previous tile's softmax + P.V to land
      __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
      wave_sync();
      issue_v_dma(cur.gb);
    }
    cls_issue(t1);                                    // next tile's slot maps: consumed after this tile's Q.K^T
    if (fast) {
      const int jb = n >> 2, wq = n & 3;
      const unsigned char* kcb = wl + L::KC + jb * L::KC_STRIDE + wq * 4;
      const unsigned char* kpb = wl + L::KP + jb * L::KP_STRIDE;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        uint32_t lo[8], hi[8];
        float sc[8], sc16[8], mn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ch = c * 32 + e * 4 + kc;
          const uint32_t wd = *reinterpret_cast<const uint32_t*>(kcb + ch * 16);
          lo[e] = wd & 0x0f0f0f0fu;
          hi[e] = wd & 0xf0f0f0f0u;
          if constexpr (KF32) {
            sc[e] = *reinterpret_cast<const float*>(kpb + ch * 4);
            mn[e] = *reinterpret_cast<const float*>(kpb + L::KP_HALF + ch * 4);
          } else {
            sc[e] = __builtin_bit_cast(float, (uint32_t) * reinterpret_cast<const uint16_t*>(kpb + ch * 2) << 16);
            mn[e] = __builtin_bit_cast(float, (uint32_t) * reinterpret_cast<const uint16_t*>(kpb + L::KP_HALF + ch * 2) << 16);
          }
          sc16[e] = sc[e] * 0.0625f;
        }
#define SVK_KL_K(I_)                                                                                            \
        {                                                                                                      \
          uint32_t kf[4];                                                                                      \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                   \
            const float x0 = add_rn(mul_rn(kl_nibble<I_>(lo[2 * e2], hi[2 * e2]), ((I_) & 1) ? sc16[2 * e2] : sc[2 * e2]), mn[2 * e2]);             \
            const float x1 = add_rn(mul_rn(kl_nibble<I_>(lo[2 * e2 + 1], hi[2 * e2 + 1]), ((I_) & 1) ? sc16[2 * e2 + 1] : sc[2 * e2 + 1]), mn[2 * e2 + 1]); \
            kf[e2] = kl_pack(x0, x1);                                                                          \
          }                                                                                                    \
          s[I_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, make_uint4(kf[0], kf[1], kf[2], kf[3])), s[I_], 0, 0, 0); \
        }
        SVK_KL_K(0) SVK_KL_K(1) SVK_KL_K(2) SVK_KL_K(3) SVK_KL_K(4) SVK_KL_K(5) SVK_KL_K(6) SVK_KL_K(7)
#undef SVK_KL_K
      }
    } else {
      tvmask = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint4 kr[NC];
        if (token_k(t0 + 8 * n + i, t1, kr)) tvmask |= 1u << i;
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[c]), s[i], 0, 0, 0);
      }
    }
    cls_mid(t1);                                      // block ids known -> their start positions travel under the softmax
    // ---- raw scores (observation layers)
    if (a.attn_score != nullptr && kc < JQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = kc * 4 + r;
        if (h < G) {
          float* dst = a.attn_score + (int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + t0 + 8 * n;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if ((tvmask >> i) & 1u) dst[i] = s[i][r];
        }
      }
    }
    // ---- online softmax; P (bf16) -> LDS [head][block j][position kc'*8 + e'] with token lt = e'*4 + kc'
    float alpha[4];
    bool rescale = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool hv = (kc * 4 + r < G);
      float x[8], tmax = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] = (hv && ((tvmask >> i) & 1u)) ? s[i][r] * sm_scale : -INFINITY;
        tmax = fmaxf(tmax, x[i]);
      }
      tmax = row16_allmax(tmax);
      const float nm = fmaxf(m[r], tmax);
      float psum = 0.f;
      uint32_t pw[4] = {0u, 0u, 0u, 0u};
      alpha[r] = 1.f;
      if (hv && nm > -INFINITY) {
        alpha[r] = __expf(m[r] - nm);
        float p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = __expf(x[i] - nm); psum += p[i]; }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) pw[i4] = kl_pack(p[i4], p[i4 + 4]);     // tokens lt = 8*wq + i4 and + 4: neighbours
        rescale |= (nm != m[r]);
        m[r] = nm;
      }
      l[r] = l[r] * alpha[r] + row16_allsum(psum);
      if (kc < JQ) {
        // group n = (block j = n >> 2, word wq = n & 3): token lt = 8*wq + i -> position (i % 4)*8 + 2*wq + (i >> 2)
        uint32_t* prow = reinterpret_cast<uint32_t*>(Pl + (kc * 4 + r) * kT + (n >> 2) * 32);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) prow[(i4 * 8 + 2 * (n & 3)) / 2] = pw[i4];
      }
    }
    if (__any(rescale)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] *= alpha[r];
    }
    const Cls nxt = cls_finish(t1);
    if (nxt.fast) issue_k_dma(nxt.gb);               // K buffer is free: this tile's Q.K^T is done
    if (fast) {
      // V(t) (issued before Q.K^T) must have landed; K(t+1) may stay in flight
      if (nxt.fast) __builtin_amdgcn_s_waitcnt(0x0f70 | (4 * (16 * D / 1024 + 1)));   // vmcnt(#K DMA instructions)
      else __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    wave_sync();
    // ---- P.V: 4 blocks of 32 tokens, 8 MFMAs each (head dims dg*8 + i); k index (kc, e) <-> token e*4 + kc
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 pa = *reinterpret_cast<const uint4*>(Pl + n * kT + 32 * j + kc * 8);
      const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, pa);
      if (fast) {
        const unsigned char* vcb = wl + L::VC + j * L::VC_STRIDE + dg * 4;
        const unsigned char* vpb = wl + L::VP + j * L::VP_STRIDE + (dg / 4) * 2;
        uint32_t lo[8], hi[8];
        float sc[8], sc16[8], mn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int lt = e * 4 + kc;
          const uint32_t wd = *reinterpret_cast<const uint32_t*>(vcb + lt * (4 * DW));
          lo[e] = wd & 0x0f0f0f0fu;
          hi[e] = wd & 0xf0f0f0f0u;
          sc[e] = __builtin_bit_cast(float, (uint32_t) * reinterpret_cast<const uint16_t*>(vpb + lt * (2 * NG)) << 16);
          mn[e] = __builtin_bit_cast(float, (uint32_t) * reinterpret_cast<const uint16_t*>(vpb + L::VP_HALF + lt * (2 * NG)) << 16);
          sc16[e] = sc[e] * 0.0625f;
        }
#define SVK_KL_V(I_)                                                                                            \
        {                                                                                                      \
          uint32_t vf[4];                                                                                      \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                   \
            const float x0 = add_rn(mul_rn(kl_nibble<I_>(lo[2 * e2], hi[2 * e2]), ((I_) & 1) ? sc16[2 * e2] : sc[2 * e2]), mn[2 * e2]);             \
            const float x1 = add_rn(mul_rn(kl_nibble<I_>(lo[2 * e2 + 1], hi[2 * e2 + 1]), ((I_) & 1) ? sc16[2 * e2 + 1] : sc[2 * e2 + 1]), mn[2 * e2 + 1]); \
            vf[e2] = kl_pack(x0, x1);                                                                          \
          }                                                                                                    \
          acc[I_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[I_], 0, 0, 0); \
        }
        SVK_KL_V(0) SVK_KL_V(1) SVK_KL_V(2) SVK_KL_V(3) SVK_KL_V(4) SVK_KL_V(5) SVK_KL_V(6) SVK_KL_V(7)
#undef SVK_KL_V
      } else {
        uint4 vr[8];                                  // vr[e] = 8 head dims of token t0 + 32j + e*4 + kc
#pragma unroll
        for (int e = 0; e < 8; ++e) vr[e] = token_v(t0 + 32 * j + e * 4 + kc, t1);
        const uint32_t* vv = reinterpret_cast<const uint32_t*>(vr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          uint32_t vf[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            vf[e2] = __builtin_amdgcn_perm(vv[(2 * e2 + 1) * 4 + i / 2], vv[(2 * e2) * 4 + i / 2], (i & 1) ? 0x07060302u : 0x05040100u);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[i], 0, 0, 0);
        }
      }
    }
    wave_sync();
    t0 = t1;
    cur = nxt;
  }
  // ---- epilogue: lane (n, kc) owns heads kc*4+r and head dims dg*8 .. +8
  if (kc < JQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = kc * 4 + r;
      if (h < G) {
        if (n == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = m[r] + __logf(l[r]);
        if (n < DW) {
          float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + dg * 8;
          const float inv = 1.0f / l[r];
          *reinterpret_cast<float4*>(o) = make_float4(acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv);
          *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4][r] * inv, acc[5][r] * inv, acc[6][r] * inv, acc[7][r] * inv);
        }
      }
    }
  }
}

template <int D, int G>
int launch_lds(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  const int wpw = (a.num_kv_heads % 2 == 0) ? 2 : 1;
  dim3 grid(nblk, a.batch, a.num_kv_heads / wpw), block(64 * wpw);
  if (a.key_param_dtype == SVK_DTYPE_F32) {
    const size_t shm = (size_t)wpw * KlLayout<D, true>::WAVE_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kivi_stage1_lds_kernel<D, G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
    hipLaunchKernelGGL((kivi_stage1_lds_kernel<D, G, true>), grid, block, shm, s, a, wpw);
  } else {
    const size_t shm = (size_t)wpw * KlLayout<D, false>::WAVE_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kivi_stage1_lds_kernel<D, G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
    hipLaunchKernelGGL((kivi_stage1_lds_kernel<D, G, false>), grid, block, shm, s, a, wpw);
  }
  return check_launch("svk_kivi_decode_stage1");
}

}  // namespace

// entry used by decode_kivi.hip's dispatcher (group_size 32 only)
int launch_kivi_lds(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  const int G = a.num_q_heads / a.num_kv_heads;
#define SVK_CASE(D_, G_) if (a.head_dim == D_ && G == G_) return launch_lds<D_, G_>(a, s);
  SVK_CASE(128, 1) SVK_CASE(128, 2) SVK_CASE(128, 3) SVK_CASE(128, 4) SVK_CASE(128, 5) SVK_CASE(128, 6) SVK_CASE(128, 7) SVK_CASE(128, 8)
  SVK_CASE(64, 1) SVK_CASE(64, 2) SVK_CASE(64, 3) SVK_CASE(64, 4) SVK_CASE(64, 5) SVK_CASE(64, 6) SVK_CASE(64, 7) SVK_CASE(64, 8)
#undef SVK_CASE
  set_error("svk_kivi_decode_stage1: unsupported head_dim %d / group %d", a.head_dim, G);
  return SVK_ERR_LAYOUT;
}

}  // namespace svk
