// The 128-token-tile KIVI decode kernel (group_size 32), shared by decode_kivi_narrow.hip (register-staged tiles: any head
// shape) and decode_kivi_wide.hip (whole-block tiles with 16-byte loads: head_dim 128, <= 4 KV heads, fp32 key
// parameters).  Two translation units so that the two sets of instantiations compile in parallel.
#pragma once

#include <stdlib.h>

#include <type_traits>

#include "svk_common.hpp"

#ifndef SVK_KV_STAMP
#define SVK_KV_STAMP(i)
#endif

// developer builds (make EXTRA=-DSVK_DEV_G7) instantiate the Qwen2.5-7B group size only: 1/8 of the compile time
#ifdef SVK_DEV_G7
#define SVK_ALL_G_CASES SVK_CASE(7)
#else
#define SVK_ALL_G_CASES SVK_CASE(1) SVK_CASE(2) SVK_CASE(3) SVK_CASE(4) SVK_CASE(5) SVK_CASE(6) SVK_CASE(7) SVK_CASE(8)
#endif

namespace svk {

// does this launch take the wide kernel?
inline bool kivi_wide_launch(const SvkKiviDecodeStage1Args& a) {
  return a.head_dim == 128 && a.group_size == 32 && a.num_kv_heads <= 4 && a.block_seq % 128 == 0 &&
         a.key_param_dtype == SVK_DTYPE_F32;
}
// decode_kivi_wide.hip / decode_kivi_narrow.hip
int launch_kivi_wide(const SvkKiviDecodeStage1Args& a, hipStream_t s);
int launch_kivi_narrow(const SvkKiviDecodeStage1Args& a, hipStream_t s);

namespace {

// ------------------------------------------------------------------------------------------------
// 128-token-tile kernel (group_size 32): both products on the matrix cores, K/V dequantised straight
// into MFMA operand registers.  V side: the scale is a bf16 value and the code a 4-bit integer (times 16 for the odd
// nibbles, against scale/16), so code*scale is exact in fp32 and fma(code, scale, min) is bit-identical to the
// reference's separately rounded multiply and add - one packed fma instead of a packed multiply and a packed add.
//
//   Q.K^T  tile = 128 tokens = 16 "groups" of 8 consecutive tokens.  MFMA i (0..7) of d-chunk c takes as
//          column n the token 8n+i, so lane (n, kc) needs nibble i of the 8 words
//          Key_Packed[block(n)][h][c*32 + kc*8 + e][word(n)], e = 0..7 - one 4-byte load per word serves all 8
//          MFMAs (the 8 tokens of a word), 32 loads per lane per tile instead of 256.
//   P.V    MFMA i of 32-token block j takes as column n the head dim n*8+i, so lane (n, kc) needs nibble i of
//          Value_Packed[block j][h][kc*8 + e][n], e = 0..7: again one word per token serves all 8 MFMAs, and the
//          accumulator of lane (n, rows g) ends up holding 8 consecutive head dims -> 32-byte output stores.
//   The S accumulators already hold, per lane, 8 consecutive tokens of 4 heads, which is what the P tile in
//   LDS ([head][token] bf16) wants as 16-byte writes; the A operand of P.V reads it back as 16-byte rows.
//   Tiles that contain raw tokens (sink, tail), a ragged end or blocks not aligned to 8 tokens take the same data
//   flow with per-token loads.
// ------------------------------------------------------------------------------------------------

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));      // v_cvt_pk_bf16_f32 (RNE)
}

template <int N>
__device__ __forceinline__ float cvt_ubyte(uint32_t x) {
  float f;
  if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(x));
  else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(x));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(x));
  return f;
}

// nibble I of `w` as a float: even nibbles come out of (w & 0x0f0f0f0f) as q, odd ones out of (w & 0xf0f0f0f0) as
// 16*q - callers multiply those by scale/16, which rounds exactly like q*scale (power-of-two scaling).
template <int I>
__device__ __forceinline__ float nibble_f32(uint32_t lo, uint32_t hi) {
  return (I & 1) ? cvt_ubyte<I / 2>(hi) : cvt_ubyte<I / 2>(lo);
}


// The 8 int4 codes of a packed word as floats n * 2^-9, two per instruction: an fp8 (e4m3) byte 0x0n IS n * 2^-9 - its
// denormals and first binade form one linear ramp (tools/probe_fp8cvt.hip) - so v_cvt_pk_f32_fp8 on the nibble-masked
// word converts two codes at once, exactly.  x[0] = (n0, n2), x[1] = (n4, n6), x[2] = (n1, n3), x[3] = (n5, n7);
// multiplying by 512 * scale (an exact power-of-two prescale) rounds exactly like code * scale.
__device__ __forceinline__ void nibbles_fp8(uint32_t w, f32x2_t (&x)[4]) {
  const uint32_t lo = w & 0x0f0f0f0fu, hi = (w >> 4) & 0x0f0f0f0fu;
  x[0] = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false);
  x[1] = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
  x[2] = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false);
  x[3] = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
}
// value of nibble I in the layout above
template <int I>
__device__ __forceinline__ float nib_of(const f32x2_t (&x)[4]) {
  return x[(I & 1) * 2 + (I >> 2)][(I >> 1) & 1];
}
__device__ __forceinline__ f32x2_t pk_mul_rn(f32x2_t x, f32x2_t y) {
#pragma clang fp contract(off)
  return x * y;
}
__device__ __forceinline__ f32x2_t pk_add_rn(f32x2_t x, f32x2_t y) {
#pragma clang fp contract(off)
  return x + y;
}

// WIDE (D = 128): tiles made of four whole, consecutive KIVI blocks fetch their codes with 16-byte-per-lane loads (1 KiB
// per instruction, 8 instructions for the tile's K codes and 8 for its V codes instead of 32 + 32 four-byte ones) into
// registers one half-tile ahead, lay them down in a per-wave LDS buffer (K and V take turns in the same 9 KiB) and read
// the operand words back with conflict-free ds_read_b32.  Workgroup ranges are shifted to block boundaries so that the
// tiles of a regular row ARE whole blocks.  Everything else (raw tokens, ragged ends, irregular maps) takes the
// narrow paths below unchanged.
constexpr int kKBlkStride = 2048 + 16;          // K codes of a block-head in LDS: [128 channels][16 B] + 16 B (bank skew)
constexpr int kVBlkStride = 2048 + 4 * 64;      // V codes: [32 tokens][64 B], 64 B of skew after every 8 tokens
constexpr int kKParStride = 512 + 16;           // per-channel K scales (or mins) of a block-head: 128 floats + 16 B (bank skew)
constexpr int kKParOff = 4 * kKBlkStride;       // Q.K^T phase: [K codes of 4 blocks | K scales of 4 blocks | K mins of 4 blocks]
constexpr int kXBufBytes = kKParOff + 8 * kKParStride;     // 12480 (the P.V phase uses the first 4 * kVBlkStride = 9216)

template <int D, int G, bool KF32, bool WIDE = false>
__global__ void __launch_bounds__(512) kivi_stage1_tile128_kernel(const SvkKiviDecodeStage1Args a) {
  constexpr int NC = D / 32, JQ = (G + 3) / 4, DW = D / 8, NG = D / 32;
  constexpr int kT = 128, GS = 32;
  static_assert(!WIDE || D == 128, "the wide path is written for head_dim 128");
  // per-wave LDS: P tile [16][128] bf16 (4 KiB) | V scales [NG][128] bf16 | V mins [NG][128] bf16 | (WIDE) code buffer
  constexpr int PST = kT + 8;            // P tile row stride in bf16: 272 B, so the 16 rows of an A-operand read hit 16 bank groups
  constexpr int P_BYTES = 16 * PST * 2, VS_BYTES = NG * kT * 2, WAVE_BYTES = P_BYTES + 2 * VS_BYTES + (WIDE ? kXBufBytes : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  if constexpr (WIDE) SVK_KV_STAMP(0);
  int b, blk;
  const int n_extra = WIDE ? a.extra_partials : 0;
  kivi_wg_to_range_extra(b, blk, n_extra);
  const int nreg = (int)gridDim.x - n_extra;
  const int extra_id = blk - nreg;                  // >= 0: one of the extra workgroups of the row
  const int n = lane & 15, kc = lane >> 4;          // MFMA column / k chunk; kc doubles as accumulator row group
  const int dg = n % DW;                            // V word (8 head dims) of this lane
  unsigned char* wl = lds_raw + w * WAVE_BYTES;
  uint16_t* Pl = reinterpret_cast<uint16_t*>(wl);
  uint16_t* Vs = reinterpret_cast<uint16_t*>(wl + P_BYTES);
  uint16_t* Vm = reinterpret_cast<uint16_t*>(wl + P_BYTES + VS_BYTES);
  unsigned char* xbuf = wl + P_BYTES + 2 * VS_BYTES;          // WIDE only
  const int len = a.context_lens[b];
  const int row = a.req_indices[b];
  const int32_t* raw_map = a.raw_slots_map + (int64_t)row * a.map_stride;
  const int32_t* blk_map = a.kivi_block_slots_map + (int64_t)row * a.map_stride;
  int start = blk * a.block_seq;
  int end_nominal = start + a.block_seq;
  if constexpr (WIDE) {
    // range i = [i*BS + f(i*BS), (i+1)*BS + f((i+1)*BS)), f(p) = distance from p to the next block boundary (0 for raw
    // tokens, boundaries and positions outside the row): still a partition of the row, position-indexed outputs and the
    // merged result do not depend on where the cuts are
    // both cuts at once and without branches: two loads deep instead of six
    const int lm = max(len - 1, 0);
    const int q0 = min(max(start, 0), lm), q1 = min(max(end_nominal, 0), lm);
    const int r0 = raw_map[q0], r1 = raw_map[q1], b0 = blk_map[q0], b1 = blk_map[q1];
    const int lt0 = start - a.kivi_block_start_pos[max(b0, 0)], lt1 = end_nominal - a.kivi_block_start_pos[max(b1, 0)];
    const bool in0 = start > 0 && start < len && r0 < 0 && b0 >= 0 && lt0 > 0 && lt0 < GS;
    const bool in1 = end_nominal > 0 && end_nominal < len && r1 < 0 && b1 >= 0 && lt1 > 0 && lt1 < GS;
    start += in0 ? GS - lt0 : 0;
    end_nominal += in1 ? GS - lt1 : 0;
  }
  int end = min(len, end_nominal);
  // WIDE: the raw head [0, H) of a row (sink tokens), its raw tail [S, len) (the not yet quantised residual) and the
  // ragged quantised piece [S', S) in front of the tail that does not fill a 128-token tile from H are narrow tiles:
  // 20-25 us of dependent round trips each against 5-6 us for a wide tile, and they used to sit in the first and the
  // last full workgroups of the row - on the critical path of a one-round launch (1 x 256 k: 73 us against 59 us for a
  // row without them).  Every regular range is cut to [H, S').  With `extra_partials` = 3 (the caller's workspace has
  // three more partial slots per row) the three pieces are the ranges of three extra workgroups of the row, dispatched
  // first, which write the partials nblk_row, nblk_row + 1, nblk_row + 2; regular workgroups past the row's length then
  // write nothing.  Without them the row's LAST regular workgroup (whose own range is the remainder len mod block_seq)
  // adds the pieces to its online-softmax state.  Either way still a partition of the row.  H and S come from capped
  // scans of the raw map (128 / 512 positions): wherever the cuts fall, the narrow paths handle what they find.
  int head_end = 0, tail_start = len, ragged_start = len;
  bool owns_ends = false;
  int slot = blk;
  if constexpr (WIDE) {
    const int nblk_row = len <= 0 ? 0 : (len + a.block_seq - 1) / a.block_seq;
    if (n_extra > 0) {
      if (extra_id < 0 && blk >= nblk_row) return;            // the extra workgroups own the slots from nblk_row on
      if (extra_id >= 0) slot = nblk_row + extra_id;
    } else {
      owns_ends = blk == nblk_row - 1;
    }
    constexpr int kHeadScan = 128, kTailScan = 512;
    const bool need_s = owns_ends || extra_id >= 0 || end + kT > len - kTailScan;
    const bool need_h = need_s || start < kHeadScan;
    const int lm1 = max(len - 1, 0);
    if (need_h) {
      const int p0 = 2 * lane, p1 = 2 * lane + 1;
      const bool q0 = p0 < len && raw_map[min(p0, lm1)] < 0, q1 = p1 < len && raw_map[min(p1, lm1)] < 0;
      const unsigned long long m0 = __ballot(q0), m1 = __ballot(q1);
      int first = min(len, kHeadScan);
      if (m0) first = min(first, 2 * (int)__builtin_ctzll(m0));
      if (m1) first = min(first, 2 * (int)__builtin_ctzll(m1) + 1);
      head_end = first;
    }
    if (need_s) {
      const int w0 = max(len - kTailScan, 0);
      int last = -1;                                   // last quantised position of the window
#pragma unroll
      for (int e = 0; e < kTailScan / 64; ++e) {
        const int p = w0 + e * 64 + lane;
        const bool qz = p < len && raw_map[min(p, lm1)] < 0;
        const unsigned long long mk = __ballot(qz);
        if (mk) last = w0 + e * 64 + 63 - (int)__builtin_clzll(mk);
      }
      tail_start = max(last < 0 ? w0 : last + 1, head_end);
      ragged_start = head_end + ((tail_start - head_end) / kT) * kT;
    }
    if (extra_id >= 0) {
      start = extra_id == 0 ? 0 : (extra_id == 1 ? ragged_start : tail_start);
      end = extra_id == 0 ? head_end : (extra_id == 1 ? tail_start : len);
    } else {
      start = max(start, head_end);
      end = min(end, ragged_start);
    }
  }
  if constexpr (WIDE) SVK_KV_STAMP(1);
  float* mid_o = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)slot * a.mid_o_stride_s;
  float* mid_lse = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + slot;
  if constexpr (WIDE) {
    if (a.new_k != nullptr && len > 0 && (n_extra > 0 ? extra_id == n_extra - 1 : owns_ends)) {
      // fused raw store: this workgroup owns position len - 1 (the newest token is a raw row, so it is in the tail range; the store does not depend on that).
      // Wave w writes its KV head's K and V rows (2 x D/8 lanes, 16 bytes each) before any read of the slot; only this
      // wave reads that slot's head row in this launch, so a workgroup-scope fence pair (= the store has completed) is
      // enough, as in decode_stage1_kernel_v3.
      const int ns = a.new_slots[b];
      if (ns >= 0 && lane < 2 * DW) {
        const bool is_v = lane >= DW;
        const int seg = lane % DW;
        const uint16_t* src = (is_v ? a.new_v : a.new_k) + (int64_t)b * a.new_stride_b + (int64_t)w * a.new_stride_h + seg * 8;
        uint16_t* dst = const_cast<uint16_t*>(is_v ? a.raw_v : a.raw_k) + (int64_t)ns * a.raw_slot_stride +
                        (int64_t)w * a.raw_head_stride + seg * 8;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  if (end <= start && !(owns_ends && (head_end > 0 || ragged_start < len))) {
    for (int h = 0; h < G; ++h) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h;
      for (int d = lane; d < D; d += 64) o[d] = 0.f;
      if (lane == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = -INFINITY;
    }
    return;
  }
  for (int i = lane; i < P_BYTES / 4; i += 64) reinterpret_cast<uint32_t*>(Pl)[i] = 0u;     // rows >= G stay zero
  // Q fragments (A operand of Q.K^T): lane (m = n, kc) holds Q[head n][c*32 + kc*8 .. +8]
  bf16x8_t qa[NC];
  {
    const uint16_t* qp = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h + kc * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (n < G) t = *reinterpret_cast<const uint4*>(qp + c * 32);
      qa[c] = __builtin_bit_cast(bf16x8_t, t);
    }
  }
  const float sm_scale = rsqrtf((float)D);
  const bool score_vec_ok = a.attn_score != nullptr && (a.score_stride_b % 4) == 0 && (a.score_stride_h % 4) == 0 &&
                            (reinterpret_cast<uintptr_t>(a.attn_score) % 16) == 0;
  bool score_vec = score_vec_ok && (start % 8) == 0;
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  f32x4_t acc[8];                                    // acc[i][r]: head kc*4+r, head dim dg*8+i
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int lim = end;          // the narrow paths stop here (the row / range end, or the next block boundary in WIDE mode)
  // K fragment of one token (generic path): 8 channels c*32 + kc*8.. of token t, raw or dequantised
  auto token_k = [&](int t, uint4 (&kr)[NC]) -> bool {
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[c] = make_uint4(0, 0, 0, 0);
    if (t >= lim) return false;
    const int rs = raw_map[t];
    if (rs >= 0) {
      const uint16_t* kp = a.raw_k + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + kc * 8;
#pragma unroll
      for (int c = 0; c < NC; ++c) kr[c] = *reinterpret_cast<const uint4*>(kp + c * 32);
      return true;
    }
    const int bs = blk_map[t];
    if (bs < 0) return false;
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= GS) return false;
    const int64_t hb = (int64_t)bs * Hkv + w;
    const int shift = (lt & 7) * 4, widx = lt >> 3;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int d0 = c * 32 + kc * 8;
      uint32_t ow[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float s0, s1, m0, m1;
        if constexpr (KF32) {
          const float* sp = reinterpret_cast<const float*>(a.key_scales) + hb * D + d0 + e2 * 2;
          const float* mp = reinterpret_cast<const float*>(a.key_mins) + hb * D + d0 + e2 * 2;
          s0 = sp[0]; s1 = sp[1]; m0 = mp[0]; m1 = mp[1];
        } else {
          const uint32_t sw = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.key_scales) + hb * D + d0 + e2 * 2);
          const uint32_t mw = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.key_mins) + hb * D + d0 + e2 * 2);
          s0 = bf16_lo(sw); s1 = bf16_hi(sw); m0 = bf16_lo(mw); m1 = bf16_hi(mw);
        }
        const int32_t* cw = a.key_packed + (hb * D + d0 + e2 * 2) * (GS / 8) + widx;
        const float q0 = (float)(((uint32_t)cw[0] >> shift) & 15u), q1 = (float)(((uint32_t)cw[GS / 8] >> shift) & 15u);
        ow[e2] = pack_bf16(add_rn(mul_rn(q0, s0), m0), add_rn(mul_rn(q1, s1), m1));
      }
      kr[c] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    return true;
  };
  // V row segment of one token (generic path): 8 head dims dg*8.. as packed bf16; zero when invalid
  auto token_v = [&](int t) -> uint4 {
    if (t >= lim) return make_uint4(0, 0, 0, 0);
    const int rs = raw_map[t];
    if (rs >= 0) return *reinterpret_cast<const uint4*>(a.raw_v + (int64_t)rs * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + dg * 8);
    const int bs = blk_map[t];
    if (bs < 0) return make_uint4(0, 0, 0, 0);
    const int lt = t - a.kivi_block_start_pos[bs];
    if (lt < 0 || lt >= GS) return make_uint4(0, 0, 0, 0);
    const int64_t tb = ((int64_t)bs * Hkv + w) * GS + lt;
    const uint32_t word = (uint32_t)a.value_packed[tb * DW + dg];
    const float sc = __builtin_bit_cast(float, (uint32_t)a.value_scales[tb * NG + dg / 4] << 16);
    const float mn = __builtin_bit_cast(float, (uint32_t)a.value_mins[tb * NG + dg / 4] << 16);
    uint32_t o[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2)
      o[e2] = pack_bf16(add_rn(mul_rn((float)((word >> (e2 * 8)) & 15u), sc), mn),
                        add_rn(mul_rn((float)((word >> (e2 * 8 + 4)) & 15u), sc), mn));
    return make_uint4(o[0], o[1], o[2], o[3]);
  };

  // ---- shared middle of a tile pass: raw scores out, online softmax over the tile, P (bf16) -> LDS, rescale of O
  auto softmax_tile = [&](f32x4_t (&s)[8], unsigned tvmask, int t0) __attribute__((always_inline)) {
    // ---- raw scores (observation layers): 8 consecutive tokens per head per lane -> two 16-byte stores when the
  //      score rows keep 16-byte alignment (t0 and 8n are multiples of 8)
  if (a.attn_score != nullptr && kc < JQ) {
    const bool vec = score_vec && tvmask == 0xffu;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = kc * 4 + r;
      if (h < G) {
        float* dst = a.attn_score + (int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + t0 + 8 * n;
        if (vec) {
          *reinterpret_cast<float4*>(dst) = make_float4(s[0][r], s[1][r], s[2][r], s[3][r]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(s[4][r], s[5][r], s[6][r], s[7][r]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if ((tvmask >> i) & 1u) dst[i] = s[i][r];
        }
      }
    }
  }
  // ---- online softmax over the tile; P (bf16) -> LDS [head][token]
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
      for (int e2 = 0; e2 < 4; ++e2) pw[e2] = pack_bf16(p[2 * e2], p[2 * e2 + 1]);
      rescale |= (nm != m[r]);
      m[r] = nm;
    }
    l[r] = l[r] * alpha[r] + row16_allsum(psum);
    if (kc < JQ) *reinterpret_cast<uint4*>(Pl + (kc * 4 + r) * PST + 8 * n) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
  }
  if (__any(rescale)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] *= alpha[r];
  }
  };
  // A tile is processed in passes over disjoint token subsets (the online softmax does not care how the tokens are
  // partitioned): MODE_FAST = the groups that are 8 aligned tokens of one KIVI block (word loads; lanes of other groups
  // masked), MODE_RAW = the groups made of raw bf16 rows (slot ids staged in LDS, straight-line vector loads),
  // MODE_TOKEN = per-token fallback for anything else.  The sink tile of a row (8 raw tokens + 120 quantised ones) used
  // to take the per-token path as a whole: ~64 us of dependent loads on the critical path of every launch.
  // The tile body is instantiated twice: HOT = all 16 groups fast (the code of the steady state, nothing else in its
  // register allocation), and the general form; the driver loop below switches between them.
  constexpr int MODE_FAST = 0, MODE_RAW = 1, MODE_TOKEN = 2;
  struct Cls { int gb, glt; bool gfast, graw, gempty; };
  auto classify = [&](int t0) -> Cls {
    // the 16 groups of 8 tokens (lanes 0..15, one group each)
    Cls c{0, 0, true, false, false};
    if (lane < 16) {
      const int tg = t0 + lane * 8;
      c.gfast = false;
      c.gempty = tg >= lim;
      int rm[8], bm[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { rm[e] = raw_map[min(tg + e, lim - 1)]; bm[e] = blk_map[min(tg + e, lim - 1)]; }
      if (!c.gempty) {
        c.graw = true;
#pragma unroll
        for (int e = 0; e < 8; ++e) c.graw = c.graw && (tg + e >= lim || rm[e] >= 0);
      }
      if (tg + 8 <= lim) {
        const int b0 = bm[0];
        bool ok = b0 >= 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) ok = ok && rm[e] < 0 && bm[e] == b0;
        if (ok) {
          const int lt0 = tg - a.kivi_block_start_pos[b0];
          c.gfast = lt0 >= 0 && (lt0 & 7) == 0 && lt0 + 8 <= GS;
          if (c.gfast) { c.gb = b0; c.glt = lt0; }
        }
      }
    }
    return c;
  };
  auto tile_body = [&](auto hot_c, int t0, const Cls& cls) __attribute__((always_inline)) {
    constexpr bool HOT = decltype(hot_c)::value;
    const int gb = cls.gb, glt = cls.glt;
    bool split_ok = true, any_fast = true, any_raw = false;
    if constexpr (!HOT) {
      split_ok = __all(cls.gfast || cls.graw || cls.gempty);      // every group has a vector path
      any_fast = __any(lane < 16 && cls.gfast);
      any_raw = __any(cls.graw);
    }
    const int npass = HOT ? 1 : (split_ok ? (int)any_fast + (int)any_raw : 1);
    for (int pass = 0; pass < npass; ++pass) {
    const int mode = HOT ? MODE_FAST : (!split_ok ? MODE_TOKEN : ((pass == 0 && any_fast) ? MODE_FAST : MODE_RAW));
    const bool fast = mode == MODE_FAST;
    int* slot_lds = reinterpret_cast<int*>(Vs);          // raw pass: the tile's 128 raw slot ids (the V scale rows are idle)
    if constexpr (!HOT) {
      if (mode == MODE_RAW) {
        const int t = t0 + 2 * lane;
        const int s0 = raw_map[min(t, lim - 1)], s1 = raw_map[min(t + 1, lim - 1)];
        // -1 = not part of this pass (past the lim, or not a raw row)
        const bool mine = __shfl((int)cls.graw, lane >> 2, 64) != 0;
        slot_lds[2 * lane] = (mine && t < lim) ? s0 : -1;
        slot_lds[2 * lane + 1] = (mine && t + 1 < lim) ? s1 : -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    f32x4_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned tvmask = 0xffu;                          // bit i: token 8n+i of this lane's column is valid
    if constexpr (!HOT) tvmask = __shfl((int)cls.gfast, n, 64) ? 0xffu : 0u;      // fast pass of a mixed tile
    uint32_t vsw[NG], vmw[NG];                        // V scale / min words of tokens 2*lane, 2*lane+1 (fast tiles)
    uint32_t vq[2][8];                                // V words of block j (double buffered)
    auto issue_v = [&](int j, uint32_t (&vw8)[8]) {
      const int vb = __shfl(gb, 4 * j + kc, 64), vlt = __shfl(glt, 4 * j + kc, 64);
      const int32_t* vw = a.value_packed + ((((int64_t)vb * Hkv + w) * GS + vlt) * DW + dg);
#pragma unroll
      for (int e = 0; e < 8; ++e) vw8[e] = (uint32_t)vw[e * DW];
    };
    if (fast) {
      const int kb = __shfl(gb, n, 64), klt = __shfl(glt, n, 64);
      const int64_t hb = (int64_t)kb * Hkv + w;
      const int32_t* kw = a.key_packed + (hb * D + kc * 8) * (GS / 8) + (klt >> 3);
      // software pipeline: chunk c+1's words / scales are in flight while chunk c is dequantised and multiplied;
      // the V scale rows and the first V block are requested under the last chunk
      uint32_t wq[2][8];
      float scq[2][8], mnq[2][8];
      auto issue_k = [&](int c, uint32_t (&wd)[8], float (&sc)[8], float (&mn)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) wd[e] = (uint32_t)kw[(c * 32 + e) * (GS / 8)];
        if constexpr (KF32) {
          const float* sp = reinterpret_cast<const float*>(a.key_scales) + hb * D + c * 32 + kc * 8;
          const float* mp = reinterpret_cast<const float*>(a.key_mins) + hb * D + c * 32 + kc * 8;
          const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
          const float4 m0 = *reinterpret_cast<const float4*>(mp), m1 = *reinterpret_cast<const float4*>(mp + 4);
          sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
          mn[0] = m0.x; mn[1] = m0.y; mn[2] = m0.z; mn[3] = m0.w; mn[4] = m1.x; mn[5] = m1.y; mn[6] = m1.z; mn[7] = m1.w;
        } else {
          const uint4 sv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.key_scales) + hb * D + c * 32 + kc * 8);
          const uint4 mv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.key_mins) + hb * D + c * 32 + kc * 8);
          sc[0] = bf16_lo(sv.x); sc[1] = bf16_hi(sv.x); sc[2] = bf16_lo(sv.y); sc[3] = bf16_hi(sv.y);
          sc[4] = bf16_lo(sv.z); sc[5] = bf16_hi(sv.z); sc[6] = bf16_lo(sv.w); sc[7] = bf16_hi(sv.w);
          mn[0] = bf16_lo(mv.x); mn[1] = bf16_hi(mv.x); mn[2] = bf16_lo(mv.y); mn[3] = bf16_hi(mv.y);
          mn[4] = bf16_lo(mv.z); mn[5] = bf16_hi(mv.z); mn[6] = bf16_lo(mv.w); mn[7] = bf16_hi(mv.w);
        }
      };
      issue_k(0, wq[0], scq[0], mnq[0]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
          issue_k(c + 1, wq[(c + 1) & 1], scq[(c + 1) & 1], mnq[(c + 1) & 1]);
        } else {
          // V scale / min rows of the tile (lane l: tokens 2l, 2l+1) and the first V block
          const int gsrc = lane >> 2;
          const int vb = __shfl(gb, gsrc, 64), vlt = __shfl(glt, gsrc, 64) + ((2 * lane) & 7);
          const int64_t tb = (((int64_t)vb * Hkv + w) * GS + vlt) * NG;
          if constexpr (NG == 4) {
            const uint4 s4 = *reinterpret_cast<const uint4*>(a.value_scales + tb), m4 = *reinterpret_cast<const uint4*>(a.value_mins + tb);
            vsw[0] = s4.x; vsw[1] = s4.y; vsw[2] = s4.z; vsw[3] = s4.w; vmw[0] = m4.x; vmw[1] = m4.y; vmw[2] = m4.z; vmw[3] = m4.w;
          } else {
            const uint2 s2 = *reinterpret_cast<const uint2*>(a.value_scales + tb), m2 = *reinterpret_cast<const uint2*>(a.value_mins + tb);
            vsw[0] = s2.x; vsw[1] = s2.y; vmw[0] = m2.x; vmw[1] = m2.y;
          }
          issue_v(0, vq[0]);
        }
        const uint32_t (&wd)[8] = wq[c & 1];
        const float (&sc)[8] = scq[c & 1];
        const float (&mn)[8] = mnq[c & 1];
        uint32_t lo[8], hi[8];
        float sc16[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = wd[e] & 0x0f0f0f0fu; hi[e] = wd[e] & 0xf0f0f0f0u; sc16[e] = sc[e] * 0.0625f; }
#define SVK_K_MFMA(I_)                                                                                         \
        {                                                                                                      \
          uint32_t kf[4];                                                                                      \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                   \
            const float x0 = add_rn(mul_rn(nibble_f32<I_>(lo[2 * e2], hi[2 * e2]), ((I_) & 1) ? sc16[2 * e2] : sc[2 * e2]), mn[2 * e2]);             \
            const float x1 = add_rn(mul_rn(nibble_f32<I_>(lo[2 * e2 + 1], hi[2 * e2 + 1]), ((I_) & 1) ? sc16[2 * e2 + 1] : sc[2 * e2 + 1]), mn[2 * e2 + 1]); \
            kf[e2] = pack_bf16(x0, x1);                                                                        \
          }                                                                                                    \
          s[I_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, make_uint4(kf[0], kf[1], kf[2], kf[3])), s[I_], 0, 0, 0); \
        }
        SVK_K_MFMA(0) SVK_K_MFMA(1) SVK_K_MFMA(2) SVK_K_MFMA(3) SVK_K_MFMA(4) SVK_K_MFMA(5) SVK_K_MFMA(6) SVK_K_MFMA(7)
#undef SVK_K_MFMA
      }
    } else if (!HOT && mode == MODE_RAW) {
      // raw rows: every row segment of a chunk at once, no per-token branch (invalid tokens read slot 0 and are masked)
      int sl[8];
      {
        const int4 s0 = *reinterpret_cast<const int4*>(slot_lds + 8 * n), s1 = *reinterpret_cast<const int4*>(slot_lds + 8 * n + 4);
        sl[0] = s0.x; sl[1] = s0.y; sl[2] = s0.z; sl[3] = s0.w; sl[4] = s1.x; sl[5] = s1.y; sl[6] = s1.z; sl[7] = s1.w;
      }
      tvmask = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (sl[i] >= 0) tvmask |= 1u << i;
        sl[i] = max(sl[i], 0);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        uint4 kr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          kr[i] = *reinterpret_cast<const uint4*>(a.raw_k + (int64_t)sl[i] * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + c * 32 + kc * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[i]), s[i], 0, 0, 0);
      }
    } else if (!HOT) {
      tvmask = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint4 kr[NC];
        if (token_k(t0 + 8 * n + i, kr)) tvmask |= 1u << i;
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[c]), s[i], 0, 0, 0);
      }
    }
    softmax_tile(s, tvmask, t0);
    // ---- V scales / mins of the tile -> LDS [group][token] (fast tiles): lane l covers tokens 2l, 2l+1
    if (fast) {
      // memory order: token 2l: groups 0..NG-1, token 2l+1: groups 0..NG-1 (2 bf16 per word)
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int flat = tk * NG + g;
          const uint32_t sword = vsw[flat >> 1], mword = vmw[flat >> 1];
          Vs[g * kT + 2 * lane + tk] = (uint16_t)((flat & 1) ? (sword >> 16) : (sword & 0xffffu));
          Vm[g * kT + 2 * lane + tk] = (uint16_t)((flat & 1) ? (mword >> 16) : (mword & 0xffffu));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- P.V: 4 blocks of 32 tokens, 8 MFMAs each (head dims dg*8 + i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 pa = *reinterpret_cast<const uint4*>(Pl + n * PST + 32 * j + kc * 8);       // A: head n, tokens 32j + kc*8..
      const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, pa);
      if (fast) {
        if (j + 1 < 4) issue_v(j + 1, vq[(j + 1) & 1]);
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t wd = vq[j & 1][e];
          lo[e] = wd & 0x0f0f0f0fu;
          hi[e] = wd & 0xf0f0f0f0u;
        }
        const uint4 s8 = *reinterpret_cast<const uint4*>(Vs + (dg / 4) * kT + 32 * j + kc * 8);
        const uint4 m8 = *reinterpret_cast<const uint4*>(Vm + (dg / 4) * kT + 32 * j + kc * 8);
        float sc[8], sc16[8], mn[8];
        sc[0] = bf16_lo(s8.x); sc[1] = bf16_hi(s8.x); sc[2] = bf16_lo(s8.y); sc[3] = bf16_hi(s8.y);
        sc[4] = bf16_lo(s8.z); sc[5] = bf16_hi(s8.z); sc[6] = bf16_lo(s8.w); sc[7] = bf16_hi(s8.w);
        mn[0] = bf16_lo(m8.x); mn[1] = bf16_hi(m8.x); mn[2] = bf16_lo(m8.y); mn[3] = bf16_hi(m8.y);
        mn[4] = bf16_lo(m8.z); mn[5] = bf16_hi(m8.z); mn[6] = bf16_lo(m8.w); mn[7] = bf16_hi(m8.w);
#pragma unroll
        for (int e = 0; e < 8; ++e) sc16[e] = sc[e] * 0.0625f;
#define SVK_V_MFMA(I_)                                                                                         \
        {                                                                                                      \
          uint32_t vf[4];                                                                                      \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                   \
            const float x0 = __builtin_fmaf(nibble_f32<I_>(lo[2 * e2], hi[2 * e2]), ((I_) & 1) ? sc16[2 * e2] : sc[2 * e2], mn[2 * e2]);             \
            const float x1 = __builtin_fmaf(nibble_f32<I_>(lo[2 * e2 + 1], hi[2 * e2 + 1]), ((I_) & 1) ? sc16[2 * e2 + 1] : sc[2 * e2 + 1], mn[2 * e2 + 1]); \
            vf[e2] = pack_bf16(x0, x1);                                                                        \
          }                                                                                                    \
          acc[I_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[I_], 0, 0, 0); \
        }
        SVK_V_MFMA(0) SVK_V_MFMA(1) SVK_V_MFMA(2) SVK_V_MFMA(3) SVK_V_MFMA(4) SVK_V_MFMA(5) SVK_V_MFMA(6) SVK_V_MFMA(7)
#undef SVK_V_MFMA
      } else if (!HOT) {
        uint4 vr[8];                                  // vr[e] = 8 head dims (i = 0..7) of token 32j + kc*8 + e
        if (mode == MODE_RAW) {
          // raw rows straight from the staged slot ids; P is zero for every token outside this pass
          const int4 s0 = *reinterpret_cast<const int4*>(slot_lds + 32 * j + kc * 8), s1 = *reinterpret_cast<const int4*>(slot_lds + 32 * j + kc * 8 + 4);
          const int sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            vr[e] = *reinterpret_cast<const uint4*>(a.raw_v + (int64_t)max(sv[e], 0) * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + dg * 8);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) vr[e] = token_v(t0 + 32 * j + kc * 8 + e);
        }
        const uint32_t* vv = reinterpret_cast<const uint32_t*>(vr);        // vv[e*4 + i/2], half i&1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          uint32_t vf[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const uint32_t w0 = vv[(2 * e2) * 4 + i / 2], w1 = vv[(2 * e2 + 1) * 4 + i / 2];
            vf[e2] = (i & 1) ? ((w0 >> 16) | (w1 & 0xffff0000u)) : ((w0 & 0xffffu) | (w1 << 16));
          }
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[i], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    }   // pass
  };

  // ------------------------------------------------------------------------------------------------------------
  // WIDE: runs of tiles made of four whole consecutive KIVI blocks
  // ------------------------------------------------------------------------------------------------------------
  struct Cls4 { bool ok; int bj; };      // bj: block slot of the 32-token block this lane's two tokens sit in (lane >> 4)
  // "tile [t, t+128) is four whole consecutive blocks" in three steps, so that inside a wide run no step waits for a
  // load it has just issued: the maps (lane l: tokens t+2l, t+2l+1) - one phase later the partial verdict and the
  // dependent block-start load - one phase later the verdict
  struct Maps4 { bool inside; int r0, r1, b0, b1; };
  struct Pend4 { bool okp; int bj, bst; };
  auto maps4 = [&](int t) -> Maps4 {
    // no branch around the loads (a tile that does not fit reads clamped positions and is rejected by `inside`): a
    // branch would end the basic block and pull the first use of the values - and its in-order wait - up to here
    Maps4 mp{t + kT <= end, 0, 0, 0, 0};
    const int last = max(end - 1, 0);
    const int p = min(t + 2 * lane, last), p1 = min(t + 2 * lane + 1, last);       // (clamped into the row whatever t is)
    mp.r0 = raw_map[p]; mp.r1 = raw_map[p1]; mp.b0 = blk_map[p]; mp.b1 = blk_map[p1];
    return mp;
  };
  auto pend4 = [&](const Maps4& mp) -> Pend4 {
    Pend4 pd{false, 0, 0};
    pd.bj = __shfl(mp.b0, lane & 48, 64);
    pd.okp = mp.inside && mp.r0 < 0 && mp.r1 < 0 && mp.b0 == pd.bj && mp.b1 == pd.bj && pd.bj >= 0;
    pd.bst = a.kivi_block_start_pos[max(pd.bj, 0)];
    return pd;
  };
  auto verdict4 = [&](const Pend4& pd, int t) -> Cls4 {
    return Cls4{(bool)__all(pd.okp && pd.bst == t + 32 * (lane >> 4)), pd.bj};
  };
  // Entered with the maps of the tiles at t0 and t0 + 128 on their way (`pa`, `pb`: block ids known, recorded block starts
  // still in flight): the first tile's K codes are requested from the block ids at once and the verdicts are taken while
  // they travel - the start of a run is three dependent round trips instead of five (in-kernel stamps: first K tile in
  // LDS 8 us after the range is known -> 5.5).  Returns false, with nothing but a few abandoned loads done, when the tile
  // at t0 turns out not to be four whole blocks on their recorded positions.
  [[maybe_unused]] auto wide_run = [&](int& t0, const Pend4& pa, const Pend4& pb) __attribute__((always_inline)) -> bool {
    if constexpr (WIDE) {
    Cls4 cls{true, pa.bj};
    // LDS addresses of this lane's operand words
    // K row d of a block sits at position 16*(d>>4) + 8*((d&7)>>2) + 4*((d>>3)&1) + (d&3): rows d and d+8 (the two
    // k-chunks of a 32-lane half) are 64 B apart, so a ds_read_b32 of the half touches 32 different banks
    const unsigned char* kl = xbuf + (n >> 2) * kKBlkStride + (16 * (kc >> 1) + 4 * (kc & 1)) * 16 + (n & 3) * 4;   // + c*512 + (8*(e>>2) + (e&3))*16
    const unsigned char* vl = xbuf + kc * 8 * 64 + kc * 64 + n * 4;                             // + j*kVBlkStride + e*64
    // (the store addresses are recomputed from a fresh lane id where they are used: six address registers held across
    // the loop were the difference between no spill and spills whose scratch reloads drain the prefetched loads)
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    u32x4_t st[8];                                                     // one half-tile of codes in flight
    u32x4_t sp[4];                                                     // K scales (lanes 0..31) / mins (32..63) of the 4 blocks
    static_assert(KF32, "the wide path stages fp32 per-channel key parameters");
    const unsigned char* kpl = xbuf + kKParOff + (n >> 2) * kKParStride + kc * 8 * 4;         // + c*128 (+ 4*kKParStride: mins)
    auto issue_k = [&](const Cls4& c) {
      // parameters first: loads return in order and the Q.K^T phase needs them with the very first chunk
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int bs = __builtin_amdgcn_readlane(c.bj, 16 * j);
        const float* base = reinterpret_cast<const float*>(lane < 32 ? a.key_scales : a.key_mins) + ((int64_t)bs * Hkv + w) * D;
        sp[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(base) + (lane & 31));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bs = __builtin_amdgcn_readlane(c.bj, 16 * (j >> 1));
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(a.key_packed + (((int64_t)bs * Hkv + w) * D + (j & 1) * 64) * (GS / 8)) + lane;
        st[j] = __builtin_nontemporal_load(src);
      }
    };
    auto issue_v = [&](const Cls4& c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bs = __builtin_amdgcn_readlane(c.bj, 16 * (j >> 1));
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(a.value_packed + (((int64_t)bs * Hkv + w) * GS + (j & 1) * 16) * DW) + lane;
        st[j] = __builtin_nontemporal_load(src);
      }
    };
    auto put_k = [&]() {
      const int ln = lane_id_fresh();
      // K row d of a block sits at position 16*(d>>4) + 8*((d&7)>>2) + 4*((d>>3)&1) + (d&3) (see kl above)
      unsigned char* kst = xbuf + (16 * (ln >> 4) + 8 * ((ln & 7) >> 2) + 4 * ((ln >> 3) & 1) + (ln & 3)) * 16;   // + (j>>1)*kKBlkStride + (j&1)*1024
      unsigned char* kpst = xbuf + kKParOff + (ln >> 5) * 4 * kKParStride + (ln & 31) * 16;                          // + j*kKParStride
      const float kpmul = ln < 32 ? 512.0f : 1.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // scales go down premultiplied by 2^9 (exact), the factor that turns an fp8-converted code n * 2^-9 back into n
        typedef __attribute__((ext_vector_type(4))) float f4_t;
        const f4_t v = __builtin_bit_cast(f4_t, sp[j]) * kpmul;
        *reinterpret_cast<f4_t*>(kpst + j * kKParStride) = v;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4_t*>(kst + (j >> 1) * kKBlkStride + (j & 1) * 1024) = st[j];
    };
    auto put_v = [&]() {
      // instruction j: tokens (j & 1) * 16 + (lane >> 2) of block j >> 1; the 64-byte skew after every 8 tokens is in vst
      const int ln = lane_id_fresh();
      unsigned char* vst = xbuf + (ln >> 2) * 64 + (ln >> 5) * 64 + (ln & 3) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4_t*>(vst + (j >> 1) * kVBlkStride + (j & 1) * (16 * 64 + 128)) = st[j];
    };
    auto lds_sync = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    issue_k(cls);
    if (!verdict4(pa, t0).ok) return false;
    Cls4 nxt = verdict4(pb, t0 + kT);
    put_k();
    lds_sync();
    SVK_KV_STAMP(3);
    [[maybe_unused]] bool first_tile = true;
    while (true) {
      // ---------------- Q.K^T of the tile: codes and per-channel scales / mins from LDS
      // in flight under this phase: the tile's V codes from the start; its V scale / min rows (lane l: tokens 2l, 2l+1)
      // from the third chunk and the maps of the tile after next from the last one (requested as late as their first
      // use allows: every register held across the whole phase is one the dequantisation cannot have)
      uint32_t vsw[NG], vmw[NG];
      issue_v(cls);
      Maps4 mp2{false, 0, 0, 0, 0};
      f32x4_t s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // the eight operand words and the per-channel parameters of a chunk are read from LDS together at the top of the
      // chunk (written out word by word inside the dequantisation, hipcc waited for every ds_read four instructions after
      // issuing it).  Reading chunk c + 1 under chunk c (software pipeline, with or without a scheduling barrier), in this
      // build or in a 256-thread-bound one that has the whole register file of a SIMD for its single wave (487 registers, no
      // spill), was measured and is not faster: 59-61 us against 57 us at 1 x 256 k, 174-184 against 165 us at 4 x
      // (profiles/r05/kivi_pipe_ab.txt).
      constexpr bool kPipe = false;
      uint32_t kwq[2][8];
      float4 kpq[2][4];
      auto read_k_chunk = [&](int c, uint32_t (&kw)[8], float4 (&kp)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kw[e] = *reinterpret_cast<const uint32_t*>(kl + c * 512 + (8 * (e >> 2) + (e & 3)) * 16);
        kp[0] = *reinterpret_cast<const float4*>(kpl + c * 128);
        kp[1] = *reinterpret_cast<const float4*>(kpl + c * 128 + 16);
        kp[2] = *reinterpret_cast<const float4*>(kpl + 4 * kKParStride + c * 128);
        kp[3] = *reinterpret_cast<const float4*>(kpl + 4 * kKParStride + c * 128 + 16);
      };
      if constexpr (kPipe) read_k_chunk(0, kwq[0], kpq[0]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c == NC - 2) {
          const int vb = __shfl(cls.bj, lane & 48, 64);
          const int64_t tb = (((int64_t)vb * Hkv + w) * GS + ((2 * lane) & 31)) * NG;
          const uint4 s4 = *reinterpret_cast<const uint4*>(a.value_scales + tb), m4 = *reinterpret_cast<const uint4*>(a.value_mins + tb);
          vsw[0] = s4.x; vsw[1] = s4.y; vsw[2] = s4.z; vsw[3] = s4.w; vmw[0] = m4.x; vmw[1] = m4.y; vmw[2] = m4.z; vmw[3] = m4.w;
        }
        if (c == NC - 1) mp2 = maps4(t0 + 2 * kT);
        if constexpr (kPipe) {
          if (c + 1 < NC) read_k_chunk(c + 1, kwq[(c + 1) & 1], kpq[(c + 1) & 1]);
        } else {
          read_k_chunk(c, kwq[c & 1], kpq[c & 1]);
        }
        float sc[8], mn[8];
        {
          const float4 s0 = kpq[c & 1][0], s1 = kpq[c & 1][1], m0 = kpq[c & 1][2], m1 = kpq[c & 1][3];
          sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
          mn[0] = m0.x; mn[1] = m0.y; mn[2] = m0.z; mn[3] = m0.w; mn[4] = m1.x; mn[5] = m1.y; mn[6] = m1.z; mn[7] = m1.w;
        }
        // channel pairs (2*e2, 2*e2+1): 16 codes -> the e2-th operand word of all 8 MFMAs (token 8n+i takes nibble i)
        uint32_t kf[8][4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          f32x2_t y0[4], y1[4];
          nibbles_fp8(kwq[c & 1][2 * e2], y0);
          nibbles_fp8(kwq[c & 1][2 * e2 + 1], y1);
          const float s0 = sc[2 * e2], s1 = sc[2 * e2 + 1];             // already x 2^9
          const f32x2_t sv0 = {s0, s0}, mv0 = {mn[2 * e2], mn[2 * e2]}, sv1 = {s1, s1}, mv1 = {mn[2 * e2 + 1], mn[2 * e2 + 1]};
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            y0[k4] = pk_add_rn(pk_mul_rn(y0[k4], sv0), mv0);
            y1[k4] = pk_add_rn(pk_mul_rn(y1[k4], sv1), mv1);
          }
          kf[0][e2] = pack_bf16(nib_of<0>(y0), nib_of<0>(y1)); kf[1][e2] = pack_bf16(nib_of<1>(y0), nib_of<1>(y1));
          kf[2][e2] = pack_bf16(nib_of<2>(y0), nib_of<2>(y1)); kf[3][e2] = pack_bf16(nib_of<3>(y0), nib_of<3>(y1));
          kf[4][e2] = pack_bf16(nib_of<4>(y0), nib_of<4>(y1)); kf[5][e2] = pack_bf16(nib_of<5>(y0), nib_of<5>(y1));
          kf[6][e2] = pack_bf16(nib_of<6>(y0), nib_of<6>(y1)); kf[7][e2] = pack_bf16(nib_of<7>(y0), nib_of<7>(y1));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, make_uint4(kf[i][0], kf[i][1], kf[i][2], kf[i][3])), s[i], 0, 0, 0);
      }
      softmax_tile(s, 0xffu, t0);
      // ---------------- the code buffer changes hands: K words are all read, V codes go down
      lds_sync();
      put_v();
      // V scales / mins -> LDS [group][token]: memory order is token 2l: groups 0..3, token 2l+1: groups 0..3, so
      // group g of both tokens is one v_perm and one dword store
      {
        uint32_t* vs32 = reinterpret_cast<uint32_t*>(Vs);
        uint32_t* vm32 = reinterpret_cast<uint32_t*>(Vm);
        vs32[0 * (kT / 2) + lane] = __builtin_amdgcn_perm(vsw[2], vsw[0], 0x05040100u);
        vs32[1 * (kT / 2) + lane] = __builtin_amdgcn_perm(vsw[2], vsw[0], 0x07060302u);
        vs32[2 * (kT / 2) + lane] = __builtin_amdgcn_perm(vsw[3], vsw[1], 0x05040100u);
        vs32[3 * (kT / 2) + lane] = __builtin_amdgcn_perm(vsw[3], vsw[1], 0x07060302u);
        vm32[0 * (kT / 2) + lane] = __builtin_amdgcn_perm(vmw[2], vmw[0], 0x05040100u);
        vm32[1 * (kT / 2) + lane] = __builtin_amdgcn_perm(vmw[2], vmw[0], 0x07060302u);
        vm32[2 * (kT / 2) + lane] = __builtin_amdgcn_perm(vmw[3], vmw[1], 0x05040100u);
        vm32[3 * (kT / 2) + lane] = __builtin_amdgcn_perm(vmw[3], vmw[1], 0x07060302u);
      }
      const Pend4 pd2 = pend4(mp2);                            // (before the conditional loads: its wait stays a counted one)
      if (nxt.ok) issue_k(nxt);                                // next tile's K codes and parameters travel under P.V
      lds_sync();
      // ---------------- P.V: 4 blocks of 32 tokens, 8 MFMAs each (head dims dg*8 + i)
      uint32_t vwq[2][8];
      uint4 vpq[2][3];                                          // P fragment, V scales, V mins of a 32-token block
      auto read_v_block = [&](int j, uint32_t (&vw)[8], uint4 (&vp)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vw[e] = *reinterpret_cast<const uint32_t*>(vl + j * kVBlkStride + e * 64);
        vp[0] = *reinterpret_cast<const uint4*>(Pl + n * PST + 32 * j + kc * 8);
        vp[1] = *reinterpret_cast<const uint4*>(Vs + (dg / 4) * kT + 32 * j + kc * 8);
        vp[2] = *reinterpret_cast<const uint4*>(Vm + (dg / 4) * kT + 32 * j + kc * 8);
      };
      if constexpr (kPipe) read_v_block(0, vwq[0], vpq[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (kPipe) {
          if (j + 1 < 4) read_v_block(j + 1, vwq[(j + 1) & 1], vpq[(j + 1) & 1]);
        } else {
          read_v_block(j, vwq[j & 1], vpq[j & 1]);
        }
        const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, vpq[j & 1][0]);
        const uint4 s8 = vpq[j & 1][1], m8 = vpq[j & 1][2];
        const uint32_t s8w[4] = {s8.x, s8.y, s8.z, s8.w}, m8w[4] = {m8.x, m8.y, m8.z, m8.w};
        // token pairs (2*e2, 2*e2+1) of the block's k-chunk: 16 codes -> the e2-th operand word of all 8 MFMAs (head dim
        // dg*8 + i takes nibble i).  bf16 scale x 4-bit code is exact in fp32, so the fused multiply-add equals the
        // reference's separately rounded multiply and add bit for bit.
        uint32_t vf[8][4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          f32x2_t y0[4], y1[4];
          nibbles_fp8(vwq[j & 1][2 * e2], y0);
          nibbles_fp8(vwq[j & 1][2 * e2 + 1], y1);
          const float s0 = bf16_lo(s8w[e2]) * 512.0f, s1 = bf16_hi(s8w[e2]) * 512.0f;
          const float m0 = bf16_lo(m8w[e2]), m1 = bf16_hi(m8w[e2]);
          const f32x2_t sv0 = {s0, s0}, mv0 = {m0, m0}, sv1 = {s1, s1}, mv1 = {m1, m1};
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            y0[k4] = __builtin_elementwise_fma(y0[k4], sv0, mv0);
            y1[k4] = __builtin_elementwise_fma(y1[k4], sv1, mv1);
          }
          vf[0][e2] = pack_bf16(nib_of<0>(y0), nib_of<0>(y1)); vf[1][e2] = pack_bf16(nib_of<1>(y0), nib_of<1>(y1));
          vf[2][e2] = pack_bf16(nib_of<2>(y0), nib_of<2>(y1)); vf[3][e2] = pack_bf16(nib_of<3>(y0), nib_of<3>(y1));
          vf[4][e2] = pack_bf16(nib_of<4>(y0), nib_of<4>(y1)); vf[5][e2] = pack_bf16(nib_of<5>(y0), nib_of<5>(y1));
          vf[6][e2] = pack_bf16(nib_of<6>(y0), nib_of<6>(y1)); vf[7][e2] = pack_bf16(nib_of<7>(y0), nib_of<7>(y1));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[i][0], vf[i][1], vf[i][2], vf[i][3])), acc[i], 0, 0, 0);
      }
      t0 += kT;
      lds_sync();                                              // V words all read
#ifdef SVK_KV_TIMING
      if (first_tile) { SVK_KV_STAMP(4); first_tile = false; }
#endif
      if (!nxt.ok) break;
      put_k();
      lds_sync();
      cls = nxt;
      nxt = verdict4(pd2, t0 + kT);
    }
    return true;
    }
    return false;
  };
  // driver
  if constexpr (WIDE) {
    // wide runs wherever four whole blocks start at t0; a narrow tile otherwise, cut at the next block start so that the
    // tiles after it are on the block grid (the sink tile of a row: 8 raw tokens, then blocks)
    auto next_block_start = [&](int t0) -> int {
      int first = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int p = t0 + 1 + 2 * lane + k;
        bool is_start = false;
        if (p < end && raw_map[p] < 0) {
          const int bs = blk_map[p];
          is_start = bs >= 0 && a.kivi_block_start_pos[bs] == p;
        }
        const unsigned long long mask = __ballot(is_start);
        if (mask) first = min(first, t0 + 1 + 2 * (int)__builtin_ctzll(mask) + k);
      }
      return first;
    };
    // up to three ranges, one copy of the loop body: the workgroup's own range, then (last workgroup of the row) the raw
    // head and the raw tail
    SVK_KV_STAMP(2);
    const int rs3[3] = {start, 0, ragged_start};
    const int re3[3] = {end, owns_ends ? head_end : 0, owns_ends ? len : 0};
#pragma nounroll
    for (int ri = 0; ri < 3; ++ri) {
      const int rs = rs3[ri], re = re3[ri];
      if (re <= rs) continue;
      start = rs;
      end = re;
      lim = re;
      score_vec = score_vec_ok && (rs % 8) == 0;
      int t0 = rs;
      while (t0 < end) {
        const Maps4 ma = maps4(t0), mb = maps4(t0 + kT);
        const Pend4 pa = pend4(ma), pb = pend4(mb);
        if (__all(pa.okp) && wide_run(t0, pa, pb)) continue;
        const int nb = next_block_start(t0);
        lim = (nb < t0 + kT && nb + kT <= end) ? nb : end;
        const Cls cls = classify(t0);
        if (__all(cls.gfast)) tile_body(std::true_type{}, t0, cls);
        else tile_body(std::false_type{}, t0, cls);
        t0 = min(t0 + kT, lim);
        lim = end;
      }
    }
  } else {
    // general tiles until an all-fast tile shows up, then the hot loop until one is not, and so on
    int t0 = start;
    Cls cls = classify(t0);
    while (t0 < end) {
      while (t0 < end && !__all(cls.gfast)) {
        tile_body(std::false_type{}, t0, cls);
        t0 += kT;
        if (t0 < end) cls = classify(t0);
      }
      while (t0 < end && __all(cls.gfast)) {
        tile_body(std::true_type{}, t0, cls);
        t0 += kT;
        if (t0 < end) cls = classify(t0);
      }
    }
  }
  if constexpr (WIDE) SVK_KV_STAMP(5);
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
  if constexpr (WIDE) SVK_KV_STAMP(6);
}

}  // namespace
}  // namespace svk
