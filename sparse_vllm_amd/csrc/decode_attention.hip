// Split-KV GQA decode attention over a token-granular paged KV cache, with the H2O /
// SnapKV token-score fused into stage 1.  gfx950 (MI355X) only.
//
// Replaces (include/svk.h cites the exact lines)
//   kernels/triton/gqa_flash_decoding_stage1.py  flash_decode_stage1{,_with_score}
//   kernels/triton/flash_decoding_stage2.py      flash_decode_stage2
//
// Design (DESIGN.md "decode stage 1"):
//   * one workgroup = (batch lane, block_seq block); one WAVE per KV head, so the four
//     256-byte head segments of each 1 KiB token row are fetched by the four waves of one
//     workgroup back to back (same DRAM page), every byte exactly once;
//   * Q.K^T on the matrix cores: v_mfma_f32_16x16x32_bf16 with A = the <=16 query heads
//     of the GQA group (rows >= G are zero) and B = 16 token rows loaded straight from
//     HBM in the B-operand layout (lane = (token, 8-element k chunk)), fp32 accumulate
//     like the reference's tl.dot;
//   * per-token score = max over heads of the raw logit: lane-local max over the 4
//     accumulator rows, then combined across the row groups and the KV-head waves
//     through LDS by ONE owner thread per token column -> plain coalesced store, no
//     float atomics (the reference issues 4 contended atomic_max per token);
//   * online softmax statistics with DPP row reductions (16 tokens live in one DPP row);
//   * P.V on the matrix cores as well: P (rounded to bf16 like `exp_logic.to(v.dtype)`) goes through a 1.25 KiB
//     per-wave LDS tile whose rows are the A operand, V rows are read as 16-byte segments in the B-operand token order.

#include <stdlib.h>
#include <type_traits>

#include "svk_common.hpp"
#include "rope_row.hpp"
#include "svk_select.hpp"

namespace svk {
namespace {

constexpr int kTileTokens = 32;    // 2 MFMA column groups of 16 tokens
constexpr int kScoreChunk = 256;   // tokens between two score-combine barriers

template <int D, int G>
struct Stage1Cfg {
  static constexpr int NC = D / 32;          // MFMA k-chunks per head row
  static constexpr int JQ = (G + 3) / 4;     // accumulator row groups holding real heads
  static constexpr int PH = JQ * 4;          // padded heads per token in the P tile
  static constexpr int DC = D / 8;           // lanes per V head row (16 B each)
  static constexpr int TQ = 64 / DC;         // tokens per V wave-load
  static constexpr int NV = kTileTokens / TQ;
  static constexpr int P_FLOATS = kTileTokens * PH;
  static constexpr int WAVE_FLOATS = P_FLOATS + 16;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <bool NT>
__device__ __forceinline__ uint4 ld16(const char* p) {
  if (NT) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)));
  return *reinterpret_cast<const uint4*>(p);
}

constexpr int kPRow = 40;                          // P tile row stride in bf16 (32 tokens + pad, 16-byte aligned rows)
constexpr int kPFloats = 16 * kPRow / 2;           // P tile [16][kPRow] bf16 expressed in floats

template <int D, int G>
struct Stage1V3Lds {
  // P tile (bf16) + 64 slot ids + Q fragments (NC x 64 lanes x 16 B)
  static constexpr int WAVE_FLOATS = kPFloats + 64 + Stage1Cfg<D, G>::NC * 64 * 4;
};

// ---------------------------------------------------------------------------------------
// v3 = v2's pipeline with P.V on the matrix cores as well: V rows are loaded as 16-byte segments in the B-operand
// token order (lane (n, jq): 8 head dims n*8.. of tokens jq*8+e) and the 8 MFMAs of a tile pick column
// "head dim n*8+i" with a byte permute, so the accumulator of a lane is 8 consecutive head dims of 4 heads
// (32 registers for any GQA group size, no cross-lane reduction in the epilogue).  The vector ALUs are left with
// the softmax only (v2 spent ~75 % of its VALU issue slots on the P.V FMAs).
// ---------------------------------------------------------------------------------------
template <int D, int G, int MODE, bool NTV, bool OFF32>
__global__ void __launch_bounds__(512)
decode_stage1_kernel_v3(const SvkFlashDecodeStage1Args a) {
  using C = Stage1Cfg<D, G>;
  constexpr int NC = C::NC, JQ = C::JQ;
  constexpr int WF = Stage1V3Lds<D, G>::WAVE_FLOATS;
  constexpr int DW = D / 8;                       // 16-byte segments per head row
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  const int blk = blockIdx.x;
  const int n = lane & 15;
  const int jq = lane >> 4;
  const int dg = n % DW;                           // V: 16-byte segment (8 head dims) of this lane's MFMA column
  constexpr int score_mode = MODE;

  const int len = a.b_seqlen[b];
  // rotated form of the fused store (unscored launches): what it needs and what depends on nothing but (b, w, lane) is
  // requested here, beside the row length - the owner of the newest token then has one round trip (cos | sin and its
  // view slot) between the length and its stores instead of four
  bool rotated = false;
  int rot_ns = -1, rot_len = 0;
  uint4 rot_k1 = make_uint4(0, 0, 0, 0), rot_k2 = rot_k1, rot_v1 = rot_k1, rot_v2 = rot_k1;
  if constexpr (MODE == SVK_SCORE_NONE) {
    rotated = a.new_cos_sin != nullptr;
    if (rotated) {
      rot_ns = a.slot_mapping[b];
      if (a.new_row_lens != nullptr) rot_len = a.new_row_lens[b];
      if (lane < D / 16) {
        const int64_t src = (int64_t)b * a.new_stride_b + (int64_t)w * a.new_stride_h + lane * 8;
        rot_k1 = *reinterpret_cast<const uint4*>(a.new_k + src);
        rot_k2 = *reinterpret_cast<const uint4*>(a.new_k + src + D / 2);
        rot_v1 = *reinterpret_cast<const uint4*>(a.new_v + src);
        rot_v2 = *reinterpret_cast<const uint4*>(a.new_v + src + D / 2);
      }
    }
  }
  const int start = blk * a.block_seq;
  const int end = min(len, start + a.block_seq);

  // per-wave LDS: P tile | 16-float broadcast pad | 2 x 32 slot ids | Q fragments (lane-linear)
  float* Pw = lds + w * WF;
  uint16_t* Pl = reinterpret_cast<uint16_t*>(Pw);                 // [16 heads][kPRow] bf16, rows >= G stay zero
  int* slot_lds = reinterpret_cast<int*>(Pw + kPFloats);
  uint4* q_lds = reinterpret_cast<uint4*>(Pw + kPFloats + 64);
  float* spart = lds + Hkv * WF;
  const int SP = Hkv * JQ;

  float* mid_o = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)blk * a.mid_o_stride_s;
  float* mid_lse = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + blk;

  uint16_t* direct_o = a.direct_o == nullptr ? nullptr : a.direct_o + (int64_t)b * a.direct_stride_b;
  if (end <= start) {
    for (int h = 0; h < G; ++h) {
      if (direct_o != nullptr) {
        for (int d = lane; d < D; d += 64) direct_o[(int64_t)(w * G + h) * a.direct_stride_h + d] = 0;
      } else {
        float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h;
        for (int d = lane; d < D; d += 64) o[d] = 0.f;
      }
      if (lane == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = -INFINITY;
    }
    return;
  }

  if (a.new_k != nullptr && end == len) {
    // fused store_kvcache: this workgroup owns the lane's newest token.  Wave w writes its kv head's K and V rows
    // (2 x D/8 lanes, 16 bytes each) before any read of the row.  Only this wave reads that slot's head row in this
    // launch and the CU cannot hold a stale line of it (L1 is write-through, invalidated at kernel start), so a
    // workgroup-scope fence pair (= the store has completed) is enough; agent scope would write back the XCD's L2
    // (+10 us per launch measured).
    if (rotated) {
      if constexpr (MODE == SVK_SCORE_NONE) {
        // rotated store (include/svk.h): the raw rows go to the pre-RoPE cache, the view's row of position len-1 gets
        // the k-normed, rotated key - lane j < D/16 owns elements 8j.. and their rotate-half partners, exactly as in
        // the view launch (rope_row.hpp), V rides with the same lanes.  Rows, slot and position were requested
        // above, beside the row length.
        constexpr int HD2 = D / 2, LPH = HD2 / 8;
        if (rot_ns >= 0 && rot_ns < a.raw_num_slots && lane < LPH) {
          const int p = lane * 8;
          const int vslot = a.req_to_tokens[(int64_t)a.b_req_idx[b] * a.req_stride + (len - 1)];
          const int pos = max(a.new_row_lens != nullptr ? rot_len - 1 : a.new_slot_to_pos[rot_ns], 0);
          float n1[8], n2[8];
          rope_row_norm<D>(rot_k1, rot_k2, a.new_k_norm_weight, a.new_k_norm_eps, p, n1, n2);
          uint4 o1, o2;
          rope_row_rotate<D>(n1, n2, a.new_cos_sin, (int64_t)pos * a.new_cos_stride, a.new_cos_dtype, p, o1, o2);
          const int64_t rd = (int64_t)rot_ns * a.raw_slot_stride + (int64_t)w * a.raw_head_stride + p;
          *reinterpret_cast<uint4*>(a.raw_k_cache + rd) = rot_k1;
          *reinterpret_cast<uint4*>(a.raw_k_cache + rd + HD2) = rot_k2;
          *reinterpret_cast<uint4*>(a.raw_v_cache + rd) = rot_v1;
          *reinterpret_cast<uint4*>(a.raw_v_cache + rd + HD2) = rot_v2;
          const int64_t vd = (int64_t)vslot * a.kv_slot_stride + (int64_t)w * a.kv_head_stride + p;
          *reinterpret_cast<uint4*>(const_cast<uint16_t*>(a.k_cache) + vd) = o1;
          *reinterpret_cast<uint4*>(const_cast<uint16_t*>(a.k_cache) + vd + HD2) = o2;
          *reinterpret_cast<uint4*>(const_cast<uint16_t*>(a.v_cache) + vd) = rot_v1;
          *reinterpret_cast<uint4*>(const_cast<uint16_t*>(a.v_cache) + vd + HD2) = rot_v2;
        }
      }
    } else {
      const int ns = a.slot_mapping[b];
      if (ns >= 0 && lane < 2 * DW) {
        const bool is_v = lane >= DW;
        const int seg = lane % DW;
        const uint16_t* src = (is_v ? a.new_v : a.new_k) + (int64_t)b * a.new_stride_b + (int64_t)w * a.new_stride_h + seg * 8;
        uint16_t* dst = const_cast<uint16_t*>(is_v ? a.v_cache : a.k_cache) + (int64_t)ns * a.kv_slot_stride +
                        (int64_t)w * a.kv_head_stride + seg * 8;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;
  // Byte addressing.  OFF32: the whole K (V) tensor spans < 4 GiB, so a row address is the
  // wave-uniform tensor base (SGPR pair) + a 32-bit per-lane byte offset: one VGPR per address
  // and 32-bit integer math instead of 64-bit (frees ~20 VGPRs in the pipelined loop).
  const char* const kt = reinterpret_cast<const char*>(a.k_cache);
  const char* const vt = reinterpret_cast<const char*>(a.v_cache);
  const int64_t slot_bytes = a.kv_slot_stride * 2;
  const int64_t k_lane_bytes = ((int64_t)w * a.kv_head_stride + jq * 8) * 2;
  const int64_t v_lane_bytes = ((int64_t)w * a.kv_head_stride + dg * 8) * 2;
  auto k_ptr = [&](int slot) -> const char* {
    if (OFF32) return kt + (size_t)((uint32_t)slot * (uint32_t)slot_bytes + (uint32_t)k_lane_bytes);
    return kt + (int64_t)slot * slot_bytes + k_lane_bytes;
  };
  auto v_ptr = [&](int slot) -> const char* {
    if (OFF32) return vt + (size_t)((uint32_t)slot * (uint32_t)slot_bytes + (uint32_t)v_lane_bytes);
    return vt + (int64_t)slot * slot_bytes + v_lane_bytes;
  };
  const float sm_scale = rsqrtf((float)D);

  // slot ids of one 32-token tile: lanes 0..31 fetch row[t0 + lane] (one coalesced 128 B read)
  // (index clamped to the last valid token: always a legal, branch-free load - a conditional
  //  load would make the compiler's vmcnt bookkeeping conservative for the whole tile body)
  // (slot_page_size > 0, a power of two: `row` holds page slots, token slot = page slot * size + offset in the page)
  const int page_shift = a.slot_page_size > 0 ? __builtin_ctz((unsigned)a.slot_page_size) : 0;
  const uint32_t page_mask = (uint32_t)max(a.slot_page_size, 1) - 1u;
  auto fetch_slots = [&](int t0) -> int {
    const uint32_t t = (uint32_t)min(t0 + (lane_id_fresh() & 31), end - 1);
    return (row[t >> page_shift] << page_shift) + (int)(t & page_mask);
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // ---- prologue
  for (int i = lane; i < kPFloats; i += 64) Pw[i] = 0.f;
  {
    const int s0 = fetch_slots(start);
    if (lane < 32) slot_lds[lane] = s0;
    const uint16_t* qp = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h + jq * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (n < G) t = *reinterpret_cast<const uint4*>(qp + c * 32);
      q_lds[c * 64 + lane] = t;
    }
  }
  int s_next = fetch_slots(start + kTileTokens);     // slot ids of tile 1, parked in a register
  wave_sync();
  uint4 kr[2][NC];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const char* kp = k_ptr(slot_lds[g * 16 + n]);
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[g][c] = ld16<false>(kp + c * 64);
  }

  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  f32x4_t acc[8];                                  // acc[i][r]: head jq*4+r, head dim dg*8+i
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int c0 = start;     // first token of the current score chunk
  int buf = 0;        // slot_lds half holding the current tile's ids
  int t0 = start;
  // One tile.  HAS_NEXT is a compile-time flag (the last tile is peeled) so that the K(i+1)
  // re-arm is straight-line code: behind a run-time branch the compiler must assume the loads
  // may not have been issued and turns every later vmcnt(N) into a wait for K(i+1) itself.
  auto tile = [&](auto has_next_c) {
    constexpr bool has_next = decltype(has_next_c)::value;
    const bool full = has_next || (t0 + kTileTokens <= end);
    const int* cur_slots = slot_lds + buf * 32;
    int* nxt_slots = slot_lds + (buf ^ 1) * 32;

    // ---- V(i) loads (slot ids from LDS)
    uint4 vr[8];                                   // vr[e] = dims dg*8.. of token jq*8+e (the k index of P.V)
#pragma unroll
    for (int e = 0; e < 8; ++e)
      vr[e] = ld16<NTV>(v_ptr(cur_slots[jq * 8 + e]));

    // ---- publish tile i+1's slot ids (fetched one iteration ago), fetch tile i+2's
    {
      const int ln = lane_id_fresh();
      if (ln < 32) nxt_slots[ln] = s_next;
    }
    s_next = fetch_slots(t0 + 2 * kTileTokens);

    // ---- S = Q K^T on K(i)
    f32x4_t s[2];
    {
      bf16x8_t qa[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) qa[c] = __builtin_bit_cast(bf16x8_t, q_lds[c * 64 + lane]);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        s[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[g][c]), s[g], 0, 0, 0);
      }
    }
    wave_sync();        // nxt_slots visible to every lane of this wave
    // ---- K registers are dead: re-arm them with K(i+1)
    if constexpr (has_next) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const char* kp = k_ptr(nxt_slots[g * 16 + n]);
#pragma unroll
        for (int c = 0; c < NC; ++c) kr[g][c] = ld16<false>(kp + c * 64);
      }
    }

    bool tv[2];
    tv[0] = full || (t0 + n < end);
    tv[1] = full || (t0 + 16 + n < end);

    if constexpr (score_mode == SVK_SCORE_PERHEAD) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int h = jq * 4 + r;
          if (h < G && tv[g])
            a.attn_score[(int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + t0 + g * 16 + n] = s[g][r];
        }
    } else if constexpr (score_mode == SVK_SCORE_HEADMAX) {
      if (jq < JQ) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float pm = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jq * 4 + r < G) pm = fmaxf(pm, s[g][r]);
          spart[(t0 - c0 + g * 16 + n) * SP + w * JQ + jq] = tv[g] ? pm : -INFINITY;
        }
      }
    }

    float p[2][4];
    float alpha[4];
    bool rescale = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool hv = (jq * 4 + r < G);
      const float x0 = (hv && tv[0]) ? s[0][r] * sm_scale : -INFINITY;
      const float x1 = (hv && tv[1]) ? s[1][r] * sm_scale : -INFINITY;
      const float tmax = row16_allmax(fmaxf(x0, x1));
      const float nm = fmaxf(m[r], tmax);
      if (hv) {
        alpha[r] = __expf(m[r] - nm);
        p[0][r] = __expf(x0 - nm);
        p[1][r] = __expf(x1 - nm);
        rescale |= (nm != m[r]);
      } else {
        alpha[r] = 1.f; p[0][r] = 0.f; p[1][r] = 0.f;
      }
      l[r] = l[r] * alpha[r] + row16_allsum(p[0][r] + p[1][r]);
      m[r] = hv ? nm : m[r];
    }

    if (jq < JQ) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pl[(jq * 4 + r) * kPRow + g * 16 + n] = (uint16_t)f32_to_bf16_bits(p[g][r]);
    }
    if (__any(rescale)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] *= alpha[r];
    }
    wave_sync();

    // P.V on the matrix cores: A = P [head n][tokens jq*8 .. +8] (one 16-byte LDS row read), B column n of MFMA i
    // = head dim dg*8+i, whose 8 k-values are the i-th halves of the 8 loaded 16-byte V segments.
    {
      const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Pl + n * kPRow + jq * 8));
      const uint32_t* vv = reinterpret_cast<const uint32_t*>(vr);            // vv[e * 4 + i / 2]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t vf[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          vf[e2] = __builtin_amdgcn_perm(vv[(2 * e2 + 1) * 4 + i / 2], vv[(2 * e2) * 4 + i / 2], (i & 1) ? 0x07060302u : 0x05040100u);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[i], 0, 0, 0);
      }
    }
    wave_sync();

    // ---- end of a score chunk (or of the block): one owner thread per token column
    if (score_mode == SVK_SCORE_HEADMAX && (!has_next || (t0 + kTileTokens - c0) == kScoreChunk)) {
      const int c1 = min(end, t0 + kTileTokens);
      __syncthreads();
      float* const dst = a.attn_score + (int64_t)b * a.score_stride_b + c0;     // wave-uniform base
      for (uint32_t t = threadIdx.x; t < (uint32_t)(c1 - c0); t += blockDim.x) {
        float mx = -INFINITY;
        for (int j = 0; j < SP; ++j) mx = fmaxf(mx, spart[t * SP + j]);
        dst[t] = a.score_overwrite ? mx : fmaxf(dst[t], mx);
      }
      __syncthreads();
      c0 = t0 + kTileTokens;
    }
    t0 += kTileTokens;
    buf ^= 1;
  };
  while (t0 + kTileTokens < end) tile(std::true_type{});
  tile(std::false_type{});

  // ---- epilogue: lane (n, jq) owns heads jq*4+r and head dims dg*8 .. +8.  The lane id is laundered through an
  // empty asm so that none of the output addresses can be hoisted above the tile loop.
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int n_e = lane_e & 15, jq_e = lane_e >> 4;
  if (jq_e < JQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = jq_e * 4 + r;
      if (h < G) {
        if (n_e == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = m[r] + __logf(l[r]);
        if (n_e < DW) {
          if (direct_o != nullptr) {
            // the single partial of the row IS the output: stage 2 would weight it by exp(lse - lse) = 1 and round
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              pk[e] = f32_to_bf16_bits(acc[2 * e][r] / l[r]) | (f32_to_bf16_bits(acc[2 * e + 1][r] / l[r]) << 16);
            *reinterpret_cast<uint4*>(direct_o + (int64_t)(w * G + h) * a.direct_stride_h + (n_e % DW) * 8) =
                make_uint4(pk[0], pk[1], pk[2], pk[3]);
          } else {
            float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + (n_e % DW) * 8;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0][r] / l[r], acc[1][r] / l[r], acc[2][r] / l[r], acc[3][r] / l[r]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4][r] / l[r], acc[5][r] / l[r], acc[6][r] / l[r], acc[7][r] / l[r]);
          }
        }
      }
    }
  }
}

template <int D, int THREADS>
__global__ void __launch_bounds__(THREADS)
decode_stage2_kernel(const SvkFlashDecodeStage2Args a) {
  // one (batch lane, q head) per workgroup.  D/4 lanes cover one partial row with 16-byte loads; the THREADS/(D/4)
  // lane groups walk the split-KV partials interleaved (long contexts have ~1000 of them).  The row's maximum lse is
  // found first, so the weighted sum has no loop-carried exp chain and the partial loads stay in flight together
  // (the online form cost one HBM latency per partial: 41 us at 1025 partials).
  constexpr int LPR = D / 4, GROUPS = THREADS / LPR, WAVES = THREADS / 64;
  __shared__ float s_red[WAVES], s_l[GROUPS];
  __shared__ __attribute__((aligned(16))) float s_acc[GROUPS][D];
  const int b = blockIdx.x, h = blockIdx.y;
  const int g = threadIdx.x / LPR, d = (threadIdx.x % LPR) * 4;
  const int len = a.b_seqlen[b];
  const int nblk = len <= 0 ? 0 : (len + a.block_seq - 1) / a.block_seq + a.extra_partials;
  const float* mo = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)h * a.mid_o_stride_h + d;
  const float* ml = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + (int64_t)h * a.mid_lse_stride_h;
  // the first kPre partial rows of this lane group (and their lse) are requested BEFORE the row maximum is reduced: they do
  // not depend on it, and a launch of a few dozen partials is otherwise two dependent round trips (lse -> maximum ->
  // partials) of ~1.2 us each in a 4.7 us launch.  Same partials, same order, same arithmetic.
  constexpr int kPre = 8;
  float4 tvp[kPre];
  float lvp[kPre];
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    const int i = g + j * GROUPS;
    tvp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    lvp[j] = 0.f;
    if (i < nblk) {
      tvp[j] = *reinterpret_cast<const float4*>(mo + (int64_t)i * a.mid_o_stride_s);
      lvp[j] = ml[i];
    }
  }
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += THREADS) mx = fmaxf(mx, ml[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if constexpr (WAVES > 1) {
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) mx = fmaxf(mx, s_red[w]);
  }
  float sum = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    if (g + j * GROUPS < nblk) {
      const float e = __expf(lvp[j] - mx);
      acc.x += e * tvp[j].x; acc.y += e * tvp[j].y; acc.z += e * tvp[j].z; acc.w += e * tvp[j].w;
      sum += e;
    }
  }
#pragma unroll 4
  for (int i = g + kPre * GROUPS; i < nblk; i += GROUPS) {
    const float4 tv = *reinterpret_cast<const float4*>(mo + (int64_t)i * a.mid_o_stride_s);
    const float e = __expf(ml[i] - mx);
    acc.x += e * tv.x; acc.y += e * tv.y; acc.z += e * tv.z; acc.w += e * tv.w;
    sum += e;
  }
  if (nblk > 1) {
    if (d == 0) s_l[g] = sum;
    *reinterpret_cast<float4*>(&s_acc[g][d]) = acc;
    __syncthreads();
    if constexpr (GROUPS > 8) {
      // two levels (1024 threads, 32 lane groups): every eighth group adds up its seven followers, then group 0 the
      // leaders - 7 + 3 dependent LDS reads instead of 31 (groups beyond the partial count hold zeros)
      if ((g & 7) == 0) {
#pragma unroll
        for (int j = 1; j < 8; ++j) {
          const float4 tv = *reinterpret_cast<const float4*>(&s_acc[g + j][d]);
          acc.x += tv.x; acc.y += tv.y; acc.z += tv.z; acc.w += tv.w;
          sum += s_l[g + j];
        }
        if (g != 0) {
          *reinterpret_cast<float4*>(&s_acc[g][d]) = acc;
          if (d == 0) s_l[g] = sum;
        }
      }
      __syncthreads();
      if (g != 0) return;
#pragma unroll
      for (int j = 8; j < GROUPS; j += 8) {
        const float4 tv = *reinterpret_cast<const float4*>(&s_acc[j][d]);
        acc.x += tv.x; acc.y += tv.y; acc.z += tv.z; acc.w += tv.w;
        sum += s_l[j];
      }
    } else {
      if (g != 0) return;
      const int used = min(nblk, GROUPS);
      for (int j = 1; j < used; ++j) {
        const float4 tv = *reinterpret_cast<const float4*>(&s_acc[j][d]);
        acc.x += tv.x; acc.y += tv.y; acc.z += tv.z; acc.w += tv.w;
        sum += s_l[j];
      }
    }
  } else if (g != 0) {
    return;
  }
  const uint32_t w0 = f32_to_bf16_bits(acc.x / sum) | (f32_to_bf16_bits(acc.y / sum) << 16);
  const uint32_t w1 = f32_to_bf16_bits(acc.z / sum) | (f32_to_bf16_bits(acc.w / sum) << 16);
  *reinterpret_cast<uint2*>(a.o + (int64_t)b * a.o_stride_b + (int64_t)h * a.o_stride_h + d) = make_uint2(w0, w1);
}

// ---- two-level merge (launches with many partials per row: KIVI full layers at 256 k tokens, one-row launches)
// The one-level kernel above gives a (row, head) ONE workgroup: 259 partials x 516 B are 134 KB through one CU, 28
// workgroups on a 256-CU chip - 29 us per KIVI layer at 1 x 256 k, 11 % of a DeltaKV decode step.  Here workgroup
// (b, h, s) merges the kSplitPPW partials [s * kSplitPPW, ...) into a second-level partial of the same form (normalised
// row + lse) in the caller's workspace, takes a ticket, and the LAST workgroup of the (b, h) to arrive merges the
// second-level partials in index order - whoever that is, the sums are formed in the same order - and writes the output.
// Tickets are the caller's [batch * heads] int32 (zero before the first launch; the last arriver leaves its ticket at zero
// again) - NOT part of the scratch, whose layout moves with the launch shape.
constexpr int kSplitPPW = 32;          // partials per first-level workgroup
constexpr int kSplitMin = 256;         // launches that may merge more partials than this per row take the two-level form

__host__ __device__ inline int64_t split_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
__host__ __device__ inline int split_groups(int max_partials) { return (max_partials + kSplitPPW - 1) / kSplitPPW; }

template <int D>
__global__ void __launch_bounds__(256)
decode_stage2_split_kernel(const SvkFlashDecodeStage2Args a) {
  constexpr int THREADS = 256, LPR = D / 4, GROUPS = THREADS / LPR, WAVES = THREADS / 64, KPRE = kSplitPPW / GROUPS;
  static_assert(kSplitPPW % GROUPS == 0, "a workgroup's partials are whole rounds of its lane groups");
  __shared__ float s_red[WAVES], s_l[GROUPS];
  __shared__ __attribute__((aligned(16))) float s_acc[GROUPS][D];
  __shared__ int s_last;
  const int b = blockIdx.x, h = blockIdx.y, s = blockIdx.z;
  const int g = threadIdx.x / LPR, d = (threadIdx.x % LPR) * 4;
  const int len = a.b_seqlen[b];
  const int nblk = len <= 0 ? 0 : (len + a.block_seq - 1) / a.block_seq + a.extra_partials;
  const int s_row = max(1, (nblk + kSplitPPW - 1) / kSplitPPW);       // first-level workgroups this row needs
  if (s >= s_row) return;
  const int smax = split_groups(a.max_partials);
  const int64_t n_bh = (int64_t)gridDim.x * gridDim.y, bh = (int64_t)b * gridDim.y + h;
  unsigned char* ws = static_cast<unsigned char*>(a.split_ws);
  int* tickets = a.split_tickets;
  float* ws_lse = reinterpret_cast<float*>(ws) + bh * smax;
  float* ws_o = reinterpret_cast<float*>(ws + split_align(n_bh * smax * 4)) + bh * smax * D;
  const float* mo = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)h * a.mid_o_stride_h + d;
  const float* ml = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + (int64_t)h * a.mid_lse_stride_h;
  int64_t stride = a.mid_o_stride_s;
  int first = s * kSplitPPW, count = min(nblk - first, kSplitPPW);
  for (int level = 0; level < 2; ++level) {
    // merge rows [first, first + count) of (mo, ml): all loads of the round in flight together, the maximum reduced under them
    float4 tv[KPRE];
    float lv[KPRE];
#pragma unroll
    for (int j = 0; j < KPRE; ++j) {
      const int i = g + j * GROUPS;
      tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      lv[j] = -INFINITY;
      if (i < count) {
        if (level == 0) {
          tv[j] = *reinterpret_cast<const float4*>(mo + (int64_t)(first + i) * stride);
          lv[j] = ml[first + i];
        } else {
          // other workgroups' second-level partials: agent-scope (sc1) loads - coherent across CUs and XCDs without an
          // acquire fence (1.7 us) in front of them
          uint64_t* p = reinterpret_cast<uint64_t*>(const_cast<float*>(mo) + (int64_t)(first + i) * stride);
          const uint64_t x0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t x1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tv[j] = make_float4(__uint_as_float((uint32_t)x0), __uint_as_float((uint32_t)(x0 >> 32)),
                              __uint_as_float((uint32_t)x1), __uint_as_float((uint32_t)(x1 >> 32)));
          lv[j] = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(const_cast<float*>(ml)) + first + i,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPRE; ++j) mx = fmaxf(mx, lv[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) mx = fmaxf(mx, s_red[w]);
    float sum = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < KPRE; ++j) {
      if (g + j * GROUPS < count) {
        const float e = __expf(lv[j] - mx);
        acc.x += e * tv[j].x; acc.y += e * tv[j].y; acc.z += e * tv[j].z; acc.w += e * tv[j].w;
        sum += e;
      }
    }
    if (d == 0) s_l[g] = sum;
    *reinterpret_cast<float4*>(&s_acc[g][d]) = acc;
    __syncthreads();
    if (g == 0) {
      const int used = min(max(count, 1), GROUPS);
      for (int j = 1; j < used; ++j) {
        const float4 t2 = *reinterpret_cast<const float4*>(&s_acc[j][d]);
        acc.x += t2.x; acc.y += t2.y; acc.z += t2.z; acc.w += t2.w;
        sum += s_l[j];
      }
      if (level == 1 || s_row == 1) {
        const uint32_t w0 = f32_to_bf16_bits(acc.x / sum) | (f32_to_bf16_bits(acc.y / sum) << 16);
        const uint32_t w1 = f32_to_bf16_bits(acc.z / sum) | (f32_to_bf16_bits(acc.w / sum) << 16);
        *reinterpret_cast<uint2*>(a.o + (int64_t)b * a.o_stride_b + (int64_t)h * a.o_stride_h + d) = make_uint2(w0, w1);
      } else {
        // write-through (sc1) stores: visible to every CU / XCD once they have completed - no release fence, whose L2
        // write-back of everything stage 1 has just written cost 13 us per launch
        uint64_t* p = reinterpret_cast<uint64_t*>(ws_o + (int64_t)s * D + d);
        const float o0 = acc.x / sum, o1 = acc.y / sum, o2 = acc.z / sum, o3 = acc.w / sum;
        __hip_atomic_store(p, (uint64_t)__float_as_uint(o0) | ((uint64_t)__float_as_uint(o1) << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, (uint64_t)__float_as_uint(o2) | ((uint64_t)__float_as_uint(o3) << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (d == 0)
          __hip_atomic_store(reinterpret_cast<uint32_t*>(ws_lse) + s, __float_as_uint(mx + __logf(sum)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (level == 1 || s_row == 1) return;
    // hand-over (cdna_hip_programming.md, guideline 16, R1): every storing wave drains its stores, then ONE lane takes the
    // ticket; the last arriver reads the others' partials with agent-scope loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      s_last = __hip_atomic_fetch_add(&tickets[bh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s_row - 1;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&tickets[bh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
    mo = ws_o + d;
    ml = ws_lse;
    stride = D;
    first = 0;
    count = s_row;                                     // <= kSplitPPW (checked by the launcher)
  }
}

template <int D, int G>
int launch_stage1(const SvkFlashDecodeStage1Args& a, hipStream_t stream) {
  using C = Stage1Cfg<D, G>;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  dim3 grid(nblk, a.batch);
  dim3 block(64 * a.num_kv_heads);
  const size_t score_floats = a.score_mode == SVK_SCORE_HEADMAX ? (size_t)kScoreChunk * a.num_kv_heads * C::JQ : 0;
  const size_t shm3 = sizeof(float) * ((size_t)a.num_kv_heads * Stage1V3Lds<D, G>::WAVE_FLOATS + score_floats);
  // 32-bit row offsets whenever the caller tells us the KV tensors span < 4 GiB
  const bool off32 = a.kv_num_slots > 0 && (a.kv_num_slots * a.kv_slot_stride * 2) < (int64_t)0xffffffffll;
#define SVK_LAUNCH_V3(MODE_)                                                                                      \
  do {                                                                                                            \
    if (off32) hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, MODE_, true, true>), grid, block, shm3, stream, a);  \
    else hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, MODE_, true, false>), grid, block, shm3, stream, a);       \
  } while (0)
  if (a.score_mode == SVK_SCORE_HEADMAX) SVK_LAUNCH_V3(SVK_SCORE_HEADMAX);
  else if (a.score_mode == SVK_SCORE_PERHEAD) SVK_LAUNCH_V3(SVK_SCORE_PERHEAD);
  else SVK_LAUNCH_V3(SVK_SCORE_NONE);
#undef SVK_LAUNCH_V3
  return check_launch("svk_flash_decode_stage1");
}

template <int D>
int dispatch_group(const SvkFlashDecodeStage1Args& a, int G, hipStream_t stream) {
  switch (G) {
    case 1: return launch_stage1<D, 1>(a, stream);
    case 2: return launch_stage1<D, 2>(a, stream);
    case 3: return launch_stage1<D, 3>(a, stream);
    case 4: return launch_stage1<D, 4>(a, stream);
    case 5: return launch_stage1<D, 5>(a, stream);
    case 6: return launch_stage1<D, 6>(a, stream);
    case 7: return launch_stage1<D, 7>(a, stream);
    case 8: return launch_stage1<D, 8>(a, stream);
    default:
      set_error("svk_flash_decode_stage1: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
}

}  // namespace
}  // namespace svk

static int validate_stage1(const SvkFlashDecodeStage1Args* a, const char* who) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "%s: null args", who);
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_LAYOUT, "%s: head_dim %d unsupported (64, 128)", who, a->head_dim);
  SVK_REQUIRE(a->block_seq > 0 && a->block_seq % 16 == 0, SVK_ERR_LAYOUT,
              "%s: block_seq %d must be a positive multiple of 16 (BLOCK_SEQ %% BLOCK_N)", who, a->block_seq);
  SVK_REQUIRE(a->num_kv_heads >= 1 && a->num_kv_heads <= 8, SVK_ERR_LAYOUT, "%s: num_kv_heads %d unsupported (1..8 per rank)", who,
              a->num_kv_heads);
  SVK_REQUIRE(a->num_q_heads % a->num_kv_heads == 0, SVK_ERR_LAYOUT, "%s: q heads %d not divisible by kv heads %d", who,
              a->num_q_heads, a->num_kv_heads);
  SVK_REQUIRE(a->score_mode == SVK_SCORE_NONE || a->score_mode == SVK_SCORE_HEADMAX || a->score_mode == SVK_SCORE_PERHEAD,
              SVK_ERR_VALUE, "%s: bad score_mode %d", who, a->score_mode);
  SVK_REQUIRE(a->slot_page_size >= 0 && (a->slot_page_size & (a->slot_page_size - 1)) == 0, SVK_ERR_VALUE,
              "%s: slot_page_size %d must be 0 or a power of two", who, a->slot_page_size);
  SVK_REQUIRE(a->score_mode == SVK_SCORE_NONE || a->attn_score != nullptr, SVK_ERR_VALUE, "%s: score_mode %d needs attn_score", who,
              a->score_mode);
  SVK_REQUIRE((a->kv_slot_stride % 8) == 0 && (a->kv_head_stride % 8) == 0 && (a->q_stride_h % 8) == 0 && (a->q_stride_b % 8) == 0,
              SVK_ERR_LAYOUT, "%s: q/k/v strides must keep 16-byte alignment", who);
  SVK_REQUIRE((a->mid_o_stride_h % 4) == 0 && (a->mid_o_stride_s % 4) == 0 && (a->mid_o_stride_b % 4) == 0, SVK_ERR_LAYOUT,
              "%s: mid_o strides must keep 16-byte alignment", who);
  if (a->new_k != nullptr || a->new_v != nullptr) {
    SVK_REQUIRE(a->new_k != nullptr && a->new_v != nullptr && a->slot_mapping != nullptr, SVK_ERR_VALUE,
                "%s: the fused store needs new_k, new_v and slot_mapping together", who);
    SVK_REQUIRE((a->new_stride_b % 8) == 0 && (a->new_stride_h % 8) == 0 && (reinterpret_cast<uintptr_t>(a->new_k) % 16) == 0 &&
                    (reinterpret_cast<uintptr_t>(a->new_v) % 16) == 0,
                SVK_ERR_LAYOUT, "%s: new_k/new_v rows must be 16-byte aligned", who);
  }
  if (a->new_cos_sin != nullptr) {
    SVK_REQUIRE(a->score_mode == SVK_SCORE_NONE && a->slot_page_size == 0, SVK_ERR_VALUE,
                "%s: the rotated store rides in unscored launches over token slots only", who);
    SVK_REQUIRE(a->new_k != nullptr && a->raw_k_cache != nullptr && a->raw_v_cache != nullptr && a->new_slot_to_pos != nullptr,
                SVK_ERR_VALUE, "%s: the rotated store needs new_k / new_v / slot_mapping, raw_k_cache, raw_v_cache and new_slot_to_pos", who);
    SVK_REQUIRE((a->raw_slot_stride % 8) == 0 && (a->raw_head_stride % 8) == 0 && a->raw_num_slots > 0, SVK_ERR_LAYOUT,
                "%s: raw cache rows must be 16-byte aligned", who);
    SVK_REQUIRE(a->new_cos_dtype == SVK_DTYPE_F32 || a->new_cos_dtype == SVK_DTYPE_BF16 || a->new_cos_dtype == SVK_DTYPE_F16,
                SVK_ERR_VALUE, "%s: new_cos_dtype %d", who, a->new_cos_dtype);
  }
  if (a->direct_o != nullptr) {
    SVK_REQUIRE(a->max_len_in_batch <= a->block_seq, SVK_ERR_VALUE,
                "%s: direct_o needs one block per sequence (max_len_in_batch %d > block_seq %d)", who, a->max_len_in_batch, a->block_seq);
    SVK_REQUIRE((a->direct_stride_b % 8) == 0 && (a->direct_stride_h % 8) == 0 && (reinterpret_cast<uintptr_t>(a->direct_o) % 16) == 0,
                SVK_ERR_LAYOUT, "%s: direct_o rows must be 16-byte aligned", who);
  }
  return SVK_OK;
}

extern "C" int svk_flash_decode_stage1(const SvkFlashDecodeStage1Args* a, svk_stream_t stream) {
  using namespace svk;
  const int rc = validate_stage1(a, "svk_flash_decode_stage1");
  if (rc != SVK_OK) return rc;
  if (a->batch <= 0 || a->max_len_in_batch <= 0) return SVK_OK;
  const int G = a->num_q_heads / a->num_kv_heads;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->head_dim == 128 ? dispatch_group<128>(*a, G, s) : dispatch_group<64>(*a, G, s);
}

extern "C" int64_t svk_flash_decode_stage2_split_workspace_bytes(int32_t batch, int32_t num_q_heads, int32_t head_dim,
                                                                  int32_t max_partials) {
  using namespace svk;
  if (batch <= 0 || num_q_heads <= 0 || max_partials <= kSplitMin || max_partials > kSplitPPW * kSplitPPW) return 0;
  const int64_t n_bh = (int64_t)batch * num_q_heads, smax = split_groups(max_partials);
  return split_align(n_bh * smax * 4) + split_align(n_bh * smax * head_dim * 4);
}

extern "C" int svk_flash_decode_stage2(const SvkFlashDecodeStage2Args* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_flash_decode_stage2: null args");
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_LAYOUT,
              "svk_flash_decode_stage2: head_dim %d unsupported (64, 128)", a->head_dim);
  SVK_REQUIRE(a->block_seq > 0, SVK_ERR_VALUE, "svk_flash_decode_stage2: block_seq must be positive");
  SVK_REQUIRE(a->extra_partials >= 0 && a->extra_partials <= 8, SVK_ERR_VALUE, "svk_flash_decode_stage2: extra_partials %d out of range", a->extra_partials);
  SVK_REQUIRE((a->mid_o_stride_h % 4) == 0 && (a->mid_o_stride_s % 4) == 0 && (a->mid_o_stride_b % 4) == 0 &&
                  (a->o_stride_b % 4) == 0 && (a->o_stride_h % 4) == 0,
              SVK_ERR_LAYOUT, "svk_flash_decode_stage2: strides must be multiples of 4 elements (16-byte partial rows)");
  if (a->batch <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(a->batch, a->num_q_heads);
  // launches that may merge more than 128 partials per row get 1024 threads per (b, head) - 32 lane groups x 8
  // prefetched rows = 256 partials in ONE round trip, where 256 threads (8 groups) walk what lies beyond their first 64
  // in further trips (B=1 rows in 32-token blocks have 132 partials: 0.368 -> 0.355 ms per step; at 69 partials the
  // 1024-thread form is the slower one - its last lane group adds up 31 others).  The count is the LAUNCH's
  // (max_partials), not the workspace's capacity: the two forms add in different orders, so the output bits of a step
  // must not depend on what an earlier, longer launch grew the shared workspace to.
  SVK_REQUIRE(a->max_partials >= 0, SVK_ERR_VALUE, "svk_flash_decode_stage2: max_partials %d must be >= 0", a->max_partials);
  if (a->split_ws != nullptr && a->split_tickets != nullptr && a->max_partials > kSplitMin && a->max_partials <= kSplitPPW * kSplitPPW &&
      a->split_ws_bytes >= svk_flash_decode_stage2_split_workspace_bytes(a->batch, a->num_q_heads, a->head_dim, a->max_partials)) {
    SVK_REQUIRE((reinterpret_cast<uintptr_t>(a->split_ws) % 256) == 0, SVK_ERR_LAYOUT,
                "svk_flash_decode_stage2: split_ws must be 256-byte aligned");
    dim3 grid3(a->batch, a->num_q_heads, split_groups(a->max_partials));
    if (a->head_dim == 128) hipLaunchKernelGGL((decode_stage2_split_kernel<128>), grid3, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((decode_stage2_split_kernel<64>), grid3, dim3(256), 0, s, *a);
    return check_launch("svk_flash_decode_stage2");
  }
  const bool wide = (a->max_partials > 0 ? (int64_t)a->max_partials : a->mid_lse_stride_h) > 128;
  if (a->head_dim == 128) {
    if (wide) hipLaunchKernelGGL((decode_stage2_kernel<128, 1024>), grid, dim3(1024), 0, s, *a);
    else hipLaunchKernelGGL((decode_stage2_kernel<128, 256>), grid, dim3(256), 0, s, *a);
  } else {
    if (wide) hipLaunchKernelGGL((decode_stage2_kernel<64, 1024>), grid, dim3(1024), 0, s, *a);
    else hipLaunchKernelGGL((decode_stage2_kernel<64, 256>), grid, dim3(256), 0, s, *a);
  }
  return check_launch("svk_flash_decode_stage2");
}
