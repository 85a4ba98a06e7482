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
//   * P.V on the vector ALUs: P (rounded to bf16 like `exp_logic.to(v.dtype)`) is
//     re-distributed through a 1 KiB per-wave LDS tile, V rows are read with fully
//     coalesced 16 B/lane loads and accumulated in fp32.

#include <stdlib.h>
#include <type_traits>

#include "svk_common.hpp"
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

template <int D, int G>
__global__ void __launch_bounds__(512)
decode_stage1_kernel_v1(const SvkFlashDecodeStage1Args a) {
  using C = Stage1Cfg<D, G>;
  constexpr int NC = C::NC, JQ = C::JQ, PH = C::PH, DC = C::DC, TQ = C::TQ, NV = C::NV;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // kv head of this wave
  const int Hkv = a.num_kv_heads;
  const int b = blockIdx.y;
  const int blk = blockIdx.x;
  const int n = lane & 15;       // token column inside a 16-token MFMA group
  const int jq = lane >> 4;      // accumulator row group / k chunk
  const int dc = lane % DC;      // V: 16-byte chunk of the head row
  const int tq = lane / DC;      // V: token inside a wave-load
  const int score_mode = a.score_mode;

  const int len = a.b_seqlen[b];
  const int start = blk * a.block_seq;
  const int end = min(len, start + a.block_seq);

  float* Pw = lds + w * C::WAVE_FLOATS;
  float* bc = Pw + C::P_FLOATS;                                   // 16 floats broadcast pad
  float* spart = lds + Hkv * C::WAVE_FLOATS;                       // [kScoreChunk][Hkv*JQ]
  const int SP = Hkv * JQ;

  float* mid_o = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)blk * a.mid_o_stride_s;
  float* mid_lse = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + blk;

  if (end <= start) {
    // empty block: neutral partial (gqa_flash_decoding_stage1.py:288-294)
    for (int h = 0; h < G; ++h) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h;
      for (int d = lane; d < D; d += 64) o[d] = 0.f;
      if (lane == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = -INFINITY;
    }
    return;
  }

  // ---- Q fragments (A operand): lane (m = n, k chunk jq) holds Q[head n][c*32 + jq*8 .. +8]
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

  const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;
  const uint16_t* kbase = a.k_cache + (int64_t)w * a.kv_head_stride + jq * 8;
  const uint16_t* vbase = a.v_cache + (int64_t)w * a.kv_head_stride + dc * 8;
  const float sm_scale = rsqrtf((float)D);

  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  float acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;

  // slot ids of the current tile (K: two 16-token groups; V: NV wave-loads)
  int sk[2], sv[NV];
  auto load_slots = [&](int t0, int (&k2)[2], int (&v2)[NV]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int t = t0 + g * 16 + n;
      k2[g] = (t < end) ? row[t] : 0;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int t = t0 + i * TQ + tq;
      v2[i] = (t < end) ? row[t] : 0;
    }
  };
  load_slots(start, sk, sv);

  for (int c0 = start; c0 < end; c0 += kScoreChunk) {
    const int c1 = min(end, c0 + kScoreChunk);
    for (int t0 = c0; t0 < c1; t0 += kTileTokens) {
      const bool full = (t0 + kTileTokens <= end);

      // ---- issue all K and V loads of this tile
      uint4 kr[2][NC];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint16_t* kp = kbase + (int64_t)sk[g] * a.kv_slot_stride;
#pragma unroll
        for (int c = 0; c < NC; ++c) kr[g][c] = *reinterpret_cast<const uint4*>(kp + c * 32);
      }
      uint4 vr[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i)
        vr[i] = *reinterpret_cast<const uint4*>(vbase + (int64_t)sv[i] * a.kv_slot_stride);

      // ---- prefetch the next tile's slot ids
      int skn[2], svn[NV];
      if (t0 + kTileTokens < end) load_slots(t0 + kTileTokens, skn, svn);
      else {
#pragma unroll
        for (int g = 0; g < 2; ++g) skn[g] = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) svn[i] = 0;
      }

      // ---- S = Q K^T  (rows = heads jq*4+r, col = token n)
      f32x4_t s[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        s[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kr[g][c]), s[g], 0, 0, 0);
      }
      bool tv[2];
      tv[0] = full || (t0 + n < end);
      tv[1] = full || (t0 + 16 + n < end);

      // ---- raw scores out
      if (score_mode == SVK_SCORE_PERHEAD) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int h = jq * 4 + r;
            if (h < G && tv[g])
              a.attn_score[(int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + t0 + g * 16 + n] = s[g][r];
          }
      } else if (score_mode == SVK_SCORE_HEADMAX) {
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

      // ---- online softmax (per head row; 16 tokens of a group live in one DPP row)
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
          alpha[r] = __expf(m[r] - nm);          // first tile: exp(-inf) = 0
          p[0][r] = __expf(x0 - nm);
          p[1][r] = __expf(x1 - nm);
          rescale |= (nm != m[r]);
        } else {
          alpha[r] = 1.f; p[0][r] = 0.f; p[1][r] = 0.f;
        }
        l[r] = l[r] * alpha[r] + row16_allsum(p[0][r] + p[1][r]);
        m[r] = hv ? nm : m[r];
      }

      // ---- P (bf16-rounded) -> per-wave LDS tile [token][head]
      if (jq < JQ) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float4 t = make_float4(bf16_round(p[g][0]), bf16_round(p[g][1]), bf16_round(p[g][2]), bf16_round(p[g][3]));
          *reinterpret_cast<float4*>(Pw + (g * 16 + n) * PH + jq * 4) = t;
        }
      }
      const bool any_rescale = __any(rescale);
      if (any_rescale && n == 0 && jq < JQ)
        *reinterpret_cast<float4*>(bc + jq * 4) = make_float4(alpha[0], alpha[1], alpha[2], alpha[3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      if (any_rescale) {
        float al[PH];
#pragma unroll
        for (int q4 = 0; q4 < JQ; ++q4) {
          float4 t = *reinterpret_cast<const float4*>(bc + q4 * 4);
          al[q4 * 4 + 0] = t.x; al[q4 * 4 + 1] = t.y; al[q4 * 4 + 2] = t.z; al[q4 * 4 + 3] = t.w;
        }
#pragma unroll
        for (int h = 0; h < G; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] *= al[h];
      }

      // ---- acc += P V
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        uint4 vv = vr[i];
        if (!full && (t0 + i * TQ + tq >= end)) vv = make_uint4(0, 0, 0, 0);
        float ph[PH];
#pragma unroll
        for (int q4 = 0; q4 < JQ; ++q4) {
          float4 t = *reinterpret_cast<const float4*>(Pw + (i * TQ + tq) * PH + q4 * 4);
          ph[q4 * 4 + 0] = t.x; ph[q4 * 4 + 1] = t.y; ph[q4 * 4 + 2] = t.z; ph[q4 * 4 + 3] = t.w;
        }
        float vf[8];
        vf[0] = bf16_lo(vv.x); vf[1] = bf16_hi(vv.x);
        vf[2] = bf16_lo(vv.y); vf[3] = bf16_hi(vv.y);
        vf[4] = bf16_lo(vv.z); vf[5] = bf16_hi(vv.z);
        vf[6] = bf16_lo(vv.w); vf[7] = bf16_hi(vv.w);
#pragma unroll
        for (int h = 0; h < G; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(ph[h], vf[e], acc[h][e]);
      }
      // the next tile's P stores must not overtake this tile's P reads
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

#pragma unroll
      for (int g = 0; g < 2; ++g) sk[g] = skn[g];
#pragma unroll
      for (int i = 0; i < NV; ++i) sv[i] = svn[i];
    }

    if (score_mode == SVK_SCORE_HEADMAX) {
      // one owner thread per token column: combine row groups and KV-head waves
      __syncthreads();
      for (int t = threadIdx.x; t < c1 - c0; t += blockDim.x) {
        float mx = -INFINITY;
        for (int j = 0; j < SP; ++j) mx = fmaxf(mx, spart[t * SP + j]);
        float* dst = a.attn_score + (int64_t)b * a.score_stride_b + c0 + t;
        *dst = fmaxf(*dst, mx);      // same combine as the reference's atomic_max
      }
      __syncthreads();
    }
  }

  // ---- epilogue: mid_lse = m + log(l);  mid_o = acc / l
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
  float lh[PH];
#pragma unroll
  for (int q4 = 0; q4 < JQ; ++q4) {
    float4 t = *reinterpret_cast<const float4*>(bc + q4 * 4);
    lh[q4 * 4 + 0] = t.x; lh[q4 * 4 + 1] = t.y; lh[q4 * 4 + 2] = t.z; lh[q4 * 4 + 3] = t.w;
  }
#pragma unroll
  for (int h = 0; h < G; ++h) {
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[h][e];
#pragma unroll
      for (int off = DC; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
      o8[e] = x / lh[h];
    }
    if (tq == 0) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + dc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
  }
}


// ---------------------------------------------------------------------------------------
// v2: software-pipelined tile loop.  Same math and layouts as v1, but HBM requests stay in
// flight while a wave computes:
//   top of tile i : issue V(i) loads (non-temporal: each V byte is read once per launch),
//                   prefetch slot ids (V slots of tile i+1, K slots of tile i+2)
//   after QK^T(i) : the K registers are dead -> re-issue them for K(i+1)
//   softmax(i), P.V(i) run under the K(i+1) loads; QK^T(i) ran under the V(i) loads.
// No extra VGPRs versus v1 (K is single-buffered, re-armed right after the MFMAs).
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int D, int G>
struct Stage1V2Lds {
  // P tile + 16-float pad + 64 slot ids + Q fragments (NC x 64 lanes x 16 B)
  static constexpr int WAVE_FLOATS = Stage1Cfg<D, G>::P_FLOATS + 16 + 64 + Stage1Cfg<D, G>::NC * 64 * 4;
};

template <bool NT>
__device__ __forceinline__ uint4 ld16(const char* p) {
  if (NT) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)));
  return *reinterpret_cast<const uint4*>(p);
}

template <int D, int G, int MODE, bool NTV, bool OFF32>
__global__ void __launch_bounds__(512)
decode_stage1_kernel_v2(const SvkFlashDecodeStage1Args a) {
  using C = Stage1Cfg<D, G>;
  constexpr int NC = C::NC, JQ = C::JQ, PH = C::PH, DC = C::DC, TQ = C::TQ, NV = C::NV;
  constexpr int WF = Stage1V2Lds<D, G>::WAVE_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  const int b = blockIdx.y;
  const int blk = blockIdx.x;
  const int n = lane & 15;
  const int jq = lane >> 4;
  const int dc = lane % DC;
  const int tq = lane / DC;
  constexpr int score_mode = MODE;

  const int len = a.b_seqlen[b];
  const int start = blk * a.block_seq;
  const int end = min(len, start + a.block_seq);

  // per-wave LDS: P tile | 16-float broadcast pad | 2 x 32 slot ids | Q fragments (lane-linear)
  float* Pw = lds + w * WF;
  float* bc = Pw + C::P_FLOATS;
  int* slot_lds = reinterpret_cast<int*>(bc + 16);
  uint4* q_lds = reinterpret_cast<uint4*>(bc + 16 + 64);
  float* spart = lds + Hkv * WF;
  const int SP = Hkv * JQ;

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

  const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;
  // Byte addressing.  OFF32: the whole K (V) tensor spans < 4 GiB, so a row address is the
  // wave-uniform tensor base (SGPR pair) + a 32-bit per-lane byte offset: one VGPR per address
  // and 32-bit integer math instead of 64-bit (frees ~20 VGPRs in the pipelined loop).
  const char* const kt = reinterpret_cast<const char*>(a.k_cache);
  const char* const vt = reinterpret_cast<const char*>(a.v_cache);
  const int64_t slot_bytes = a.kv_slot_stride * 2;
  const int64_t k_lane_bytes = ((int64_t)w * a.kv_head_stride + jq * 8) * 2;
  const int64_t v_lane_bytes = ((int64_t)w * a.kv_head_stride + dc * 8) * 2;
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
  auto fetch_slots = [&](int t0) -> int { return row[(uint32_t)min(t0 + (lane_id_fresh() & 31), end - 1)]; };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // ---- prologue
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
  float acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;

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
    uint4 vr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      vr[i] = ld16<NTV>(v_ptr(cur_slots[i * TQ + tq]));

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
      for (int g = 0; g < 2; ++g) {
        float4 t = make_float4(bf16_round(p[g][0]), bf16_round(p[g][1]), bf16_round(p[g][2]), bf16_round(p[g][3]));
        *reinterpret_cast<float4*>(Pw + (g * 16 + n) * PH + jq * 4) = t;
      }
    }
    const bool any_rescale = __any(rescale);
    if (any_rescale && n == 0 && jq < JQ)
      *reinterpret_cast<float4*>(bc + jq * 4) = make_float4(alpha[0], alpha[1], alpha[2], alpha[3]);
    wave_sync();

    if (any_rescale) {
      float al[PH];
#pragma unroll
      for (int q4 = 0; q4 < JQ; ++q4) {
        float4 t = *reinterpret_cast<const float4*>(bc + q4 * 4);
        al[q4 * 4 + 0] = t.x; al[q4 * 4 + 1] = t.y; al[q4 * 4 + 2] = t.z; al[q4 * 4 + 3] = t.w;
      }
#pragma unroll
      for (int h = 0; h < G; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] *= al[h];
    }

    // P.V with the P rows double-buffered through registers: the next token group's P is
    // read from LDS while the current one is multiplied.  The sched_barrier keeps the compiler
    // from hoisting all NV P reads to the top (that costs 7*NV live VGPRs and spills).
    float ph[2][PH];
    auto read_p = [&](int i, float (&dst)[PH]) {
#pragma unroll
      for (int q4 = 0; q4 < JQ; ++q4) {
        float4 t = *reinterpret_cast<const float4*>(Pw + (i * TQ + tq) * PH + q4 * 4);
        dst[q4 * 4 + 0] = t.x; dst[q4 * 4 + 1] = t.y; dst[q4 * 4 + 2] = t.z; dst[q4 * 4 + 3] = t.w;
      }
    };
    read_p(0, ph[0]);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i + 1 < NV) read_p(i + 1, ph[(i + 1) & 1]);
      uint4 vv = vr[i];
      if (!full && (t0 + i * TQ + tq >= end)) vv = make_uint4(0, 0, 0, 0);
      float vf[8];
      vf[0] = bf16_lo(vv.x); vf[1] = bf16_hi(vv.x);
      vf[2] = bf16_lo(vv.y); vf[3] = bf16_hi(vv.y);
      vf[4] = bf16_lo(vv.z); vf[5] = bf16_hi(vv.z);
      vf[6] = bf16_lo(vv.w); vf[7] = bf16_hi(vv.w);
#pragma unroll
      for (int h = 0; h < G; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(ph[i & 1][h], vf[e], acc[h][e]);
      __builtin_amdgcn_sched_barrier(0);
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
        dst[t] = fmaxf(dst[t], mx);
      }
      __syncthreads();
      c0 = t0 + kTileTokens;
    }
    t0 += kTileTokens;
    buf ^= 1;
  };
  while (t0 + kTileTokens < end) tile(std::true_type{});
  tile(std::false_type{});

  // ---- epilogue.  The lane id is laundered through an empty asm so that none of the output
  // addresses below can be hoisted above the tile loop (they would stay live across it and
  // push the loop over the 256-VGPR budget of 2 waves/SIMD).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int n_e = lane_e & 15, jq_e = lane_e >> 4, dc_e = lane_e % DC, tq_e = lane_e / DC;
  if (n_e == 0 && jq_e < JQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = jq_e * 4 + r;
      if (h < G) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = m[r] + __logf(l[r]);
    }
    *reinterpret_cast<float4*>(bc + jq_e * 4) = make_float4(l[0], l[1], l[2], l[3]);
  }
  wave_sync();
  float lh[PH];
#pragma unroll
  for (int q4 = 0; q4 < JQ; ++q4) {
    float4 t = *reinterpret_cast<const float4*>(bc + q4 * 4);
    lh[q4 * 4 + 0] = t.x; lh[q4 * 4 + 1] = t.y; lh[q4 * 4 + 2] = t.z; lh[q4 * 4 + 3] = t.w;
  }
#pragma unroll
  for (int h = 0; h < G; ++h) {
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[h][e];
#pragma unroll
      for (int off = DC; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
      o8[e] = x / lh[h];
    }
    if (tq_e == 0) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + dc_e * 8;
      *reinterpret_cast<float4*>(o) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
  }
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
// ---------------------------------------------------------------------------------------
// Single-launch H2O decode layer: every workgroup of a row takes a ticket after publishing its partials and token
// scores; the last one to arrive merges the split-KV partials of the row's heads (stage 2) and normalises /
// accumulates the row's token scores (the `h2o_decode_finish` work) - no second launch, no grid barrier.
// Release: each thread fences its stores at agent scope before the ticket; acquire: the finishing workgroup
// invalidates its vector L1 after the ticket, so the other CUs' partials are read from L2.
// ---------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void fused_row_finish(const SvkFlashDecodeStage1Args& a, const SvkH2oDecodeScoreArgs& fs, uint16_t* fo,
                                                 int64_t fo_stride_b, int64_t fo_stride_h, int32_t* tickets, int b, float* lds) {
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(&tickets[b], 1);
    s_last = (t == (int)gridDim.x - 1);
    if (s_last) tickets[b] = 0;                       // self-cleaning for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int len = a.b_seqlen[b];
  // ---- stage 2 (flash_decoding_stage2.py:19-46): one wave per q head, lane -> 2 (D=128) or 1 (D=64) dims; partials are
  //      fetched 8 at a time so the L2 round trips overlap instead of chaining
  const int nblk = len <= 0 ? 0 : (len + a.block_seq - 1) / a.block_seq;
  for (int h = w; h < a.num_q_heads; h += nw) {
    const float* mo = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)h * a.mid_o_stride_h;
    const float* ml = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + (int64_t)h * a.mid_lse_stride_h;
    const int d = D == 128 ? lane * 2 : lane;
    float sum = 0.f, mxl = -INFINITY, a0 = 0.f, a1 = 0.f;
    for (int i0 = 0; i0 < nblk; i0 += 8) {
      float t0[8], t1[8], tl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = min(i0 + j, nblk - 1);
        tl[j] = ml[i];
        if (D == 128) {
          const float2 tv = *reinterpret_cast<const float2*>(mo + (int64_t)i * a.mid_o_stride_s + d);
          t0[j] = tv.x; t1[j] = tv.y;
        } else {
          t0[j] = mo[(int64_t)i * a.mid_o_stride_s + d];
          t1[j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i0 + j < nblk) {
          const float nm = fmaxf(tl[j], mxl);
          const float os = __expf(mxl - nm);
          const float e = __expf(tl[j] - nm);
          a0 = a0 * os + e * t0[j];
          a1 = a1 * os + e * t1[j];
          sum = sum * os + e;
          mxl = nm;
        }
      }
    }
    uint16_t* o = fo + (int64_t)b * fo_stride_b + (int64_t)h * fo_stride_h + d;
    if (D == 128) *reinterpret_cast<uint32_t*>(o) = f32_to_bf16_bits(a0 / sum) | (f32_to_bf16_bits(a1 / sum) << 16);
    else *o = (uint16_t)f32_to_bf16_bits(a0 / sum);
  }
  // ---- token scores: x *= scale; softmax over the full width; cum = pad(prev, 1) + p  (sparse_controller.py:762-767,
  //      h2o.py:957-1038).  The row is held in registers when it fits (<= 32 elements per thread).
  float* red = lds;                                   // the tile loop is over: reuse the dynamic LDS
  float* x = fs.attn_score + (int64_t)b * fs.score_stride_b;
  const int W = fs.width;
  float* cum = (fs.cum_score != nullptr && !(fs.b_new_slot != nullptr && fs.b_new_slot[b] < 0))   // padded lanes: no update
                   ? fs.cum_score + (int64_t)fs.b_req_idx[b] * fs.cum_stride : nullptr;
  const int nt = blockDim.x;
  if (W <= 32 * nt) {
    float v[32], c[32];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int t = threadIdx.x + i * nt;
      v[i] = t < W ? mul_rn(x[t], fs.scale) : -INFINITY;
      c[i] = (cum != nullptr && t < len - 1) ? cum[t] : 0.f;
      mx = fmaxf(mx, v[i]);
    }
    mx = block_allmax(mx, red);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      v[i] = expf(v[i] - mx);
      sum += v[i];
    }
    sum = block_allsum(sum, red);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int t = threadIdx.x + i * nt;
      if (t < W) {
        const float p = v[i] / sum;
        x[t] = p;
        if (cum != nullptr && t < len) cum[t] = c[i] + p;         // pad(prev, 1): the newest position starts from 0
      }
    }
    return;
  }
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < W; t += nt) mx = fmaxf(mx, mul_rn(x[t], fs.scale));
  mx = block_allmax(mx, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < W; t += nt) sum += expf(mul_rn(x[t], fs.scale) - mx);
  sum = block_allsum(sum, red);
  for (int t = threadIdx.x; t < W; t += nt) {
    const float p = expf(mul_rn(x[t], fs.scale) - mx) / sum;
    x[t] = p;
    if (cum != nullptr && t < len) cum[t] = (t == len - 1) ? p : cum[t] + p;
  }
}

// One token-score row of the PREVIOUS layer inside this layer's stage-1 launch (svk_flash_decode_stage1_deferred):
// x *= scale; softmax over the full width; cum = pad(prev, 1) + p (sparse_controller.py:762-767, h2o.py:957-1038) - the
// arithmetic of h2o_decode_score_kernel, row held in registers when it fits.
__device__ __forceinline__ void deferred_score_row(const SvkH2oDecodeScoreArgs& fs, int b, float* red) {
  float* x = fs.attn_score + (int64_t)b * fs.score_stride_b;
  const int W = fs.width;
  const int nt = blockDim.x;
  float* cum = nullptr;
  int len = 0;
  if (fs.cum_score != nullptr && !(fs.b_new_slot != nullptr && fs.b_new_slot[b] < 0)) {   // padded graph lanes: no update
    cum = fs.cum_score + (int64_t)fs.b_req_idx[b] * fs.cum_stride;
    len = fs.b_seqlen[b];
  }
  if (W <= 32 * nt) {
    float v[32], c[32];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int t = threadIdx.x + i * nt;
      v[i] = t < W ? mul_rn(x[t], fs.scale) : -INFINITY;
      c[i] = (cum != nullptr && t < len - 1) ? cum[t] : 0.f;
      mx = fmaxf(mx, v[i]);
    }
    mx = block_allmax(mx, red);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      v[i] = expf(v[i] - mx);
      sum += v[i];
    }
    sum = block_allsum(sum, red);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int t = threadIdx.x + i * nt;
      if (t < W) {
        const float p = v[i] / sum;
        x[t] = p;
        if (cum != nullptr && t < len) cum[t] = (t == len - 1) ? p : c[i] + p;
      }
    }
    return;
  }
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < W; t += nt) mx = fmaxf(mx, mul_rn(x[t], fs.scale));
  mx = block_allmax(mx, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < W; t += nt) sum += expf(mul_rn(x[t], fs.scale) - mx);
  sum = block_allsum(sum, red);
  for (int t = threadIdx.x; t < W; t += nt) {
    const float p = expf(mul_rn(x[t], fs.scale) - mx) / sum;
    x[t] = p;
    if (cum != nullptr && t < len) cum[t] = (t == len - 1) ? p : cum[t] + p;
  }
}

template <int D, int G, int MODE, bool NTV, bool OFF32, bool FUSED>
__global__ void __launch_bounds__(512)
decode_stage1_kernel_v3(const SvkFlashDecodeStage1Args a, const SvkH2oDecodeScoreArgs fs, uint16_t* fo, int64_t fo_stride_b,
                        int64_t fo_stride_h, int32_t* tickets) {
  using C = Stage1Cfg<D, G>;
  constexpr int NC = C::NC, JQ = C::JQ;
  constexpr int WF = Stage1V3Lds<D, G>::WAVE_FLOATS;
  constexpr int DW = D / 8;                       // 16-byte segments per head row
  extern __shared__ __attribute__((aligned(16))) float lds[];

  int b = blockIdx.y;
  if constexpr (!FUSED) {
    // the first gridDim.y - batch grid rows: the deferred score epilogue of the previous layer (one workgroup per row,
    // the others of that grid row leave at once).  It rides with this launch's 256+ streaming workgroups instead of
    // having its own latency-bound launch between two layers - and FIRST in dispatch order, so that its 8 us of
    // dependent latency run under the streaming work instead of trailing it (as the last grid rows it added 9 us to
    // a 176 us launch).
    const int n_def = (int)gridDim.y - a.batch;
    if (b < n_def) {
      if (blockIdx.x == 0) deferred_score_row(fs, b, lds);
      return;
    }
    b -= n_def;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  const int blk = blockIdx.x;
  const int n = lane & 15;
  const int jq = lane >> 4;
  const int dg = n % DW;                           // V: 16-byte segment (8 head dims) of this lane's MFMA column
  constexpr int score_mode = MODE;

  const int len = a.b_seqlen[b];
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
    if constexpr (FUSED) fused_row_finish<D>(a, fs, fo, fo_stride_b, fo_stride_h, tickets, b, lds);
    return;
  }

  if (a.new_k != nullptr && end == len) {
    // fused store_kvcache: this workgroup owns the lane's newest token.  Wave w writes its kv head's K and V rows
    // (2 x D/8 lanes, 16 bytes each) before any read of the row.  Only this wave reads that slot's head row in this
    // launch and the CU cannot hold a stale line of it (L1 is write-through, invalidated at kernel start), so a
    // workgroup-scope fence pair (= the store has completed) is enough; agent scope would write back the XCD's L2
    // (+10 us per launch measured).
    const int ns = a.slot_mapping[b];
    if (ns >= 0 && lane < 2 * DW) {
      const bool is_v = lane >= DW;
      const int seg = lane % DW;
      const uint16_t* src = (is_v ? a.new_v : a.new_k) + (int64_t)b * a.new_stride_b + (int64_t)w * a.new_stride_h + seg * 8;
      uint16_t* dst = const_cast<uint16_t*>(is_v ? a.v_cache : a.k_cache) + (int64_t)ns * a.kv_slot_stride +
                      (int64_t)w * a.kv_head_stride + seg * 8;
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
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
  auto fetch_slots = [&](int t0) -> int { return row[(uint32_t)min(t0 + (lane_id_fresh() & 31), end - 1)]; };
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
        dst[t] = fmaxf(dst[t], mx);
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
  if constexpr (FUSED) fused_row_finish<D>(a, fs, fo, fo_stride_b, fo_stride_h, tickets, b, lds);
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
#pragma unroll 4
  for (int i = g; i < nblk; i += GROUPS) {
    const float4 tv = *reinterpret_cast<const float4*>(mo + (int64_t)i * a.mid_o_stride_s);
    const float e = __expf(ml[i] - mx);
    acc.x += e * tv.x; acc.y += e * tv.y; acc.z += e * tv.z; acc.w += e * tv.w;
    sum += e;
  }
  if (nblk > 1) {
    if (d == 0) s_l[g] = sum;
    *reinterpret_cast<float4*>(&s_acc[g][d]) = acc;
    __syncthreads();
    if (g != 0) return;
    const int used = min(nblk, GROUPS);
    for (int j = 1; j < used; ++j) {
      const float4 tv = *reinterpret_cast<const float4*>(&s_acc[j][d]);
      acc.x += tv.x; acc.y += tv.y; acc.z += tv.z; acc.w += tv.w;
      sum += s_l[j];
    }
  } else if (g != 0) {
    return;
  }
  const uint32_t w0 = f32_to_bf16_bits(acc.x / sum) | (f32_to_bf16_bits(acc.y / sum) << 16);
  const uint32_t w1 = f32_to_bf16_bits(acc.z / sum) | (f32_to_bf16_bits(acc.w / sum) << 16);
  *reinterpret_cast<uint2*>(a.o + (int64_t)b * a.o_stride_b + (int64_t)h * a.o_stride_h + d) = make_uint2(w0, w1);
}

#include "decode_stage1_dma.hpp"

// SVK_STAGE1_VARIANT=1|2 selects the earlier kernels (A/B runs); 3 = default; 4 = the LDS-DMA kernel where it applies
// (Qwen2.5-7B heads: head_dim 128, GQA group 7, block_seq <= kV4MaxRange, KV tensors < 4 GiB), v3 elsewhere.  v4 is
// parity-green and streams at the same rate as v3 (both sit at the ~6.0-6.3 TB/s this gather reaches
// on the chip), so it stays opt-in.
inline int stage1_variant() {
  static const int variant = getenv("SVK_STAGE1_VARIANT") ? atoi(getenv("SVK_STAGE1_VARIANT")) : 3;
  return variant;
}
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// v4 launch: K ring depth S by how many workgroups share a CU (the ring lives in LDS); SVK_STAGE1_KSTAGES / SVK_STAGE1_KNT
// override the depth and the K cache policy for A/B runs.
template <int G, int MODE, int S, bool NT>
void launch_stage1_v4_one(const SvkFlashDecodeStage1Args& a, dim3 grid, dim3 block, size_t shm, hipStream_t stream) {
  auto kfn = decode_stage1_kernel_v4<G, MODE, S, 2, NT, true>;
  static bool attr_set = false;
  if (!attr_set) {      // more than 64 KiB of dynamic LDS needs the opt-in, once per kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, block, shm, stream, a);
}

template <int G, int MODE>
bool launch_stage1_v4(const SvkFlashDecodeStage1Args& a, dim3 grid, dim3 block, bool off32, hipStream_t stream) {
  using C = Stage1Cfg<128, G>;
  static const int kstages = env_int("SVK_STAGE1_KSTAGES", 0), knt = env_int("SVK_STAGE1_KNT", 1);
  if (!off32 || a.block_seq > kV4MaxRange) return false;
  const int range32 = ((a.block_seq + kTileTokens - 1) / kTileTokens) * kTileTokens;
  const long wgs = (long)grid.x * grid.y;
  int S = kstages >= 4 ? 4 : (kstages > 0 ? 2 : (wgs <= 256 ? 4 : 2));
  auto total = [&](int s_) { return Stage1V4Lds(range32, a.num_kv_heads, C::JQ, s_, MODE == SVK_SCORE_HEADMAX).total; };
  if (S == 4 && total(4) > 160 * 1024) S = 2;
  if (total(S) > 160 * 1024) return false;
  const size_t shm = (size_t)total(S);
  if (S == 4) {
    if (knt) launch_stage1_v4_one<G, MODE, 4, true>(a, grid, block, shm, stream);
    else launch_stage1_v4_one<G, MODE, 4, false>(a, grid, block, shm, stream);
  } else {
    if (knt) launch_stage1_v4_one<G, MODE, 2, true>(a, grid, block, shm, stream);
    else launch_stage1_v4_one<G, MODE, 2, false>(a, grid, block, shm, stream);
  }
  return true;
}

template <int D, int G>
int launch_stage1(const SvkFlashDecodeStage1Args& a, hipStream_t stream, const SvkH2oDecodeScoreArgs* deferred = nullptr) {
  using C = Stage1Cfg<D, G>;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  const bool carry = deferred != nullptr && stage1_variant() == 3;     // only v3 carries the deferred rows
  if (deferred != nullptr && !carry) {
    const int rc = svk_h2o_decode_score_update(deferred, stream);
    if (rc != SVK_OK) return rc;
  }
  dim3 grid(nblk, a.batch + (carry ? deferred->batch : 0));
  const SvkH2oDecodeScoreArgs dfs = carry ? *deferred : SvkH2oDecodeScoreArgs{};
  dim3 block(64 * a.num_kv_heads);
  const size_t score_floats = a.score_mode == SVK_SCORE_HEADMAX ? (size_t)kScoreChunk * a.num_kv_heads * C::JQ : 0;
  const size_t shm3 = sizeof(float) * ((size_t)a.num_kv_heads * Stage1V3Lds<D, G>::WAVE_FLOATS + score_floats);
  // 32-bit row offsets whenever the caller tells us the KV tensors span < 4 GiB
  const bool off32 = a.kv_num_slots > 0 && (a.kv_num_slots * a.kv_slot_stride * 2) < (int64_t)0xffffffffll;
  const int variant = stage1_variant();
  if constexpr (D == 128 && G == 7) {
    if (variant == 4) {
      bool done = false;
      if (a.score_mode == SVK_SCORE_HEADMAX) done = launch_stage1_v4<G, SVK_SCORE_HEADMAX>(a, grid, block, off32, stream);
      else if (a.score_mode == SVK_SCORE_PERHEAD) done = launch_stage1_v4<G, SVK_SCORE_PERHEAD>(a, grid, block, off32, stream);
      else done = launch_stage1_v4<G, SVK_SCORE_NONE>(a, grid, block, off32, stream);
      if (done) return check_launch("svk_flash_decode_stage1");
    }
  }
  if (variant == 1 || (variant == 2 && G == 8)) {
    const size_t shm1 = sizeof(float) * ((size_t)a.num_kv_heads * C::WAVE_FLOATS + score_floats);
    hipLaunchKernelGGL((decode_stage1_kernel_v1<D, G>), grid, block, shm1, stream, a);
    return check_launch("svk_flash_decode_stage1");
  }
  if (variant == 2) {
    if constexpr (G != 8) {
      const size_t shm2 = sizeof(float) * ((size_t)a.num_kv_heads * Stage1V2Lds<D, G>::WAVE_FLOATS + score_floats);
#define SVK_LAUNCH_V2(MODE_)                                                                                      \
  do {                                                                                                            \
    if (off32) hipLaunchKernelGGL((decode_stage1_kernel_v2<D, G, MODE_, true, true>), grid, block, shm2, stream, a);  \
    else hipLaunchKernelGGL((decode_stage1_kernel_v2<D, G, MODE_, true, false>), grid, block, shm2, stream, a);       \
  } while (0)
      if (a.score_mode == SVK_SCORE_HEADMAX) SVK_LAUNCH_V2(SVK_SCORE_HEADMAX);
      else if (a.score_mode == SVK_SCORE_PERHEAD) SVK_LAUNCH_V2(SVK_SCORE_PERHEAD);
      else SVK_LAUNCH_V2(SVK_SCORE_NONE);
#undef SVK_LAUNCH_V2
    }
    return check_launch("svk_flash_decode_stage1");
  }
#define SVK_LAUNCH_V3(MODE_)                                                                                      \
  do {                                                                                                            \
    if (off32) hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, MODE_, true, true, false>), grid, block, shm3, stream, a, dfs, (uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int32_t*)nullptr);  \
    else hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, MODE_, true, false, false>), grid, block, shm3, stream, a, dfs, (uint16_t*)nullptr, (int64_t)0, (int64_t)0, (int32_t*)nullptr);       \
  } while (0)
  if (a.score_mode == SVK_SCORE_HEADMAX) SVK_LAUNCH_V3(SVK_SCORE_HEADMAX);
  else if (a.score_mode == SVK_SCORE_PERHEAD) SVK_LAUNCH_V3(SVK_SCORE_PERHEAD);
  else SVK_LAUNCH_V3(SVK_SCORE_NONE);
#undef SVK_LAUNCH_V3
  return check_launch("svk_flash_decode_stage1");
}

template <int D, int G>
int launch_fused(const SvkH2oDecodeFusedArgs& f, hipStream_t stream) {
  using C = Stage1Cfg<D, G>;
  const SvkFlashDecodeStage1Args& a = f.stage1;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  dim3 grid(nblk, a.batch), block(64 * a.num_kv_heads);
  const size_t shm3 = sizeof(float) * ((size_t)a.num_kv_heads * Stage1V3Lds<D, G>::WAVE_FLOATS + (size_t)kScoreChunk * a.num_kv_heads * C::JQ);
  const bool off32 = a.kv_num_slots > 0 && (a.kv_num_slots * a.kv_slot_stride * 2) < (int64_t)0xffffffffll;
  if (off32) hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, SVK_SCORE_HEADMAX, true, true, true>), grid, block, shm3, stream, a, f.score, f.o, f.o_stride_b, f.o_stride_h, f.tickets);
  else hipLaunchKernelGGL((decode_stage1_kernel_v3<D, G, SVK_SCORE_HEADMAX, true, false, true>), grid, block, shm3, stream, a, f.score, f.o, f.o_stride_b, f.o_stride_h, f.tickets);
  return check_launch("svk_h2o_decode_fused");
}

template <int D>
int dispatch_fused(const SvkH2oDecodeFusedArgs& f, int G, hipStream_t stream) {
  switch (G) {
    case 1: return launch_fused<D, 1>(f, stream);
    case 2: return launch_fused<D, 2>(f, stream);
    case 3: return launch_fused<D, 3>(f, stream);
    case 4: return launch_fused<D, 4>(f, stream);
    case 5: return launch_fused<D, 5>(f, stream);
    case 6: return launch_fused<D, 6>(f, stream);
    case 7: return launch_fused<D, 7>(f, stream);
    case 8: return launch_fused<D, 8>(f, stream);
    default:
      set_error("svk_h2o_decode_fused: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
}

template <int D>
int dispatch_group(const SvkFlashDecodeStage1Args& a, int G, hipStream_t stream, const SvkH2oDecodeScoreArgs* deferred = nullptr) {
  switch (G) {
    case 1: return launch_stage1<D, 1>(a, stream, deferred);
    case 2: return launch_stage1<D, 2>(a, stream, deferred);
    case 3: return launch_stage1<D, 3>(a, stream, deferred);
    case 4: return launch_stage1<D, 4>(a, stream, deferred);
    case 5: return launch_stage1<D, 5>(a, stream, deferred);
    case 6: return launch_stage1<D, 6>(a, stream, deferred);
    case 7: return launch_stage1<D, 7>(a, stream, deferred);
    case 8: return launch_stage1<D, 8>(a, stream, deferred);
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
    SVK_REQUIRE(stage1_variant() >= 3, SVK_ERR_VALUE, "%s: the fused store is only built into stage-1 variants 3 and 4", who);
  }
  if (a->direct_o != nullptr) {
    SVK_REQUIRE(a->max_len_in_batch <= a->block_seq, SVK_ERR_VALUE,
                "%s: direct_o needs one block per sequence (max_len_in_batch %d > block_seq %d)", who, a->max_len_in_batch, a->block_seq);
    SVK_REQUIRE(stage1_variant() == 3, SVK_ERR_VALUE, "%s: direct_o is only built into stage-1 variant 3", who);
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

extern "C" int svk_flash_decode_stage1_deferred(const SvkFlashDecodeStage1Args* a, const SvkH2oDecodeScoreArgs* prev,
                                                svk_stream_t stream) {
  using namespace svk;
  if (prev == nullptr || prev->batch <= 0) return svk_flash_decode_stage1(a, stream);
  const int rc = validate_stage1(a, "svk_flash_decode_stage1_deferred");
  if (rc != SVK_OK) return rc;
  SVK_REQUIRE(prev->attn_score != nullptr && prev->width > 0, SVK_ERR_VALUE, "svk_flash_decode_stage1_deferred: bad score args");
  SVK_REQUIRE(prev->cum_score == nullptr || (prev->b_req_idx != nullptr && prev->b_seqlen != nullptr), SVK_ERR_VALUE,
              "svk_flash_decode_stage1_deferred: cum_score needs b_req_idx and b_seqlen");
  if (a->batch <= 0 || a->max_len_in_batch <= 0) return svk_h2o_decode_score_update(prev, stream);
  const int G = a->num_q_heads / a->num_kv_heads;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->head_dim == 128 ? dispatch_group<128>(*a, G, s, prev) : dispatch_group<64>(*a, G, s, prev);
}

extern "C" int svk_h2o_decode_fused(const SvkH2oDecodeFusedArgs* f, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(f != nullptr && f->o != nullptr && f->tickets != nullptr && f->score.attn_score != nullptr, SVK_ERR_VALUE,
              "svk_h2o_decode_fused: null args");
  const SvkFlashDecodeStage1Args* a = &f->stage1;
  int rc = validate_stage1(a, "svk_h2o_decode_fused");
  if (rc != SVK_OK) return rc;
  SVK_REQUIRE(a->score_mode == SVK_SCORE_HEADMAX && a->attn_score == f->score.attn_score && a->score_stride_b == f->score.score_stride_b,
              SVK_ERR_VALUE, "svk_h2o_decode_fused: stage 1 must write the head-max scores the finish step normalises");
  SVK_REQUIRE(f->score.batch == a->batch && f->score.width > 0, SVK_ERR_VALUE, "svk_h2o_decode_fused: score batch/width mismatch");
  SVK_REQUIRE(f->score.cum_score == nullptr || f->score.b_req_idx != nullptr, SVK_ERR_VALUE, "svk_h2o_decode_fused: cum_score needs b_req_idx");
  SVK_REQUIRE((f->o_stride_b % 2) == 0 && (f->o_stride_h % 2) == 0, SVK_ERR_LAYOUT, "svk_h2o_decode_fused: output strides must be even");
  if (a->batch <= 0 || a->max_len_in_batch <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int G = a->num_q_heads / a->num_kv_heads;
  return a->head_dim == 128 ? dispatch_fused<128>(*f, G, s) : dispatch_fused<64>(*f, G, s);
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
  // the lse row stride is the workspace's partial capacity: long-context workspaces get 1024 threads per (b, head)
  const bool wide = a->mid_lse_stride_h >= 256;
  if (a->head_dim == 128) {
    if (wide) hipLaunchKernelGGL((decode_stage2_kernel<128, 1024>), grid, dim3(1024), 0, s, *a);
    else hipLaunchKernelGGL((decode_stage2_kernel<128, 256>), grid, dim3(256), 0, s, *a);
  } else {
    if (wide) hipLaunchKernelGGL((decode_stage2_kernel<64, 1024>), grid, dim3(1024), 0, s, *a);
    else hipLaunchKernelGGL((decode_stage2_kernel<64, 256>), grid, dim3(256), 0, s, *a);
  }
  return check_launch("svk_flash_decode_stage2");
}
