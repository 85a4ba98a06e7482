// Prefill token scores on the matrix cores (gfx950).  Replaces kernels/triton/prefill_score.py.
//
// Row space of one (score range, KV head): row = g * Wpad + qi, g = head of the GQA group,
// qi = query inside the window, Wpad = window padded to 16 -> every 16-row MFMA tile belongs to one
// head.  A workgroup = 4 waves x 64 rows; it walks a 256-key range 16 keys at a time with
// register double-buffered B fragments (keys are loaded straight from the paged cache in the
// B-operand layout, as in decode stage 1).
//   probability: pass 1 keeps per-lane online (max, sum) for its 16 rows, reduces across the
//                16 key lanes once per range and writes partial stats; a tiny reduce kernel merges
//                the partials; pass 2 recomputes Q.K^T (cheaper than spilling 58 MB of logits),
//                forms probabilities, sums them over each head's queries and max-combines heads.
//   logits:      one pass, per-token max over rows.
// The only atomics are the final per-token max across workgroups (float max as integer max).

#include <algorithm>

#include "svk_common.hpp"
#include "lds_dma.hpp"

namespace svk {
namespace {

constexpr int kRowsPerWg = 256;
constexpr int kKeysPerWg = 256;
constexpr float kMasked = -1.0e20f;

__device__ __forceinline__ void atomic_max_nonneg(float* p, float v) {
  atomicMax(reinterpret_cast<int*>(p), __builtin_bit_cast(int, v));      // valid for v >= 0, *p >= 0
}
__device__ __forceinline__ void atomic_max_any(float* p, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __builtin_bit_cast(int, v));
  else atomicMin(reinterpret_cast<unsigned int*>(p), __builtin_bit_cast(unsigned int, v));
}

struct RangeInfo {
  int start_loc, cache_len, ctx_len, chunk_len, q_start, q_end, cand_end;
  const int32_t* row;
};

__device__ __forceinline__ RangeInfo load_range(const SvkPrefillScoreArgs& a, int i) {
  RangeInfo r;
  const int s = a.batch_indices ? a.batch_indices[i] : i;
  r.start_loc = a.b_start_loc[s];
  r.cache_len = a.b_prompt_cache_len[s];
  r.ctx_len = a.b_seq_len[s];
  r.chunk_len = r.ctx_len - r.cache_len;
  r.q_start = a.score_q_start[i];
  r.q_end = a.score_q_end[i];
  r.cand_end = min(max(a.candidate_start, r.ctx_len - a.num_recent_tokens), a.score_cols);
  r.row = a.req_to_tokens + (int64_t)a.b_req_idx[s] * a.req_stride;
  return r;
}

// PASS: 0 = partial stats, 1 = final probabilities, 2 = logits
template <int D, int PASS>
__global__ void __launch_bounds__(256) prefill_score_kernel(const SvkPrefillScoreArgs a, int G, int Wpad, int q_limit,
                                                            int RB, int NKB, float* part_m, float* part_l,
                                                            const float* glob_m, const float* glob_l) {
  constexpr int NC = D / 32;
  __shared__ float tsum[16][kKeysPerWg];       // per row tile column sums / maxima
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, jq = lane >> 4;
  const int kb = blockIdx.x;
  int grp = blockIdx.y;
  const int rb = grp % RB; grp /= RB;
  const int h = grp % a.num_kv_heads;
  const int i = grp / a.num_kv_heads;
  const RangeInfo R = load_range(a, i);
  const int k0 = kb * kKeysPerWg;
  const int key_lo = max(k0, a.candidate_start), key_hi = min(k0 + kKeysPerWg, R.cand_end);
  const int rows_total = G * Wpad;
  const int wrow0 = rb * kRowsPerWg + w * 64;          // first row of this wave
  const float sm_scale = rsqrtf((float)D);

  // absolute query position of the 16 rows this lane sees in the C layout (-1 = masked row)
  int rowpos[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wrow0 + rt * 16 + jq * 4 + r;
      const int qi = row % Wpad;
      const int qa = R.q_start + qi;
      const int rel = qa - R.cache_len;
      const bool ok = row < rows_total && qi < q_limit && qa < R.q_end && rel >= 0 && rel < R.chunk_len;
      rowpos[rt][r] = ok ? qa : -1;
    }
  // A fragments: lane (m = n, k chunk jq) holds Q[row wrow0 + rt*16 + n][c*32 + jq*8 ..]
  bf16x8_t qa_frag[4][NC];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int row = wrow0 + rt * 16 + n;
    const int g = row / Wpad, qi = row % Wpad;
    const int qabs = R.q_start + qi;
    const int rel = qabs - R.cache_len;
    const bool ok = row < rows_total && qi < q_limit && qabs < R.q_end && rel >= 0 && rel < R.chunk_len;
    const uint16_t* qp = a.q + (int64_t)(R.start_loc + rel) * a.q_stride_t + (int64_t)(h * G + g) * a.q_stride_h + jq * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (ok) t = *reinterpret_cast<const uint4*>(qp + c * 32);
      qa_frag[rt][c] = __builtin_bit_cast(bf16x8_t, t);
    }
  }
  float gm[4][4], gl[4][4];      // pass 1: running stats; pass 2: global stats of the row
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PASS == 1) {
        const int row = rb * kRowsPerWg + w * 64 + rt * 16 + jq * 4 + r;
        const int64_t o = ((int64_t)(i * a.num_kv_heads + h) * RB + rb) * kRowsPerWg + (row - rb * kRowsPerWg);
        gm[rt][r] = glob_m[o];
        const float l = glob_l[o];
        gl[rt][r] = l > 0.f ? l : 1.f;
      } else {
        gm[rt][r] = kMasked;
        gl[rt][r] = 0.f;
      }
    }
  if (PASS != 0) {
    for (int t = threadIdx.x; t < 16 * kKeysPerWg; t += blockDim.x) (&tsum[0][0])[t] = PASS == 2 ? -INFINITY : 0.f;
    __syncthreads();
  }

  const uint16_t* kbase = a.k_cache + (int64_t)h * a.kv_head_stride + jq * 8;
  auto load_keys = [&](int t0, uint4 (&kr)[NC]) {
    const int t = t0 + n;
    const bool in = t >= key_lo && t < key_hi;
    const int slot = in ? R.row[t] : 0;
    const uint16_t* kp = kbase + (int64_t)slot * a.kv_slot_stride;
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[c] = *reinterpret_cast<const uint4*>(kp + c * 32);
  };
  if (key_lo < key_hi) {
    const int t_first = (key_lo / 16) * 16;
    uint4 kcur[NC], knxt[NC];
    load_keys(t_first, kcur);
    for (int t0 = t_first; t0 < key_hi; t0 += 16) {
      if (t0 + 16 < key_hi) load_keys(t0 + 16, knxt);
      const int t = t0 + n;
      const bool kin = t >= key_lo && t < key_hi;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa_frag[rt][c], __builtin_bit_cast(bf16x8_t, kcur[c]), s, 0, 0, 0);
        if (PASS == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = kin && rowpos[rt][r] >= t;
            const float x = valid ? s[r] * sm_scale : kMasked;
            const float nm = fmaxf(gm[rt][r], x);
            gl[rt][r] = gl[rt][r] * __expf(gm[rt][r] - nm) + (valid ? __expf(x - nm) : 0.f);
            gm[rt][r] = nm;
          }
        } else if (PASS == 1) {
          float colsum = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = kin && rowpos[rt][r] >= t;
            colsum += valid ? __expf(s[r] * sm_scale - gm[rt][r]) / gl[rt][r] : 0.f;
          }
          colsum += __shfl_xor(colsum, 16, 64);
          colsum += __shfl_xor(colsum, 32, 64);
          if (jq == 0 && kin) tsum[w * 4 + rt][t - k0] = colsum;
        } else {
          float colmax = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = kin && rowpos[rt][r] >= t;
            colmax = fmaxf(colmax, valid ? s[r] : kMasked);
          }
          colmax = fmaxf(colmax, __shfl_xor(colmax, 16, 64));
          colmax = fmaxf(colmax, __shfl_xor(colmax, 32, 64));
          if (jq == 0 && kin) tsum[w * 4 + rt][t - k0] = colmax;
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) kcur[c] = knxt[c];
    }
  }

  if (PASS == 0) {
    // merge the 16 key lanes of every row: (m, l) pairs combine as M = max m, L = sum l*exp(m-M)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float M = row16_allmax(gm[rt][r]);
        const float L = row16_allsum(gl[rt][r] * __expf(gm[rt][r] - M));
        if (n == 0) {
          const int row_in_wg = w * 64 + rt * 16 + jq * 4 + r;
          const int64_t o = (((int64_t)(i * a.num_kv_heads + h) * RB + rb) * NKB + kb) * kRowsPerWg + row_in_wg;
          part_m[o] = M;
          part_l[o] = L;
        }
      }
    return;
  }
  __syncthreads();
  // one thread per key column: combine the row tiles of each head, then heads, then publish
  const int tiles_per_head = Wpad / 16;
  for (int c = threadIdx.x; c < kKeysPerWg; c += blockDim.x) {
    const int t = k0 + c;
    if (t < key_lo || t >= key_hi) continue;
    float* dst = a.attn_score + (int64_t)i * a.score_stride + t;
    if (PASS == 2) {
      float mx = -INFINITY;
      for (int tl = 0; tl < 16; ++tl)
        if (rb * kRowsPerWg + tl * 16 < rows_total) mx = fmaxf(mx, tsum[tl][c]);
      atomic_max_any(dst, mx);
    } else {
      const float inv_len = 1.f / (float)max(R.q_end - R.q_start, 1);
      // tiles of this workgroup that belong to head g: global tile index in [g*tph, (g+1)*tph)
      const int tile0 = rb * 16;
      float best = 0.f;
      for (int tl = 0; tl < 16;) {
        const int gt = tile0 + tl;
        if (gt * 16 >= rows_total) break;
        const int g = gt / tiles_per_head;
        float hs = 0.f;
        while (tl < 16 && (tile0 + tl) / tiles_per_head == g && (tile0 + tl) * 16 < rows_total) hs += tsum[tl++][c];
        // NOTE: a head whose tiles straddle two workgroups (Wpad = 128 with an odd row-block
        // boundary never happens: 256 % Wpad == 0) is always complete here
        best = fmaxf(best, hs * inv_len);
      }
      atomic_max_nonneg(dst, best);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// v2 (head_dim 128, GQA group <= 8, window <= 128): the path prefill_score_fwd takes for the Qwen / Llama shapes.
//   * workgroup = (128-key block, score range, KV head); wave g = query head g of the GQA group, walking the head's
//     window in blocks of 32 query rows.  The K tile (128 keys x 256 B) is fetched ONCE per workgroup into LDS by
//     `global_load_lds_dwordx4` (v1: every wave loaded its own B fragments from the paged cache, 16 keys at a time), chunks
//     swizzled on the DMA source address so that the fragment reads are conflict-free.
//   * 32x32x16 MFMA in the orientation that makes the reduction of the pass lane-local:
//       pass 0 (softmax statistics of a query row over the keys): S^T = K Q^T, a lane holds 16 keys of ONE query row -
//         block maximum + one exp2 per logit (v1: per-lane online pairs with two exps per logit);
//       pass 1 / logits (sum / maximum over the query rows of a key): S = Q K^T, a lane holds 16 query rows of ONE key -
//         the per-key accumulator is a register, no shuffles; one multiply-add, one exp2, one multiply-add per logit.
//   * statistics travel between the passes in the base-2 domain (u = s D^-1/2 log2 e) and as 1 / l.
// ------------------------------------------------------------------------------------------------
constexpr int kPsKeys = 128;
constexpr int kPsRowB = 256;

__device__ __forceinline__ float ps_max16(const f32x16_t& x) {
  return vmax(vmax3(vmax3(x[0], x[1], x[2]), vmax3(x[3], x[4], x[5]), x[15]),
              vmax3(vmax3(x[6], x[7], x[8]), vmax3(x[9], x[10], x[11]), vmax3(x[12], x[13], x[14])));
}

template <int PASS>
__global__ void __launch_bounds__(512, 4) prefill_score_kernel_v2(const SvkPrefillScoreArgs a, int G, int Wpad32, int q_limit, int NKB,
                                                               float* part_m, float* part_l, const float* glob_ml) {
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // K tile [128][256 B] | hs[G][128] f32
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);               // head g of the GQA group
  const int lq = lane & 31, half = lane >> 5;
  const int kb = blockIdx.x;
  const int h = blockIdx.y % a.num_kv_heads, i = blockIdx.y / a.num_kv_heads;
  const RangeInfo R = load_range(a, i);
  const int k0 = kb * kPsKeys;
  const int key_lo = max(k0, a.candidate_start), key_hi = min(k0 + kPsKeys, R.cand_end);
  const int ROWS = G * Wpad32;
  const int nrb = Wpad32 / 32;
  const int64_t grp = (int64_t)i * a.num_kv_heads + h;
  float* hs = reinterpret_cast<float*>(lds_raw + kPsKeys * kPsRowB);
  if (PASS == 0 && h == 0) {
    // the score row starts from 0 (pass 1 publishes by atomic max): this range's workgroup of KV head 0 clears its
    // 128 columns here, so the probability mode needs no fill launch
    float* dst = a.attn_score + (int64_t)i * a.score_stride;
    for (int c = k0 + (int)threadIdx.x; c < min(k0 + kPsKeys, a.score_cols); c += blockDim.x) dst[c] = 0.f;
  }
  if (key_lo >= key_hi) {
    if (PASS == 0 && half == 0)
      for (int rb = 0; rb < nrb; ++rb) {
        const int64_t o = (grp * NKB + kb) * ROWS + w * Wpad32 + rb * 32 + lq;
        part_m[o] = kMasked;
        part_l[o] = 0.f;
      }
    return;
  }
  // Q fragments (lane = (row lq of a 32-row block, 8-element half of every 16-wide k-step); the same registers serve as
  // the B operand of K Q^T and the A operand of Q K^T) and, for pass 1, the row constants m + log2 l in the
  // "register = row" layout of the accumulator.  The first block is requested before the K tile lands.  (Requesting
  // every block one ahead costs 32 more registers and the second resident workgroup per CU with them: 28.6 -> 40.8 us
  // in the logits mode.)
  bf16x8_t qn[D / 16];
  float rn[16];
  auto load_block = [&](int rb) __attribute__((always_inline)) {
    const int qi = rb * 32 + lq;
    const int qabs = R.q_start + qi;
    const int rel = qabs - R.cache_len;
    const bool ok = qi < q_limit && qabs < R.q_end && rel >= 0 && rel < R.chunk_len;
    const uint16_t* qp = a.q + (int64_t)(R.start_loc + (ok ? rel : 0)) * a.q_stride_t + (int64_t)(h * G + w) * a.q_stride_h + half * 8;
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (ok) t = *reinterpret_cast<const uint4*>(qp + ds * 16);
      qn[ds] = __builtin_bit_cast(bf16x8_t, t);
    }
    if (PASS == 1) {
      const float* gm = glob_ml + grp * ROWS + w * Wpad32 + rb * 32 + 4 * half;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 x = *reinterpret_cast<const float4*>(gm + 8 * j);
        rn[4 * j] = x.x; rn[4 * j + 1] = x.y; rn[4 * j + 2] = x.z; rn[4 * j + 3] = x.w;
      }
    }
  };
  load_block(0);
  // ---- K tile -> LDS: 32 DMA instructions (4 keys each) = 8 groups of four behind one M0 write, dealt to the waves
  {
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw);
    const char* kt = reinterpret_cast<const char*>(a.k_cache) + (int64_t)h * a.kv_head_stride * 2;
    const int64_t slot_bytes = a.kv_slot_stride * 2;
    const int lane_row = lane >> 4, lane_pos = lane & 15;
    for (int g8 = w; g8 < 8; g8 += G) {
      int slot[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) slot[j] = R.row[min(k0 + 16 * g8 + 4 * j + lane_row, key_hi - 1)];
      const char* src[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        src[j] = kt + ((int64_t)slot[j] * slot_bytes + (int64_t)((lane_pos ^ ((4 * j + lane_row) & 15)) << 4) - 1024 * j);
      pa_dma4x16(src, lds0 + g8 * 4096);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const float c2 = rsqrtf((float)D) * 1.4426950408889634f;                      // logits -> base-2 exponent domain
  const uint32_t kfrag0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + lq * kPsRowB;
  float acc[4];                                                                 // pass 1 / 2: per key sub-block, lane = key
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) acc[sb] = PASS == 2 ? -INFINITY : 0.f;

  for (int rb = 0; rb < nrb; ++rb) {
    // query row rb*32 + lq of head w (lanes lq and lq + 32 hold the two 8-element halves of every 16-wide k-step)
    const int qi = rb * 32 + lq;
    const int qabs = R.q_start + qi;
    const int rel = qabs - R.cache_len;
    const bool ok = qi < q_limit && qabs < R.q_end && rel >= 0 && rel < R.chunk_len;
    const bool rows_ok = __all(ok);
    const int minpos = R.q_start + rb * 32;                                     // position of the block's first row
    if (rb > 0) load_block(rb);
    bf16x8_t (&qf)[D / 16] = qn;
    float (&rm)[16] = rn;
    float m2 = kMasked, l = 0.f;                                                // pass 0: this lane's (row, key half) pair
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const int t0 = k0 + sb * 32;
      if (t0 + 32 <= key_lo || t0 >= key_hi) continue;
      pa_u32x4_t kf[D / 16];
#pragma unroll
      for (int ds = 0; ds < D / 16; ++ds)
        kf[ds] = *reinterpret_cast<const __attribute__((address_space(3))) pa_u32x4_t*>(kfrag0 + sb * 32 * kPsRowB + (((ds * 2 + half) ^ (lq & 15)) << 4));
#pragma unroll
      for (int ds = 0; ds < D / 16; ++ds) asm volatile("" : "+v"(kf[ds]));     // one batch of reads, one wait
      f32x16_t s;
#pragma unroll
      for (int ds = 0; ds < D / 16; ++ds) {
        if (PASS == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[ds]), qf[ds], ds == 0 ? f32x16_t{} : s, 0, 0, 0);
        else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ds], __builtin_bit_cast(bf16x8_t, kf[ds]), ds == 0 ? f32x16_t{} : s, 0, 0, 0);
      }
      // every logit of the block is live: keys inside the candidate range, rows real, all keys at or before the rows
      const bool fast = t0 >= key_lo && t0 + 32 <= key_hi && rows_ok && t0 + 31 <= minpos;
      if (PASS == 0) {
        // register r = key t0 + (r&3) + 8(r>>2) + 4 half of query row lq
        if (!fast) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool valid = ok && t >= key_lo && t < key_hi && qabs >= t;
            s[r] = valid ? s[r] * c2 : kMasked;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] *= c2;
        }
        const float nm = vmax(m2, ps_max16(s));
        float psum = 0.f;
        if (!fast) {
#pragma unroll
          for (int r = 0; r < 16; ++r) psum += s[r] > 0.5f * kMasked ? __builtin_amdgcn_exp2f(s[r] - nm) : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(s[r] - nm);
        }
        l = l * __builtin_amdgcn_exp2f(m2 - nm) + psum;
        m2 = nm;
      } else {
        // lane = key t0 + lq, register r = query row rb*32 + (r&3) + 8(r>>2) + 4 half
        const int t = t0 + lq;
        const bool kin = t >= key_lo && t < key_hi;
        if (PASS == 1) {
          float cs = 0.f;
          if (fast) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -rm[r]));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int qr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              const int qa_r = R.q_start + qr, rel_r = qa_r - R.cache_len;
              const bool valid = kin && qr < q_limit && qa_r < R.q_end && rel_r >= 0 && rel_r < R.chunk_len && qa_r >= t;
              cs += valid ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -rm[r])) : 0.f;
            }
          }
          acc[sb] += cs;
        } else {
          float cm = -INFINITY;
          if (fast) {
            cm = ps_max16(s);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int qr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              const int qa_r = R.q_start + qr, rel_r = qa_r - R.cache_len;
              const bool valid = kin && qr < q_limit && qa_r < R.q_end && rel_r >= 0 && rel_r < R.chunk_len && qa_r >= t;
              cm = fmaxf(cm, valid ? s[r] : kMasked);
            }
          }
          acc[sb] = fmaxf(acc[sb], cm);
        }
      }
    }
    if (PASS == 0) {
      // the two halves of a row saw different keys: (m, l) pairs combine as M = max m, L = sum l 2^(m - M)
      const float mo = lane_xor32(m2), lo = lane_xor32(l);
      const float M = vmax(m2, mo);
      const float L = l * __builtin_amdgcn_exp2f(m2 - M) + lo * __builtin_amdgcn_exp2f(mo - M);
      if (half == 0) {
        const int64_t o = (grp * NKB + kb) * ROWS + w * Wpad32 + rb * 32 + lq;
        part_m[o] = M;
        part_l[o] = L;
      }
    }
  }
  if (PASS == 0) return;
  // ---- per key: the head's sum over its window rows (both halves), then the maximum over the heads of the group
  const float inv_len = 1.f / (float)max(R.q_end - R.q_start, 1);
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) {
    const float o = lane_xor32(acc[sb]);
    const float v = PASS == 1 ? (acc[sb] + o) * inv_len : fmaxf(acc[sb], o);
    if (half == 0) hs[w * kPsKeys + sb * 32 + lq] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < kPsKeys; c += blockDim.x) {
    const int t = k0 + c;
    if (t < key_lo || t >= key_hi) continue;
    float best = hs[c];
    for (int g = 1; g < G; ++g) best = fmaxf(best, hs[g * kPsKeys + c]);
    float* dst = a.attn_score + (int64_t)i * a.score_stride + t;
    if (PASS == 1) atomic_max_nonneg(dst, fmaxf(best, 0.f));
    else atomic_max_any(dst, best);
  }
}

// merge the NKB partial (m, l) pairs of pass 0 into m + log2 l (base-2 domain): 32 rows x 8 slices of the key blocks
// per workgroup (one thread per row walking 128 partials one dependent load after the other took 80 us)
__global__ void __launch_bounds__(256) prefill_score_reduce_kernel_v2(const float* part_m, const float* part_l, float* glob_ml,
                                                                       int NKB, int ROWS) {
  __shared__ float red[8][32];
  const int sl = threadIdx.x >> 5, r = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool live = r < ROWS;
  const float* pm = part_m + (int64_t)blockIdx.y * NKB * ROWS + (live ? r : 0);
  const float* pl = part_l + (int64_t)blockIdx.y * NKB * ROWS + (live ? r : 0);
  // this thread's slice: key blocks sl, sl + 8, ... in rounds of 16 whose 32 loads are in flight together
  float M = kMasked, L = 0.f;
  for (int b0 = sl; b0 < NKB; b0 += 128) {
    float vm[16], vl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = b0 + 8 * j;
      vm[j] = b < NKB ? pm[(int64_t)b * ROWS] : kMasked;
      vl[j] = b < NKB ? pl[(int64_t)b * ROWS] : 0.f;
    }
    float m = M;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, vm[j]);
    float l = L * __builtin_amdgcn_exp2f(M - m);
#pragma unroll
    for (int j = 0; j < 16; ++j) l += vl[j] * __builtin_amdgcn_exp2f(vm[j] - m);
    M = m;
    L = l;
  }
  red[sl][threadIdx.x & 31] = M;
  __syncthreads();
  float Mg = M;
#pragma unroll
  for (int j = 0; j < 8; ++j) Mg = fmaxf(Mg, red[j][threadIdx.x & 31]);
  __syncthreads();
  red[sl][threadIdx.x & 31] = L * __builtin_amdgcn_exp2f(M - Mg);
  __syncthreads();
  if (sl == 0 && live) {
    float Lg = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) Lg += red[j][threadIdx.x & 31];
    // p = 2^(u - m) / l = 2^(u - (m + log2 l)): one constant per row for pass 1
    glob_ml[(int64_t)blockIdx.y * ROWS + r] = Mg + __builtin_amdgcn_logf(Lg > 0.f ? Lg : 1.f);
  }
}

__global__ void __launch_bounds__(256) prefill_score_reduce_kernel(const float* part_m, const float* part_l, float* glob_m,
                                                                    float* glob_l, int NKB) {
  // one thread per (group, row): merge the NKB partial (m, l) pairs
  const int64_t o = (int64_t)blockIdx.x * kRowsPerWg + threadIdx.x;
  const float* pm = part_m + (int64_t)blockIdx.x * NKB * kRowsPerWg + threadIdx.x;
  const float* pl = part_l + (int64_t)blockIdx.x * NKB * kRowsPerWg + threadIdx.x;
  float M = kMasked;
  for (int b = 0; b < NKB; ++b) M = fmaxf(M, pm[(int64_t)b * kRowsPerWg]);
  float L = 0.f;
  for (int b = 0; b < NKB; ++b) L += pl[(int64_t)b * kRowsPerWg] * __expf(pm[(int64_t)b * kRowsPerWg] - M);
  glob_m[o] = M;
  glob_l[o] = L;
}

__global__ void __launch_bounds__(256) fill_rows_kernel(float* dst, int64_t stride, int cols, float v) {
  float* row = dst + (int64_t)blockIdx.y * stride;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) row[c] = v;
}

struct Tiling { int G, Wpad, q_limit, RB, NKB; int64_t groups; };

inline int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

inline Tiling make_tiling(int n_ranges, int Hq, int Hkv, int max_q, int cols, int mode) {
  Tiling t;
  t.G = Hq / Hkv;
  if (mode == SVK_PREFILL_SCORE_LOGITS) {
    const int block_m = std::min(32, std::max(16, next_pow2(max_q)));     // prefill_score.py:493-495
    t.q_limit = ((max_q + block_m - 1) / block_m) * block_m;
  } else {
    t.q_limit = std::max(16, next_pow2(max_q));                          // :497
  }
  t.Wpad = ((std::min(t.q_limit, std::max(max_q, 1)) + 15) / 16) * 16;
  if (kRowsPerWg % t.Wpad != 0 && t.Wpad < kRowsPerWg) t.Wpad = next_pow2(t.Wpad);   // keep heads inside one workgroup
  t.RB = (t.G * t.Wpad + kRowsPerWg - 1) / kRowsPerWg;
  t.NKB = (cols + kKeysPerWg - 1) / kKeysPerWg;
  t.groups = (int64_t)n_ranges * Hkv * t.RB;
  return t;
}

}  // namespace
}  // namespace svk

namespace svk {
namespace {
// v2: rows of a (range, KV head) group = G heads x the window padded to 32; key blocks of 128
struct Tiling2 { int Wpad32, NKB, ROWS; int64_t groups; };
inline Tiling2 make_tiling2(const Tiling& t, int n_ranges, int Hkv, int cols) {
  Tiling2 u;
  u.Wpad32 = ((t.Wpad + 31) / 32) * 32;
  u.NKB = (cols + kPsKeys - 1) / kPsKeys;
  u.ROWS = t.G * u.Wpad32;
  u.groups = (int64_t)n_ranges * Hkv;
  return u;
}
}  // namespace
}  // namespace svk

extern "C" int32_t svk_prefill_score_window_pad(int32_t num_q_heads, int32_t num_kv_heads, int32_t max_query_len) {
  using namespace svk;
  if (num_kv_heads <= 0 || max_query_len <= 0) return 0;
  const Tiling t = make_tiling(1, num_q_heads, num_kv_heads, max_query_len, 1, SVK_PREFILL_SCORE_PROBABILITY);
  return make_tiling2(t, 1, num_kv_heads, 1).Wpad32;
}

extern "C" int64_t svk_prefill_score_workspace_bytes(int32_t n_ranges, int32_t num_q_heads, int32_t num_kv_heads,
                                                     int32_t max_query_len, int32_t score_cols) {
  using namespace svk;
  if (n_ranges <= 0 || num_kv_heads <= 0 || max_query_len <= 0 || score_cols <= 0) return 0;
  const Tiling t = make_tiling(n_ranges, num_q_heads, num_kv_heads, max_query_len, score_cols, SVK_PREFILL_SCORE_PROBABILITY);
  const Tiling2 u = make_tiling2(t, n_ranges, num_kv_heads, score_cols);
  const int64_t v1 = (int64_t)sizeof(float) * 2 * t.groups * kRowsPerWg * ((int64_t)t.NKB + 1);
  const int64_t v2 = (int64_t)sizeof(float) * 2 * u.groups * u.ROWS * ((int64_t)u.NKB + 1);
  return v1 > v2 ? v1 : v2;
}

extern "C" int svk_prefill_score(const SvkPrefillScoreArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_prefill_score: null args");
  SVK_REQUIRE(a->score_mode == SVK_PREFILL_SCORE_PROBABILITY || a->score_mode == SVK_PREFILL_SCORE_LOGITS, SVK_ERR_VALUE,
              "prefill score_mode must be 'probability' or 'logits', got %d.", a->score_mode);
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_LAYOUT, "svk_prefill_score: head_dim %d unsupported (64, 128)", a->head_dim);
  SVK_REQUIRE(a->num_kv_heads > 0 && a->num_q_heads % a->num_kv_heads == 0, SVK_ERR_VALUE,
              "num query heads must be divisible by num kv heads: q=%d k=%d", a->num_q_heads, a->num_kv_heads);
  if (a->max_query_len <= 0 || a->score_cols <= 0 || a->n_ranges <= 0) return SVK_OK;
  if (a->score_mode == SVK_PREFILL_SCORE_PROBABILITY) {
    SVK_REQUIRE(std::max(16, next_pow2(a->max_query_len)) <= 128, SVK_ERR_VALUE,
                "probability prefill score query range is too large for this kernel: %d > 128", a->max_query_len);
    SVK_REQUIRE(a->workspace != nullptr || a->row_stats != nullptr, SVK_ERR_VALUE, "svk_prefill_score: probability mode needs a workspace");
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Tiling t = make_tiling(a->n_ranges, a->num_q_heads, a->num_kv_heads, a->max_query_len, a->score_cols, a->score_mode);
  const bool logits = a->score_mode == SVK_PREFILL_SCORE_LOGITS;
  const bool v2 = a->head_dim == 128 && t.G <= 8 && t.Wpad <= 128;
  if (a->row_stats != nullptr) {
    SVK_REQUIRE(!logits && v2, SVK_ERR_LAYOUT, "svk_prefill_score: row_stats serve the head_dim 128 probability kernel only");
    SVK_REQUIRE(a->candidate_start == 0 && a->num_recent_tokens == 0 && a->batch_indices == nullptr, SVK_ERR_VALUE,
                "svk_prefill_score: row_stats are the attention's statistics over ALL causal keys: candidate_start = 0, "
                "num_recent_tokens = 0 and range i <-> sequence i are required");
  }
  if (logits || !v2)      // (the v2 probability path clears the row inside its first pass)
    hipLaunchKernelGGL(fill_rows_kernel, dim3(std::min(64, (a->score_cols + 255) / 256), a->n_ranges), dim3(256), 0, s,
                       a->attn_score, a->score_stride, a->score_cols, logits ? -INFINITY : 0.f);
  if (v2) {
    const Tiling2 u = make_tiling2(t, a->n_ranges, a->num_kv_heads, a->score_cols);
    dim3 grid2(u.NKB, (unsigned)u.groups), block2(64 * t.G);
    const size_t shm = (size_t)kPsKeys * kPsRowB + (size_t)t.G * kPsKeys * sizeof(float);
    if (logits) {
      hipLaunchKernelGGL((prefill_score_kernel_v2<2>), grid2, block2, shm, s, *a, t.G, u.Wpad32, t.q_limit, u.NKB, nullptr, nullptr,
                         nullptr);
    } else if (a->row_stats != nullptr) {
      // the chunk's attention launch left the window rows' statistics and a cleared score row: final pass only
      hipLaunchKernelGGL((prefill_score_kernel_v2<1>), grid2, block2, shm, s, *a, t.G, u.Wpad32, t.q_limit, u.NKB, nullptr, nullptr,
                         a->row_stats);
    } else {
      const int64_t part2 = u.groups * u.NKB * u.ROWS, glob2 = u.groups * u.ROWS;
      float* pm2 = a->workspace;
      float* pl2 = pm2 + part2;
      float* gm2 = pl2 + part2;
      (void)glob2;
      hipLaunchKernelGGL((prefill_score_kernel_v2<0>), grid2, block2, shm, s, *a, t.G, u.Wpad32, t.q_limit, u.NKB, pm2, pl2, nullptr);
      hipLaunchKernelGGL(prefill_score_reduce_kernel_v2, dim3((u.ROWS + 31) / 32, (unsigned)u.groups), dim3(256), 0, s, pm2, pl2,
                         gm2, u.NKB, u.ROWS);
      hipLaunchKernelGGL((prefill_score_kernel_v2<1>), grid2, block2, shm, s, *a, t.G, u.Wpad32, t.q_limit, u.NKB, nullptr, nullptr,
                         gm2);
    }
    return check_launch("svk_prefill_score");
  }
  dim3 grid(t.NKB, (unsigned)t.groups), block(256);
  const int64_t part = t.groups * t.NKB * kRowsPerWg, glob = t.groups * kRowsPerWg;
  float* pm = a->workspace;
  float* pl = pm ? pm + part : nullptr;
  float* gmx = pl ? pl + part : nullptr;
  float* glx = gmx ? gmx + glob : nullptr;
#define SVK_PS(D_)                                                                                                         \
  do {                                                                                                                     \
    if (logits) {                                                                                                          \
      hipLaunchKernelGGL((prefill_score_kernel<D_, 2>), grid, block, 0, s, *a, t.G, t.Wpad, t.q_limit, t.RB, t.NKB, nullptr, \
                         nullptr, nullptr, nullptr);                                                                       \
    } else {                                                                                                               \
      hipLaunchKernelGGL((prefill_score_kernel<D_, 0>), grid, block, 0, s, *a, t.G, t.Wpad, t.q_limit, t.RB, t.NKB, pm, pl,  \
                         nullptr, nullptr);                                                                                \
      hipLaunchKernelGGL(prefill_score_reduce_kernel, dim3((unsigned)t.groups), dim3(kRowsPerWg), 0, s, pm, pl, gmx, glx,  \
                         t.NKB);                                                                                           \
      hipLaunchKernelGGL((prefill_score_kernel<D_, 1>), grid, block, 0, s, *a, t.G, t.Wpad, t.q_limit, t.RB, t.NKB, nullptr, \
                         nullptr, gmx, glx);                                                                               \
    }                                                                                                                      \
  } while (0)
  if (a->head_dim == 128) SVK_PS(128); else SVK_PS(64);
#undef SVK_PS
  return check_launch("svk_prefill_score");
}
