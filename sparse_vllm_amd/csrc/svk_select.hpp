// Block-wide primitives shared by the selection kernels (H2O heavy hitters, Quest page top-k,
// SnapKV / DeltaKV top-k): reductions, 1-bit prefix counts and an exact radix select.
#pragma once

#include "svk_common.hpp"

namespace svk {

// ------------------------------------------------------------------------------------
// block-wide helpers (blockDim.x multiple of 64, <= 1024)
// ------------------------------------------------------------------------------------

__device__ __forceinline__ float block_allmax(float x, float* red) {
  x = wave_allmax(x);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = x;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

__device__ __forceinline__ float block_allsum(float x, float* red) {
  x = wave_allsum(x);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = x;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// exclusive prefix count of a 1-bit flag over the block, plus the block total
__device__ __forceinline__ int block_excl_count(bool flag, int* wsum, int& total) {
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wsum[w] = __popcll(bal);
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    const int c = wsum[i];
    if (i < w) base += c;
    tot += c;
  }
  total = tot;
  return base + in_wave;
}

// Descending-order key: smaller key == larger score; -0.0 and +0.0 compare equal like
// torch's comparison-based stable sort.
__device__ __forceinline__ uint32_t desc_key(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u << 1) == 0u) u = 0u;                                   // canonical +0
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}


struct SelectScratch {
  int hist[256];
  int wsum[16];
  uint32_t prefix;
  int k;
};

// Exact top-k of sc[0:n) under the order (score descending, index ascending) -- the order of a
// stable descending argsort, i.e. ties keep the LOWER index.  Requires 1 <= k <= n and a block of
// 64..1024 threads, all of which must call.  `emit(pos, idx)` is called exactly k times with
// pos = rank of idx among the selected indices in ASCENDING index order.
// Method: 4-pass MSB radix select (8 bits per pass) of the k-th smallest desc_key gives the
// threshold key T and how many elements equal to T are taken; one ordered pass then emits.
// histogram add with wave-level aggregation: scores that cluster into a few radix bins (probabilities, bf16-valued
// scores whose low key bytes are all equal) would otherwise serialise ~n LDS atomics on one address.  Two leader
// rounds fold the dominant bins into one atomic per wave each; the remaining lanes add individually.
__device__ __forceinline__ void hist_add_aggregated(int* hist, uint32_t bin, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (todo == 0ull) break;
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lb = (uint32_t)__shfl((int)bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == lb) & todo;
    if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1);
}

// Core on descending-order keys `key_at(i)` (smaller key = better; callers stage keys in LDS when the row fits, so the
// five sweeps do not pay a dependent global-load latency each).  Each thread works on strips of kStrip consecutive
// elements so the ordered emit needs one block scan per kStrip * blockDim elements.
template <typename KeyAt, typename Emit>
__device__ __forceinline__ void block_select_topk_ordered_keys(KeyAt key_at, int n, int k, SelectScratch& S, Emit emit) {
  constexpr int kStrip = 4;
  const int tid = threadIdx.x, nt = blockDim.x;
  uint32_t prefix = 0;
  int kk = k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < 256; i += nt) S.hist[i] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += nt * kStrip) {
#pragma unroll
      for (int e = 0; e < kStrip; ++e) {
        const int i = c0 + tid * kStrip + e;
        uint32_t key = 0u;
        const bool in = i < n;
        if (in) key = key_at(i);
        hist_add_aggregated(S.hist, (key >> shift) & 255u, in && (key & himask) == prefix);
      }
    }
    __syncthreads();
    if (tid < 64) {
      int c[4];
      int local = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c[j] = S.hist[tid * 4 + j]; local += c[j]; }
      int incl = local;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (tid >= o) incl += v;
      }
      const int excl = incl - local;
      if (kk > excl && kk <= incl) {
        int run = excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (kk > run && kk <= run + c[j]) {
            S.prefix = prefix | ((uint32_t)(tid * 4 + j) << shift);
            S.k = kk - run;
          }
          run += c[j];
        }
      }
    }
    __syncthreads();
    prefix = S.prefix;
    kk = S.k;
    __syncthreads();
  }
  const uint32_t T = prefix;   // threshold key
  const int take_eq = kk;      // elements equal to T that are taken, lowest index first
  int out_base = 0, eq_base = 0;
  const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
  for (int c0 = 0; c0 < n; c0 += nt * kStrip) {
    uint32_t key[kStrip];
    int n_lt = 0, n_eq = 0;
#pragma unroll
    for (int e = 0; e < kStrip; ++e) {
      const int i = c0 + tid * kStrip + e;
      key[e] = 0xffffffffu;
      if (i < n) {
        key[e] = key_at(i);
        n_lt += key[e] < T;
        n_eq += key[e] == T;
      }
    }
    // block-wide exclusive prefix of (n_lt, n_eq) packed in one int (each < 2^15 per block of <= 4096 elements)
    const int packed = (n_lt << 16) | n_eq;
    int incl = packed;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    __syncthreads();
    if (lane == 63) S.wsum[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int j = 0; j < nw; ++j) {
      const int c = S.wsum[j];
      if (j < w) base += c;
      tot += c;
    }
    const int excl = base + incl - packed;
    int lt_before = excl >> 16, eq_before = eq_base + (excl & 0xffff);
    // selected-before = (all < T before) + (equal-to-T before that are taken)
#pragma unroll
    for (int e = 0; e < kStrip; ++e) {
      const int i = c0 + tid * kStrip + e;
      if (i < n) {
        const bool lt = key[e] < T, eq = key[e] == T;
        if (lt || (eq && eq_before < take_eq)) emit(out_base + lt_before + min(eq_before, take_eq) - min(eq_base, take_eq), i);
        lt_before += lt;
        eq_before += eq;
      }
    }
    out_base += (tot >> 16) + (min(eq_base + (tot & 0xffff), take_eq) - min(eq_base, take_eq));
    eq_base += tot & 0xffff;
  }
}

// `score_at(i)` form (callers that mask or remap scores on the fly)
template <typename ScoreAt, typename Emit>
__device__ __forceinline__ void block_select_topk_ordered_fn(ScoreAt score_at, int n, int k, SelectScratch& S, Emit emit) {
  block_select_topk_ordered_keys([&](int i) { return desc_key(score_at(i)); }, n, k, S, emit);
}

template <typename Emit>
__device__ __forceinline__ void block_select_topk_ordered(const float* sc, int n, int k, SelectScratch& S, Emit emit) {
  block_select_topk_ordered_keys([sc](int i) { return desc_key(sc[i]); }, n, k, S, emit);
}

}  // namespace svk
