// Block-wide primitives shared by the selection kernels (H2O heavy hitters, Quest page top-k,
// SnapKV / DeltaKV top-k): reductions, 1-bit prefix counts and an exact radix select.
#pragma once

#include "svk_common.hpp"

#ifndef SVK_SEL_STAMP
#define SVK_SEL_STAMP(i)      // developer timing hook (quest.hip, -DSVK_QV_TIMING)
#endif

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

// inclusive prefix sum over the 64 lanes on the DPP network (a __shfl_up scan is six ds_bpermute round trips through the
// LDS crossbar): Hillis-Steele inside each row of 16 lanes (sources outside the row read as 0), then the row totals
// ripple with the two row broadcasts
__device__ __forceinline__ int wave_incl_scan_add(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return x;
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
// torch's comparison-based stable sort.  Sign-magnitude -> offset binary by ADDING / SUBTRACTING the magnitude
// (0x80000000 +- |f|) rather than by complementing the negatives: the order is the same, and bits that are zero in
// every magnitude stay shared by all keys - bf16-valued scores of mixed sign keep their 16 low key bits equal
// (complementing turns them into 0x0000 / 0xffff), so the select sees 16 varying bits and runs 2 digit passes, not 4.
__device__ __forceinline__ uint32_t desc_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  const uint32_t mag = u & 0x7fffffffu;                         // (-0.0 -> +0.0: both map to 0x80000000)
  const uint32_t asc = (u & 0x80000000u) ? 0x80000000u - mag : 0x80000000u + mag;
  return ~asc;
}


struct SelectScratch {
  int hist[256];
  int wsum[16], wsum2[16];
  uint32_t prefix;
  int k, total;
  uint32_t or_bits, and_bits;      // OR / AND of all keys: key bytes every element shares need no radix pass
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
  // cheap duplicate detector first (one DPP move + ballot): only when a quarter of the wave's neighbouring lanes
  // carry the same bin are the leader rounds worth their ~2 x (shuffle + ballot) latency; spread digits go straight
  // to individual LDS atomics (22 k keys per pass: 15 us -> 8 us)
  const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp((int)~bin, (int)bin, 0x111, 0xf, 0xf, false);   // row_shr:1
  if (__popcll(__ballot(active && nb == bin)) >= 16) {
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
    active = (todo >> lane) & 1ull;
  }
  if (active) atomicAdd(&hist[bin], 1);
}

// The same for a histogram of 16-bit digits held as two 16-bit counts per LDS word (65 536 bins in 128 KB).  Bins 2p and
// 2p + 1 are the halves of word (p & 31) * 1024 + (p >> 5): the 32 words of the 64-bin segment s = p >> 5 lie 1024 words
// apart, so thread s of a 1024-thread block sums its segment with conflict-free reads.  Counts must stay below 65 536.
__device__ __forceinline__ int hist16_word(uint32_t bin) { return (int)(((bin >> 1) & 31u) * 1024u + (bin >> 6)); }
__device__ __forceinline__ void hist16_add_aggregated(uint32_t* hist16, uint32_t bin, bool active) {
  const int lane = threadIdx.x & 63;
  const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp((int)~bin, (int)bin, 0x111, 0xf, 0xf, false);   // row_shr:1
  if (__popcll(__ballot(active && nb == bin)) >= 16) {
    unsigned long long todo = __ballot(active);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (todo == 0ull) break;
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t lb = (uint32_t)__shfl((int)bin, leader, 64);
      const unsigned long long same = __ballot(active && bin == lb) & todo;
      if (lane == leader) atomicAdd(&hist16[hist16_word(lb)], (uint32_t)__popcll(same) << ((lb & 1u) * 16u));
      todo &= ~same;
    }
    active = (todo >> lane) & 1ull;
  }
  if (active) atomicAdd(&hist16[hist16_word(bin)], 1u << ((bin & 1u) * 16u));
}

// Core on descending-order keys `key_at(i)` (smaller key = better; callers stage keys in LDS when the row fits, so the
// sweeps do not pay a dependent global-load latency each).
// OR / AND of all keys, accumulated in the scratch (workgroup-wide): begin (includes a barrier), add per thread, then a
// barrier before the select reads them
__device__ __forceinline__ void select_bits_begin(SelectScratch& S) {
  if (threadIdx.x == 0) { S.or_bits = 0u; S.and_bits = 0xffffffffu; }
  __syncthreads();
}
__device__ __forceinline__ void select_bits_add(SelectScratch& S, uint32_t o, uint32_t an) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    o |= (uint32_t)__shfl_xor((int)o, off, 64);
    an &= (uint32_t)__shfl_xor((int)an, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { atomicOr(&S.or_bits, o); atomicAnd(&S.and_bits, an); }
}

// The bin of S.hist[0:256) that holds the kk-th key (1-based) of a digit at `shift`: prefix |= bin << shift, kk = the rank
// wanted inside that bin; returns the histogram's total.  The histogram is complete (a barrier behind the caller's
// adds); all threads call; on return every thread has read the result and S.hist is still intact.
__device__ __forceinline__ int select_find_bin(SelectScratch& S, uint32_t& prefix, int& kk, int shift) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    int c[4];
    int local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = S.hist[tid * 4 + j]; local += c[j]; }
    const int incl = wave_incl_scan_add(local);
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
    if (tid == 63) S.total = incl;
  }
  __syncthreads();
  prefix = S.prefix;
  kk = S.k;
  const int total = S.total;
  __syncthreads();
  return total;
}

// The radix passes of the select: digits of up to 8 bits from bit `top` down to bit `low`; keys whose bits above the
// current digit differ from `prefix` are out of the race.  In: prefix = the decided / shared bits, kk = rank wanted
// among the keys still in the race (1-based).  Out: prefix = the kk-th key, kk = how many keys equal to it are taken.
// `sweep(shift, dmask, himask, prefix)` adds every key still in the race to S.hist (hist_add_aggregated, all lanes of
// a wave together).  All threads of the block call.
template <typename Sweep>
__device__ __forceinline__ void select_radix_passes_sweep(Sweep sweep, SelectScratch& S, uint32_t& prefix, int& kk, int top,
                                                          int low) {
  const int tid = threadIdx.x, nt = blockDim.x;
  while (top >= low) {
    const int shift = max(top - 7, 0);
    const int width = top - shift + 1;
    const uint32_t dmask = (width == 32 ? 0xffffffffu : ((1u << width) - 1u));
    const uint32_t himask = top == 31 ? 0u : (0xffffffffu << (top + 1));
    prefix &= ~(dmask << shift);                               // this digit is decided now
    top = shift - 1;
    for (int i = tid; i < 256; i += nt) S.hist[i] = 0;
    __syncthreads();
    sweep(shift, dmask, himask, prefix);
    __syncthreads();
    select_find_bin(S, prefix, kk, shift);
    SVK_SEL_STAMP(8 + (31 - shift) / 8);
  }
  SVK_SEL_STAMP(2);
}

// sweep over key_at(tid), key_at(tid + nt), ...
template <typename KeyAt>
__device__ __forceinline__ void select_radix_passes(KeyAt key_at, int n, SelectScratch& S, uint32_t& prefix, int& kk, int top,
                                                    int low) {
  select_radix_passes_sweep([&](int shift, uint32_t dmask, uint32_t himask, uint32_t pfx) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int c0 = 0; c0 < n; c0 += nt) {                        // all threads iterate together (wave-wide ballots inside)
      const int i = c0 + tid;
      uint32_t key = 0u;
      const bool in = i < n;
      if (in) key = key_at(i);
      hist_add_aggregated(S.hist, (key >> shift) & dmask, in && (key & himask) == (pfx & himask));
    }
  }, S, prefix, kk, top, low);
}

// Block scan behind the ordered emit: (n_lt, n_eq) = this thread's counts over the contiguous index range it owns ->
// the counts of all lower-numbered threads.
__device__ __forceinline__ void select_emit_offsets(int n_lt, int n_eq, SelectScratch& S, int& lt_before, int& eq_before) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int il = wave_incl_scan_add(n_lt), ie = wave_incl_scan_add(n_eq);
  __syncthreads();
  if (lane == 63) { S.wsum[w] = il; S.wsum2[w] = ie; }
  __syncthreads();
  lt_before = il - n_lt;
  eq_before = ie - n_eq;
  for (int j = 0; j < w; ++j) { lt_before += S.wsum[j]; eq_before += S.wsum2[j]; }
}

// ordered emit over key_at(0..n): T = threshold key, take_eq = elements equal to T that are taken, lowest index first
template <typename KeyAt, typename Emit>
__device__ __forceinline__ void select_ordered_emit(KeyAt key_at, int n, uint32_t T, int take_eq, SelectScratch& S, Emit emit) {
  const int tid = threadIdx.x, nt = blockDim.x;
  // ordered emit: thread t owns the contiguous index range [t*chunk, (t+1)*chunk) (odd chunk: conflict-free LDS
  // strides), so ONE block scan of (#keys < T, #keys == T) gives every element its rank among the selected
  const int chunk = ((n + nt - 1) / nt) | 1;
  const int i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  int n_lt = 0, n_eq = 0;
  for (int i = i0; i < i1; ++i) {
    const uint32_t key = key_at(i);
    n_lt += key < T;
    n_eq += key == T;
  }
  int lt_before, eq_before;
  select_emit_offsets(n_lt, n_eq, S, lt_before, eq_before);
  // selected-before = (all < T before) + (equal-to-T before that are taken)
  for (int i = i0; i < i1; ++i) {
    const uint32_t key = key_at(i);
    const bool lt = key < T, eq = key == T;
    if (lt || (eq && eq_before < take_eq)) emit(lt_before + min(eq_before, take_eq), i);
    lt_before += lt;
    eq_before += eq;
  }
}


// The select with the keys in registers: thread t owns the `per` (<= CH, the same for all threads) consecutive indices
// from t * per (key[j] = key of index t * per + j; indices >= n are ignored whatever their key), so neither the passes
// nor the emit read memory; `emit(pos, idx, key)`.
// (threshold half: T = the k-th key, kk = how many keys equal to T are taken)
// `hist16` (optional): 32 768 words of LDS, blockDim.x == 1024.  Keys that differ in their 16 high bits only (bf16-valued
// scores: Quest page scores) are then counted by that whole 16-bit digit in ONE pass of LDS atomics spread over thousands
// of bins, and the threshold is read off a block scan of the 1024 segment sums + a wave scan of the
// segment's 64 bins -
// instead of a pass per 8-bit digit whose first digit (sign + 7 exponent bits) piles a row's keys onto a handful of
// bins (8191 keys: ~4 us for that pass alone, ~1.5 us for the next).
template <int CH>
__device__ __forceinline__ void select_owned_threshold(const uint32_t (&key)[CH], int per, int n, int k, SelectScratch& S,
                                                       uint32_t& T_out, int& kk_out, uint32_t* hist16 = nullptr) {
  const int base = threadIdx.x * per;
  if (hist16 != nullptr)                   // cleared under the loads and the barriers of the OR / AND reduction (used or not:
    for (int i = threadIdx.x * 4; i < 32768; i += blockDim.x * 4)      //  clearing it after the decision costs 0.8 us more)
      *reinterpret_cast<uint4*>(hist16 + i) = make_uint4(0u, 0u, 0u, 0u);
  select_bits_begin(S);
  uint32_t o = 0u, an = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < CH; ++j)
    if (j < per && base + j < n) { o |= key[j]; an &= key[j]; }
  select_bits_add(S, o, an);
  __syncthreads();
  SVK_SEL_STAMP(1);
  const uint32_t all_and = S.and_bits;
  const uint32_t varying = S.or_bits ^ all_and;
  uint32_t prefix = all_and;
  int kk = k;
  // (only when the 8-bit digits would need two passes or more: keys that differ in <= 8 bit positions - real page
  //  scores are positive and sit in a narrow band of bf16 values - are settled by ONE 8-bit pass, 1.6 us, while their few
  //  hundred distinct values would pile onto as many of the 65 536 bins: 3.0 us of atomics + the 1.5 us scan)
  if (hist16 != nullptr && varying != 0u && (varying & 0xffffu) == 0u && n < 65536 && blockDim.x == 1024 &&
      (31 - __builtin_clz(varying)) - __builtin_ctz(varying) >= 8) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = 16;
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j < per) hist16_add_aggregated(hist16, key[j] >> 16, base + j < n);
    __syncthreads();
    SVK_SEL_STAMP(8);
    int local = 0;                                              // thread t sums segment t (bins [64 t, 64 t + 64))
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const uint32_t c = hist16[j * 1024 + tid];
      local += (int)(c & 0xffffu) + (int)(c >> 16);
    }
    const int incl = wave_incl_scan_add(local);
    if (lane == 63) S.wsum[w] = incl;
    __syncthreads();
    int excl = incl - local;
    for (int j = 0; j < nw; ++j) excl += j < w ? S.wsum[j] : 0;
    if (kk > excl && kk <= excl + local) { S.wsum2[0] = tid; S.wsum2[1] = kk - excl; }     // the kk-th key lies in this segment
    __syncthreads();
    if (w == 0) {                                               // its 64 bins, one per lane
      const int seg = S.wsum2[0], k2 = S.wsum2[1];
      const uint32_t c = hist16[(lane >> 1) * 1024 + seg];
      const int cnt = (lane & 1) ? (int)(c >> 16) : (int)(c & 0xffffu);
      const int inc = wave_incl_scan_add(cnt);
      if (k2 > inc - cnt && k2 <= inc) { S.prefix = (uint32_t)(seg * 64 + lane); S.k = k2 - (inc - cnt); }
    }
    __syncthreads();
    T_out = (S.prefix << 16) | (all_and & 0xffffu);
    kk_out = S.k;
    __syncthreads();
    SVK_SEL_STAMP(2);
    return;
  }
  auto sweep = [&](int shift, uint32_t dmask, uint32_t himask, uint32_t pfx) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j < per) hist_add_aggregated(S.hist, (key[j] >> shift) & dmask, base + j < n && (key[j] & himask) == (pfx & himask));
  };
  const int top = varying ? 31 - __builtin_clz(varying) : -1, low = varying ? __builtin_ctz(varying) : 0;
  if (top - low >= 8) {
    // first digit by histogram; if the threshold bin then holds few keys (the usual case: n / 256 on average) the
    // remaining digits are settled by ranking those keys against each other instead of three more histogram passes
    // (each ~1.3 us of barriers for a handful of keys)
    const int shift1 = top - 7;
    select_radix_passes_sweep(sweep, S, prefix, kk, top, shift1);
    const int in_bin = S.hist[(prefix >> shift1) & 0xffu];
    __syncthreads();                                           // everyone has read the bin count: the histogram is free
    if (in_bin <= 256) {
      uint32_t* cand = reinterpret_cast<uint32_t*>(S.hist);
      if (threadIdx.x == 0) S.k = 0;
      __syncthreads();
      const uint32_t himask = 0xffffffffu << shift1;
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (j < per && base + j < n && (key[j] & himask) == (prefix & himask)) cand[atomicAdd(&S.k, 1)] = key[j];
      __syncthreads();
      if ((int)threadIdx.x < in_bin) {
        const uint32_t mine = cand[threadIdx.x];
        int lt = 0, eq = 0;
        for (int i = 0; i < in_bin; ++i) {
          const uint32_t o = cand[i];
          lt += o < mine;
          eq += o == mine;
        }
        if (lt < kk && kk <= lt + eq) { S.prefix = mine; S.wsum2[0] = kk - lt; }      // equal keys write equal values
      }
      __syncthreads();
      prefix = S.prefix;
      kk = S.wsum2[0];
      __syncthreads();
    } else {
      select_radix_passes_sweep(sweep, S, prefix, kk, shift1 - 1, low);
    }
  } else {
    select_radix_passes_sweep(sweep, S, prefix, kk, top, low);
  }
  T_out = prefix;
  kk_out = kk;
}

// (emit half: the keys < T and the first kk keys == T, `emit(pos, idx, key, j)` with pos = rank in ascending index order
//  and j = the key's register index, a constant after unrolling)
template <int CH, typename Emit>
__device__ __forceinline__ void select_owned_emit(const uint32_t (&key)[CH], int per, int n, uint32_t T, int kk,
                                                  SelectScratch& S, Emit emit) {
  const int base = threadIdx.x * per;
  int n_lt = 0, n_eq = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const bool in = j < per && base + j < n;
    n_lt += in && key[j] < T;
    n_eq += in && key[j] == T;
  }
  int lt_before, eq_before;
  select_emit_offsets(n_lt, n_eq, S, lt_before, eq_before);
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    if (j < per) {
      const bool in = base + j < n;
      const bool lt = in && key[j] < T, eq = in && key[j] == T;
      if (lt || (eq && eq_before < kk)) emit(lt_before + min(eq_before, kk), base + j, key[j], j);
      lt_before += lt;
      eq_before += eq;
    }
  }
}

template <int CH, typename Emit>
__device__ __forceinline__ void block_select_topk_ordered_owned(const uint32_t (&key)[CH], int per, int n, int k,
                                                                SelectScratch& S, Emit emit, uint32_t* hist16 = nullptr) {
  uint32_t T;
  int kk;
  select_owned_threshold<CH>(key, per, n, k, S, T, kk, hist16);
  select_owned_emit<CH>(key, per, n, T, kk, S, emit);
}

template <typename KeyAt, typename Emit>
__device__ __forceinline__ void block_select_topk_ordered_keys(KeyAt key_at, int n, int k, SelectScratch& S, Emit emit,
                                                               bool bits_ready = false) {
  const int tid = threadIdx.x, nt = blockDim.x;
  uint32_t prefix = 0;
  int kk = k;
  // bytes shared by every key (bf16-valued scores: the two low bytes; probabilities: most of the top byte) are
  // found with one atomic-free sweep and skip their radix pass
  // (callers that stage the keys themselves fold this sweep into their staging loop: `select_bits_begin` before,
  //  `select_bits_add` with each thread's OR / AND, then `bits_ready` = true)
  if (!bits_ready) {
    select_bits_begin(S);
    uint32_t o = 0u, an = 0xffffffffu;
    for (int i = tid; i < n; i += nt) {
      const uint32_t key = key_at(i);
      o |= key;
      an &= key;
    }
    select_bits_add(S, o, an);
    __syncthreads();
  }
  const uint32_t all_and = S.and_bits;
  const uint32_t varying = S.or_bits ^ all_and;              // bit positions on which the keys differ
  // 8-bit digits are laid from the highest varying bit down to the lowest one (not on byte boundaries): the first
  // digit then spreads over the bins even when the keys share their leading bits (probabilities share sign and the
  // upper exponent bits and would pile into a handful of bins - serialised LDS atomics), and digits made of shared
  // bits only are never histogrammed (bf16-valued scores: 2 passes instead of 4)
  prefix = all_and;                                           // shared bits are those of every key, the k-th too
  const int top = varying ? 31 - __builtin_clz(varying) : -1;       // highest bit still undecided
  const int low = varying ? __builtin_ctz(varying) : 0;
  select_radix_passes(key_at, n, S, prefix, kk, top, low);
  select_ordered_emit(key_at, n, prefix, kk, S, emit);
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
