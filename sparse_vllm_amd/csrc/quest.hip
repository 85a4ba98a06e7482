// Quest (query-aware page top-k) kernels for gfx950: page min/max metadata, page scoring on the
// matrix cores with the reference's bf16 rounding points, exact top-k + packed view, paging.
// HBM-bound scans of the [pages, Hkv, D] min/max metadata (1/16 of the K bytes each).

#include "svk_common.hpp"
// Developer build (make EXTRA=-DSVK_QV_TIMING, then tools/qv_timing.py): s_memrealtime stamps (100 MHz) of workgroup 0 of
// quest_build_view_kernel: 0 entry, 1 keys staged, 8.. after each radix pass, 2 threshold known, 3 pages emitted, 4 view written.
#ifdef SVK_QV_TIMING
__device__ unsigned long long g_qv_stamps[16];
#define SVK_SEL_STAMP(i)                                                                                    \
  do {                                                                                                      \
    __syncthreads();                                                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_qv_stamps[i] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)
extern "C" int svk_debug_quest_view_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qv_stamps), sizeof(g_qv_stamps));
}
#endif
#include "svk_select.hpp"

namespace svk {
namespace {

// ------------------------------------------------------------------------------------
// page min / max
// ------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t bf16x2_max(uint32_t a, uint32_t b) {
  const float al = bf16_lo(a), ah = bf16_hi(a), bl = bf16_lo(b), bh = bf16_hi(b);
  return (al >= bl ? (a & 0xffffu) : (b & 0xffffu)) | (ah >= bh ? (a & 0xffff0000u) : (b & 0xffff0000u));
}
__device__ __forceinline__ uint32_t bf16x2_min(uint32_t a, uint32_t b) {
  const float al = bf16_lo(a), ah = bf16_hi(a), bl = bf16_lo(b), bh = bf16_hi(b);
  return (al <= bl ? (a & 0xffffu) : (b & 0xffffu)) | (ah <= bh ? (a & 0xffff0000u) : (b & 0xffff0000u));
}

__global__ void __launch_bounds__(128) quest_page_minmax_kernel(const SvkQuestPageMinmaxArgs a, int chunks_per_row) {
  const int page_i = blockIdx.x, layer = blockIdx.y;
  const int64_t page = a.page_slots[page_i];
  const uint16_t* k = a.k_cache + (int64_t)layer * a.k_layer_stride + page * a.page_size * (int64_t)a.row_elems;
  uint16_t* mx = a.metadata + (int64_t)layer * a.meta_layer_stride + page * (int64_t)a.row_elems;
  uint16_t* mn = mx + a.meta_kind_stride;
  for (int ch = threadIdx.x; ch < chunks_per_row; ch += blockDim.x) {
    uint4 hi = *reinterpret_cast<const uint4*>(k + ch * 8);
    uint4 lo = hi;
    for (int t = 1; t < a.page_size; ++t) {
      const uint4 v = *reinterpret_cast<const uint4*>(k + (int64_t)t * a.row_elems + ch * 8);
      hi.x = bf16x2_max(hi.x, v.x); hi.y = bf16x2_max(hi.y, v.y); hi.z = bf16x2_max(hi.z, v.z); hi.w = bf16x2_max(hi.w, v.w);
      lo.x = bf16x2_min(lo.x, v.x); lo.y = bf16x2_min(lo.y, v.y); lo.z = bf16x2_min(lo.z, v.z); lo.w = bf16x2_min(lo.w, v.w);
    }
    *reinterpret_cast<uint4*>(mx + ch * 8) = hi;
    *reinterpret_cast<uint4*>(mn + ch * 8) = lo;
  }
}

// ------------------------------------------------------------------------------------
// page scoring: one wave per KV head, 16 pages per MFMA group
//   S+ = Q+ . Max^T, S- = Q- . Min^T (fp32 accumulate), s = bf16(bf16(S+) + bf16(S-)), max over heads
// ------------------------------------------------------------------------------------

constexpr int kPagesPerBlock = 32;

template <int D, int G>
__global__ void __launch_bounds__(512) quest_score_pages_kernel(const SvkQuestScorePagesArgs a) {
  constexpr int NC = D / 32, JQ = (G + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [kPagesPerBlock][Hkv*JQ]
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * kPagesPerBlock;
  const int n = lane & 15, jq = lane >> 4;
  const int SP = Hkv * JQ;
  const int p1 = min(p0 + kPagesPerBlock, a.n_prev);
  float* out = a.page_scores + (int64_t)b * a.score_stride;
  // the block's page slots are requested before the row's length is looked at (index clamped to the scored columns, which
  // the table row always has): the length's round trip then runs under this one instead of in front of it
  const int32_t* ptab = a.page_table + (int64_t)a.req_indices[b] * a.page_table_stride;
  constexpr int NGRP = kPagesPerBlock / 16;
  int slot[NGRP];
#pragma unroll
  for (int g = 0; g < NGRP; ++g) slot[g] = ptab[min(p0 + g * 16 + n, a.n_prev - 1)];
  const int len = a.context_lens[b];
  const int num_pages = max(1, (len + a.page_size - 1) / a.page_size);
  const int n_valid = min(a.n_prev, num_pages - 1);        // previous pages that exist
  if (p0 >= n_valid) {                                       // nothing valid in this block
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) out[p] = -INFINITY;
    return;
  }
  // Q+ / Q- fragments (A operand rows = heads of this KV group)
  bf16x8_t qp[NC], qn[NC];
  {
    const uint16_t* qptr = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h + jq * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (n < G) t = *reinterpret_cast<const uint4*>(qptr + c * 32);
      uint32_t v[4] = {t.x, t.y, t.z, t.w}, pp[4], nn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // clamp_min(0) / clamp_max(0) per bf16 element: keep the element iff its sign bit says so
        const uint32_t lo = v[i] & 0xffffu, hi = v[i] & 0xffff0000u;
        const bool lo_neg = (lo & 0x8000u) != 0, hi_neg = (hi & 0x80000000u) != 0;
        pp[i] = (lo_neg ? 0u : lo) | (hi_neg ? 0u : hi);
        nn[i] = (lo_neg ? lo : 0u) | (hi_neg ? hi : 0u);
      }
      qp[c] = __builtin_bit_cast(bf16x8_t, make_uint4(pp[0], pp[1], pp[2], pp[3]));
      qn[c] = __builtin_bit_cast(bf16x8_t, make_uint4(nn[0], nn[1], nn[2], nn[3]));
    }
  }
  const int64_t head_off = (int64_t)w * D + jq * 8;
  const int64_t row_elems = (int64_t)Hkv * D;
  // all page slots of the block first (above), then every group's metadata loads before the first MFMA: the scan is
  // latency-bound (two dependent round trips per 16 pages), so keep the whole block's requests in flight together
#pragma unroll
  for (int g = 0; g < NGRP; ++g)
    if (p0 + g * 16 + n >= n_valid) slot[g] = 0;
  uint4 vmax[NGRP][NC], vmin[NGRP][NC];
#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    const uint16_t* pm = a.page_max + (int64_t)max(slot[g], 0) * row_elems + head_off;
    const uint16_t* pn = a.page_min + (int64_t)max(slot[g], 0) * row_elems + head_off;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      vmax[g][c] = *reinterpret_cast<const uint4*>(pm + c * 32);
      vmin[g][c] = *reinterpret_cast<const uint4*>(pn + c * 32);
    }
  }
#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    if (p0 + g * 16 >= p1) break;
    f32x4_t sp = {0.f, 0.f, 0.f, 0.f}, sn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      sp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qp[c], __builtin_bit_cast(bf16x8_t, vmax[g][c]), sp, 0, 0, 0);
      sn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qn[c], __builtin_bit_cast(bf16x8_t, vmin[g][c]), sn, 0, 0, 0);
    }
    if (jq < JQ) {
      float best = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (jq * 4 + r < G) best = fmaxf(best, bf16_round(bf16_round(sp[r]) + bf16_round(sn[r])));
      lds[(g * 16 + n) * SP + w * JQ + jq] = best;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < p1 - p0; t += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < SP; ++j) mx = fmaxf(mx, lds[t * SP + j]);
    out[p0 + t] = (p0 + t < n_valid) ? mx : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------
// top-k + packed view, one workgroup per batch lane
// ------------------------------------------------------------------------------------

// CH > 0: thread t holds the keys of per = ceil(n_prev / blockDim) <= CH consecutive pages in registers for the whole select;
// CH == 0: keys in LDS (`lds_keys`) or re-read from memory on every sweep.
template <int CH>
__global__ void __launch_bounds__(1024) quest_build_view_kernel(const SvkQuestBuildViewArgs a, int lds_keys) {
  __shared__ SelectScratch scratch;
  extern __shared__ __attribute__((aligned(16))) int sel_pages[];          // [prev_budget] selected logical pages, ascending
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* sc = a.page_scores + (int64_t)b * a.score_stride;
  constexpr int C = CH > 0 ? CH : 4;
  // pages per thread, rounded up to a multiple of 4: every thread's range starts 16-byte aligned (vector loads of the
  // scores and page slots for any row length; the last threads own nothing)
  [[maybe_unused]] const int per = ((a.n_prev + nt - 1) / nt + 3) & ~3;
  [[maybe_unused]] const int base = tid * per;
  [[maybe_unused]] float v[C];
  if constexpr (CH > 0) {
    // the thread's scores first (16-byte loads when the row allows): they do not depend on the row's length or table row,
    // so their round trip overlaps those loads; then no memory in the select
    if ((per & 3) == 0 && (reinterpret_cast<uintptr_t>(sc) & 15u) == 0u && base + per <= a.n_prev) {
#pragma unroll
      for (int j = 0; j < C; j += 4) {
        if (j < per) {
          const float4 f = *reinterpret_cast<const float4*>(sc + base + j);
          v[j] = f.x; v[j + 1] = f.y; v[j + 2] = f.z; v[j + 3] = f.w;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) v[j] = j < per && base + j < a.n_prev ? sc[base + j] : 0.f;
    }
  }
  const int len = a.context_lens[b];
  const int row = a.req_indices[b];
  const int ps = a.page_size;
  const int num_pages = max(1, (len + ps - 1) / ps);
  int32_t* packed = a.packed_slots + (int64_t)b * a.packed_stride;
  const int32_t* ttab = a.token_table + (int64_t)row * a.token_table_stride;
  const int32_t* ptab = a.page_table + (int64_t)row * a.page_table_stride;
  if (tid == 0) a.local_req[b] = b;
  const bool dense = !a.is_long_text && (len <= a.token_budget || num_pages <= a.page_budget_base);
  SVK_SEL_STAMP(0);
  if (dense) {
    if (a.emit_page_slots) {
      for (int j = tid; j < (a.max_keep + ps - 1) / ps; j += nt) packed[j] = j < num_pages ? max(ptab[j], 0) : 0;
    } else {
      for (int i = tid; i < a.max_keep; i += nt) packed[i] = ttab[i];
    }
    if (tid == 0) a.local_lens[b] = len;
    return;
  }
  if constexpr (CH > 0) {
    // the page slots of the thread's own pages, in flight during the select: a selected page is then written straight
    // from the emit (no list of selected pages, no gather pass behind a barrier)
    int pslot[C];
    if ((per & 3) == 0 && (reinterpret_cast<uintptr_t>(ptab) & 15u) == 0u && base + per <= a.n_prev) {
#pragma unroll
      for (int j = 0; j < C; j += 4) {
        if (j < per) {
          const int4 t = *reinterpret_cast<const int4*>(ptab + base + j);
          pslot[j] = t.x; pslot[j + 1] = t.y; pslot[j + 2] = t.z; pslot[j + 3] = t.w;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) pslot[j] = j < per && base + j < a.n_prev ? ptab[base + j] : 0;
    }
    const int last_slot = max(ptab[num_pages - 1], 0);
    uint32_t key[C];
#pragma unroll
    for (int j = 0; j < C; ++j) key[j] = desc_key(v[j]);
    // (lds_keys == 2: 128 KB of LDS for the select's 16-bit histogram)
    uint32_t* hist16 = lds_keys == 2 ? reinterpret_cast<uint32_t*>(sel_pages) : nullptr;
    // page-slot view: written straight from the emit.  Token-slot view: the selected pages' slots go to LDS and the block
    // writes the page_size x longer view together, coalesced (16-byte stores from the few scattered emitting lanes
    // measured 5 us for 292 pages)
    int* sel_slots = sel_pages + (lds_keys == 2 ? 32768 : 0);
    uint32_t T;
    int take_eq;
    select_owned_threshold<C>(key, per, a.n_prev, a.prev_budget, scratch, T, take_eq, hist16);
    // (pin the page-slot loads between the two halves of the select: they have had the threshold search to land; left
    //  alone, the compiler sinks each one into the emit branch that uses it - up to `per` dependent round trips)
#pragma unroll
    for (int j = 0; j < C; ++j) asm volatile("" : "+v"(pslot[j]));
    select_owned_emit<C>(key, per, a.n_prev, T, take_eq, scratch, [&](int pos, int, uint32_t, int j) {
      if (a.emit_page_slots) packed[pos] = max(pslot[j], 0);
      else sel_slots[pos] = max(pslot[j], 0);
    });
    SVK_SEL_STAMP(3);
    if (tid == 0) a.local_lens[b] = a.prev_budget * ps + (len - (num_pages - 1) * ps);
    if (a.emit_page_slots) {
      if (tid == 0) packed[a.prev_budget] = last_slot;
      for (int j = a.prev_budget + 1 + tid; j < (a.max_keep + ps - 1) / ps; j += nt) packed[j] = 0;
    } else {
      __syncthreads();
      const int sparse_keep = (a.prev_budget + 1) * ps;
      for (int i = tid; i < sparse_keep; i += nt) {
        const int j = i / ps;
        packed[i] = (j < a.prev_budget ? sel_slots[j] : last_slot) * ps + (i - j * ps);
      }
      if (!a.is_long_text)
        for (int i = sparse_keep + tid; i < a.max_keep; i += nt) packed[i] = ttab[i];
    }
    SVK_SEL_STAMP(4);
    return;
  } else if (lds_keys) {
    // keys staged in LDS (8 scores per thread and trip with the loads first, the select's OR / AND sweep folded in)
    uint32_t* keys = reinterpret_cast<uint32_t*>(sel_pages + a.prev_budget);
    select_bits_begin(scratch);
    uint32_t o_bits = 0u, a_bits = 0xffffffffu;
    for (int i0 = 0; i0 < a.n_prev; i0 += 8 * nt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * nt + tid;
        v[j] = i < a.n_prev ? sc[i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * nt + tid;
        if (i < a.n_prev) {
          const uint32_t key = desc_key(v[j]);
          keys[i] = key;
          o_bits |= key;
          a_bits &= key;
        }
      }
    }
    select_bits_add(scratch, o_bits, a_bits);
    __syncthreads();
    block_select_topk_ordered_keys([keys](int i) { return keys[i]; }, a.n_prev, a.prev_budget, scratch,
                                   [&](int pos, int idx) { sel_pages[pos] = idx; }, true);
  } else {
    block_select_topk_ordered(sc, a.n_prev, a.prev_budget, scratch, [&](int pos, int idx) { sel_pages[pos] = idx; });
  }
  __syncthreads();
  SVK_SEL_STAMP(3);
  if (tid == 0) a.local_lens[b] = a.prev_budget * ps + (len - (num_pages - 1) * ps);
  if (a.emit_page_slots) {
    for (int j = tid; j < (a.max_keep + ps - 1) / ps; j += nt)
      packed[j] = j <= a.prev_budget ? max(ptab[j < a.prev_budget ? sel_pages[j] : num_pages - 1], 0) : 0;
    SVK_SEL_STAMP(4);
    return;
  }
  const int sparse_keep = (a.prev_budget + 1) * ps;
  for (int i0 = 0; i0 < sparse_keep; i0 += 8 * nt) {          // page-table gathers of 8 entries per thread in flight together
    int slot[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nt + tid;
      const int j = i / ps;
      slot[u] = i < sparse_keep ? ptab[j < a.prev_budget ? sel_pages[j] : num_pages - 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nt + tid;
      if (i < sparse_keep) packed[i] = max(slot[u], 0) * ps + (i - (i / ps) * ps);
    }
  }
  if (!a.is_long_text)
    for (int i = sparse_keep + tid; i < a.max_keep; i += nt) packed[i] = ttab[i];
  SVK_SEL_STAMP(4);
}

__global__ void __launch_bounds__(256) quest_decode_alloc_kernel(const SvkQuestDecodeAllocArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.graph_batch) return;
  if (b >= a.batch) {
    a.slot_mapping[b] = -1;
    a.context_lens[b] = a.cur_lens[0] + 1;
    a.req_indices[b] = a.row_ids[0];
    return;
  }
  const int row = a.row_ids[b], cur = a.cur_lens[b];
  const int page = cur / a.page_size, off = cur - page * a.page_size;
  int32_t* ptab = a.page_table + (int64_t)row * a.page_table_stride;
  int page_slot;
  if (off == 0) {
    page_slot = a.new_page_slots[b];
    ptab[page] = page_slot;
  } else {
    page_slot = ptab[page];
  }
  const int slot = page_slot * a.page_size + off;
  a.token_table[(int64_t)row * a.token_table_stride + cur] = slot;
  a.slot_mapping[b] = slot;
  a.context_lens[b] = cur + 1;
  a.req_indices[b] = row;
}

// ------------------------------------------------------------------------------------
// device-resident decode bookkeeping (svk.h SvkQuestDeviceStepArgs)
// ------------------------------------------------------------------------------------

// one workgroup: lane b of the batch.  Lanes whose row starts a new page are ranked in lane order (ballot + wave sums)
__global__ void __launch_bounds__(256) quest_device_begin_kernel(const SvkQuestDeviceStepArgs a) {
  __shared__ int s_wave[4], s_cur0, s_row0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ptr = *a.free_page_ptr;
  if (tid == 0) { s_row0 = a.row_ids[0]; s_cur0 = a.row_len[a.row_ids[0]]; }
  int base = 0;                                       // new pages taken by the lanes of earlier rounds
  for (int b0 = 0; b0 < a.graph_batch; b0 += blockDim.x) {
    const int b = b0 + tid;
    const bool real = b < a.batch;
    const int row = real ? a.row_ids[b] : 0;
    const int cur = real ? a.row_len[row] : 0;
    const bool need = real && (cur % a.page_size) == 0;
    const unsigned long long bal = __ballot(need);
    __syncthreads();                                  // (also orders s_cur0 / s_row0 and the previous round's s_wave reads)
    if (lane == 0) s_wave[w] = __popcll(bal);
    __syncthreads();
    int before = base + __popcll(bal & ((1ull << lane) - 1ull));
    int total = 0;
    for (int i = 0; i < 4; ++i) {
      if (i < w) before += s_wave[i];
      total += s_wave[i];
    }
    base += total;
    if (b < a.graph_batch) {
      if (!real) {                                    // padded graph lanes (quest_decode_alloc_kernel)
        a.slot_mapping[b] = -1;
        a.context_lens[b] = s_cur0 + 1;
        a.req_indices[b] = s_row0;
      } else {
        const int page = cur / a.page_size, off = cur - page * a.page_size;
        int32_t* ptab = a.page_table + (int64_t)row * a.page_table_stride;
        int page_slot;
        if (need) {
          page_slot = a.free_pages[ptr - 1 - before];
          ptab[page] = page_slot;
        } else {
          page_slot = ptab[page];
        }
        const int slot = page_slot * a.page_size + off;
        a.token_table[(int64_t)row * a.token_table_stride + cur] = slot;
        a.slot_mapping[b] = slot;
        a.context_lens[b] = cur + 1;
        a.req_indices[b] = row;
      }
    }
  }
  __syncthreads();                                    // every lane has read its length / the pointer
  for (int b = tid; b < a.batch; b += blockDim.x) a.row_len[a.row_ids[b]] += 1;
  if (tid == 0) *a.free_page_ptr = ptr - base;
}

// grid (lane, layer): the page a lane's row has just completed, if any
__global__ void __launch_bounds__(128) quest_device_end_kernel(const SvkQuestDeviceStepArgs a, int chunks_per_row) {
  const int b = blockIdx.x, layer = blockIdx.y;
  const int row = a.row_ids[b];
  const int n = a.row_len[row];
  if (n <= 0 || (n % a.page_size) != 0) return;
  const int64_t page = a.page_table[(int64_t)row * a.page_table_stride + n / a.page_size - 1];
  const uint16_t* k = a.k_cache + (int64_t)layer * a.k_layer_stride + page * a.page_size * (int64_t)a.row_elems;
  uint16_t* mx = a.metadata + (int64_t)layer * a.meta_layer_stride + page * (int64_t)a.row_elems;
  uint16_t* mn = mx + a.meta_kind_stride;
  for (int ch = threadIdx.x; ch < chunks_per_row; ch += blockDim.x) {
    uint4 hi = *reinterpret_cast<const uint4*>(k + ch * 8);
    uint4 lo = hi;
    for (int t = 1; t < a.page_size; ++t) {
      const uint4 v = *reinterpret_cast<const uint4*>(k + (int64_t)t * a.row_elems + ch * 8);
      hi.x = bf16x2_max(hi.x, v.x); hi.y = bf16x2_max(hi.y, v.y); hi.z = bf16x2_max(hi.z, v.z); hi.w = bf16x2_max(hi.w, v.w);
      lo.x = bf16x2_min(lo.x, v.x); lo.y = bf16x2_min(lo.y, v.y); lo.z = bf16x2_min(lo.z, v.z); lo.w = bf16x2_min(lo.w, v.w);
    }
    *reinterpret_cast<uint4*>(mx + ch * 8) = hi;
    *reinterpret_cast<uint4*>(mn + ch * 8) = lo;
  }
}

template <int D>
int dispatch_score(const SvkQuestScorePagesArgs& a, hipStream_t s) {
  const int G = a.num_q_heads / a.num_kv_heads;
  dim3 grid((a.n_prev + kPagesPerBlock - 1) / kPagesPerBlock, a.batch), block(64 * a.num_kv_heads);
  const size_t shm = sizeof(float) * kPagesPerBlock * a.num_kv_heads * ((G + 3) / 4);
  switch (G) {
#define SVK_CASE(G_) case G_: hipLaunchKernelGGL((quest_score_pages_kernel<D, G_>), grid, block, shm, s, a); break;
    SVK_CASE(1) SVK_CASE(2) SVK_CASE(3) SVK_CASE(4) SVK_CASE(5) SVK_CASE(6) SVK_CASE(7) SVK_CASE(8)
#undef SVK_CASE
    default:
      set_error("svk_quest_score_pages: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
  return check_launch("svk_quest_score_pages");
}

}  // namespace
}  // namespace svk

extern "C" int svk_quest_page_minmax(const SvkQuestPageMinmaxArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_quest_page_minmax: null args");
  SVK_REQUIRE(a->row_elems > 0 && a->row_elems % 8 == 0, SVK_ERR_LAYOUT, "svk_quest_page_minmax: row_elems %d must be a multiple of 8", a->row_elems);
  SVK_REQUIRE(a->page_size > 0, SVK_ERR_VALUE, "quest_chunk_size must be > 0");
  if (a->n_pages <= 0 || a->n_layers <= 0) return SVK_OK;
  const int cpr = a->row_elems / 8;
  hipLaunchKernelGGL(quest_page_minmax_kernel, dim3(a->n_pages, a->n_layers), dim3(cpr >= 128 ? 128 : 64), 0,
                     static_cast<hipStream_t>(stream), *a, cpr);
  return check_launch("svk_quest_page_minmax");
}

extern "C" int svk_quest_score_pages(const SvkQuestScorePagesArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_quest_score_pages: null args");
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_LAYOUT, "svk_quest_score_pages: head_dim %d unsupported (64, 128)", a->head_dim);
  SVK_REQUIRE(a->num_kv_heads >= 1 && a->num_kv_heads <= 8 && a->num_q_heads % a->num_kv_heads == 0, SVK_ERR_LAYOUT,
              "svk_quest_score_pages: unsupported head configuration %d/%d", a->num_q_heads, a->num_kv_heads);
  SVK_REQUIRE(a->page_size > 0, SVK_ERR_VALUE, "quest_chunk_size must be > 0");
  if (a->batch <= 0 || a->n_prev <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->head_dim == 128 ? dispatch_score<128>(*a, s) : dispatch_score<64>(*a, s);
}

extern "C" int svk_quest_build_view(const SvkQuestBuildViewArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_quest_build_view: null args");
  SVK_REQUIRE(a->prev_budget >= 1 && a->prev_budget <= a->n_prev, SVK_ERR_VALUE,
              "svk_quest_build_view: prev_budget %d must be in [1, n_prev=%d]", a->prev_budget, a->n_prev);
  SVK_REQUIRE(a->max_keep >= (a->prev_budget + 1) * a->page_size || a->is_long_text, SVK_ERR_VALUE,
              "svk_quest_build_view: max_keep %d smaller than the sparse view", a->max_keep);
  if (a->batch <= 0) return SVK_OK;
  // page-score keys live in registers (up to 32 per thread); longer rows stage them in LDS whenever the row fits
  size_t shm = sizeof(int) * a->prev_budget;
  const int lds_keys = (shm + sizeof(uint32_t) * (size_t)a->n_prev) <= 128 * 1024;
  if (lds_keys) shm += sizeof(uint32_t) * (size_t)a->n_prev;
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[] = {reinterpret_cast<const void*>(quest_build_view_kernel<0>), reinterpret_cast<const void*>(quest_build_view_kernel<4>),
                         reinterpret_cast<const void*>(quest_build_view_kernel<8>), reinterpret_cast<const void*>(quest_build_view_kernel<12>),
                         reinterpret_cast<const void*>(quest_build_view_kernel<16>), reinterpret_cast<const void*>(quest_build_view_kernel<24>),
                         reinterpret_cast<const void*>(quest_build_view_kernel<32>)};
    for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr_set = true;
  }
  const int nt = a->n_prev > 2048 ? 1024 : 256;
  const int per_thread = ((a->n_prev + nt - 1) / nt + 3) & ~3;           // (the kernel's `per`)
  hipStream_t s = static_cast<hipStream_t>(stream);
  // keys in registers; rows of the 1024-thread form also get 128 KB of LDS for the select's 16-bit histogram
  const size_t sel_bytes = a->emit_page_slots ? 0 : sizeof(int) * (size_t)a->prev_budget;
  const bool h16 = nt == 1024 && 128 * 1024 + sel_bytes <= 156 * 1024;
  const size_t shm_owned = (h16 ? 128 * 1024 : 0) + sel_bytes;
  const int mode = h16 ? 2 : 0;
  if (per_thread > 32) hipLaunchKernelGGL(quest_build_view_kernel<0>, dim3(a->batch), dim3(nt), shm, s, *a, lds_keys);
  else if (per_thread <= 4) hipLaunchKernelGGL(quest_build_view_kernel<4>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  else if (per_thread <= 8) hipLaunchKernelGGL(quest_build_view_kernel<8>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  else if (per_thread <= 12) hipLaunchKernelGGL(quest_build_view_kernel<12>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  else if (per_thread <= 16) hipLaunchKernelGGL(quest_build_view_kernel<16>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  else if (per_thread <= 24) hipLaunchKernelGGL(quest_build_view_kernel<24>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  else hipLaunchKernelGGL(quest_build_view_kernel<32>, dim3(a->batch), dim3(nt), shm_owned, s, *a, mode);
  return check_launch("svk_quest_build_view");
}

extern "C" int svk_quest_decode_alloc(const SvkQuestDecodeAllocArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_quest_decode_alloc: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  hipLaunchKernelGGL(quest_decode_alloc_kernel, dim3((a->graph_batch + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_quest_decode_alloc");
}

extern "C" int svk_quest_device_step_begin(const SvkQuestDeviceStepArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->row_len != nullptr && a->free_pages != nullptr && a->free_page_ptr != nullptr && a->row_ids != nullptr,
              SVK_ERR_VALUE, "svk_quest_device_step_begin: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  SVK_REQUIRE(a->page_size > 0, SVK_ERR_VALUE, "quest_chunk_size must be > 0");
  hipLaunchKernelGGL(quest_device_begin_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_quest_device_step_begin");
}

extern "C" int svk_quest_device_step_end(const SvkQuestDeviceStepArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->row_len != nullptr && a->row_ids != nullptr && a->k_cache != nullptr && a->metadata != nullptr,
              SVK_ERR_VALUE, "svk_quest_device_step_end: null args");
  SVK_REQUIRE(a->row_elems > 0 && a->row_elems % 8 == 0, SVK_ERR_LAYOUT, "svk_quest_device_step_end: row_elems %d must be a multiple of 8", a->row_elems);
  SVK_REQUIRE(a->page_size > 0, SVK_ERR_VALUE, "quest_chunk_size must be > 0");
  if (a->batch <= 0 || a->n_layers <= 0) return SVK_OK;
  const int cpr = a->row_elems / 8;
  hipLaunchKernelGGL(quest_device_end_kernel, dim3(a->batch, a->n_layers), dim3(cpr >= 128 ? 128 : 64), 0,
                     static_cast<hipStream_t>(stream), *a, cpr);
  return check_launch("svk_quest_device_step_end");
}
