// H2O / SnapKV-family bookkeeping kernels: score normalisation + accumulation, exact
// top-k selection with the reference's tie rule, slot-table compaction, decode slot
// allocation.  gfx950 only.  Integer results are bit-exact with the reference's torch
// code (include/svk.h cites the lines); these are HBM/L2-bound index kernels.

#include "svk_common.hpp"
#include "svk_select.hpp"

namespace svk {
namespace {

// ------------------------------------------------------------------------------------
// fill
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) fill_f32_kernel(float* dst, int64_t n, float v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(dst + i) & 15) == 0)) {
      *reinterpret_cast<float4*>(dst + i) = make_float4(v, v, v, v);
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) dst[j] = v;
    }
  }
}

// ------------------------------------------------------------------------------------
// decode score: x *= scale; softmax over the full width; optional cumulative update
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(1024) h2o_decode_score_kernel(const SvkH2oDecodeScoreArgs a) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  float* x = a.attn_score + (int64_t)b * a.score_stride_b;
  const int W = a.width;
  const int vlen = a.mask_by_len ? a.b_seqlen[b] : W;      // positions >= vlen read as the -1e20 fill (SvkH2oDecodeScoreArgs)
  auto raw = [&](int t) { return t < vlen ? x[t] : -1e20f; };
  float mx = -INFINITY;
  // __fmul_rn keeps `x * scale` a separately rounded product like torch's mul_ (no fma contraction)
  for (int t = threadIdx.x; t < W; t += blockDim.x) mx = fmaxf(mx, mul_rn(raw(t), a.scale));
  mx = block_allmax(mx, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < W; t += blockDim.x) sum += expf(mul_rn(raw(t), a.scale) - mx);
  sum = block_allsum(sum, red);
  float* cum = nullptr;
  int len = 0;
  if (a.cum_score != nullptr && !(a.b_new_slot != nullptr && a.b_new_slot[b] < 0)) {   // padded graph lanes: no update
    cum = a.cum_score + (int64_t)a.b_req_idx[b] * a.cum_stride;
    len = a.b_seqlen[b];
  }
  for (int t = threadIdx.x; t < W; t += blockDim.x) {
    const float p = expf(mul_rn(raw(t), a.scale) - mx) / sum;
    x[t] = p;
    if (cum != nullptr && t < len) cum[t] = (t == len - 1) ? p : cum[t] + p;   // pad(prev, 1) + p
  }
}

// All layers of a decode step in one launch (grid = (batch lane, layer)): the arithmetic of h2o_decode_score_kernel
// with the layer's pointers.  The row lives in registers (one read, one write of the raw scores and of the cumulative
// row).  256 threads x 16-byte accesses: eight workgroups per CU keep 8 x 256 x 2 x E4 loads in flight (with 1024 scalar
// threads per row two workgroups fit a CU and the 3584 rows of a B=128 step took 77 us = 3.1 TB/s).
template <int E4, bool VEC>
__global__ void __launch_bounds__(256) h2o_decode_score_layers_kernel(const SvkH2oDecodeScoreArgs a, int64_t score_stride_layer,
                                                                      int64_t cum_stride_layer, int64_t new_slot_stride_layer,
                                                                      int64_t req_stride_layer, int64_t seqlen_stride_layer) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const int64_t l = blockIdx.y;
  float* x = a.attn_score + l * score_stride_layer + (int64_t)b * a.score_stride_b;
  const int W = a.width;                  // VEC: a multiple of 4 and every row 16-byte aligned (launcher)
  float* cum = nullptr;
  int len = 0;
  if (a.cum_score != nullptr && !(a.b_new_slot != nullptr && a.b_new_slot[l * new_slot_stride_layer + b] < 0)) {   // padded graph lanes
    cum = a.cum_score + l * cum_stride_layer + (int64_t)a.b_req_idx[l * req_stride_layer + b] * a.cum_stride;
    len = a.b_seqlen[l * seqlen_stride_layer + b];
  }
  // mask_by_len: the raw scores were stored without a -1e20 pre-fill; positions at or beyond the row's length read as the
  // fill value whatever the buffer holds (bit-identical to the pre-filled form)
  const int vlen = a.mask_by_len ? a.b_seqlen[l * seqlen_stride_layer + b] : W;
  float v[E4][4], c[E4][4];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    const int t = (threadIdx.x + i * 256) * 4;
    if constexpr (VEC) {
      float4 xv = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), cv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < W) xv = *reinterpret_cast<const float4*>(x + t);
      if (t + 0 >= vlen) xv.x = -1e20f;
      if (t + 1 >= vlen) xv.y = -1e20f;
      if (t + 2 >= vlen) xv.z = -1e20f;
      if (t + 3 >= vlen) xv.w = -1e20f;
      if (cum != nullptr && t < len - 1) cv = *reinterpret_cast<const float4*>(cum + t);   // (elements >= len - 1 are not used)
      v[i][0] = t < W ? mul_rn(xv.x, a.scale) : -INFINITY; v[i][1] = t < W ? mul_rn(xv.y, a.scale) : -INFINITY;
      v[i][2] = t < W ? mul_rn(xv.z, a.scale) : -INFINITY; v[i][3] = t < W ? mul_rn(xv.w, a.scale) : -INFINITY;
      c[i][0] = cv.x; c[i][1] = cv.y; c[i][2] = cv.z; c[i][3] = cv.w;
    } else {
      // same element -> thread mapping and summation order with scalar accesses (any width / alignment: the eager path
      // sizes the rows to the current maximum length): bit-identical to the 16-byte form
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = t + e < W ? mul_rn(t + e < vlen ? x[t + e] : -1e20f, a.scale) : -INFINITY;
        c[i][e] = (cum != nullptr && t + e < len - 1) ? cum[t + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[i][e]);
  }
  mx = block_allmax(mx, red);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < E4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = expf(v[i][e] - mx);          // exp(-inf) = 0 for the padding lanes
      sum += v[i][e];
    }
  sum = block_allsum(sum, red);
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    const int t = (threadIdx.x + i * 256) * 4;
    if (t < W) {
      float p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = v[i][e] / sum;
      if constexpr (VEC) {
        *reinterpret_cast<float4*>(x + t) = make_float4(p[0], p[1], p[2], p[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e < W) x[t + e] = p[e];
      }
      if (cum != nullptr && t < len) {
        // pad(prev, 1) + p: positions < len - 1 add, position len - 1 starts at p, positions >= len stay untouched
        if (VEC && t + 3 < len - 1) {
          *reinterpret_cast<float4*>(cum + t) = make_float4(c[i][0] + p[0], c[i][1] + p[1], c[i][2] + p[2], c[i][3] + p[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e < len) cum[t + e] = (t + e == len - 1) ? p[e] : c[i][e] + p[e];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// exact H2O selection
// ------------------------------------------------------------------------------------

__device__ __forceinline__ void h2o_select_row(const float* sc, int64_t* keep, int kv_len, int budget, int recent,
                                               SelectScratch& scratch) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (kv_len <= budget) {
    for (int i = tid; i < kv_len; i += nt) keep[i] = i;
    return;
  }
  const int heavy = budget - recent;
  const int rs = kv_len - recent;      // recent_start; heavy < rs always holds here
  for (int i = tid; i < recent; i += nt) keep[heavy + i] = rs + i;
  if (heavy <= 0) return;
  // top-`heavy` of sc[0:rs] by (score desc, index asc), emitted in ascending index order
  block_select_topk_ordered(sc, rs, heavy, scratch, [&](int pos, int idx) { keep[pos] = idx; });
}

__global__ void __launch_bounds__(256) h2o_select_kernel(const SvkH2oSelectArgs a) {
  __shared__ SelectScratch scratch;
  const int rowi = blockIdx.x;
  h2o_select_row(a.scores + (int64_t)rowi * a.score_stride, a.keep + (int64_t)rowi * a.keep_stride, a.kv_len, a.budget,
                 a.recent_count, scratch);
}

__global__ void __launch_bounds__(256) select_topk_kernel(const SvkSelectTopkArgs a) {
  __shared__ SelectScratch scratch;
  const float* sc = a.scores + (int64_t)blockIdx.x * a.score_stride;
  int64_t* keep = a.keep + (int64_t)blockIdx.x * a.keep_stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int mid = a.kv_len - a.suffix - a.prefix;
  for (int i = tid; i < a.prefix; i += nt) keep[i] = i;
  for (int i = tid; i < a.suffix; i += nt) keep[a.prefix + a.topk + i] = a.kv_len - a.suffix + i;
  if (a.topk <= 0) return;
  if (a.topk >= mid) {
    for (int i = tid; i < mid; i += nt) keep[a.prefix + i] = a.prefix + i;
    return;
  }
  block_select_topk_ordered(sc + a.prefix, mid, a.topk, scratch,
                            [&](int pos, int idx) { keep[a.prefix + pos] = a.prefix + idx; });
}

// ------------------------------------------------------------------------------------
// slot-table compaction
// ------------------------------------------------------------------------------------

__device__ __forceinline__ void compact_row(int32_t* tab, int32_t* stack, float* pay, const int64_t* keep, int K, int cur) {
  const int tid = threadIdx.x, nt = blockDim.x;
  // (A) dropped slots -> free stack, ascending position.  Position q with j = |{keep < q}|
  //     kept entries before it lands at out[q - j]   (snapkv.py:1754-1768 row-major mask order)
  for (int q = tid; q < cur; q += nt) {
    int lo = 0, hi = K;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keep[mid] < q) lo = mid + 1; else hi = mid;
    }
    if (!(lo < K && keep[lo] == q)) stack[q - lo] = tab[q];
  }
  __syncthreads();

  // (B) in-place gather new[j] = old[keep[j]]: keep is ascending so keep[j] >= j; a chunk
  //     reads only positions >= its own first index, hence chunk-by-chunk is hazard free.
  for (int c0 = 0; c0 < K; c0 += nt) {
    const int j = c0 + tid;
    int32_t s = 0;
    float p = 0.f;
    if (j < K) {
      const int64_t src = keep[j];
      s = tab[src];
      if (pay) p = pay[src];
    }
    __syncthreads();
    if (j < K) {
      tab[j] = s;
      if (pay) pay[j] = p;
    }
    __syncthreads();
  }
  // (C) zero the tail (snapkv.py:1790-1799)
  for (int q = K + tid; q < cur; q += nt) {
    tab[q] = 0;
    if (pay) pay[q] = 0.f;
  }
}

__global__ void __launch_bounds__(256) compact_rows_kernel(const SvkCompactRowsArgs a) {
  const int lane_i = blockIdx.x, li = blockIdx.y;
  const int layer = a.layer_ids[li];
  const int row = a.row_ids[(int64_t)li * a.n_lanes + lane_i];
  const int K = a.keep_len, cur = a.cur_len;
  const int64_t* keep = a.keep + ((int64_t)li * a.n_lanes + lane_i) * K;
  int32_t* tab = a.slot_table + (int64_t)layer * a.table_stride_layer + (int64_t)row * a.table_stride_row;
  int32_t* stack = a.free_stack + (int64_t)layer * a.stack_stride + a.free_base[li] + (int64_t)lane_i * (cur - K);
  float* pay = a.row_payload ? a.row_payload + (int64_t)layer * a.payload_stride_layer + (int64_t)row * a.payload_stride_row
                             : nullptr;
  compact_row(tab, stack, pay, keep, K, cur);
}

// ------------------------------------------------------------------------------------
// device-resident decode bookkeeping (svk.h SvkH2oDeviceStepArgs)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) h2o_device_begin_kernel(const SvkH2oDeviceStepArgs a) {
  const int l = blockIdx.x, tid = threadIdx.x;
  int32_t* lens = a.row_len + (int64_t)l * a.rows_total;
  const int64_t ptr = a.free_ptr[l];
  const int row0 = a.row_ids[0];
  const int cur0 = lens[row0];                                         // (read before any lane of this layer writes it)
  __syncthreads();
  int32_t* sm = a.slot_mapping + (int64_t)l * a.out_stride;
  int32_t* cl = a.context_lens + (int64_t)l * a.out_stride;
  int32_t* ri = a.req_indices + (int64_t)l * a.out_stride;
  for (int b = tid; b < a.graph_batch; b += blockDim.x) {
    if (b >= a.batch) {                                               // padded graph lanes (h2o.py:419-424)
      sm[b] = -1;
      cl[b] = cur0 + 1;
      ri[b] = row0;
      continue;
    }
    const int row = a.row_ids[b];
    const int cur = lens[row];
    const int32_t slot = a.free_stack[(int64_t)l * a.stack_stride + ptr - a.batch + b];
    a.slot_table[(int64_t)l * a.table_stride_layer + (int64_t)row * a.table_stride_row + cur] = slot;
    sm[b] = slot;
    cl[b] = cur + 1;
    ri[b] = row;
    lens[row] = cur + 1;
  }
  if (tid == 0) a.free_ptr[l] = ptr - a.batch;
}

__global__ void __launch_bounds__(256) h2o_device_select_kernel(const SvkH2oDeviceStepArgs a) {
  __shared__ SelectScratch scratch;
  const int b = blockIdx.x, l = blockIdx.y;
  const int row = a.row_ids[b];
  if (a.row_len[(int64_t)l * a.rows_total + row] != a.trigger_len) return;
  int64_t* keep = a.keep + ((int64_t)l * a.batch + b) * a.budget;
  if (a.select_mode == SVK_DEVICE_SELECT_SNAPKV) {
    // sink ++ top-k of the middle by this step's head-max raw scores (lane-indexed rows) ++ recent
    // (SparseController._snapkv_select_indices_batch, sparse_controller.py:1670-1747), ascending
    const float* sc = a.scores + (int64_t)l * a.score_stride_layer + (int64_t)b * a.score_stride_row;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int prefix = a.prefix_count, suffix = a.recent_count, topk = a.budget - prefix - suffix;
    const int mid = a.trigger_len - suffix - prefix;
    for (int i = tid; i < prefix; i += nt) keep[i] = i;
    for (int i = tid; i < suffix; i += nt) keep[prefix + topk + i] = a.trigger_len - suffix + i;
    if (topk <= 0) return;
    block_select_topk_ordered(sc + prefix, mid, topk, scratch, [&](int pos, int idx) { keep[prefix + pos] = prefix + idx; });
    return;
  }
  if (a.select_mode == SVK_DEVICE_SELECT_WINDOW) {
    // sink + recent window (StreamingLLM, sparse_controller.py:1661-1668): [0, budget - recent) ++ [len - recent, len)
    const int prefix = a.budget - a.recent_count;
    for (int i = threadIdx.x; i < a.budget; i += blockDim.x) keep[i] = i < prefix ? i : a.trigger_len - a.budget + i;
    return;
  }
  h2o_select_row(a.scores + (int64_t)l * a.score_stride_layer + (int64_t)row * a.score_stride_row, keep, a.trigger_len,
                 a.budget, a.recent_count, scratch);
}

__global__ void __launch_bounds__(256) h2o_device_compact_kernel(const SvkH2oDeviceStepArgs a) {
  __shared__ int s_rank[4];
  const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int32_t* lens = a.row_len + (int64_t)l * a.rows_total;
  const int row = a.row_ids[b];
  if (lens[row] != a.trigger_len) return;
  // lanes in front of this one that evict too: the host-driven burst hands the lanes of a group consecutive pieces of
  // the free stack in lane order
  int cnt = 0;
  for (int j = tid; j < b; j += blockDim.x) cnt += lens[a.row_ids[j]] == a.trigger_len ? 1 : 0;
  cnt = (int)wave_allsum((float)cnt);
  if ((tid & 63) == 0) s_rank[tid >> 6] = cnt;
  __syncthreads();
  const int rank = s_rank[0] + s_rank[1] + s_rank[2] + s_rank[3];
  const int K = a.budget, cur = a.trigger_len;
  int32_t* tab = a.slot_table + (int64_t)l * a.table_stride_layer + (int64_t)row * a.table_stride_row;
  int32_t* stack = a.free_stack + (int64_t)l * a.stack_stride + a.free_ptr[l] + (int64_t)rank * (cur - K);
  // (SnapKV's scores are this step's lane-indexed scratch rows, not a payload that lives with the slot table)
  float* pay = (a.scores && a.select_mode == SVK_DEVICE_SELECT_H2O)
                   ? a.scores + (int64_t)l * a.score_stride_layer + (int64_t)row * a.score_stride_row : nullptr;
  compact_row(tab, stack, pay, a.keep + ((int64_t)l * a.batch + b) * K, K, cur);
}

__global__ void __launch_bounds__(256) h2o_device_commit_kernel(const SvkH2oDeviceStepArgs a) {
  __shared__ int s_cnt[4];
  const int l = blockIdx.x, tid = threadIdx.x;
  int32_t* lens = a.row_len + (int64_t)l * a.rows_total;
  int cnt = 0;
  for (int b = tid; b < a.batch; b += blockDim.x) {
    const int row = a.row_ids[b];
    if (lens[row] == a.trigger_len) {
      lens[row] = a.budget;
      ++cnt;
    }
  }
  cnt = (int)wave_allsum((float)cnt);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (total > 0) a.free_ptr[l] += (int64_t)total * (a.trigger_len - a.budget);
  }
}

// The three burst phases in ONE launch (round 4; grid = (lane, layer) like the select / compact kernels): a workgroup
// whose row fired selects and compacts it exactly as above - its rank among the fired lanes, their number and the layer's
// stack pointer are read before anything of the layer changes - and then takes a ticket; the last of the layer's FIRED
// workgroups to arrive has seen the others finish, so it alone commits the layer (row_len of the fired rows, free_ptr) and
// resets the ticket.  A workgroup whose row did not fire returns on its first load and touches no ticket (with every
// workgroup taking one, the idle launch of 127 steps out of 128 cost 79 us at B=256: 256 same-address device-scope
// atomics per layer, ~300 ns each).  Nothing crosses between workgroups except the ticket count: no fence, no hand-over
// of data.  Two graph nodes less per decode step.
__global__ void __launch_bounds__(256) h2o_device_burst_kernel(const SvkH2oDeviceStepArgs a) {
  __shared__ SelectScratch scratch;
  __shared__ int s_rank[4], s_all[4], s_last;
  const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  int n_fired = 0;
  int32_t* lens = a.row_len + (int64_t)l * a.rows_total;
  const int row = a.row_ids[b];
  if (lens[row] != a.trigger_len) return;
  {
    int64_t* keep = a.keep + ((int64_t)l * a.batch + b) * a.budget;
    if (a.select_mode == SVK_DEVICE_SELECT_SNAPKV) {
      const float* sc = a.scores + (int64_t)l * a.score_stride_layer + (int64_t)b * a.score_stride_row;
      const int nt = blockDim.x;
      const int prefix = a.prefix_count, suffix = a.recent_count, topk = a.budget - prefix - suffix;
      const int mid = a.trigger_len - suffix - prefix;
      for (int i = tid; i < prefix; i += nt) keep[i] = i;
      for (int i = tid; i < suffix; i += nt) keep[prefix + topk + i] = a.trigger_len - suffix + i;
      if (topk > 0)
        block_select_topk_ordered(sc + prefix, mid, topk, scratch, [&](int pos, int idx) { keep[prefix + pos] = prefix + idx; });
    } else if (a.select_mode == SVK_DEVICE_SELECT_WINDOW) {
      const int prefix = a.budget - a.recent_count;
      for (int i = tid; i < a.budget; i += blockDim.x) keep[i] = i < prefix ? i : a.trigger_len - a.budget + i;
    } else {
      h2o_select_row(a.scores + (int64_t)l * a.score_stride_layer + (int64_t)row * a.score_stride_row, keep, a.trigger_len,
                     a.budget, a.recent_count, scratch);
    }
    __syncthreads();                                  // the row's keep list is complete (this workgroup wrote all of it)
    int cnt = 0, all = 0;
    for (int j = tid; j < a.batch; j += blockDim.x) {
      const int f = lens[a.row_ids[j]] == a.trigger_len ? 1 : 0;
      cnt += j < b ? f : 0;
      all += f;
    }
    cnt = (int)wave_allsum((float)cnt);
    all = (int)wave_allsum((float)all);
    if ((tid & 63) == 0) { s_rank[tid >> 6] = cnt; s_all[tid >> 6] = all; }
    __syncthreads();
    const int rank = s_rank[0] + s_rank[1] + s_rank[2] + s_rank[3];
    n_fired = s_all[0] + s_all[1] + s_all[2] + s_all[3];
    const int K = a.budget, cur = a.trigger_len;
    int32_t* tab = a.slot_table + (int64_t)l * a.table_stride_layer + (int64_t)row * a.table_stride_row;
    int32_t* stack = a.free_stack + (int64_t)l * a.stack_stride + a.free_ptr[l] + (int64_t)rank * (cur - K);
    float* pay = (a.scores && a.select_mode == SVK_DEVICE_SELECT_H2O)
                     ? a.scores + (int64_t)l * a.score_stride_layer + (int64_t)row * a.score_stride_row : nullptr;
    compact_row(tab, stack, pay, keep, K, cur);
  }
  __syncthreads();                                    // every thread of this workgroup has done its reads of lens / free_ptr
  if (tid == 0) s_last = atomicAdd(&a.tickets[l], 1) == n_fired - 1;
  __syncthreads();
  if (!s_last) return;
  // the layer's last fired workgroup: the others have finished (their tickets are in), nobody reads lens / free_ptr any more
  int cnt = 0;
  for (int j = tid; j < a.batch; j += blockDim.x) {
    const int r = a.row_ids[j];
    if (lens[r] == a.trigger_len) {
      lens[r] = a.budget;
      ++cnt;
    }
  }
  cnt = (int)wave_allsum((float)cnt);
  __syncthreads();
  if ((tid & 63) == 0) s_rank[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    const int total = s_rank[0] + s_rank[1] + s_rank[2] + s_rank[3];
    if (total > 0) a.free_ptr[l] += (int64_t)total * (a.trigger_len - a.budget);
    a.tickets[l] = 0;                                 // self-cleaning for the next launch
  }
}

// ------------------------------------------------------------------------------------
// decode slot allocation (all layers)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) decode_alloc_kernel(const SvkDecodeAllocArgs a) {
  const int li = blockIdx.y;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.graph_batch) return;
  const int layer = a.layer_ids[li];
  int32_t* sm = a.slot_mapping + (int64_t)li * a.out_stride;
  int32_t* cl = a.context_lens + (int64_t)li * a.out_stride;
  int32_t* ri = a.req_indices + (int64_t)li * a.out_stride;
  // non-uniform layers (snapkv.py:2656-2673): rows, lengths and the stack pointer are per layer
  const int32_t* row_ids = a.row_ids + (int64_t)li * a.meta_stride_layer;
  const int32_t* cur_lens = a.cur_lens + (int64_t)li * a.meta_stride_layer;
  const int64_t free_ptr = a.free_ptrs != nullptr ? a.free_ptrs[li] : a.free_ptr;
  if (b >= a.batch) {
    // padded graph lanes: slot -1, metadata of lane 0 (h2o.py:419-424 index_fill_)
    sm[b] = -1;
    cl[b] = cur_lens[0] + 1;
    ri[b] = row_ids[0];
    return;
  }
  const int32_t slot = a.free_stack[(int64_t)layer * a.stack_stride + free_ptr - a.batch + b];
  const int row = row_ids[b];
  const int cur = cur_lens[b];
  a.slot_table[(int64_t)layer * a.table_stride_layer + (int64_t)row * a.table_stride_row + cur] = slot;
  sm[b] = slot;
  cl[b] = cur + 1;
  ri[b] = row;
}

}  // namespace
}  // namespace svk

extern "C" int svk_fill_f32(float* dst, int64_t n, float value, svk_stream_t stream) {
  using namespace svk;
  if (n <= 0) return SVK_OK;
  SVK_REQUIRE(dst != nullptr, SVK_ERR_VALUE, "svk_fill_f32: null dst");
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dst, n, value);
  return check_launch("svk_fill_f32");
}

extern "C" int svk_h2o_decode_score_update(const SvkH2oDecodeScoreArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->attn_score != nullptr, SVK_ERR_VALUE, "svk_h2o_decode_score_update: null args");
  SVK_REQUIRE(a->width > 0, SVK_ERR_VALUE, "svk_h2o_decode_score_update: width must be positive");
  SVK_REQUIRE(a->cum_score == nullptr || (a->b_req_idx != nullptr && a->b_seqlen != nullptr), SVK_ERR_VALUE,
              "svk_h2o_decode_score_update: cum_score needs b_req_idx and b_seqlen");
  if (a->batch <= 0) return SVK_OK;
  const int threads = a->width >= 4096 ? 1024 : (a->width >= 1024 ? 512 : 256);
  hipLaunchKernelGGL(h2o_decode_score_kernel, dim3(a->batch), dim3(threads), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_h2o_decode_score_update");
}

extern "C" int svk_h2o_decode_score_update_layers(const SvkH2oDecodeScoreArgs* first, int32_t n_layers, int64_t score_stride_layer,
                                                  int64_t cum_stride_layer, int64_t new_slot_stride_layer, int64_t req_stride_layer,
                                                  int64_t seqlen_stride_layer, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(first != nullptr && first->attn_score != nullptr, SVK_ERR_VALUE, "svk_h2o_decode_score_update_layers: null args");
  SVK_REQUIRE(first->width > 0 && n_layers >= 0, SVK_ERR_VALUE, "svk_h2o_decode_score_update_layers: bad shape");
  SVK_REQUIRE(!first->mask_by_len || first->b_seqlen != nullptr, SVK_ERR_VALUE, "svk_h2o_decode_score_update_layers: mask_by_len needs b_seqlen");
  SVK_REQUIRE(first->cum_score == nullptr || (first->b_req_idx != nullptr && first->b_seqlen != nullptr), SVK_ERR_VALUE,
              "svk_h2o_decode_score_update_layers: cum_score needs b_req_idx and b_seqlen");
  if (first->batch <= 0 || n_layers == 0) return SVK_OK;
  const bool vec_ok = first->width % 4 == 0 && first->score_stride_b % 4 == 0 && score_stride_layer % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(first->attn_score) & 15) == 0 &&
                      (first->cum_score == nullptr || (first->cum_stride % 4 == 0 && cum_stride_layer % 4 == 0 &&
                                                       (reinterpret_cast<uintptr_t>(first->cum_score) & 15) == 0));
  if (first->width > 1024 * 32) {      // rows that do not fit the register-resident form: layer by layer
    for (int l = 0; l < n_layers; ++l) {
      SvkH2oDecodeScoreArgs a = *first;
      a.attn_score += (int64_t)l * score_stride_layer;
      if (a.cum_score != nullptr) a.cum_score += (int64_t)l * cum_stride_layer;
      if (a.b_new_slot != nullptr) a.b_new_slot += (int64_t)l * new_slot_stride_layer;
      if (a.b_req_idx != nullptr) a.b_req_idx += (int64_t)l * req_stride_layer;
      if (a.b_seqlen != nullptr) a.b_seqlen += (int64_t)l * seqlen_stride_layer;
      const int rc = svk_h2o_decode_score_update(&a, stream);
      if (rc != SVK_OK) return rc;
    }
    return SVK_OK;
  }
  dim3 grid(first->batch, n_layers);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int e4 = ((first->width + 3) / 4 + 255) / 256;
#define SVK_SCORE_LAYERS(E4_)                                                                                              \
  do {                                                                                                                     \
    if (vec_ok)                                                                                                            \
      hipLaunchKernelGGL((h2o_decode_score_layers_kernel<E4_, true>), grid, dim3(256), 0, s, *first, score_stride_layer,   \
                         cum_stride_layer, new_slot_stride_layer, req_stride_layer, seqlen_stride_layer);                  \
    else                                                                                                                   \
      hipLaunchKernelGGL((h2o_decode_score_layers_kernel<E4_, false>), grid, dim3(256), 0, s, *first, score_stride_layer,  \
                         cum_stride_layer, new_slot_stride_layer, req_stride_layer, seqlen_stride_layer);                  \
  } while (0)
  if (e4 <= 2) SVK_SCORE_LAYERS(2);
  else if (e4 <= 5) SVK_SCORE_LAYERS(5);
  else if (e4 <= 8) SVK_SCORE_LAYERS(8);
  else if (e4 <= 16) SVK_SCORE_LAYERS(16);
  else SVK_SCORE_LAYERS(32);
#undef SVK_SCORE_LAYERS
  return check_launch("svk_h2o_decode_score_update_layers");
}

extern "C" int svk_h2o_select_indices(const SvkH2oSelectArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_h2o_select_indices: null args");
  SVK_REQUIRE(a->budget > 0, SVK_ERR_VALUE, "H2O budget must be positive, got %d.", a->budget);
  SVK_REQUIRE(a->kv_len >= 0 && a->rows >= 0, SVK_ERR_VALUE, "svk_h2o_select_indices: negative shape");
  if (a->kv_len > a->budget) {
    SVK_REQUIRE(a->recent_count >= 1 && a->recent_count <= a->budget && a->recent_count <= a->kv_len, SVK_ERR_VALUE,
                "svk_h2o_select_indices: recent_count %d out of range (budget %d, kv_len %d)", a->recent_count, a->budget, a->kv_len);
  }
  if (a->rows == 0 || a->kv_len == 0) return SVK_OK;
  hipLaunchKernelGGL(h2o_select_kernel, dim3(a->rows), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_h2o_select_indices");
}

extern "C" int svk_select_prefix_topk_suffix(const SvkSelectTopkArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_select_prefix_topk_suffix: null args");
  SVK_REQUIRE(a->prefix >= 0 && a->topk >= 0 && a->suffix >= 0, SVK_ERR_VALUE, "svk_select_prefix_topk_suffix: negative counts");
  SVK_REQUIRE(a->prefix + a->suffix <= a->kv_len && a->topk <= a->kv_len - a->prefix - a->suffix, SVK_ERR_VALUE,
              "svk_select_prefix_topk_suffix: prefix %d + topk %d + suffix %d exceeds kv_len %d", a->prefix, a->topk, a->suffix, a->kv_len);
  if (a->rows <= 0 || a->prefix + a->topk + a->suffix == 0) return SVK_OK;
  hipLaunchKernelGGL(select_topk_kernel, dim3(a->rows), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_select_prefix_topk_suffix");
}

extern "C" int svk_compact_rows(const SvkCompactRowsArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_compact_rows: null args");
  SVK_REQUIRE(a->keep_len > 0, SVK_ERR_STATE, "free_part_slots got empty keep_indices");
  SVK_REQUIRE(a->keep_len <= a->cur_len, SVK_ERR_STATE, "svk_compact_rows: keep_len %d exceeds cur_len %d", a->keep_len, a->cur_len);
  if (a->n_layers <= 0 || a->n_lanes <= 0) return SVK_OK;
  hipLaunchKernelGGL(compact_rows_kernel, dim3(a->n_lanes, a->n_layers), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_compact_rows");
}

extern "C" int svk_decode_alloc_slots(const SvkDecodeAllocArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_decode_alloc_slots: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  SVK_REQUIRE(a->free_ptrs != nullptr || a->free_ptr >= a->batch, SVK_ERR_STATE,
              "Out of KV cache slots in static decode: need=%d free=%lld.", a->batch, (long long)a->free_ptr);
  if (a->n_layers <= 0) return SVK_OK;
  hipLaunchKernelGGL(decode_alloc_kernel, dim3((a->graph_batch + 255) / 256, a->n_layers), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_decode_alloc_slots");
}

extern "C" int svk_h2o_device_step_begin(const SvkH2oDeviceStepArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->row_len != nullptr && a->free_ptr != nullptr && a->row_ids != nullptr, SVK_ERR_VALUE,
              "svk_h2o_device_step_begin: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  if (a->n_layers <= 0) return SVK_OK;
  hipLaunchKernelGGL(h2o_device_begin_kernel, dim3(a->n_layers), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_h2o_device_step_begin");
}

extern "C" int svk_h2o_device_burst(const SvkH2oDeviceStepArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->row_len != nullptr && a->free_ptr != nullptr && a->row_ids != nullptr && a->keep != nullptr,
              SVK_ERR_VALUE, "svk_h2o_device_burst: null args");
  SVK_REQUIRE(a->select_mode == SVK_DEVICE_SELECT_H2O || a->select_mode == SVK_DEVICE_SELECT_WINDOW ||
                  a->select_mode == SVK_DEVICE_SELECT_SNAPKV, SVK_ERR_VALUE, "svk_h2o_device_burst: bad select_mode %d", a->select_mode);
  SVK_REQUIRE(a->select_mode == SVK_DEVICE_SELECT_WINDOW || a->scores != nullptr, SVK_ERR_VALUE,
              "svk_h2o_device_burst: score-based selections need the score tensor");
  SVK_REQUIRE(a->select_mode != SVK_DEVICE_SELECT_SNAPKV ||
                  (a->prefix_count >= 0 && a->prefix_count + a->recent_count <= a->budget &&
                   a->budget - a->prefix_count - a->recent_count <= a->trigger_len - a->prefix_count - a->recent_count), SVK_ERR_VALUE,
              "svk_h2o_device_burst: sink %d + recent %d do not fit the budget %d", a->prefix_count, a->recent_count, a->budget);
  SVK_REQUIRE(a->budget > 0 && a->trigger_len > a->budget, SVK_ERR_VALUE,
              "svk_h2o_device_burst: trigger_len %d must exceed the budget %d", a->trigger_len, a->budget);
  SVK_REQUIRE(a->recent_count >= 1 && a->recent_count <= a->budget, SVK_ERR_VALUE,
              "svk_h2o_device_burst: recent_count %d out of range (budget %d)", a->recent_count, a->budget);
  if (a->n_layers <= 0 || a->batch <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->tickets != nullptr) {     // one launch: select + compact + ticketed commit
    hipLaunchKernelGGL(h2o_device_burst_kernel, dim3(a->batch, a->n_layers), dim3(256), 0, s, *a);
    return check_launch("svk_h2o_device_burst");
  }
  hipLaunchKernelGGL(h2o_device_select_kernel, dim3(a->batch, a->n_layers), dim3(256), 0, s, *a);
  hipLaunchKernelGGL(h2o_device_compact_kernel, dim3(a->batch, a->n_layers), dim3(256), 0, s, *a);
  hipLaunchKernelGGL(h2o_device_commit_kernel, dim3(a->n_layers), dim3(256), 0, s, *a);
  return check_launch("svk_h2o_device_burst");
}
