// KV payload movement: scatter of new rows, slot-to-slot copies.  gfx950 only.
// HBM-bound byte work: 16 B per lane, one 1 KiB token row per wave-instruction.

#include "svk_common.hpp"

namespace svk {
namespace {

__global__ void __launch_bounds__(256)
store_kvcache_kernel(const SvkStoreKvcacheArgs a, int chunks_per_row) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.n_tokens * chunks_per_row;
  if (idx >= total) return;
  const int tok = (int)(idx / chunks_per_row);
  const int ch = (int)(idx % chunks_per_row);
  const int slot = a.slot_mapping[tok];
  if (slot == -1) return;   // store_kvcache.py:22 `if slot == -1: return`
  const uint4 kv = *reinterpret_cast<const uint4*>(a.key + (int64_t)tok * a.key_stride + ch * 8);
  const uint4 vv = *reinterpret_cast<const uint4*>(a.value + (int64_t)tok * a.value_stride + ch * 8);
  *reinterpret_cast<uint4*>(a.k_cache + (int64_t)slot * a.row_elems + ch * 8) = kv;
  *reinterpret_cast<uint4*>(a.v_cache + (int64_t)slot * a.row_elems + ch * 8) = vv;
}

// phase 0: workspace[kv][i] = cache[src[i]];  phase 1: cache[dst[i]] = workspace[kv][i]
__global__ void __launch_bounds__(256)
copy_slots_kernel(const SvkCopySlotsArgs a, int chunks_per_row, int phase) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.n * chunks_per_row;
  if (idx >= total) return;
  const int i = (int)(idx / chunks_per_row);
  const int ch = (int)(idx % chunks_per_row);
  uint16_t* wk = a.workspace + (int64_t)i * a.row_elems + ch * 8;
  uint16_t* wv = wk + (int64_t)a.n * a.row_elems;
  if (phase == 0) {
    const int64_t s = a.src_slots[i];
    *reinterpret_cast<uint4*>(wk) = *reinterpret_cast<const uint4*>(a.k_cache + s * a.row_elems + ch * 8);
    *reinterpret_cast<uint4*>(wv) = *reinterpret_cast<const uint4*>(a.v_cache + s * a.row_elems + ch * 8);
  } else {
    const int64_t d = a.dst_slots[i];
    *reinterpret_cast<uint4*>(a.k_cache + d * a.row_elems + ch * 8) = *reinterpret_cast<const uint4*>(wk);
    *reinterpret_cast<uint4*>(a.v_cache + d * a.row_elems + ch * 8) = *reinterpret_cast<const uint4*>(wv);
  }
}

// one workgroup per decode lane: the first violation found anywhere in the launch is recorded (atomicCAS on status[0])
__global__ void __launch_bounds__(256) check_slot_table_kernel(const SvkCheckSlotTableArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int row = a.req_indices[b], len = a.context_lens[b];
  auto report = [&](int kind, int pos, int slot) {
    if (atomicCAS(&a.status[0], 0, kind) == 0) {
      a.status[1] = b; a.status[2] = row; a.status[3] = pos; a.status[4] = slot; a.status[5] = len;
    }
  };
  if (row < 0 || row >= a.num_rows) {
    if (tid == 0) report(SVK_SLOT_CHECK_ROW, -1, -1);
    return;
  }
  if (len > a.width * (a.slot_page_size > 1 ? a.slot_page_size : 1)) {
    if (tid == 0) report(SVK_SLOT_CHECK_WIDTH, -1, -1);
    return;
  }
  const int n = a.slot_page_size > 1 ? (len + a.slot_page_size - 1) / a.slot_page_size : len;
  const int32_t* tab = a.slot_table + (int64_t)row * a.table_stride;
  for (int p = tid; p < n; p += 256) {
    const int s = tab[p];
    if (s < 0 || s >= a.slot_cap) { report(SVK_SLOT_CHECK_SLOT, p, s); break; }
  }
}

}  // namespace
}  // namespace svk

extern "C" int svk_check_slot_table(const SvkCheckSlotTableArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->slot_table != nullptr && a->req_indices != nullptr && a->context_lens != nullptr && a->status != nullptr,
              SVK_ERR_VALUE, "svk_check_slot_table: null args");
  SVK_REQUIRE(a->slot_cap > 0 && a->num_rows > 0 && a->width > 0, SVK_ERR_VALUE, "svk_check_slot_table: empty table / pool");
  if (a->batch <= 0) return SVK_OK;
  hipLaunchKernelGGL(check_slot_table_kernel, dim3(a->batch), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_check_slot_table");
}

extern "C" int svk_store_kvcache(const SvkStoreKvcacheArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_store_kvcache: null args");
  SVK_REQUIRE(a->row_elems > 0 && a->row_elems % 8 == 0, SVK_ERR_LAYOUT,
              "svk_store_kvcache: row_elems %d must be a positive multiple of 8", a->row_elems);
  SVK_REQUIRE(a->key_stride % 8 == 0 && a->value_stride % 8 == 0, SVK_ERR_LAYOUT,
              "svk_store_kvcache: key/value token strides must keep 16-byte alignment");
  if (a->n_tokens <= 0) return SVK_OK;
  const int cpr = a->row_elems / 8;
  const int64_t total = (int64_t)a->n_tokens * cpr;
  hipLaunchKernelGGL(store_kvcache_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a, cpr);
  return check_launch("svk_store_kvcache");
}

extern "C" int svk_copy_slots(const SvkCopySlotsArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_copy_slots: null args");
  SVK_REQUIRE(a->row_elems > 0 && a->row_elems % 8 == 0, SVK_ERR_LAYOUT,
              "svk_copy_slots: row_elems %d must be a positive multiple of 8", a->row_elems);
  if (a->n <= 0) return SVK_OK;
  const int cpr = a->row_elems / 8;
  const int64_t total = (int64_t)a->n * cpr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(copy_slots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *a, cpr, 0);
  hipLaunchKernelGGL(copy_slots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *a, cpr, 1);
  return check_launch("svk_copy_slots");
}
