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

}  // namespace
}  // namespace svk

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
