// DeltaKV sparse-layer attention view: gather the active slots into a contiguous K/V copy, rotating the
// raw (pre-RoPE) keys on the way (gfx950).  HBM-bound gather: one lane moves 16 B of each rotate-half
// partner (elements p..p+7 and p+D/2..), D/16 lanes per (token, head), so each K/V row is read and
// written as full 2*D-byte runs; k-norm statistics are reduced over those D/16 lanes with DPP-free shuffles.

#include "rope_row.hpp"

namespace svk {
namespace {

template <int D>
__global__ void __launch_bounds__(256) materialize_kernel(const SvkDeltakvMaterializeArgs a) {
  constexpr int HD2 = D / 2, LPH = HD2 / 8;            // lanes per (token, head)
  const int lanes_per_token = LPH * a.num_kv_heads;
  const int tokens_per_block = blockDim.x / lanes_per_token;
  const int tl = threadIdx.x / lanes_per_token;
  const int r = threadIdx.x % lanes_per_token;
  const int h = r / LPH, p = (r % LPH) * 8;
  const int64_t total = (int64_t)a.batch * a.width;
  const int64_t n = (int64_t)blockIdx.x * tokens_per_block + tl;
  const bool live = tl < tokens_per_block && n < total;
  // blockIdx.y: one of layer_count consecutive layers that share the slot table (caches, views and k-norm weights at
  // their layer strides)
  const int64_t ly = blockIdx.y;
  uint16_t* const k_cache = a.k_cache + ly * a.kv_layer_stride;
  uint16_t* const v_cache = a.v_cache + ly * a.kv_layer_stride;
  uint16_t* const out_k = a.out_k + ly * a.out_layer_stride;
  uint16_t* const out_v = a.out_v + ly * a.out_layer_stride;
  const float* const k_norm_weight = a.k_norm_weight == nullptr ? nullptr : a.k_norm_weight + ly * a.k_norm_layer_stride;
  uint4 v1 = make_uint4(0, 0, 0, 0), v2 = v1, rk1 = v1, rk2 = v1;
  int pos = 0;
  bool copy = false, skip = false;
  if (a.new_k != nullptr && tl < tokens_per_block && n >= total && n - total < a.batch) {
    // the blocks behind the view: this step's raw rows into the cache (what store_kvcache did in a launch of its own)
    const int b = (int)(n - total);
    const int slot = a.new_slots[b];
    if (slot >= 0 && slot < a.num_slots) {
      const int64_t src = (int64_t)b * a.new_token_stride + (int64_t)h * a.new_head_stride + p;
      const int64_t dst = (int64_t)slot * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      *reinterpret_cast<uint4*>(k_cache + dst) = *reinterpret_cast<const uint4*>(a.new_k + src);
      *reinterpret_cast<uint4*>(k_cache + dst + HD2) = *reinterpret_cast<const uint4*>(a.new_k + src + HD2);
      *reinterpret_cast<uint4*>(v_cache + dst) = *reinterpret_cast<const uint4*>(a.new_v + src);
      *reinterpret_cast<uint4*>(v_cache + dst + HD2) = *reinterpret_cast<const uint4*>(a.new_v + src + HD2);
    }
  }
  if (live) {
    const int b = (int)(n / a.width), w = (int)(n % a.width);
    const int slot = a.active_slots[(int64_t)b * a.active_stride + w];
    const bool valid = slot >= 0 && slot < a.num_slots;
    const int safe = min(max(slot, 0), a.num_slots - 1);
    pos = max(a.slot_to_pos[safe], 0);
    copy = valid && a.postrope_mask != nullptr && a.postrope_mask[safe] != 0;
    if (a.temp_slots != nullptr) {
      const int j = w - a.temp_offset;
      copy = valid && j >= 0 && j < a.temp_count && a.temp_slots[(int64_t)b * a.temp_stride + j] == slot;
      skip = copy && a.skip_temp != 0;          // the reconstruction wrote this row of the view itself
    }
    const uint16_t* ks = k_cache;
    const uint16_t* vs = v_cache;
    int64_t base = (int64_t)safe * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
    if (a.new_slots != nullptr && valid && slot == a.new_slots[b]) {   // the row the store blocks are writing right now
      if (a.skip_new != 0) {
        skip = true;                            // the attention launch of the layer writes this row (rotated store)
      } else {
        ks = a.new_k;
        vs = a.new_v;
        base = (int64_t)b * a.new_token_stride + (int64_t)h * a.new_head_stride + p;
      }
    }
    if (!skip) {
      rk1 = *reinterpret_cast<const uint4*>(ks + base);
      rk2 = *reinterpret_cast<const uint4*>(ks + base + HD2);
      v1 = *reinterpret_cast<const uint4*>(vs + base);
      v2 = *reinterpret_cast<const uint4*>(vs + base + HD2);
    }
  }
  float n1[8], n2[8];
  rope_row_norm<D>(rk1, rk2, k_norm_weight, a.k_norm_eps, p, n1, n2);
  if (!live || skip) return;
  uint4 o1 = rk1, o2 = rk2;
  if (!copy) rope_row_rotate<D>(n1, n2, a.cos_sin, (int64_t)pos * a.cos_stride, a.cos_dtype, p, o1, o2);
  const int64_t ob = n * a.out_slot_stride + (int64_t)h * a.out_head_stride + p;
  *reinterpret_cast<uint4*>(out_k + ob) = o1;
  *reinterpret_cast<uint4*>(out_k + ob + HD2) = o2;
  *reinterpret_cast<uint4*>(out_v + ob) = v1;
  *reinterpret_cast<uint4*>(out_v + ob + HD2) = v2;
}

}  // namespace
}  // namespace svk

extern "C" int svk_deltakv_materialize_sparse_view(const SvkDeltakvMaterializeArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_deltakv_materialize_sparse_view: null args");
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_VALUE,
              "svk_deltakv_materialize_sparse_view: head_dim must be 64 or 128, got %d", a->head_dim);
  SVK_REQUIRE(a->num_kv_heads >= 1 && a->num_kv_heads <= 8, SVK_ERR_LAYOUT,
              "svk_deltakv_materialize_sparse_view: 1..8 KV heads per rank, got %d", a->num_kv_heads);
  SVK_REQUIRE(a->kv_slot_stride % 8 == 0 && a->kv_head_stride % 8 == 0 && a->out_slot_stride % 8 == 0 && a->out_head_stride % 8 == 0,
              SVK_ERR_LAYOUT, "svk_deltakv_materialize_sparse_view: K/V strides must keep 16-byte alignment");
  const int64_t total = (int64_t)a->batch * a->width;
  const int layers = a->layer_count > 1 ? a->layer_count : 1;
  SVK_REQUIRE(a->skip_new == 0 || (a->new_slots != nullptr && a->new_k == nullptr && a->new_v == nullptr), SVK_ERR_VALUE,
              "svk_deltakv_materialize_sparse_view: skip_new needs new_slots and no new_k / new_v (the attention launch stores the row)");
  SVK_REQUIRE(layers == 1 || (a->new_k == nullptr && a->kv_layer_stride % 8 == 0 && a->out_layer_stride % 8 == 0), SVK_ERR_LAYOUT,
              "svk_deltakv_materialize_sparse_view: a multi-layer launch carries no store and needs 16-byte aligned layer strides");
  const bool store = a->new_slots != nullptr && a->skip_new == 0;
  if (store) {
    SVK_REQUIRE(a->new_k != nullptr && a->new_v != nullptr, SVK_ERR_VALUE,
                "svk_deltakv_materialize_sparse_view: new_slots needs new_k and new_v");
    SVK_REQUIRE(a->new_token_stride % 8 == 0 && a->new_head_stride % 8 == 0, SVK_ERR_LAYOUT,
                "svk_deltakv_materialize_sparse_view: new_k/new_v strides must keep 16-byte alignment");
  }
  const int64_t entries = total + (store ? a->batch : 0);          // the store rows ride behind the view entries
  if (entries <= 0) return SVK_OK;
  const int lpt = (a->head_dim / 16) * a->num_kv_heads;
  const int tpb = 256 / lpt;
  const unsigned grid = (unsigned)((entries + tpb - 1) / tpb);
  hipStream_t s = static_cast<hipStream_t>(stream);
  SvkDeltakvMaterializeArgs k = *a;
  if (layers == 1) k.kv_layer_stride = k.out_layer_stride = k.k_norm_layer_stride = 0;
  if (a->head_dim == 128) hipLaunchKernelGGL(materialize_kernel<128>, dim3(grid, layers), dim3(256), 0, s, k);
  else hipLaunchKernelGGL(materialize_kernel<64>, dim3(grid, layers), dim3(256), 0, s, k);
  return check_launch("svk_deltakv_materialize_sparse_view");
}
