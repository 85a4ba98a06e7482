// DeltaKV sparse-layer attention view: gather the active slots into a contiguous K/V copy, rotating the
// raw (pre-RoPE) keys on the way (gfx950).  HBM-bound gather: one lane moves 16 B of each rotate-half
// partner (elements p..p+7 and p+D/2..), D/16 lanes per (token, head), so each K/V row is read and
// written as full 2*D-byte runs; k-norm statistics are reduced over those D/16 lanes with DPP-free shuffles.

#include "svk_common.hpp"

namespace svk {
namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
  f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16), f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16),
                    f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16), f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16));
}

__device__ __forceinline__ void load8(const void* base, int64_t off, int dtype, float (&f)[8]) {
  if (dtype == SVK_DTYPE_F32) {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else if (dtype == SVK_DTYPE_BF16) {
    unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off), f);
  } else {
    const _Float16* h = reinterpret_cast<const _Float16*>(base) + off;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
  }
}

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
  float k1[8], k2[8];
  uint4 v1 = make_uint4(0, 0, 0, 0), v2 = v1, rk1 = v1, rk2 = v1;
  int pos = 0;
  bool copy = false, skip = false;
  if (a.new_slots != nullptr && tl < tokens_per_block && n >= total && n - total < a.batch) {
    // the blocks behind the view: this step's raw rows into the cache (what store_kvcache did in a launch of its own)
    const int b = (int)(n - total);
    const int slot = a.new_slots[b];
    if (slot >= 0 && slot < a.num_slots) {
      const int64_t src = (int64_t)b * a.new_token_stride + (int64_t)h * a.new_head_stride + p;
      const int64_t dst = (int64_t)slot * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      *reinterpret_cast<uint4*>(a.k_cache + dst) = *reinterpret_cast<const uint4*>(a.new_k + src);
      *reinterpret_cast<uint4*>(a.k_cache + dst + HD2) = *reinterpret_cast<const uint4*>(a.new_k + src + HD2);
      *reinterpret_cast<uint4*>(a.v_cache + dst) = *reinterpret_cast<const uint4*>(a.new_v + src);
      *reinterpret_cast<uint4*>(a.v_cache + dst + HD2) = *reinterpret_cast<const uint4*>(a.new_v + src + HD2);
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
    const uint16_t* ks = a.k_cache;
    const uint16_t* vs = a.v_cache;
    int64_t base = (int64_t)safe * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
    if (a.new_slots != nullptr && valid && slot == a.new_slots[b]) {   // the row the store blocks are writing right now
      ks = a.new_k;
      vs = a.new_v;
      base = (int64_t)b * a.new_token_stride + (int64_t)h * a.new_head_stride + p;
    }
    if (!skip) {
      rk1 = *reinterpret_cast<const uint4*>(ks + base);
      rk2 = *reinterpret_cast<const uint4*>(ks + base + HD2);
      v1 = *reinterpret_cast<const uint4*>(vs + base);
      v2 = *reinterpret_cast<const uint4*>(vs + base + HD2);
    }
  }
  unpack8(rk1, k1);
  unpack8(rk2, k2);
  float n1[8], n2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { n1[e] = k1[e]; n2[e] = k2[e]; }
  if (a.k_norm_weight != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += k1[e] * k1[e] + k2[e] * k2[e];
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) ss += __shfl_xor(ss, off, 64);
    const float rstd = rsqrtf(ss / (float)D + a.k_norm_eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      n1[e] = k1[e] * rstd * a.k_norm_weight[p + e];
      n2[e] = k2[e] * rstd * a.k_norm_weight[p + HD2 + e];
    }
  }
  if (!live || skip) return;
  uint4 o1 = rk1, o2 = rk2;
  if (!copy) {
    float c[8], s[8], r1[8], r2[8];
    load8(a.cos_sin, (int64_t)pos * a.cos_stride + p, a.cos_dtype, c);
    load8(a.cos_sin, (int64_t)pos * a.cos_stride + p + HD2, a.cos_dtype, s);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r1[e] = n1[e] * c[e] - n2[e] * s[e];
      r2[e] = n2[e] * c[e] + n1[e] * s[e];
    }
    o1 = pack8(r1);
    o2 = pack8(r2);
  }
  const int64_t ob = n * a.out_slot_stride + (int64_t)h * a.out_head_stride + p;
  *reinterpret_cast<uint4*>(a.out_k + ob) = o1;
  *reinterpret_cast<uint4*>(a.out_k + ob + HD2) = o2;
  *reinterpret_cast<uint4*>(a.out_v + ob) = v1;
  *reinterpret_cast<uint4*>(a.out_v + ob + HD2) = v2;
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
  const bool store = a->new_slots != nullptr;
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
  if (a->head_dim == 128) hipLaunchKernelGGL(materialize_kernel<128>, dim3(grid), dim3(256), 0, s, *a);
  else hipLaunchKernelGGL(materialize_kernel<64>, dim3(grid), dim3(256), 0, s, *a);
  return check_launch("svk_deltakv_materialize_sparse_view");
}
