// DeltaKV compression side: residual quantise/pack, KIVI block store, father top-k and
// mean of father rows.  All four are small HBM-bound passes that run once per `recent` decoded tokens per sequence;
// the GEMMs around them (L2 ranking scores, compress_down) are plain library calls on the host side.

#include "svk_common.hpp"

namespace svk {
namespace {

template <int DT> struct Lowp;
template <> struct Lowp<SVK_DTYPE_F32> {
  using T = float;
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return reinterpret_cast<const float*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Lowp<SVK_DTYPE_BF16> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return __builtin_bit_cast(float, (uint32_t) reinterpret_cast<const uint16_t*>(p)[i] << 16);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)f32_to_bf16_bits(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf16_round(v); }
};
template <> struct Lowp<SVK_DTYPE_F16> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return (float) reinterpret_cast<const _Float16*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { reinterpret_cast<_Float16*>(p)[i] = (_Float16)v; }
  static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

__device__ __forceinline__ float sub_rn(float x, float y) {
#pragma clang fp contract(off)
  return x - y;
}
__device__ __forceinline__ float div_rn(float x, float y) { return __fdiv_rn(x, y); }

// one thread per (row, group): reference kernel quant.py:29-76
template <int DT>
__global__ void __launch_bounds__(256) quant_pack_kernel(const SvkQuantPackArgs a) {
  using L = Lowp<DT>;
  const int groups = a.features / a.group_size;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.rows * groups) return;
  const int r = (int)(gid / groups), g = (int)(gid % groups);
  const int64_t src = (int64_t)r * a.data_stride + (int64_t)g * a.group_size;
  float mx = -INFINITY, mn = INFINITY;
  for (int i = 0; i < a.group_size; ++i) {
    const float x = L::load(a.data, src + i);
    mx = fmaxf(mx, x);
    mn = fminf(mn, x);
  }
  const float qmax = (float)((1 << a.bits) - 1);
  const float scale = L::rnd(div_rn(sub_rn(mx, mn), qmax));       // fp32 statistics, one rounding at the store
  const float denom = add_rn(scale, 1.0e-6f);
  const int64_t dr = a.dst_rows ? a.dst_rows[r] : r;
  L::store(a.scale, dr * a.scale_stride + g, scale);
  L::store(a.mn, dr * a.scale_stride + g, mn);
  const int fpi = 32 / a.bits;
  int32_t* code = a.code + dr * a.code_stride + (int64_t)g * (a.group_size / fpi);
  for (int w0 = 0; w0 < a.group_size; w0 += fpi) {
    uint32_t word = 0;
    for (int j = 0; j < fpi; ++j) {
      const float x = L::load(a.data, src + w0 + j);
      const float nrm = div_rn(L::rnd(sub_rn(x, mn)), denom);
      const float q = rintf(fminf(fmaxf(nrm, 0.f), qmax));         // round half to even
      word |= (uint32_t)(int)q << (j * a.bits);
    }
    code[w0 / fpi] = (int32_t)word;
  }
}

// q of one value with torch's bf16 op-by-op arithmetic (quant.py:283-287)
__device__ __forceinline__ uint32_t torch_q4(float x, float mn, float denom_b) {
  const float nrm = bf16_round(div_rn(bf16_round(sub_rn(x, mn)), denom_b));
  return (uint32_t)(int)rintf(fminf(fmaxf(nrm, 0.f), 15.f));
}

// workgroup = (block, kv head); D threads: thread d quantises channel d of K over the block's G tokens, then
// (token t, group gi) pairs quantise V
template <bool KF32>
__global__ void __launch_bounds__(128) kivi_store_kernel(const SvkKiviStoreArgs a) {
  extern __shared__ int s_slots[];                 // [G]
  const int blk = blockIdx.x, h = blockIdx.y, D = a.head_dim, G = a.group_size, H = a.num_kv_heads;
  for (int i = threadIdx.x; i < G; i += blockDim.x) s_slots[i] = a.raw_slots[(int64_t)blk * G + i];
  __syncthreads();
  const int64_t dst = a.block_slots[blk];
  const int64_t hb = dst * H + h;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float mx = -INFINITY, mn = INFINITY;
    for (int t = 0; t < G; ++t) {
      const float x = __builtin_bit_cast(float, (uint32_t)a.k_cache[(int64_t)s_slots[t] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + d] << 16);
      mx = fmaxf(mx, x);
      mn = fminf(mn, x);
    }
    const float scale = bf16_round(div_rn(bf16_round(sub_rn(mx, mn)), 15.f));
    const float denom = bf16_round(add_rn(scale, 1.0e-6f));
    if (KF32) {
      reinterpret_cast<float*>(a.key_scales)[hb * D + d] = scale;
      reinterpret_cast<float*>(a.key_mins)[hb * D + d] = mn;
    } else {
      reinterpret_cast<uint16_t*>(a.key_scales)[hb * D + d] = (uint16_t)f32_to_bf16_bits(scale);
      reinterpret_cast<uint16_t*>(a.key_mins)[hb * D + d] = (uint16_t)f32_to_bf16_bits(mn);
    }
    for (int w0 = 0; w0 < G; w0 += 8) {
      uint32_t word = 0;
      for (int j = 0; j < 8; ++j) {
        const float x = __builtin_bit_cast(float, (uint32_t)a.k_cache[(int64_t)s_slots[w0 + j] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + d] << 16);
        word |= torch_q4(x, mn, denom) << (j * 4);
      }
      a.key_packed[(hb * D + d) * (G / 8) + w0 / 8] = (int32_t)word;
    }
  }
  const int ng = D / G;                             // V groups per token
  for (int idx = threadIdx.x; idx < G * ng; idx += blockDim.x) {
    const int t = idx / ng, gi = idx % ng;
    const uint16_t* vp = a.v_cache + (int64_t)s_slots[t] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + gi * G;
    float mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < G; ++i) {
      const float x = __builtin_bit_cast(float, (uint32_t)vp[i] << 16);
      mx = fmaxf(mx, x);
      mn = fminf(mn, x);
    }
    const float scale = bf16_round(div_rn(bf16_round(sub_rn(mx, mn)), 15.f));
    const float denom = bf16_round(add_rn(scale, 1.0e-6f));
    const int64_t tb = hb * G + t;
    a.value_scales[tb * ng + gi] = (uint16_t)f32_to_bf16_bits(scale);
    a.value_mins[tb * ng + gi] = (uint16_t)f32_to_bf16_bits(mn);
    for (int w0 = 0; w0 < G; w0 += 8) {
      uint32_t word = 0;
      for (int j = 0; j < 8; ++j)
        word |= torch_q4(__builtin_bit_cast(float, (uint32_t)vp[w0 + j] << 16), mn, denom) << (j * 4);
      a.value_packed[tb * (D / 8) + (gi * G + w0) / 8] = (int32_t)word;
    }
  }
}

// one wave per row: lanes keep a sorted local top-k over their strided columns, then k rounds of wave arg-max
constexpr int kMaxFathers = 8;

__global__ void __launch_bounds__(256) cluster_topk_kernel(const SvkClusterTopkArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= a.rows) return;
  const int row_abs = a.row_offset + r;
  float bs[kMaxFathers];
  int bi[kMaxFathers];
#pragma unroll
  for (int j = 0; j < kMaxFathers; ++j) { bs[j] = -INFINITY; bi[j] = 0x7fffffff; }
  for (int c = lane; c < a.m; c += 64) {
    float s;
    if (a.score_dtype == SVK_DTYPE_F32) s = reinterpret_cast<const float*>(a.scores)[(int64_t)r * a.score_stride + c];
    else if (a.score_dtype == SVK_DTYPE_BF16)
      s = __builtin_bit_cast(float, (uint32_t) reinterpret_cast<const uint16_t*>(a.scores)[(int64_t)r * a.score_stride + c] << 16);
    else s = (float) reinterpret_cast<const _Float16*>(a.scores)[(int64_t)r * a.score_stride + c];
    if (c >= a.m0 && a.new_center_rel[c - a.m0] > row_abs) s = -INFINITY;
    // insert (s, c) into the local list ordered by (score desc, index asc); columns arrive in ascending order
    float cs = s;
    int ci = c;
#pragma unroll
    for (int j = 0; j < kMaxFathers; ++j) {
      if (j < a.k && (cs > bs[j] || bi[j] == 0x7fffffff)) {
        const float ts = bs[j]; const int ti = bi[j];
        bs[j] = cs; bi[j] = ci;
        cs = ts; ci = ti;
      }
    }
  }
  int head = 0;
  for (int j = 0; j < a.k; ++j) {
    float s = -INFINITY;
    int i = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < kMaxFathers; ++q)
      if (q == head) { s = bs[q]; i = bi[q]; }
    float ws = s;
    int wi = i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(ws, off, 64);
      const int oi = __shfl_xor(wi, off, 64);
      if (os > ws || (os == ws && oi < wi)) { ws = os; wi = oi; }
    }
    if (lane == 0) a.topk[(int64_t)r * a.topk_stride + j] = wi;
    if (i == wi && s == ws) ++head;
  }
}

// one workgroup per row; thread = 8 consecutive elements of concat(K row, V row)
__global__ void __launch_bounds__(256) gather_mean_kernel(const SvkGatherMeanArgs a) {
  const int r = blockIdx.x;
  const int HD = a.num_kv_heads * a.head_dim;
  __shared__ int s_f[kMaxFathers];
  if (threadIdx.x < a.k) {
    const int f = a.center_slots[a.topk[(int64_t)r * a.topk_stride + threadIdx.x]];
    s_f[threadIdx.x] = f;
  }
  __syncthreads();
  if (a.father_slots != nullptr && threadIdx.x < a.k_out)
    a.father_slots[(int64_t)r * a.father_stride + threadIdx.x] = threadIdx.x < a.k ? s_f[threadIdx.x] : s_f[0];
  const float kf = (float)a.k;
  for (int e0 = threadIdx.x * 8; e0 < 2 * HD; e0 += blockDim.x * 8) {
    const bool is_v = e0 >= HD;
    const int e = is_v ? e0 - HD : e0;
    const int h = e / a.head_dim, d = e % a.head_dim;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.k; ++j) {
      const uint16_t* p = (is_v ? a.v_cache : a.k_cache) + (int64_t)s_f[j] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + d;
      const uint4 v = *reinterpret_cast<const uint4*>(p);
      acc[0] += bf16_lo(v.x); acc[1] += bf16_hi(v.x); acc[2] += bf16_lo(v.y); acc[3] += bf16_hi(v.y);
      acc[4] += bf16_lo(v.z); acc[5] += bf16_hi(v.z); acc[6] += bf16_lo(v.w); acc[7] += bf16_hi(v.w);
    }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = f32_to_bf16_bits(div_rn(acc[2 * q], kf)) | (f32_to_bf16_bits(div_rn(acc[2 * q + 1], kf)) << 16);
    *reinterpret_cast<uint4*>(a.base + (int64_t)r * a.base_stride + e0) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace
}  // namespace svk

extern "C" int svk_quantize_pack_grouped(const SvkQuantPackArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_quantize_pack_grouped: null args");
  SVK_REQUIRE(a->bits == 2 || a->bits == 4 || a->bits == 8, SVK_ERR_VALUE, "Packed quantization supports bits=(2, 4, 8), got %d.", a->bits);
  SVK_REQUIRE(a->group_size > 0 && a->group_size % (32 / a->bits) == 0, SVK_ERR_VALUE,
              "2D int%d quantization requires group_size to be a positive multiple of %d, got %d.", a->bits, 32 / a->bits, a->group_size);
  SVK_REQUIRE(a->features % a->group_size == 0, SVK_ERR_VALUE,
              "2D int4 quantization requires D divisible by group_size, got D=%d, group=%d.", a->features, a->group_size);
  if (a->rows <= 0) return SVK_OK;
  const int64_t total = (int64_t)a->rows * (a->features / a->group_size);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->data_dtype == SVK_DTYPE_F32) hipLaunchKernelGGL(quant_pack_kernel<SVK_DTYPE_F32>, grid, block, 0, s, *a);
  else if (a->data_dtype == SVK_DTYPE_BF16) hipLaunchKernelGGL(quant_pack_kernel<SVK_DTYPE_BF16>, grid, block, 0, s, *a);
  else if (a->data_dtype == SVK_DTYPE_F16) hipLaunchKernelGGL(quant_pack_kernel<SVK_DTYPE_F16>, grid, block, 0, s, *a);
  else { set_error("svk_quantize_pack_grouped: unsupported dtype %d", a->data_dtype); return SVK_ERR_VALUE; }
  return check_launch("svk_quantize_pack_grouped");
}

extern "C" int svk_kivi_store_blocks(const SvkKiviStoreArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_kivi_store_blocks: null args");
  SVK_REQUIRE(a->group_size > 0 && a->group_size % 8 == 0 && a->head_dim % a->group_size == 0, SVK_ERR_VALUE,
              "Full-layer KIVI int4 packing requires group_size divisible by 8 and head_dim divisible by group_size; got %d/%d.",
              a->group_size, a->head_dim);
  SVK_REQUIRE(a->key_param_dtype == SVK_DTYPE_F32 || a->key_param_dtype == SVK_DTYPE_BF16, SVK_ERR_VALUE,
              "svk_kivi_store_blocks: key scale/min dtype must be f32 or bf16, got %d", a->key_param_dtype);
  if (a->blocks <= 0) return SVK_OK;
  const dim3 grid(a->blocks, a->num_kv_heads), block(128);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->key_param_dtype == SVK_DTYPE_F32) hipLaunchKernelGGL(kivi_store_kernel<true>, grid, block, sizeof(int) * a->group_size, s, *a);
  else hipLaunchKernelGGL(kivi_store_kernel<false>, grid, block, sizeof(int) * a->group_size, s, *a);
  return check_launch("svk_kivi_store_blocks");
}

extern "C" int svk_cluster_topk(const SvkClusterTopkArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_cluster_topk: null args");
  SVK_REQUIRE(a->k >= 1 && a->k <= kMaxFathers && a->k <= a->m, SVK_ERR_VALUE, "svk_cluster_topk: k %d out of range (1..%d, m=%d)", a->k, kMaxFathers, a->m);
  SVK_REQUIRE(a->m0 >= 0 && a->m0 <= a->m, SVK_ERR_VALUE, "svk_cluster_topk: m0 %d out of range (m=%d)", a->m0, a->m);
  if (a->rows <= 0) return SVK_OK;
  hipLaunchKernelGGL(cluster_topk_kernel, dim3((a->rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_cluster_topk");
}

extern "C" int svk_gather_mean_fathers(const SvkGatherMeanArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_gather_mean_fathers: null args");
  SVK_REQUIRE(a->k >= 1 && a->k <= kMaxFathers && a->k_out <= kMaxFathers, SVK_ERR_VALUE, "svk_gather_mean_fathers: k %d / k_out %d out of range", a->k, a->k_out);
  SVK_REQUIRE(a->head_dim % 8 == 0 && a->kv_slot_stride % 8 == 0 && a->kv_head_stride % 8 == 0 && a->base_stride % 8 == 0, SVK_ERR_LAYOUT,
              "svk_gather_mean_fathers: rows must keep 16-byte alignment");
  if (a->rows <= 0) return SVK_OK;
  hipLaunchKernelGGL(gather_mean_kernel, dim3(a->rows), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_gather_mean_fathers");
}
