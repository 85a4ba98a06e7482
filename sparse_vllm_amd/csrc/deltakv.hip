// DeltaKV decode-side kernels for gfx950: static decode plan, residual (de)quantisation, father-mean
// reconstruct with RoPE write-back, observation-layer token scores and sorted top-k.
// All HBM/L2-bound integer / byte / elementwise work; no matrix cores.

#include "svk_common.hpp"
#include "svk_select.hpp"

#include <stdlib.h>
#include <string.h>

namespace svk {
namespace {

__device__ __forceinline__ float load_scalar(const void* p, int64_t i, int dtype) {
  if (dtype == SVK_DTYPE_F32) return reinterpret_cast<const float*>(p)[i];
  const uint16_t h = reinterpret_cast<const uint16_t*>(p)[i];
  if (dtype == SVK_DTYPE_BF16) return __builtin_bit_cast(float, (uint32_t)h << 16);
  return (float)__builtin_bit_cast(_Float16, h);
}

// ------------------------------------------------------------------------------------
// static decode plan: grid (ceil(S / 256), batch) - every output column is independent and costs two dependent gathers
// into the 262 k-entry slot maps, so a row is spread over workgroups instead of looped by one (19 -> 6 us)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) deltakv_plan_kernel(const SvkDeltakvPlanArgs a) {
  const int b = blockIdx.y;
  const int K = a.k_max, SINK = a.sink;
  const int S = SINK + K + a.max_buffer;
  const int max_pos = a.max_positions - 1;
  const int row = a.req_indices[b];
  const int ctx = a.context_lens[b];
  const int clen = a.compressed_lens[b];
  const int top_len = min(max(clen, 0), K);
  const int32_t* raw = a.raw_slots_map + (int64_t)row * a.raw_stride;
  const int32_t* lat = a.latent_slots_map + (int64_t)row * a.latent_stride;
  const int safe = SINK > 0 ? max(raw[0], 0) : 0;
  const int buf_start = SINK + clen;
  const int buf_len = min(max(ctx - buf_start, 0), a.max_buffer);
  const int start_out = SINK + top_len;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < S; c += gridDim.x * blockDim.x) {
    int o_slot = safe, o_pos = 0;
    if (c < SINK) {
      const int sp = min(c, max_pos);
      o_slot = raw[sp];
      o_pos = sp;
    }
    if (c >= SINK && c < SINK + K) {
      const int j = c - SINK;
      const bool in_top = j < top_len;
      int rel = -1;
      if (true) rel = a.active_compressed[(int64_t)b * a.active_stride + j];
      const int top_pos = rel + SINK;
      const bool valid = in_top && rel >= 0 && rel < clen && top_pos < ctx;
      const int sp = min(max(top_pos, 0), max_pos);
      const int r = in_top ? raw[sp] : 0;
      const int l = in_top ? lat[sp] : -1;
      const int t = in_top ? a.temp_slots[(int64_t)b * a.temp_stride + j] : 0;
      const bool need = valid && l >= 0;
      if (in_top) {
        o_slot = need ? t : (valid ? max(r, 0) : safe);
        o_pos = valid ? top_pos : 0;
      }
      a.recon_pos_out[b * K + j] = need ? top_pos : -1;
      a.recon_latent_out[b * K + j] = need ? l : -1;
      a.recon_out_slot_out[b * K + j] = need ? t : -1;
    }
    if (c >= start_out) {           // recent buffer, compacted right after the visible top slots
      const int j = c - start_out;
      const int pos = min(max(buf_start + j, 0), max_pos);
      const bool ok = j < buf_len;
      o_slot = ok ? max(raw[pos], 0) : safe;
      o_pos = ok ? pos : 0;
    }
    a.active_slots_out[(int64_t)b * a.out_stride + c] = o_slot;
    a.active_pos_out[(int64_t)b * a.pos_stride + c] = o_pos;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.new_context_lens_out[b] = SINK + top_len + buf_len;
}

// ------------------------------------------------------------------------------------
// grouped dequantisation
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) dequant_grouped_kernel(const SvkDequantGroupedArgs a) {
  const int r = blockIdx.y;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.features) return;
  const int fpi = 32 / a.bits;
  const int64_t src = a.row_index ? max(a.row_index[r], 0) : r;
  const uint32_t word = (uint32_t)a.packed[src * a.packed_stride + f / fpi];
  const float q = (float)((word >> ((f % fpi) * a.bits)) & ((1u << a.bits) - 1u));
  const int g = f / a.group_size;
  // two roundings (q*scale, then +min) like the un-fused reference arithmetic the fixtures pin
  const float v = add_rn(mul_rn(q, load_scalar(a.scale, src * a.scale_stride + g, a.scale_dtype)),
                            load_scalar(a.mn, src * a.scale_stride + g, a.scale_dtype));
  if (a.out_dtype == SVK_DTYPE_F32) reinterpret_cast<float*>(a.out)[(int64_t)r * a.out_stride + f] = v;
  else if (a.out_dtype == SVK_DTYPE_BF16) reinterpret_cast<uint16_t*>(a.out)[(int64_t)r * a.out_stride + f] = (uint16_t)f32_to_bf16_bits(v);
  else reinterpret_cast<_Float16*>(a.out)[(int64_t)r * a.out_stride + f] = (_Float16)v;
}

// ------------------------------------------------------------------------------------
// reconstruct + RoPE write-back.  One thread = (entry, kv head, pair p) -> elements p and p + D/2
// (the rotate-half partner); the D/2 threads of a head are consecutive lanes, so every father row
// segment is read as two fully coalesced D-byte runs.
// ------------------------------------------------------------------------------------

__device__ __forceinline__ float delta_at(const SvkDeltakvReconstructArgs& a, int n, int latent, int feat) {
  if (a.delta_bits == 0) return load_scalar(a.delta, (int64_t)n * a.delta_stride + feat, a.delta_dtype);
  const int fpi = 32 / a.delta_bits;
  const uint32_t word = (uint32_t)reinterpret_cast<const int32_t*>(a.delta)[(int64_t)latent * a.delta_stride + feat / fpi];
  const float q = (float)((word >> ((feat % fpi) * a.delta_bits)) & ((1u << a.delta_bits) - 1u));
  const int g = feat / a.group_size;
  return q * load_scalar(a.scale, (int64_t)latent * a.scale_stride + g, a.scale_dtype) +
         load_scalar(a.mn, (int64_t)latent * a.scale_stride + g, a.scale_dtype);
}

__global__ void __launch_bounds__(256) deltakv_reconstruct_kernel(const SvkDeltakvReconstructArgs a) {
  extern __shared__ float red[];       // [blockDim.x] sum of squares for the k-norm
  const int D = a.head_dim, HD2 = D / 2, H = a.num_kv_heads;
  const int per_entry = H * HD2;
  const int entries_per_block = blockDim.x / per_entry;
  const int e = threadIdx.x / per_entry;
  const int n = blockIdx.x * entries_per_block + e;
  const int t = threadIdx.x % per_entry;
  const int h = t / HD2, p = t % HD2;
  bool active = e < entries_per_block && n < a.n;
  int latent = -1, out_slot = 0, out_pos = 0;
  if (active) {
    out_slot = a.out_slots[n];
    out_pos = a.out_pos[n];
    if (a.delta_bits == 0) active = out_slot >= 0 && out_pos >= 0;
    else { latent = a.latent_slots[n]; active = latent >= 0; }
  }
  float k1 = 0.f, k2 = 0.f, v1 = 0.f, v2 = 0.f;
  if (active) {
    float ak1 = 0.f, ak2 = 0.f, av1 = 0.f, av2 = 0.f;
    const int32_t* fathers = a.father_table ? a.father_table + (int64_t)max(a.father_index[n], 0) * a.father_table_stride
                                            : a.father_slots + (int64_t)n * a.father_stride;
    for (int kk = 0; kk < a.k_fathers; ++kk) {
      const int fs = a.father_table ? max(fathers[kk], 0) : fathers[kk];
      const int64_t base = (int64_t)fs * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      const float y1 = __builtin_bit_cast(float, (uint32_t)a.k_cache[base] << 16);
      const float y2 = __builtin_bit_cast(float, (uint32_t)a.k_cache[base + HD2] << 16);
      if (a.raw_k_cache) {
        ak1 += y1; ak2 += y2;
      } else {
        const int fp = a.slot_to_pos[fs];
        const float c = load_scalar(a.cos_sin, (int64_t)fp * a.cos_stride + p, a.cos_dtype);
        const float s = load_scalar(a.cos_sin, (int64_t)fp * a.cos_stride + p + HD2, a.cos_dtype);
        ak1 += y1 * c + y2 * s;
        ak2 += y2 * c - y1 * s;
      }
      av1 += __builtin_bit_cast(float, (uint32_t)a.v_cache[base] << 16);
      av2 += __builtin_bit_cast(float, (uint32_t)a.v_cache[base + HD2] << 16);
    }
    const float inv = 1.0f / (float)a.k_fathers;
    const int Dtot = H * D;
    const int fk1 = h * D + p;
    k1 = delta_at(a, n, latent, fk1) + ak1 * inv;
    k2 = delta_at(a, n, latent, fk1 + HD2) + ak2 * inv;
    v1 = delta_at(a, n, latent, Dtot + fk1) + av1 * inv;
    v2 = delta_at(a, n, latent, Dtot + fk1 + HD2) + av2 * inv;
  }
  if (a.k_norm_weight != nullptr && !a.store_raw_k) {
    // RMS norm over the head's D elements = the HD2 consecutive threads of this head
    red[threadIdx.x] = active ? k1 * k1 + k2 * k2 : 0.f;
    __syncthreads();
    if (active) {
      const int h0 = threadIdx.x - p;
      float ss = 0.f;
      for (int i = 0; i < HD2; ++i) ss += red[h0 + i];
      const float rstd = rsqrtf(ss / (float)D + a.k_norm_eps);
      k1 = k1 * rstd * a.k_norm_weight[p];
      k2 = k2 * rstd * a.k_norm_weight[p + HD2];
    }
  }
  if (!active) return;
  float o1 = k1, o2 = k2;
  if (!a.store_raw_k) {
    const float c = load_scalar(a.cos_sin, (int64_t)out_pos * a.cos_stride + p, a.cos_dtype);
    const float s = load_scalar(a.cos_sin, (int64_t)out_pos * a.cos_stride + p + HD2, a.cos_dtype);
    o1 = k1 * c - k2 * s;
    o2 = k2 * c + k1 * s;
  }
  const int64_t ob = (int64_t)out_slot * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
  a.k_cache[ob] = (uint16_t)f32_to_bf16_bits(o1);
  a.k_cache[ob + HD2] = (uint16_t)f32_to_bf16_bits(o2);
  a.v_cache[ob] = (uint16_t)f32_to_bf16_bits(v1);
  a.v_cache[ob + HD2] = (uint16_t)f32_to_bf16_bits(v2);
}

// 16-byte-lane form of the reconstruct kernel for the decode path (dense bf16 delta = compress_up output, bf16 cache,
// fp32 cos|sin): D/16 lanes per (entry, head), each owning elements p..p+7 of both rotate-half partners, so every father
// row is fetched as 16-byte pieces (4 loads per father per lane instead of 32 two-byte ones).  Element arithmetic and
// order are those of the scalar kernel above; only the k-norm sum of squares is reduced in a different order.
template <int D>
__global__ void __launch_bounds__(256) deltakv_reconstruct_vec_kernel(const SvkDeltakvReconstructArgs a_in,
                                                                      const SvkDeltakvReconstructBatch lb) {
  SvkDeltakvReconstructArgs a = a_in;
  if (gridDim.y > 1) {                  // layer y of a batched launch (dense bf16 delta): per-layer tensors advance
    const int64_t z = blockIdx.y;
    a.delta = reinterpret_cast<const uint16_t*>(a.delta) + z * lb.delta_stride_batch;
    if (a.father_table != nullptr) a.father_table += z * lb.father_table_stride_batch;
    a.k_cache += z * lb.kv_cache_stride_batch;
    a.v_cache += z * lb.kv_cache_stride_batch;
    if (a.k_norm_weight != nullptr) a.k_norm_weight += z * lb.k_norm_stride_batch;
    if (a.out_k_cache != nullptr) { a.out_k_cache += z * lb.out_cache_stride_batch; a.out_v_cache += z * lb.out_cache_stride_batch; }
  }
  constexpr int HD2 = D / 2, LPH = HD2 / 8;
  const int H = a.num_kv_heads;
  const int lanes_per_entry = LPH * H;
  const int entries_per_block = blockDim.x / lanes_per_entry;
  const int e = threadIdx.x / lanes_per_entry;
  const int n = blockIdx.x * entries_per_block + e;
  const int t = threadIdx.x % lanes_per_entry;
  const int h = t / LPH, p = (t % LPH) * 8;
  bool active = e < entries_per_block && n < a.n;
  int out_slot = 0, out_pos = 0;
  if (active) {
    out_slot = a.out_slots[n];
    out_pos = a.out_pos[n];
    active = out_slot >= 0 && out_pos >= 0;
  }
  auto up8 = [](const uint4& v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
  };
  auto ld8f = [](const float* q, float (&f)[8]) {
    const float4 x = *reinterpret_cast<const float4*>(q), y = *reinterpret_cast<const float4*>(q + 4);
    f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w; f[4] = y.x; f[5] = y.y; f[6] = y.z; f[7] = y.w;
  };
  float k1[8], k2[8], v1[8], v2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k1[i] = k2[i] = v1[i] = v2[i] = 0.f;
  if (active) {
    float ak1[8], ak2[8], av1[8], av2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ak1[i] = ak2[i] = av1[i] = av2[i] = 0.f;
    const int32_t* fathers = a.father_table ? a.father_table + (int64_t)max(a.father_index[n], 0) * a.father_table_stride
                                            : a.father_slots + (int64_t)n * a.father_stride;
    for (int kk = 0; kk < a.k_fathers; ++kk) {
      const int fs = a.father_table ? max(fathers[kk], 0) : fathers[kk];
      const int64_t base = (int64_t)fs * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      float y1[8], y2[8], w1[8], w2[8];
      up8(*reinterpret_cast<const uint4*>(a.k_cache + base), y1);
      up8(*reinterpret_cast<const uint4*>(a.k_cache + base + HD2), y2);
      up8(*reinterpret_cast<const uint4*>(a.v_cache + base), w1);
      up8(*reinterpret_cast<const uint4*>(a.v_cache + base + HD2), w2);
      if (a.raw_k_cache) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { ak1[i] += y1[i]; ak2[i] += y2[i]; }
      } else {
        const int fp = a.slot_to_pos[fs];
        float c[8], sn[8];
        ld8f(reinterpret_cast<const float*>(a.cos_sin) + (int64_t)fp * a.cos_stride + p, c);
        ld8f(reinterpret_cast<const float*>(a.cos_sin) + (int64_t)fp * a.cos_stride + p + HD2, sn);
#pragma unroll
        for (int i = 0; i < 8; ++i) { ak1[i] += y1[i] * c[i] + y2[i] * sn[i]; ak2[i] += y2[i] * c[i] - y1[i] * sn[i]; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { av1[i] += w1[i]; av2[i] += w2[i]; }
    }
    const float inv = 1.0f / (float)a.k_fathers;
    const int Dtot = H * D;
    const uint16_t* dl = reinterpret_cast<const uint16_t*>(a.delta) + (int64_t)n * a.delta_stride + h * D + p;
    float d1[8], d2[8], e1[8], e2[8];
    up8(*reinterpret_cast<const uint4*>(dl), d1);
    up8(*reinterpret_cast<const uint4*>(dl + HD2), d2);
    up8(*reinterpret_cast<const uint4*>(dl + Dtot), e1);
    up8(*reinterpret_cast<const uint4*>(dl + Dtot + HD2), e2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      k1[i] = d1[i] + ak1[i] * inv; k2[i] = d2[i] + ak2[i] * inv;
      v1[i] = e1[i] + av1[i] * inv; v2[i] = e2[i] + av2[i] * inv;
    }
  }
  if (a.k_norm_weight != nullptr && !a.store_raw_k) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += k1[i] * k1[i] + k2[i] * k2[i];
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) ss += __shfl_xor(ss, off, 64);
    const float rstd = rsqrtf(ss / (float)D + a.k_norm_eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) { k1[i] = k1[i] * rstd * a.k_norm_weight[p + i]; k2[i] = k2[i] * rstd * a.k_norm_weight[p + HD2 + i]; }
  }
  if (!active) return;
  float o1[8], o2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { o1[i] = k1[i]; o2[i] = k2[i]; }
  if (!a.store_raw_k) {
    float c[8], sn[8];
    ld8f(reinterpret_cast<const float*>(a.cos_sin) + (int64_t)out_pos * a.cos_stride + p, c);
    ld8f(reinterpret_cast<const float*>(a.cos_sin) + (int64_t)out_pos * a.cos_stride + p + HD2, sn);
#pragma unroll
    for (int i = 0; i < 8; ++i) { o1[i] = k1[i] * c[i] - k2[i] * sn[i]; o2[i] = k2[i] * c[i] + k1[i] * sn[i]; }
  }
  auto pk8 = [](const float (&f)[8]) {
    return make_uint4(f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16), f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16),
                      f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16), f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16));
  };
  uint16_t* ok = a.k_cache;
  uint16_t* ov = a.v_cache;
  int64_t ob = (int64_t)out_slot * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
  if (a.out_k_cache != nullptr) {                       // straight into the entry's row of the attention view
    const int64_t row = (int64_t)(n / a.out_entries_per_row) * a.out_view_width + a.out_view_offset + n % a.out_entries_per_row;
    ok = a.out_k_cache;
    ov = a.out_v_cache;
    ob = row * a.out_slot_stride + (int64_t)h * a.out_head_stride + p;
  }
  *reinterpret_cast<uint4*>(ok + ob) = pk8(o1);
  *reinterpret_cast<uint4*>(ok + ob + HD2) = pk8(o2);
  *reinterpret_cast<uint4*>(ov + ob) = pk8(v1);
  *reinterpret_cast<uint4*>(ov + ob + HD2) = pk8(v2);
}

// ------------------------------------------------------------------------------------
// observation-layer token scores
// ------------------------------------------------------------------------------------

constexpr int kTokenScoreChunk = 4096;       // score elements per statistics workgroup

// partial (max, sum exp(x - max)) of one chunk of one head's candidate range -> workspace[b][h][chunk][2]
// VEC: the chunk is whole and its first element 16-byte aligned - every lane takes four float4 instead of sixteen
// 4-byte elements (a wave-load of 1 KiB instead of 256 B: 38.2 -> 22.2 us at 4 x 262 k tokens, 11.1 -> 9.5 us at 1 x).
__global__ void __launch_bounds__(256) token_score_stats_kernel(const SvkDeltakvTokenScoresArgs a, int nchunk) {
  __shared__ float red[16];
  const int b = blockIdx.x, h = blockIdx.y, c = blockIdx.z;
  const int len = min(max(a.candidate_lens[b], 0), a.length - a.candidate_start);
  const int t0 = c * kTokenScoreChunk, t1 = min(len, t0 + kTokenScoreChunk);
  const float* x = a.raw_scores + (int64_t)b * a.raw_stride_b + (int64_t)h * a.raw_stride_h + a.candidate_start;
  float v[kTokenScoreChunk / 256];
  float mx = -INFINITY;
  if (t1 - t0 == kTokenScoreChunk && (reinterpret_cast<uintptr_t>(x + t0) & 15u) == 0u) {
#pragma unroll
    for (int j = 0; j < kTokenScoreChunk / 1024; ++j) {
      const float4 f = *reinterpret_cast<const float4*>(x + t0 + (j * 256 + threadIdx.x) * 4);
      v[4 * j] = mul_rn(f.x, a.scale); v[4 * j + 1] = mul_rn(f.y, a.scale);
      v[4 * j + 2] = mul_rn(f.z, a.scale); v[4 * j + 3] = mul_rn(f.w, a.scale);
    }
#pragma unroll
    for (int j = 0; j < kTokenScoreChunk / 256; ++j) mx = fmaxf(mx, v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < kTokenScoreChunk / 256; ++j) {
      const int t = t0 + j * 256 + threadIdx.x;
      v[j] = t < t1 ? mul_rn(x[t], a.scale) : -INFINITY;
      mx = fmaxf(mx, v[j]);
    }
  }
  mx = block_allmax(mx, red);
  float sum = 0.f;
  if (mx > -INFINITY) {
#pragma unroll
    for (int j = 0; j < kTokenScoreChunk / 256; ++j) sum += expf(v[j] - mx);     // exp(-inf) == 0 for the padding
  }
  sum = block_allsum(sum, red);
  float* row_ws = a.workspace + ((int64_t)b * a.num_heads + h) * (nchunk + 2) * 2;
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    // write-through (agent-scope) stores + a ticket: the LAST chunk of a (row, head) to arrive combines the row's chunk
    // statistics itself (round 6; a launch of its own before: 5.3 us per observation layer).  No release fence - it would
    // write back everything stage 1 has just left dirty in this XCD's L2 (docs/KERNELS.md 4.2).
    __hip_atomic_store(reinterpret_cast<uint32_t*>(row_ws) + 2 * c, __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<uint32_t*>(row_ws) + 2 * c + 1, __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* ticket = reinterpret_cast<int*>(row_ws + 2 * (nchunk + 1));
    s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nchunk - 1;
    if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // self-cleaning
  }
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  // combine (one wave): the order of the sums is fixed by the chunk index, whoever arrives last
  const int lane = threadIdx.x;
  float gm = -INFINITY;
  for (int cc = lane; cc < nchunk; cc += 64)
    gm = fmaxf(gm, __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(row_ws) + 2 * cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  gm = wave_allmax(gm);
  float gs = 0.f;
  for (int cc = lane; cc < nchunk; cc += 64) {
    const float m = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(row_ws) + 2 * cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float sv = __uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(row_ws) + 2 * cc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (m > -INFINITY) gs += sv * expf(m - gm);
  }
  gs = wave_allsum(gs);
  if (lane == 0) { row_ws[2 * nchunk] = gm; row_ws[2 * nchunk + 1] = gs; }
}

// two tokens per thread (256 apart), 16 heads per trip: the kernel is a latency chain per trip (loads -> exp / divide), so
// it wants few trips with many loads in flight (28 heads: 2 trips of 32 loads).  (Round 5: four consecutive tokens per
// thread as one float4 per head - half the waves, the same bytes in flight - measured 14.4 -> 18.8 us at 1 x 262 k and
// 33.5 -> 39.2 us at 4 x; not kept.)
__global__ void __launch_bounds__(256) token_score_final_kernel(const SvkDeltakvTokenScoresArgs a, int nchunk) {
  extern __shared__ float stats[];      // [H][2] global max / sum per head
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * a.num_heads; i += 256)
    stats[i] = a.workspace[(((int64_t)b * a.num_heads + (i >> 1)) * (nchunk + 2) + nchunk) * 2 + (i & 1)];
  __syncthreads();
  const int len = min(max(a.candidate_lens[b], 0), a.length - a.candidate_start);
  const int t0 = blockIdx.x * 512 + threadIdx.x;
  bool in[2];
  float best[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rel = t0 + u * 256 - a.candidate_start;
    in[u] = t0 + u * 256 < a.length && rel >= 0 && rel < len;
    best[u] = 0.f;
  }
  const float* x = a.raw_scores + (int64_t)b * a.raw_stride_b + t0;
  if (in[0] || in[1]) {
    for (int h0 = 0; h0 < a.num_heads; h0 += 16) {
      float xv[16][2];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int h = min(h0 + j, a.num_heads - 1);            // a repeated last head leaves the maximum unchanged
#pragma unroll
        for (int u = 0; u < 2; ++u) xv[j][u] = in[u] ? x[(int64_t)h * a.raw_stride_h + u * 256] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int h = min(h0 + j, a.num_heads - 1);
        const float mx = stats[2 * h], sum = stats[2 * h + 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) best[u] = fmaxf(best[u], expf(mul_rn(xv[j][u], a.scale) - mx) / sum);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = t0 + u * 256;
    if (t >= a.length) continue;
    float out = a.fill_value;
    if (in[u]) {
      out = best[u];
      if (a.round_dtype == SVK_DTYPE_BF16) out = bf16_round(out);
      else if (a.round_dtype == SVK_DTYPE_F16) out = (float)(_Float16)out;
    }
    a.token_scores[(int64_t)b * a.out_stride + t] = out;
  }
}

// ------------------------------------------------------------------------------------
// sorted top-k: radix select on LDS-staged keys, then a bitonic sort of the k winners in LDS.
// Rows longer than one LDS stage are split over several workgroups: every chunk emits its own top-k
// candidates (ascending index), one workgroup per row then selects among the chunks' candidates - the
// global top-k under (score desc, index asc) is a subset of the union of the chunk top-k's.
// ------------------------------------------------------------------------------------

constexpr int kTopkStage = 24576;     // keys staged in LDS by one workgroup (96 KiB) next to a <= 32 KiB sort buffer

// Ascending bitonic sort of keys[0:kpad) in LDS (kpad a power of two, all threads of the workgroup call).  Wave w owns the
// aligned block of kpad / waves elements: every compare-exchange step whose stride stays inside that block needs only
// the wave's own LDS ordering, so only the log2(waves) widest strides of each merge phase take a workgroup barrier
// (2048 keys on 16 waves: 10 barriers instead of 66; 18 -> 7 us).
__device__ __forceinline__ void bitonic_sort_keys(unsigned long long* keys, int kpad) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
  const int epw = kpad / nw;                           // elements per wave block (0 or 1: everything goes the wide way)
  auto cmpx = [&](int i, int size, int stride) {
    const int lo = 2 * i - (i & (stride - 1));
    const int hi = lo + stride;
    const bool up = (lo & size) == 0;
    const unsigned long long x = keys[lo], y = keys[hi];
    if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
  };
  for (int size = 2; size <= kpad; size <<= 1) {
    int stride = size >> 1;
    for (; stride > 0 && stride >= epw; stride >>= 1) {                  // partners in different wave blocks
      for (int i = tid; i < kpad / 2; i += nt) cmpx(i, size, stride);
      __syncthreads();
    }
    for (; stride > 0; stride >>= 1) {                                   // partners inside the wave's block
      for (int i = lane; i < epw / 2; i += 64) cmpx(w * (epw / 2) + i, size, stride);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();                                   // the next phase starts with cross-block strides (or the caller reads)
  }
}

// grid (chunks, rows).  chunks == 1: final indices; else candidates[(row * chunks + c) * k ..] = (key << 32 | index)
__global__ void __launch_bounds__(1024) topk_stage_kernel(const SvkTopkSortedArgs a, int kpad, int chunk, int chunks,
                                                          unsigned long long* cand) {
  __shared__ SelectScratch scratch;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned long long* sorted = reinterpret_cast<unsigned long long*>(dyn);                  // [kpad] (chunks == 1)
  uint32_t* keys = reinterpret_cast<uint32_t*>(dyn + (chunks == 1 ? sizeof(unsigned long long) * kpad : 0));
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
  const float* sc = a.scores + (int64_t)r * a.score_stride;
  const int vlen = a.valid_len ? min(max(a.valid_len[r], 0), a.n) : a.n;
  const int i0 = c * chunk, m = min(a.n, i0 + chunk) - i0;
  const float masked = a.masked_value;
  // staging with 8 loads in flight per thread and the select's OR / AND sweep folded in
  select_bits_begin(scratch);
  uint32_t o_bits = 0u, a_bits = 0xffffffffu;
  for (int j0 = 0; j0 < m; j0 += 8 * nt) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = j0 + u * nt + tid;
      v[u] = (i < m && i0 + i < vlen) ? sc[i0 + i] : masked;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = j0 + u * nt + tid;
      if (i < m) {
        const uint32_t key = desc_key(v[u]);
        keys[i] = key;
        o_bits |= key;
        a_bits &= key;
      }
    }
  }
  select_bits_add(scratch, o_bits, a_bits);
  if (chunks == 1)
    for (int i = tid; i < kpad; i += nt) sorted[i] = ~0ull;
  __syncthreads();
  if (chunks == 1) {
    block_select_topk_ordered_keys([keys](int i) { return keys[i]; }, m, a.k, scratch, [&](int pos, int i) {
      sorted[pos] = ((unsigned long long)keys[i] << 32) | (unsigned)i;
    }, true);
    __syncthreads();
    bitonic_sort_keys(sorted, kpad);
    for (int i = tid; i < a.k; i += nt) a.indices[(int64_t)r * a.index_stride + i] = (int32_t)(sorted[i] & 0xffffffffull);
    return;
  }
  unsigned long long* out = cand + ((int64_t)r * chunks + c) * a.k;
  if (m <= a.k) {                       // the whole chunk survives; pad with "worst" entries
    for (int i = tid; i < a.k; i += nt) out[i] = i < m ? (((unsigned long long)keys[i] << 32) | (unsigned)(i0 + i)) : ~0ull;
    return;
  }
  block_select_topk_ordered_keys([keys](int i) { return keys[i]; }, m, a.k, scratch, [&](int pos, int i) {
    out[pos] = ((unsigned long long)keys[i] << 32) | (unsigned)(i0 + i);
  }, true);
}

// ------------------------------------------------------------------------------------
// long rows (n > kTopkStage), histogram plan: four launches that every CU takes part in (round 2 ran per-chunk selects
// + one merge workgroup: 30 + 34 us at 262 k scores, k = 2048 -> DESIGN.md 4.8):
//   topk_prep_kernel     zeroes the row's histogram and leaves the OR / AND of each 4096-key chunk's valid keys
//   topk_hist_kernel     4096 keys per workgroup -> 4096-bin histogram (LDS, then global atomics) of the 12 key bits
//                        below the highest bit on which the valid keys differ (probabilities / bf16-valued scores
//                        share their leading bits: a fixed top-12 window would put most of the row into one bin)
//   topk_collect_kernel  every workgroup finds the threshold bin T (the bin that holds the k-th key) from the histogram
//                        and writes its keys of the bins <= T in ascending index order into its own region + their count
//   topk_final_kernel    one workgroup per row: the regions concatenated are all candidates in ascending index order
//                        (k <= m < k + |bin T|), the usual ordered select + bitonic sort finishes
// Entries at index >= valid_len + k are never looked at: they compare as `masked_value`, and the k masked entries in
// front of them tie with them at lower indices.  The masked key does not widen the window: a key whose bits above the
// window are below / above the valid keys' common prefix falls into the first / last bin (still monotone in the key).
// ------------------------------------------------------------------------------------

constexpr int kTopkHistBits = 12, kTopkHistBins = 1 << kTopkHistBits, kTopkHistChunk = 4096;

__device__ __forceinline__ int topk_effective_n(const SvkTopkSortedArgs& a, int r, int& vlen) {
  vlen = a.valid_len ? min(max(a.valid_len[r], 0), a.n) : a.n;
  return (int)min((int64_t)a.n, (int64_t)vlen + a.k);
}

// workspace: [rows][kTopkHistBins] u32 histograms | [rows][pad4(nwg)] u32 candidate counts | [rows][pad4(2 nwg)] u32 chunk
// OR / AND | [rows][nwg][kTopkHistChunk] u64 candidates
struct TopkHistWs {
  uint32_t* hist;
  uint32_t* counts;
  uint32_t* bits;
  unsigned long long* cand;
  const unsigned long long* flat = nullptr;    // two-level plan: the candidates once more as one contiguous list,
  const uint32_t* flat_count = nullptr;        // and how many were appended (may exceed the list's capacity)
};
__host__ __device__ __forceinline__ int64_t topk_hist_bytes(int rows, int nwg) {
  return (int64_t)rows * ((int64_t)sizeof(uint32_t) * (kTopkHistBins + ((nwg + 3) & ~3) + ((2 * nwg + 3) & ~3)) +
                          (int64_t)sizeof(unsigned long long) * nwg * kTopkHistChunk);
}
__device__ __forceinline__ TopkHistWs topk_hist_ws(void* workspace, int rows, int r, int nwg) {
  uint32_t* base = static_cast<uint32_t*>(workspace);
  const int cpad = (nwg + 3) & ~3, bpad = (2 * nwg + 3) & ~3;
  TopkHistWs w;
  w.hist = base + (int64_t)r * kTopkHistBins;
  w.counts = base + (int64_t)rows * kTopkHistBins + (int64_t)r * cpad;
  w.bits = base + (int64_t)rows * (kTopkHistBins + cpad) + (int64_t)r * bpad;
  w.cand = reinterpret_cast<unsigned long long*>(base + (int64_t)rows * (kTopkHistBins + cpad + bpad)) + (int64_t)r * nwg * kTopkHistChunk;
  return w;
}

// workspace of the two-level plan (see topk2_* below)
struct Topk2Ws {
  uint32_t* hist1;
  uint32_t* hist2;
  uint32_t* counts;      // [nwg] final candidates per region (what topk_rank_kernel reads)
  uint32_t* kept;        // [nwg] keys of the threshold bin kept aside
  uint32_t* meta;        // [4]: T1, keys in bins below T1, entries appended to `flat`
  unsigned long long* flat;   // [kTopk2Flat] the same candidates as the regions' fronts, unordered, contiguous (rank launch)
  unsigned long long* cand;
};
constexpr int kTopk2Flat = 4096;          // = kTopkRankCap: what the rank launch can take
__host__ __device__ __forceinline__ int64_t topk2_bytes(int rows, int nwg) {
  const int cpad = (nwg + 3) & ~3;
  return (int64_t)rows * ((int64_t)sizeof(uint32_t) * (2 * kTopkHistBins + 2 * cpad + 4) +
                          (int64_t)sizeof(unsigned long long) * ((int64_t)nwg * kTopkHistChunk + kTopk2Flat));
}
__device__ __forceinline__ Topk2Ws topk2_ws(void* workspace, int rows, int r, int nwg) {
  uint32_t* base = static_cast<uint32_t*>(workspace);
  const int cpad = (nwg + 3) & ~3;
  Topk2Ws w;
  w.hist1 = base + (int64_t)r * kTopkHistBins;                                  // (all rows' level-1 histograms first: the zero contract)
  w.hist2 = base + (int64_t)rows * kTopkHistBins + (int64_t)r * kTopkHistBins;
  w.counts = base + (int64_t)rows * 2 * kTopkHistBins + (int64_t)r * cpad;
  w.kept = base + (int64_t)rows * (2 * kTopkHistBins + cpad) + (int64_t)r * cpad;
  w.meta = base + (int64_t)rows * (2 * kTopkHistBins + 2 * cpad) + (int64_t)r * 4;
  unsigned long long* wide = reinterpret_cast<unsigned long long*>(base + (int64_t)rows * (2 * kTopkHistBins + 2 * cpad + 4));
  w.flat = wide + (int64_t)r * kTopk2Flat;
  w.cand = wide + (int64_t)rows * kTopk2Flat + (int64_t)r * nwg * kTopkHistChunk;
  return w;
}
// the same regions / counts seen through the one-level layout's accessor names, for topk_rank_kernel and the fallback
__device__ __forceinline__ TopkHistWs topk2_as_hist_ws(const Topk2Ws& w) {
  TopkHistWs h;
  h.hist = w.hist1; h.counts = w.counts; h.bits = w.meta; h.cand = w.cand;
  h.flat = w.flat; h.flat_count = w.meta + 2;
  return h;
}

__device__ __forceinline__ TopkHistWs topk_any_ws(void* workspace, int rows, int r, int nwg, int two_level) {
  return two_level ? topk2_as_hist_ws(topk2_ws(workspace, rows, r, nwg)) : topk_hist_ws(workspace, rows, r, nwg);
}

// the bin window of a row: bin(key) for keys whose bits above the window equal `hi`
struct TopkWindow {
  int shift;          // lowest key bit of the window
  uint32_t hi;        // the valid keys' common bits above the window (key >> (shift + 12)), 0 when the window reaches bit 31
};
__device__ __forceinline__ uint32_t topk_bin(const TopkWindow& w, uint32_t key) {
  const int up = w.shift + kTopkHistBits;
  const uint32_t hi = up >= 32 ? 0u : key >> up;
  if (hi < w.hi) return 0u;
  if (hi > w.hi) return kTopkHistBins - 1;
  return (key >> w.shift) & (kTopkHistBins - 1);
}
// all threads of the workgroup call; `red` = 32 words of LDS
__device__ __forceinline__ TopkWindow topk_window(const TopkHistWs& ws, int nwg, uint32_t* red) {
  uint32_t o = 0u, an = 0xffffffffu;
  for (int c = threadIdx.x; c < nwg; c += blockDim.x) { o |= ws.bits[2 * c]; an &= ws.bits[2 * c + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    o |= (uint32_t)__shfl_xor((int)o, off, 64);
    an &= (uint32_t)__shfl_xor((int)an, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { red[2 * w] = o; red[2 * w + 1] = an; }
  __syncthreads();
  o = 0u; an = 0xffffffffu;
  for (int j = 0; j < nw; ++j) { o |= red[2 * j]; an &= red[2 * j + 1]; }
  const uint32_t varying = o ^ an;
  TopkWindow win;
  const int top = varying ? 31 - __builtin_clz(varying) : kTopkHistBits - 1;
  win.shift = max(top - (kTopkHistBits - 1), 0);
  const int up = win.shift + kTopkHistBits;
  win.hi = up >= 32 ? 0u : an >> up;                 // above the highest varying bit every valid key equals the AND
  return win;
}

__device__ __forceinline__ void topk_load4(const SvkTopkSortedArgs& a, int r, int i0, int vlen, uint32_t (&key)[4]) {
  const float* sc = a.scores + (int64_t)r * a.score_stride;
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * 1024 + (int)threadIdx.x;
    v[u] = i < vlen ? sc[i] : a.masked_value;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) key[u] = desc_key(v[u]);
}

__global__ void __launch_bounds__(1024) topk_prep_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ uint32_t red[32];
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const TopkHistWs ws = topk_hist_ws(workspace, (int)gridDim.y, r, nwg);
  const int per = (kTopkHistBins + nwg - 1) / nwg;               // this workgroup's slice of the histogram to zero
  for (int j = tid; j < per; j += 1024)
    if (c * per + j < kTopkHistBins) ws.hist[c * per + j] = 0u;
  int vlen;
  const int ne = topk_effective_n(a, r, vlen);
  const int i0 = c * kTopkHistChunk;
  uint32_t o = 0u, an = 0xffffffffu;
  if (i0 < ne) {
    uint32_t key[4];
    topk_load4(a, r, i0, vlen, key);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024 + tid;
      if (i < ne && (i < vlen || vlen == 0)) { o |= key[u]; an &= key[u]; }      // masked entries only when nothing is valid
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    o |= (uint32_t)__shfl_xor((int)o, off, 64);
    an &= (uint32_t)__shfl_xor((int)an, off, 64);
  }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = o; red[2 * (tid >> 6) + 1] = an; }
  __syncthreads();
  if (tid == 0) {
    for (int j = 1; j < 16; ++j) { o |= red[2 * j]; an &= red[2 * j + 1]; }
    ws.bits[2 * c] = o;
    ws.bits[2 * c + 1] = an;
  }
}

__global__ void __launch_bounds__(1024) topk_hist_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ int hist[kTopkHistBins];
  __shared__ uint32_t red[32];
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  int vlen;
  const int ne = topk_effective_n(a, r, vlen);
  const int i0 = c * kTopkHistChunk;
  if (i0 >= ne) return;
  const TopkHistWs ws = topk_hist_ws(workspace, (int)gridDim.y, r, nwg);
  uint32_t key[4];
  topk_load4(a, r, i0, vlen, key);
#pragma unroll
  for (int u = 0; u < 4; ++u) hist[u * 1024 + tid] = 0;
  const TopkWindow win = topk_window(ws, nwg, red);              // (its barriers also publish the zeroed bins)
#pragma unroll
  for (int u = 0; u < 4; ++u) hist_add_aggregated(hist, topk_bin(win, key[u]), i0 + u * 1024 + tid < ne);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = hist[u * 1024 + tid];
    if (n != 0) atomicAdd(&ws.hist[u * 1024 + tid], (uint32_t)n);
  }
}

__global__ void __launch_bounds__(1024) topk_collect_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ int wsum[16];
  __shared__ uint32_t red[32];
  __shared__ int s_T;
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int vlen;
  const int ne = topk_effective_n(a, r, vlen);
  const TopkHistWs ws = topk_hist_ws(workspace, (int)gridDim.y, r, nwg);
  const int i0 = c * kTopkHistChunk;
  if (i0 >= ne) {
    if (tid == 0) ws.counts[c] = 0u;
    return;
  }
  uint32_t key[4];
  topk_load4(a, r, i0, vlen, key);
  const TopkWindow win = topk_window(ws, nwg, red);
  // threshold bin: thread t owns bins 4t .. 4t+3
  {
    const uint4 h = *reinterpret_cast<const uint4*>(ws.hist + tid * 4);
    const int cb[4] = {(int)h.x, (int)h.y, (int)h.z, (int)h.w};
    const int local = cb[0] + cb[1] + cb[2] + cb[3];
    const int incl_w = wave_incl_scan_add(local);
    if (lane == 63) wsum[w] = incl_w;
    __syncthreads();
    int base = 0;
    for (int j = 0; j < w; ++j) base += wsum[j];
    const int incl = base + incl_w, excl = incl - local;
    if (a.k > excl && a.k <= incl) {
      int run = excl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (a.k > run && a.k <= run + cb[j]) s_T = tid * 4 + j;
        run += cb[j];
      }
    }
    __syncthreads();
  }
  const uint32_t T = (uint32_t)s_T;
  unsigned long long* out = ws.cand + (int64_t)c * kTopkHistChunk;
  int written = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * 1024 + tid;
    const bool take = i < ne && topk_bin(win, key[u]) <= T;
    int total;
    const int rank = block_excl_count(take, wsum, total);
    if (take) out[written + rank] = ((unsigned long long)key[u] << 32) | (unsigned)i;
    written += total;
  }
  if (tid == 0) ws.counts[c] = (uint32_t)written;
}

// one workgroup per row; `lds_cap` = candidates whose keys fit the dynamic LDS next to the sort buffer and the prefix table
// (any block size of 64..1024 threads; `dyn` = the kernel's dynamic LDS, `rows` = rows of the launch)
__device__ __forceinline__ void topk_final_select_sort(const SvkTopkSortedArgs& a, int r, int rows, int kpad, void* workspace, int nwg,
                                                       int lds_cap, SelectScratch& scratch, unsigned char* dyn, int two_level) {
  unsigned long long* sorted = reinterpret_cast<unsigned long long*>(dyn);                    // [kpad]
  int* prefix = reinterpret_cast<int*>(dyn + sizeof(unsigned long long) * kpad);               // [nwg + 1] exclusive
  uint32_t* keys = reinterpret_cast<uint32_t*>(prefix + ((nwg + 2) & ~1));                     // [lds_cap]
  const int tid = threadIdx.x, nt = blockDim.x;
  const TopkHistWs ws = topk_any_ws(workspace, rows, r, nwg, two_level);
  // exclusive prefix of the region counts: one coalesced load into LDS (a thread summing straight from memory chains
  // a round trip per region: 20 us at 64 regions), then every thread adds up what is in front of its regions
  int* cnt = reinterpret_cast<int*>(keys);           // the key stage is not in use yet
  for (int c = tid; c < nwg; c += nt) cnt[c] = (int)ws.counts[c];
  __syncthreads();
  for (int c = tid; c <= nwg; c += nt) {
    int run = 0;
    for (int j = 0; j < c; ++j) run += cnt[j];
    prefix[c] = run;
  }
  for (int i = tid; i < kpad; i += nt) sorted[i] = ~0ull;
  __syncthreads();
  const int m = prefix[nwg];
  auto cand_at = [&](int i) {
    int lo = 0, hi = nwg;                       // largest c with prefix[c] <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= i) lo = mid; else hi = mid;
    }
    return ws.cand[(int64_t)lo * kTopkHistChunk + (i - prefix[lo])];
  };
  if (m <= lds_cap) {
    select_bits_begin(scratch);
    uint32_t o_bits = 0u, a_bits = 0xffffffffu;
    for (int j0 = 0; j0 < m; j0 += 4 * nt) {
      unsigned long long c4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = j0 + u * nt + tid;
        c4[u] = i < m ? cand_at(i) : ~0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = j0 + u * nt + tid;
        if (i < m) {
          const uint32_t key = (uint32_t)(c4[u] >> 32);
          keys[i] = key;
          o_bits |= key;
          a_bits &= key;
        }
      }
    }
    select_bits_add(scratch, o_bits, a_bits);
    __syncthreads();
    block_select_topk_ordered_keys([keys](int i) { return keys[i]; }, m, a.k, scratch, [&](int pos, int i) {
      sorted[pos] = cand_at(i);
    }, true);
  } else {
    block_select_topk_ordered_keys([&](int i) { return (uint32_t)(cand_at(i) >> 32); }, m, a.k, scratch, [&](int pos, int i) {
      sorted[pos] = cand_at(i);
    });
  }
  __syncthreads();
  bitonic_sort_keys(sorted, kpad);
  for (int i = tid; i < a.k; i += nt) a.indices[(int64_t)r * a.index_stride + i] = (int32_t)(sorted[i] & 0xffffffffull);
}

__global__ void __launch_bounds__(1024) topk_final_kernel(const SvkTopkSortedArgs a, int kpad, void* workspace, int nwg, int lds_cap) {
  __shared__ SelectScratch scratch;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  topk_final_select_sort(a, (int)blockIdx.x, (int)gridDim.x, kpad, workspace, nwg, lds_cap, scratch, dyn, 0);
}

// The final stage as a RANK launch (round 6; `SVK_TOPK_FINAL=select` keeps the single-workgroup select + bitonic sort
// above: 25 us at 262 k scores, k = 2048).  The candidates of a row - every key of the bins <= T, k <= m < k + |bin T| -
// are few: each of them can simply COUNT the candidates in front of it in the (score desc, index asc) order.  That count is
// its output position; the candidates with a count below k are the result, already in order - exact selection and sort in
// one barrier-free sweep.  The sweep is m^2 compare-and-count operations: spread thin - a workgroup of 256 threads stages
// all m candidates in LDS (all of a thread's loads in flight at once) and ranks SIXTEEN of them, sixteen lanes per candidate
// each over every sixteenth entry of the list - it is ~130 LDS reads per lane on ~130 CUs (the first form, 128 candidates
// per workgroup and two lanes per candidate, kept 68 SIMDs busy for 10 us).  Rows with more than kTopkRankCap candidates (a huge tie group at the
// threshold) take the select + sort above inside the row's first workgroup.
constexpr int kTopkRankCap = 4096, kTopkRankPerWg = 16;

__global__ void __launch_bounds__(256) topk_rank_kernel(const SvkTopkSortedArgs a, int kpad, void* workspace, int nwg, int lds_cap,
                                                        int two_level) {
  __shared__ SelectScratch scratch;
  __shared__ int s_prefix[260];
  __shared__ int wtot[4];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const int r = blockIdx.y, tid = threadIdx.x;
  const TopkHistWs ws = topk_any_ws(workspace, (int)gridDim.y, r, nwg, two_level);
  int m = -1;
  const bool flat = ws.flat != nullptr;
  if (flat) {
    m = (int)*ws.flat_count;                    // (one scalar load: no prefix over the regions on this path)
  } else if (nwg <= 256) {
    // exclusive prefix of the region counts (one thread per region, wave scans)
    const int cnt = tid < nwg ? (int)ws.counts[tid] : 0;
    const int incl = wave_incl_scan_add(cnt);
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int j = 0; j < (tid >> 6); ++j) base += wtot[j];
    if (tid < nwg) s_prefix[tid] = base + incl - cnt;
    if (tid == 255) s_prefix[256] = base + incl;
    __syncthreads();
    m = s_prefix[256];
    __syncthreads();
    if (tid == 0) s_prefix[nwg] = m;
    __syncthreads();
  }
  if (m < 0 || m > kTopkRankCap) {
    if (blockIdx.x == 0) topk_final_select_sort(a, r, (int)gridDim.y, kpad, workspace, nwg, lds_cap, scratch, dyn, two_level);
    return;
  }
  if ((int)blockIdx.x * kTopkRankPerWg >= m) return;
  // all m candidates -> LDS as (key, index) pairs of 32-bit words (every load of the sweep in flight at once)
  uint2* cand = reinterpret_cast<uint2*>(dyn);                  // [m] (<= 32 KB)
  {
    unsigned long long c16[kTopkRankCap / 256];
#pragma unroll
    for (int u = 0; u < kTopkRankCap / 256; ++u) {
      const int i = u * 256 + tid;
      c16[u] = 0ull;
      if (i < m) {
        if (flat) {
          c16[u] = ws.flat[i];
        } else {
          int lo = 0, hi = nwg;                 // largest c with prefix[c] <= i
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_prefix[mid] <= i) lo = mid; else hi = mid;
          }
          c16[u] = ws.cand[(int64_t)lo * kTopkHistChunk + (i - s_prefix[lo])];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kTopkRankCap / 256; ++u) {
      const int i = u * 256 + tid;
      if (i < m) cand[i] = make_uint2((uint32_t)(c16[u] >> 32), (uint32_t)c16[u]);
    }
  }
  __syncthreads();
  // sixteen lanes per candidate, each over every sixteenth candidate of the list
  const int mine_i = (int)blockIdx.x * kTopkRankPerWg + (tid >> 4), part = tid & 15;
  const uint2 mine = cand[mine_i < m ? mine_i : 0];
  int rank = 0;
#pragma unroll 4
  for (int j = part; j < m; j += 16) {
    const uint2 c = cand[j];
    rank += (int)(c.x < mine.x) + (int)((c.x == mine.x) & (c.y < mine.y));
  }
  rank += __shfl_xor(rank, 8, 64);
  rank += __shfl_xor(rank, 4, 64);
  rank += __shfl_xor(rank, 2, 64);
  rank += __shfl_xor(rank, 1, 64);
  if (part == 0 && mine_i < m && rank < a.k) a.indices[(int64_t)r * a.index_stride + rank] = (int32_t)mine.y;
}

// ------------------------------------------------------------------------------------
// long rows, TWO-LEVEL plan (round 6, default; `SVK_TOPK_PLAN=hist` keeps the one-level plan above).  The one-level
// threshold bin still holds 5-12 k keys of a 262 k row (fp32 scores spread over ~60 of the 4096 bins), which the final
// workgroup then had to radix-select and sort: 22-25 us.  Two fixed 12-bit levels - key bits [31:20], then [19:8] inside the
// threshold bin - leave k + a handful of candidates, few enough for every candidate to rank itself (topk_rank_kernel):
//   topk2_hist1_kernel    4096 keys per workgroup -> level-1 histogram (LDS, then global atomics); zeroes level 2
//   topk2_split_kernel    threshold bin T1 from level 1; keys below T1 are winners (front of the chunk's region, ascending
//                         index), keys IN T1 are kept aside (back of the region) and counted into the level-2 histogram
//   topk2_refine_kernel   threshold sub-bin T2 from level 2; the kept keys with sub-bin <= T2 join the region's front
//                         (still ascending index among themselves); zeroes level 1 for the next launch
//   topk_rank_kernel      every candidate counts the candidates in front of it = its output position
// Candidates that tie on all 24 bits beyond kTopkRankCap (rows of equal scores) fall back to the ordered select + sort in
// the rank launch's first workgroup; equal keys lie in ascending index order in the concatenated regions, which is what
// that select needs.  The level-1 histogram must be ZERO when the launch starts and is zero again when it ends.
// ------------------------------------------------------------------------------------
// thread t owns bins 4t .. 4t+3 of a 4096-bin histogram (1024 threads): the bin that holds the `want`-th key (1-based)
// and the number of keys in the bins before it.  All threads call; `wsum` = 16 ints, `res` = 2 ints of LDS.
__device__ __forceinline__ void topk2_find_bin(const uint32_t* hist, int want, int* wsum, int* res) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint4 h = *reinterpret_cast<const uint4*>(hist + tid * 4);
  const int cb[4] = {(int)h.x, (int)h.y, (int)h.z, (int)h.w};
  const int local = cb[0] + cb[1] + cb[2] + cb[3];
  const int incl_w = wave_incl_scan_add(local);
  if (lane == 63) wsum[w] = incl_w;
  __syncthreads();
  int base = 0;
  for (int j = 0; j < w; ++j) base += wsum[j];
  const int incl = base + incl_w, excl = incl - local;
  if (want > excl && want <= incl) {
    int run = excl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (want > run && want <= run + cb[j]) { res[0] = tid * 4 + j; res[1] = run; }
      run += cb[j];
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(1024) topk2_hist1_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ int hist[kTopkHistBins];
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const Topk2Ws ws = topk2_ws(workspace, (int)gridDim.y, r, nwg);
  const int per = (kTopkHistBins + nwg - 1) / nwg;               // this workgroup's slice of the level-2 histogram to zero
  for (int j = tid; j < per; j += 1024)
    if (c * per + j < kTopkHistBins) ws.hist2[c * per + j] = 0u;
  if (c == 0 && tid == 0) ws.meta[2] = 0u;
  int vlen;
  const int ne = topk_effective_n(a, r, vlen);
  const int i0 = c * kTopkHistChunk;
  if (i0 >= ne) return;
  uint32_t key[4];
  topk_load4(a, r, i0, vlen, key);
#pragma unroll
  for (int u = 0; u < 4; ++u) hist[u * 1024 + tid] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) hist_add_aggregated(hist, key[u] >> 20, i0 + u * 1024 + tid < ne);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = hist[u * 1024 + tid];
    if (n != 0) atomicAdd(&ws.hist1[u * 1024 + tid], (uint32_t)n);
  }
}

__global__ void __launch_bounds__(1024) topk2_split_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ int hist[kTopkHistBins];
  __shared__ int wsum[16], res[2];
  __shared__ int cnt[4][16];
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int vlen;
  const int ne = topk_effective_n(a, r, vlen);
  const Topk2Ws ws = topk2_ws(workspace, (int)gridDim.y, r, nwg);
  const int i0 = c * kTopkHistChunk;
  if (i0 >= ne) {
    if (tid == 0) { ws.counts[c] = 0u; ws.kept[c] = 0u; }
    return;
  }
  uint32_t key[4];
  topk_load4(a, r, i0, vlen, key);
#pragma unroll
  for (int u = 0; u < 4; ++u) hist[u * 1024 + tid] = 0;
  topk2_find_bin(ws.hist1, a.k, wsum, res);                      // (its barriers also publish the zeroed bins)
  const uint32_t T1 = (uint32_t)res[0];
  if (c == 0 && tid == 0) { ws.meta[0] = T1; ws.meta[1] = (uint32_t)res[1]; }
  // winners (bins below T1) to the front of the region, the threshold bin's keys to its back, both in ascending index
  // order: four ballots per wave, ONE barrier
  bool win[4], keep[4];
  unsigned long long bw[4], bk[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool in = i0 + u * 1024 + tid < ne;
    const uint32_t b1 = key[u] >> 20;
    win[u] = in && b1 < T1;
    keep[u] = in && b1 == T1;
    bw[u] = __ballot(win[u]);
    bk[u] = __ballot(keep[u]);
    if (lane == 0) cnt[u][w] = __popcll(bw[u]) | (__popcll(bk[u]) << 16);
    hist_add_aggregated(hist, (key[u] >> 8) & (kTopkHistBins - 1), keep[u]);
  }
  __syncthreads();
  __shared__ int s_flat_base;
  if (tid == 0) {
    int total = 0;
    for (int u = 0; u < 4; ++u)
      for (int j = 0; j < 16; ++j) total += cnt[u][j] & 0xffff;
    s_flat_base = total ? (int)atomicAdd(&ws.meta[2], (uint32_t)total) : 0;
  }
  __syncthreads();
  const int flat_base = s_flat_base;
  unsigned long long* out = ws.cand + (int64_t)c * kTopkHistChunk;
  int nwin = 0, nkeep = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int bw_base = 0, bk_base = 0, tw = 0, tk = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int v = cnt[u][j];
      if (j < w) { bw_base += v & 0xffff; bk_base += v >> 16; }
      tw += v & 0xffff;
      tk += v >> 16;
    }
    const unsigned long long item = ((unsigned long long)key[u] << 32) | (unsigned)(i0 + u * 1024 + tid);
    if (win[u]) {
      const int pos = nwin + bw_base + __popcll(bw[u] & below);
      out[pos] = item;
      if (flat_base + pos < kTopk2Flat) ws.flat[flat_base + pos] = item;
    }
    if (keep[u]) out[kTopkHistChunk - 1 - (nkeep + bk_base + __popcll(bk[u] & below))] = item;
    nwin += tw;
    nkeep += tk;
  }
  if (tid == 0) { ws.counts[c] = (uint32_t)nwin; ws.kept[c] = (uint32_t)nkeep; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = hist[u * 1024 + tid];
    if (n != 0) atomicAdd(&ws.hist2[u * 1024 + tid], (uint32_t)n);
  }
}

__global__ void __launch_bounds__(1024) topk2_refine_kernel(const SvkTopkSortedArgs a, void* workspace, int nwg) {
  __shared__ int wsum[16], res[2];
  __shared__ int cnt[4][16];
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const Topk2Ws ws = topk2_ws(workspace, (int)gridDim.y, r, nwg);
  const int per = (kTopkHistBins + nwg - 1) / nwg;               // level 1 is not read any more: zero it for the next launch
  for (int j = tid; j < per; j += 1024)
    if (c * per + j < kTopkHistBins) ws.hist1[c * per + j] = 0u;
  const int nkeep = (int)ws.kept[c];
  if (nkeep == 0) return;                                        // (uniform: nothing of this chunk lies in the threshold bin)
  const int need = a.k - (int)ws.meta[1];                        // keys still wanted from the threshold bin (>= 1)
  topk2_find_bin(ws.hist2, need, wsum, res);
  const uint32_t T2 = (uint32_t)res[0];
  unsigned long long* out = ws.cand + (int64_t)c * kTopkHistChunk;
  int nwin = (int)ws.counts[c];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int j0 = 0; j0 < nkeep; j0 += 4 * 1024) {
    unsigned long long item[4], bal[4];
    bool take[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 1024 + tid;
      item[u] = j < nkeep ? out[kTopkHistChunk - 1 - j] : ~0ull;
      take[u] = j < nkeep && (((uint32_t)(item[u] >> 32) >> 8) & (kTopkHistBins - 1)) <= T2;
      bal[u] = __ballot(take[u]);
      if (lane == 0) cnt[u][w] = __popcll(bal[u]);
    }
    __syncthreads();                                             // (also: every kept item of this trip is in registers)
    __shared__ int s_flat_base;
    if (tid == 0) {
      int total = 0;
      for (int u = 0; u < 4; ++u)
        for (int j = 0; j < 16; ++j) total += cnt[u][j];
      s_flat_base = total ? (int)atomicAdd(&ws.meta[2], (uint32_t)total) : 0;
    }
    __syncthreads();
    int flat_pos = s_flat_base;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int base = 0, tot = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < w) base += cnt[u][j];
        tot += cnt[u][j];
      }
      if (take[u]) {
        const int pos = base + __popcll(bal[u] & below);
        out[nwin + pos] = item[u];
        if (flat_pos + pos < kTopk2Flat) ws.flat[flat_pos + pos] = item[u];
      }
      nwin += tot;
      flat_pos += tot;
    }
    __syncthreads();
  }
  if (tid == 0) ws.counts[c] = (uint32_t)nwin;
}

// fallback for shapes whose candidates do not fit one LDS stage: single workgroup, keys re-read from memory
__global__ void __launch_bounds__(1024) topk_sorted_kernel(const SvkTopkSortedArgs a, int kpad) {
  __shared__ SelectScratch scratch;
  extern __shared__ unsigned long long keys[];     // [kpad]
  const int r = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* sc = a.scores + (int64_t)r * a.score_stride;
  const int vlen = a.valid_len ? min(max(a.valid_len[r], 0), a.n) : a.n;
  const float masked = a.masked_value;
  auto score_at = [&](int i) { return i < vlen ? sc[i] : masked; };
  for (int i = tid; i < kpad; i += nt) keys[i] = ~0ull;
  __syncthreads();
  block_select_topk_ordered_fn(score_at, a.n, a.k, scratch, [&](int pos, int i) {
    keys[pos] = ((unsigned long long)desc_key(score_at(i)) << 32) | (unsigned)i;
  });
  __syncthreads();
  bitonic_sort_keys(keys, kpad);
  for (int i = tid; i < a.k; i += nt) a.indices[(int64_t)r * a.index_stride + i] = (int32_t)(keys[i] & 0xffffffffull);
}

__global__ void __launch_bounds__(256) deltakv_decode_alloc_kernel(const SvkDeltakvDecodeAllocArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.graph_batch) return;
  const int src = b < a.batch ? b : 0;               // padded graph lanes mirror lane 0
  const int row = a.meta[src], cur = a.meta[a.meta_stride + src];
  const int clen = a.meta[4 * a.meta_stride + src];
  int fs = -1, ss = -1;
  if (b < a.batch) {
    fs = a.meta[2 * a.meta_stride + b];
    ss = a.meta[3 * a.meta_stride + b];
    a.full_slots_map[(int64_t)row * a.full_map_stride + cur] = fs;
    a.full_slot_to_pos[fs] = cur;
    a.sparse_raw_slots_map[(int64_t)row * a.sparse_map_stride + cur] = ss;
    a.sparse_slot_to_pos[ss] = cur;
  }
  a.context_lens[b] = cur + 1;
  a.req_indices[b] = row;
  a.slot_mapping[b] = fs;
  a.sparse_slot_mapping[b] = ss;
  a.compressed_lens[b] = clen;
}

// The same step from device-resident state (one workgroup: the stack pointers move once, behind a barrier).
__global__ void __launch_bounds__(256) deltakv_device_begin_kernel(const SvkDeltakvDeviceStepArgs a) {
  const int fp = *a.full_ptr, sp = *a.sparse_ptr;
  const int row0 = a.rows[0];
  const int cur0 = a.row_len[row0], clen0 = a.compressed_len[row0];
  __syncthreads();                                   // every lane has read lane 0's length before its row moves on
  for (int b = threadIdx.x; b < a.graph_batch; b += blockDim.x) {
    int row = row0, cur = cur0, clen = clen0, fs = -1, ss = -1;
    if (b < a.batch) {
      row = a.rows[b];
      cur = a.row_len[row];
      clen = a.compressed_len[row];
      fs = a.full_stack[fp - a.batch + b];
      ss = a.sparse_stack[sp - a.batch + b];
      a.full_slots_map[(int64_t)row * a.full_map_stride + cur] = fs;
      a.full_slot_to_pos[fs] = cur;
      a.sparse_raw_slots_map[(int64_t)row * a.sparse_map_stride + cur] = ss;
      a.sparse_slot_to_pos[ss] = cur;
      a.row_len[row] = cur + 1;
    }
    a.context_lens[b] = cur + 1;
    a.req_indices[b] = row;
    a.slot_mapping[b] = fs;
    a.sparse_slot_mapping[b] = ss;
    a.compressed_lens[b] = clen;
  }
  if (threadIdx.x == 0) {
    *a.full_ptr = fp - a.batch;
    *a.sparse_ptr = sp - a.batch;
  }
}

}  // namespace
}  // namespace svk

extern "C" int svk_deltakv_device_step_begin(const SvkDeltakvDeviceStepArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->rows != nullptr && a->row_len != nullptr && a->full_ptr != nullptr && a->sparse_ptr != nullptr,
              SVK_ERR_VALUE, "svk_deltakv_device_step_begin: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static DeltaKV decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static DeltaKV decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  hipLaunchKernelGGL(deltakv_device_begin_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_deltakv_device_step_begin");
}

extern "C" int svk_deltakv_decode_alloc(const SvkDeltakvDecodeAllocArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->meta != nullptr, SVK_ERR_VALUE, "svk_deltakv_decode_alloc: null args");
  SVK_REQUIRE(a->batch > 0, SVK_ERR_VALUE, "Static DeltaKV decode requires a non-empty real decode batch.");
  SVK_REQUIRE(a->graph_batch >= a->batch, SVK_ERR_VALUE,
              "Static DeltaKV decode graph batch is smaller than the real decode batch: graph=%d, real=%d.", a->graph_batch, a->batch);
  SVK_REQUIRE(a->meta_stride >= a->batch, SVK_ERR_VALUE, "svk_deltakv_decode_alloc: meta_stride %lld < batch %d",
              (long long)a->meta_stride, a->batch);
  hipLaunchKernelGGL(deltakv_decode_alloc_kernel, dim3((a->graph_batch + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_deltakv_decode_alloc");
}

extern "C" int svk_deltakv_static_decode_plan(const SvkDeltakvPlanArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_deltakv_static_decode_plan: null args");
  SVK_REQUIRE(a->k_max >= 0 && a->sink >= 0 && a->max_buffer >= 0 && a->max_positions > 0, SVK_ERR_VALUE,
              "svk_deltakv_static_decode_plan: bad shape parameters");
  if (a->batch <= 0) return SVK_OK;
  const int plan_cols = a->sink + a->k_max + a->max_buffer;
  hipLaunchKernelGGL(deltakv_plan_kernel, dim3((plan_cols + 255) / 256, a->batch), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_deltakv_static_decode_plan");
}

extern "C" int svk_dequantize_grouped(const SvkDequantGroupedArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_dequantize_grouped: null args");
  SVK_REQUIRE(a->bits == 2 || a->bits == 4 || a->bits == 8, SVK_ERR_VALUE,
              "Packed quantization supports bits=(2, 4, 8), got %d.", a->bits);
  SVK_REQUIRE(a->group_size > 0 && a->features % a->group_size == 0, SVK_ERR_VALUE,
              "dequantization requires output_dim divisible by group_size, got output_dim=%d, group_size=%d.", a->features, a->group_size);
  SVK_REQUIRE(a->features % (32 / a->bits) == 0, SVK_ERR_VALUE, "features %d not divisible by %d", a->features, 32 / a->bits);
  if (a->rows <= 0) return SVK_OK;
  hipLaunchKernelGGL(dequant_grouped_kernel, dim3((a->features + 255) / 256, a->rows), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_dequantize_grouped");
}

namespace svk {
namespace {
int launch_reconstruct(const SvkDeltakvReconstructArgs* a, const SvkDeltakvReconstructBatch& lb, svk_stream_t stream) {
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_deltakv_reconstruct_writeback: null args");
  SVK_REQUIRE(a->head_dim % 2 == 0 && a->head_dim > 0, SVK_ERR_LAYOUT, "head_dim must be even");
  SVK_REQUIRE(a->delta_bits == 0 || a->delta_bits == 2 || a->delta_bits == 4 || a->delta_bits == 8, SVK_ERR_VALUE,
              "DeltaKV fused residual reconstruction supports quant_bits=2, 4 or 8, got %d.", a->delta_bits);
  SVK_REQUIRE(a->k_fathers > 0, SVK_ERR_VALUE, "svk_deltakv_reconstruct_writeback: k_fathers must be positive");
  const int per_entry = a->num_kv_heads * (a->head_dim / 2);
  SVK_REQUIRE(per_entry <= 1024, SVK_ERR_LAYOUT, "svk_deltakv_reconstruct_writeback: Hkv*D/2 = %d exceeds 1024", per_entry);
  if (a->delta_bits != 0) {
    const int feats = 2 * a->num_kv_heads * a->head_dim;
    SVK_REQUIRE(a->group_size > 0 && feats % a->group_size == 0, SVK_ERR_VALUE,
                "DeltaKV fused residual reconstruction requires 2*D divisible by group_size; 2*D=%d, group_size=%d.", feats, a->group_size);
    SVK_REQUIRE(a->latent_slots != nullptr && a->scale != nullptr && a->mn != nullptr, SVK_ERR_VALUE,
                "svk_deltakv_reconstruct_writeback: packed residuals need latent_slots, scale and mn");
  }
  if (a->n <= 0) return SVK_OK;
  if (a->out_k_cache != nullptr) {
    SVK_REQUIRE(a->out_v_cache != nullptr && a->out_entries_per_row > 0 && a->out_view_width >= a->out_view_offset + a->out_entries_per_row &&
                    a->out_view_offset >= 0 && a->out_slot_stride % 8 == 0 && a->out_head_stride % 8 == 0 &&
                    reinterpret_cast<uintptr_t>(a->out_k_cache) % 16 == 0 && reinterpret_cast<uintptr_t>(a->out_v_cache) % 16 == 0,
                SVK_ERR_LAYOUT, "svk_deltakv_reconstruct_writeback: bad view destination (width %d, offset %d, entries per row %d)",
                a->out_view_width, a->out_view_offset, a->out_entries_per_row);
    SVK_REQUIRE(lb.n_batch == 1 || lb.out_cache_stride_batch % 8 == 0, SVK_ERR_LAYOUT,
                "svk_deltakv_reconstruct_writeback_batched: per-layer view stride must keep 16-byte alignment");
  }
  // decode path: dense bf16 delta, fp32 cos|sin, 16-byte aligned rows -> 16-byte lanes
  if (a->delta_bits == 0 && a->delta_dtype == SVK_DTYPE_BF16 && a->cos_dtype == SVK_DTYPE_F32 && (a->head_dim == 64 || a->head_dim == 128) &&
      a->num_kv_heads <= 8 && a->delta_stride % 8 == 0 && a->kv_slot_stride % 8 == 0 && a->kv_head_stride % 8 == 0 && a->cos_stride % 4 == 0 &&
      reinterpret_cast<uintptr_t>(a->delta) % 16 == 0 && reinterpret_cast<uintptr_t>(a->cos_sin) % 16 == 0) {
    const int lpe = (a->head_dim / 16) * a->num_kv_heads;
    const int epb2 = 256 / lpe;
    const dim3 grid((a->n + epb2 - 1) / epb2, lb.n_batch), block(256);
    if (a->head_dim == 128) hipLaunchKernelGGL(deltakv_reconstruct_vec_kernel<128>, grid, block, 0, static_cast<hipStream_t>(stream), *a, lb);
    else hipLaunchKernelGGL(deltakv_reconstruct_vec_kernel<64>, grid, block, 0, static_cast<hipStream_t>(stream), *a, lb);
    return check_launch("svk_deltakv_reconstruct_writeback");
  }
  SVK_REQUIRE(lb.n_batch == 1, SVK_ERR_LAYOUT, "svk_deltakv_reconstruct_writeback_batched: only the dense bf16 16-byte form is batched");
  SVK_REQUIRE(a->out_k_cache == nullptr, SVK_ERR_LAYOUT, "svk_deltakv_reconstruct_writeback: the view destination needs the dense bf16 16-byte form");
  int threads = per_entry;
  if (threads < 256) threads = (256 / per_entry) * per_entry;
  const int epb = threads / per_entry;
  hipLaunchKernelGGL(deltakv_reconstruct_kernel, dim3((a->n + epb - 1) / epb), dim3(threads), sizeof(float) * threads,
                     static_cast<hipStream_t>(stream), *a);
  return check_launch("svk_deltakv_reconstruct_writeback");
}
}  // namespace
}  // namespace svk

extern "C" int svk_deltakv_reconstruct_writeback(const SvkDeltakvReconstructArgs* a, svk_stream_t stream) {
  SvkDeltakvReconstructBatch one = {};
  one.n_batch = 1;
  return svk::launch_reconstruct(a, one, stream);
}

extern "C" int svk_deltakv_reconstruct_writeback_batched(const SvkDeltakvReconstructArgs* first, const SvkDeltakvReconstructBatch* b,
                                                         svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(b != nullptr && b->n_batch >= 1 && b->n_batch <= 65535, SVK_ERR_VALUE, "svk_deltakv_reconstruct_writeback_batched: bad batch description");
  SVK_REQUIRE(b->n_batch == 1 || (b->delta_stride_batch % 8 == 0 && b->kv_cache_stride_batch % 8 == 0), SVK_ERR_LAYOUT,
              "svk_deltakv_reconstruct_writeback_batched: per-layer strides must keep 16-byte alignment");
  return launch_reconstruct(first, *b, stream);
}

// statistics slots per (row, head) of the workspace: one per 4096-element chunk + one for the combined (max, sum) + one
// whose first word is the (row, head)'s ticket (zero before the first launch, zero again after every launch)
extern "C" int svk_deltakv_token_scores_chunks(int32_t length) {
  return (length <= 0 ? 1 : (length + svk::kTokenScoreChunk - 1) / svk::kTokenScoreChunk) + 2;
}

extern "C" int svk_deltakv_token_scores(const SvkDeltakvTokenScoresArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr && a->workspace != nullptr, SVK_ERR_VALUE, "svk_deltakv_token_scores: null args/workspace");
  SVK_REQUIRE(a->candidate_start >= 0 && a->candidate_start <= a->length, SVK_ERR_VALUE,
              "candidate_start must be within score length; got %d for L=%d.", a->candidate_start, a->length);
  if (a->batch <= 0 || a->length <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nchunk = svk_deltakv_token_scores_chunks(a->length) - 2;
  hipLaunchKernelGGL(token_score_stats_kernel, dim3(a->batch, a->num_heads, nchunk), dim3(256), 0, s, *a, nchunk);
  hipLaunchKernelGGL(token_score_final_kernel, dim3((a->length + 511) / 512, a->batch), dim3(256),
                     sizeof(float) * 2 * a->num_heads, s, *a, nchunk);
  return check_launch("svk_deltakv_token_scores");
}

extern "C" int64_t svk_topk_sorted_workspace_bytes(int32_t rows, int32_t n, int32_t k) {
  (void)k;
  if (n <= svk::kTopkStage) return 0;
  const int nwg = (n + svk::kTopkHistChunk - 1) / svk::kTopkHistChunk;
  const int64_t one = svk::topk_hist_bytes(rows, nwg), two = svk::topk2_bytes(rows, nwg);
  return one > two ? one : two;
}

extern "C" int svk_topk_sorted_desc(const SvkTopkSortedArgs* a, void* workspace, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_topk_sorted_desc: null args");
  SVK_REQUIRE(a->k >= 0 && a->k <= a->n, SVK_ERR_VALUE, "svk_topk_sorted_desc: k %d out of range (n=%d)", a->k, a->n);
  SVK_REQUIRE(a->k <= 4096, SVK_ERR_VALUE, "svk_topk_sorted_desc: k %d > 4096 unsupported", a->k);
  if (a->rows <= 0 || a->k == 0) return SVK_OK;
  int kpad = 2;
  while (kpad < a->k) kpad <<= 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr_set = true;
  }
  if (a->n > kTopkStage && workspace != nullptr) {
    const int nwg = (a->n + kTopkHistChunk - 1) / kTopkHistChunk;
    static bool final_attr = false;
    if (!final_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      final_attr = true;
    }
    const int prefix_ints = (nwg + 2) & ~1;
    const int cap = kTopkStage;
    const size_t final_lds = sizeof(unsigned long long) * kpad + sizeof(int) * (size_t)prefix_ints + sizeof(uint32_t) * (size_t)cap;
    // the rank launch keeps its LDS small (four workgroups per CU: 256 short workgroups per row): its fallback - more than
    // kTopkRankCap candidates, i.e. thousands of keys that tie on 24 bits - selects with the keys re-read from memory
    // (lds_cap = 0) instead of staged
    const size_t rank_sort = sizeof(unsigned long long) * kpad + sizeof(int) * (size_t)(prefix_ints + nwg + 2);   // (+ the count stage)
    const size_t rank_lds = rank_sort > sizeof(unsigned long long) * (kTopkRankCap + 2) ? rank_sort : sizeof(unsigned long long) * (kTopkRankCap + 2);
    const int plan = [] {                              // 2: two-level + rank (default); 1: one-level + rank; 0: one-level + select
      const char* e = getenv("SVK_TOPK_PLAN");
      const char* f = getenv("SVK_TOPK_FINAL");
      if (f != nullptr && strcmp(f, "select") == 0) return 0;
      return (e != nullptr && strcmp(e, "hist") == 0) ? 1 : 2;
    }();
    if (plan == 2) {
      // (the level-1 histograms - the first rows x 16 KB of the workspace - are zero on entry and zero again on exit)
      hipLaunchKernelGGL(topk2_hist1_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      hipLaunchKernelGGL(topk2_split_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      hipLaunchKernelGGL(topk2_refine_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      hipLaunchKernelGGL(topk_rank_kernel, dim3(kTopkRankCap / kTopkRankPerWg, a->rows), dim3(256), rank_lds, s, *a, kpad, workspace, nwg, 0, 1);
    } else {
      hipLaunchKernelGGL(topk_prep_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      hipLaunchKernelGGL(topk_hist_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      hipLaunchKernelGGL(topk_collect_kernel, dim3(nwg, a->rows), dim3(1024), 0, s, *a, workspace, nwg);
      if (plan == 0) {
        hipLaunchKernelGGL(topk_final_kernel, dim3(a->rows), dim3(1024), final_lds, s, *a, kpad, workspace, nwg, cap);
      } else {
        hipLaunchKernelGGL(topk_rank_kernel, dim3(kTopkRankCap / kTopkRankPerWg, a->rows), dim3(256), rank_lds, s, *a, kpad, workspace, nwg, 0, 0);
      }
    }
  } else if (a->n <= kTopkStage) {
    hipLaunchKernelGGL(topk_stage_kernel, dim3(1, a->rows), dim3(a->n > 2048 ? 1024 : 256),
                       sizeof(unsigned long long) * kpad + sizeof(uint32_t) * (size_t)a->n, s, *a, kpad, a->n, 1,
                       static_cast<unsigned long long*>(nullptr));
  } else {
    hipLaunchKernelGGL(topk_sorted_kernel, dim3(a->rows), dim3((kpad >= 2048 || a->n > 8192) ? 1024 : 256),
                       sizeof(unsigned long long) * kpad, s, *a, kpad);
  }
  return check_launch("svk_topk_sorted_desc");
}
