// DeltaKV compression side: the L2 ranking product, the causal mask over the block's own centres
// and the per-token top-k of `_cluster_compress` as ONE MFMA launch that never writes the [tokens, centres] score matrix
// (the reference's fused form: `_deltakv_l2_topk_block_kernel`, kernels/triton/deltakv_kernels.py:3945-4045, wrapper
// :4048-4134; merge of the per-block candidates deltakv_base.py:3352-3358).
//
// Work split.  A wave owns 16 tokens of the block and a range of 16-centre tiles.  Its 16 token rows (kv_dim = 2*Hkv*D
// bf16 values each) are read ONCE into registers as MFMA B-operand fragments (kv_dim / 32 x 16 bytes per lane: 128 VGPRs
// at kv_dim 1024); every centre tile is streamed from the layer's K and V caches BY SLOT straight into A-operand
// fragments (16 bytes per lane per 32-deep step: the four k-groups of a centre row form one 64-byte segment per load),
// two tiles in flight.  `v_mfma_f32_16x16x32_bf16` computes S^T = centres x tokens^T, so a lane ends up with four
// centres of ONE token (token = lane % 16): the running top-k of a token lives in registers of the four lanes that share
// it and is merged through LDS once, at the end.  Splits of the centre range across workgroups (so that a block of 128
// tokens still fills the chip) leave k candidates per (token, split) in a small workspace that a second launch merges.
//
// Arithmetic = the runtime's `_metric_l2` (deltakv_base.py:2168-2190) on bf16 tensors: dot accumulated in fp32 and rounded
// to bf16 (the GEMM's output dtype), times 2 (exact), minus bf16(||b||^2) with one more bf16 rounding; ||b||^2 is the fp32
// sum of the bf16-rounded squares (`(b * b).sum(dim=1, dtype=float32)`).  Order: score descending, lower centre column on
// ties - `svk_cluster_topk`'s order, which this launch replaces together with the library GEMM in front of it.

#include <stdlib.h>

#include "svk_common.hpp"

namespace svk {
namespace {

constexpr int kMaxK = 8;
constexpr int kTokTile = 16;

// (score, column) as one unsigned key: larger key = better candidate (score descending, column ascending)
__device__ __forceinline__ uint64_t cand_key(float score, int col) {
  uint32_t u = __builtin_bit_cast(uint32_t, score);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((uint64_t)u << 32) | (uint32_t)(~(uint32_t)col);
}
__device__ __forceinline__ int key_col(uint64_t key) { return (int)(~(uint32_t)key); }

template <int K>
__device__ __forceinline__ void insert(uint64_t (&best)[K], uint64_t cur) {
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint64_t b = best[j];
    const bool up = cur > b;
    best[j] = up ? cur : b;
    cur = up ? b : cur;
  }
}

// one wave per centre: ||b||^2 as torch computes it on a bf16 row - squares rounded to bf16, summed in fp32, rounded to bf16
__global__ void __launch_bounds__(256) center_norm_kernel(const SvkClusterL2TopkArgs a, float* __restrict__ center_norms) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= a.m) return;
  const int64_t slot = a.center_slots[c];
  float sum = 0.f;
  for (int i = lane * 8; i < 2 * a.half_dim; i += 64 * 8) {
    const uint16_t* src = i < a.half_dim ? a.k_cache + slot * a.kv_slot_stride + i : a.v_cache + slot * a.kv_slot_stride + (i - a.half_dim);
    const uint4 w = *reinterpret_cast<const uint4*>(src);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = bf16_lo(ws[j]), hi = bf16_hi(ws[j]);
      sum = add_rn(sum, bf16_round(mul_rn(lo, lo)));
      sum = add_rn(sum, bf16_round(mul_rn(hi, hi)));
    }
  }
  sum = wave_allsum(sum);
  if (lane == 0) center_norms[c] = bf16_round(sum);
}

template <int K>
__global__ void __launch_bounds__(64) cluster_l2_topk_kernel(const SvkClusterL2TopkArgs a, const float* __restrict__ center_norms,
                                                              uint64_t* __restrict__ partial) {
  constexpr int kMaxSteps = 32;                     // kv_dim <= 1024
  const int lane = threadIdx.x, tok = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * kTokTile + tok;
  const int row_abs = a.row_offset + row;
  const int steps = 2 * a.half_dim / 32, half_steps = a.half_dim / 32;
  __shared__ uint64_t s_best[64][K];

  bf16x8_t tf[kMaxSteps];
  {
    const uint16_t* trow = a.tokens + (int64_t)(row < a.rows ? row : a.rows - 1) * a.token_stride + g * 8;
#pragma unroll
    for (int s = 0; s < kMaxSteps; ++s)
      if (s < steps) tf[s] = *reinterpret_cast<const bf16x8_t*>(trow + s * 32);
  }
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 0;

  const int tiles = (a.m + 15) / 16;
  const int t0 = (int)((int64_t)tiles * blockIdx.y / gridDim.y), t1 = (int)((int64_t)tiles * (blockIdx.y + 1) / gridDim.y);

  auto score_tile = [&](int t, const f32x4_t acc) {
    const int c0 = t * 16 + 4 * g;
    const float4 nrm = *reinterpret_cast<const float4*>(center_norms + c0);     // (workspace padded to whole tiles)
    const float n4[4] = {nrm.x, nrm.y, nrm.z, nrm.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + r;
      if (c >= a.m) continue;
      const float dot = bf16_round(acc[r]);
      float s = bf16_round(mul_rn(dot, 2.0f) - n4[r]);
      if (c >= a.m0 && a.new_center_rel[c - a.m0] > row_abs) s = -INFINITY;
      insert<K>(best, cand_key(s, c));
    }
  };

  for (int t = t0; t < t1; t += 2) {
    const bool two = t + 1 < t1;
    const int ca = t * 16 + tok, cb = (t + 1) * 16 + tok;
    const int64_t sa = a.center_slots[ca < a.m ? ca : 0], sb = a.center_slots[(two && cb < a.m) ? cb : 0];
    const uint16_t* ka = a.k_cache + sa * a.kv_slot_stride + g * 8;
    const uint16_t* va = a.v_cache + sa * a.kv_slot_stride + g * 8;
    const uint16_t* kb = a.k_cache + sb * a.kv_slot_stride + g * 8;
    const uint16_t* vb = a.v_cache + sb * a.kv_slot_stride + g * 8;
    f32x4_t acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < kMaxSteps; s0 += 8) {
      if (s0 >= steps) break;
      bf16x8_t fa[8], fb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s = s0 + i;
        if (s < steps) {                               // (narrow rows: kv_dim below 256)
          fa[i] = *reinterpret_cast<const bf16x8_t*>(s < half_steps ? ka + s * 32 : va + (s - half_steps) * 32);
          fb[i] = *reinterpret_cast<const bf16x8_t*>(s < half_steps ? kb + s * 32 : vb + (s - half_steps) * 32);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (s0 + i < steps) {
          acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], tf[s0 + i], acc_a, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], tf[s0 + i], acc_b, 0, 0, 0);
        }
      }
    }
    score_tile(t, acc_a);
    if (two) score_tile(t + 1, acc_b);
  }

  // the four lanes of a token hold disjoint centre subsets: merge their lists
#pragma unroll
  for (int j = 0; j < K; ++j) s_best[lane][j] = best[j];
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int gg = 1; gg < 4; ++gg)
#pragma unroll
      for (int j = 0; j < K; ++j) insert<K>(best, s_best[gg * 16 + tok][j]);
    if (row < a.rows) {
      if (gridDim.y == 1) {
#pragma unroll
        for (int j = 0; j < K; ++j)
          if (j < a.k) a.topk[(int64_t)row * a.topk_stride + j] = key_col(best[j]);
      } else {
        uint64_t* dst = partial + ((int64_t)row * gridDim.y + blockIdx.y) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) dst[j] = best[j];
      }
    }
  }
}

// one wave per token row: lanes take the (token, split) candidates 64 apart into sorted local lists, then k rounds of a wave
// arg-max pop the winners in order (a thread per token walking its splits x K candidates one dependent load at a time took
// 100 us at 125 splits)
template <int K>
__global__ void __launch_bounds__(256) cluster_merge_kernel(const SvkClusterL2TopkArgs a, const uint64_t* __restrict__ partial, int splits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 0;
  const uint64_t* src = partial + (int64_t)row * splits * K;
  const int n = splits * K;
  for (int i0 = 0; i0 < n; i0 += 4 * 64) {
    uint64_t c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = i0 + u * 64 + lane < n ? src[i0 + u * 64 + lane] : 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) insert<K>(best, c[u]);
  }
  for (int j = 0; j < a.k; ++j) {
    uint64_t top = best[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(top >> 32), off, 64) << 32) | (uint32_t)__shfl_xor((int)top, off, 64);
      top = o > top ? o : top;
    }
    if (lane == 0) a.topk[(int64_t)row * a.topk_stride + j] = key_col(top);
    if (best[0] == top) {                      // (keys are unique: exactly one lane pops)
#pragma unroll
      for (int q = 0; q + 1 < K; ++q) best[q] = best[q + 1];
      best[K - 1] = 0;
    }
  }
}

// centre-range splits: about one single-wave workgroup per SIMD, at least two tiles per split
int choose_splits(int rows, int m) {
  const int groups = (rows + kTokTile - 1) / kTokTile, tiles = (m + 15) / 16;
  // measured (tools/kbench_cluster.py, 128 rows x 8000 centres): >= 2 tiles per split and ~1024 waves: 44 us; 4 tiles / 2048
  // waves 43; 1 tile / 4096 waves 72 (the 16 token rows are re-read by every split); 8 tiles 52
  static const int min_tiles = [] { const char* e = getenv("SVK_CLUSTER_SPLIT_TILES"); return e ? atoi(e) : 2; }();   // developer knobs
  static const int waves = [] { const char* e = getenv("SVK_CLUSTER_WAVES"); return e ? atoi(e) : 1024; }();
  int s = (waves + groups - 1) / groups;
  if (s > tiles / min_tiles) s = tiles / min_tiles;
  return s < 1 ? 1 : s;
}

int kernel_k(int k) { return k <= 4 ? 4 : 8; }

}  // namespace
}  // namespace svk

extern "C" int64_t svk_cluster_l2_topk_workspace_bytes(int32_t rows, int32_t m, int32_t k) {
  using namespace svk;
  if (rows <= 0 || m <= 0 || k <= 0 || k > kMaxK) return 0;
  const int64_t norms = ((int64_t)(m + 15) / 16) * 16 * sizeof(float);
  const int splits = choose_splits(rows, m);
  const int64_t partial = splits > 1 ? (int64_t)rows * splits * kernel_k(k) * sizeof(uint64_t) : 0;
  return norms + partial;
}

extern "C" int svk_cluster_l2_topk(const SvkClusterL2TopkArgs* in, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(in != nullptr, SVK_ERR_VALUE, "svk_cluster_l2_topk: null args");
  const SvkClusterL2TopkArgs a = *in;
  SVK_REQUIRE(a.k >= 1 && a.k <= kMaxK && a.k <= a.m, SVK_ERR_VALUE, "svk_cluster_l2_topk: k %d out of range (1..%d, m=%d)", a.k, kMaxK, a.m);
  SVK_REQUIRE(a.m0 >= 0 && a.m0 <= a.m, SVK_ERR_VALUE, "svk_cluster_l2_topk: m0 %d out of range (m=%d)", a.m0, a.m);
  SVK_REQUIRE(a.half_dim > 0 && a.half_dim % 32 == 0 && a.half_dim <= 512, SVK_ERR_LAYOUT,
              "svk_cluster_l2_topk: Hkv*D = %d (served: multiples of 32 up to 512; wider rows keep the library product)", a.half_dim);
  SVK_REQUIRE(a.token_stride % 8 == 0 && a.kv_slot_stride % 8 == 0, SVK_ERR_LAYOUT, "svk_cluster_l2_topk: rows must keep 16-byte alignment");
  SVK_REQUIRE(a.workspace != nullptr && a.workspace_bytes >= svk_cluster_l2_topk_workspace_bytes(a.rows, a.m, a.k), SVK_ERR_VALUE,
              "svk_cluster_l2_topk: workspace of %lld bytes is too small", (long long)a.workspace_bytes);
  if (a.rows <= 0) return SVK_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int splits = choose_splits(a.rows, a.m);
  float* norms = reinterpret_cast<float*>(a.workspace);
  uint64_t* partial = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(a.workspace) + ((int64_t)(a.m + 15) / 16) * 16 * sizeof(float));
  hipLaunchKernelGGL(center_norm_kernel, dim3((a.m + 3) / 4), dim3(256), 0, s, a, norms);
  const dim3 grid((a.rows + kTokTile - 1) / kTokTile, splits);
  if (kernel_k(a.k) == 4) hipLaunchKernelGGL(cluster_l2_topk_kernel<4>, grid, dim3(64), 0, s, a, norms, partial);
  else hipLaunchKernelGGL(cluster_l2_topk_kernel<8>, grid, dim3(64), 0, s, a, norms, partial);
  if (splits > 1) {
    if (kernel_k(a.k) == 4) hipLaunchKernelGGL(cluster_merge_kernel<4>, dim3((a.rows + 3) / 4), dim3(256), 0, s, a, partial, splits);
    else hipLaunchKernelGGL(cluster_merge_kernel<8>, dim3((a.rows + 3) / 4), dim3(256), 0, s, a, partial, splits);
  }
  return check_launch("svk_cluster_l2_topk");
}
