// DeltaKV residual load, first half of `compress_up` fused with the latent dequantisation (gfx950):
//   h = gelu(bf16(dequant_int4(latent[row_index]) . W1^T + b1))         [rows, N] bf16
// replacing three launches of the reference flow (deltakv_less_memory.py:2841-2848 `_load_residual`):
// triton_dequantize_2d_int4_grouped (quant.py:160-216), nn.Linear and nn.GELU of utils/compressor.py:69-73.
// MFMA-bound: rows x K x N x 2 flop (2.1 GFLOP at 2048 x 256 x 2048) against 1.3 MB of operands; the library GEMM this
// replaces ran at 17 us for this skinny-K shape, plus 5 us each for the dequant and GELU launches.
//
// Workgroup = 128 rows x 64 output features, 4 waves of 64 rows x 32 features.  The A tile (the workgroup's 128 latent rows) is
// dequantised ONCE into LDS as bf16 with the reference's two roundings (q*scale, +min) and the bf16 cast of the
// dequant output; the MFMA loop computes C^T = W1 . x^T (A operand = 16 weight rows straight from global/L2, B operand
// = 16 latent rows from LDS), so a lane's 4 accumulator rows are 4 consecutive output features of one token: bias,
// bf16 rounding of the linear output, erf-GELU in fp32 (torch's formula and operation order) and 8-byte stores.

#include "svk_common.hpp"

namespace svk {
namespace {

constexpr int kBM = 128, kBN = 64;       // 2 workgroups per CU: one's erf epilogue runs under the other's staging / MFMAs
constexpr int kNI = kBN / 32;           // 16-feature MFMA tiles per wave

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2m_t;
typedef __attribute__((ext_vector_type(2))) float f32x2m_t;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const f32x2m_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2m_t));      // v_cvt_pk_bf16_f32 (RNE)
}

__device__ __forceinline__ float load_param(const void* p, int64_t i, int dtype) {
  if (dtype == SVK_DTYPE_F32) return reinterpret_cast<const float*>(p)[i];
  if (dtype == SVK_DTYPE_BF16) return __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16);
  return (float)reinterpret_cast<const _Float16*>(p)[i];
}

// erf for the GELU epilogue: 1 - erfc(|x|) with erfc(x) = t exp(-x^2 + P(t)), t = 1 / (1 + x/2) (the Chebyshev fit of
// Numerical Recipes' erfcc, fractional error of erfc < 1.2e-7 before fp32 evaluation; measured |erf error| < 4e-7).  GELU
// only uses 1 + erf, rounded to bf16 afterwards: over 4 M normal(0, 1.5) bf16 inputs the bf16 GELU outputs equal those of
// a correctly rounded fp32 erf on every element (tests/test_gpu_deltakv.py keeps the comparison with the double-precision
// formula).  One v_rcp_f32, one v_exp_f32 and 11 FMAs instead of the ~90 instructions of the library erff with its two
// divergent branches: the epilogue was 3/4 of this kernel's time.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, ax, 1.0f));
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  const float ec = t * __expf(fmaf(-ax, ax, p));
  return copysignf(1.0f - ec, x);
}

// The GELU of a bf16-rounded Linear output is a function of 16 bits.  Kernels whose threads each evaluate a few hundred
// elements build a table of it in LDS once - with the very formula above, so every output is bit-identical to the
// arithmetic form - for the exponents 2^-16 .. 2^4 of both signs (5120 entries, 10 KB) and gather from it: ~6 integer
// instructions and one ds_read_u16 per element instead of ~22 vector instructions with two transcendentals.  Values outside
// the table (|x| < 2^-16, |x| >= 16, inf / nan) take the arithmetic form on the spot.
constexpr int kGeluE0 = 111, kGeluE1 = 131;                     // exponent fields [E0, E1): 2^-16 .. 2^4
constexpr int kGeluHalf = (kGeluE1 - kGeluE0) * 128, kGeluN = 2 * kGeluHalf;

__device__ __forceinline__ float gelu_erf(float y) {
  return mul_rn(mul_rn(y, 0.5f), add_rn(1.0f, erf_fast(mul_rn(y, 0.70710678118654752440f))));
}

__device__ __forceinline__ void gelu_table_build(uint16_t* tab, int tid, int nt) {
  for (int i = tid; i < kGeluN; i += nt) {
    const uint32_t neg = i >= kGeluHalf;
    const uint32_t bits = (uint32_t)(i - (int)neg * kGeluHalf + (kGeluE0 << 7)) | (neg << 15);
    tab[i] = (uint16_t)pack2_bf16(gelu_erf(__builtin_bit_cast(float, bits << 16)), 0.f);
  }
}

// bf16 pair (low | high << 16) -> bf16 pair of their GELUs
__device__ __forceinline__ uint32_t gelu_table_pair(const uint16_t* tab, uint32_t pair) {
  const uint32_t u0 = pair & 0xffffu, u1 = pair >> 16;
  const uint32_t i0 = (u0 & 0x7fffu) - (uint32_t)(kGeluE0 << 7), i1 = (u1 & 0x7fffu) - (uint32_t)(kGeluE0 << 7);
  const bool in0 = i0 < (uint32_t)kGeluHalf, in1 = i1 < (uint32_t)kGeluHalf;
  uint32_t g0 = tab[in0 ? i0 + (u0 >> 15) * kGeluHalf : 0u];
  uint32_t g1 = tab[in1 ? i1 + (u1 >> 15) * kGeluHalf : 0u];
  if (__builtin_expect(!(in0 && in1), 0)) {
    if (!in0) g0 = pack2_bf16(gelu_erf(__builtin_bit_cast(float, u0 << 16)), 0.f) & 0xffffu;
    if (!in1) g1 = pack2_bf16(gelu_erf(__builtin_bit_cast(float, u1 << 16)), 0.f) & 0xffffu;
  }
  return g0 | (g1 << 16);
}

template <bool GELU, int KT>
__global__ void __launch_bounds__(256) dequant_linear_act_kernel(const SvkDequantLinearArgs a_in, const SvkDequantLinearBatch lb) {
  extern __shared__ __attribute__((aligned(16))) uint16_t xs[];      // [kBM][K + 8] bf16
  SvkDequantLinearArgs a = a_in;
  if (gridDim.z > 1) {                  // layer z of a batched launch: every per-layer tensor advances by its stride
    const int64_t z = blockIdx.z;
    a.packed += z * lb.packed_stride_batch;
    const int64_t es = a.scale_dtype == SVK_DTYPE_F32 ? 4 : 2;
    a.scale = reinterpret_cast<const char*>(a.scale) + z * lb.scale_stride_batch * es;
    a.mn = reinterpret_cast<const char*>(a.mn) + z * lb.scale_stride_batch * es;
    a.weight += z * lb.weight_stride_batch;
    if (a.bias != nullptr) a.bias += z * lb.bias_stride_batch;
    a.out += z * lb.out_stride_batch;
  }
  const int K = KT > 0 ? KT : a.k, ldx = K + 8;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w & 1, wm = w >> 1;
  const int fr = lane & 15, kc = lane >> 4;
  const uint16_t* wrow[kNI];
#pragma unroll
  for (int i = 0; i < kNI; ++i) {
    const int n = min(n0 + wn * (kBN / 2) + i * 16 + fr, a.n - 1);           // clamped: rows past N are never stored
    wrow[i] = a.weight + (int64_t)n * a.weight_stride + kc * 8;
  }
  // compile-time K: every weight fragment of the wave is requested before the dequantisation below, so the L2 round
  // trips of the A operand run under the staging instead of once per k-step
  constexpr int KS = KT > 0 ? KT / 32 : 1;
  uint4 wf[KS][kNI];
  if constexpr (KT > 0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < kNI; ++i) wf[ks][i] = *reinterpret_cast<const uint4*>(wrow[i] + ks * 32);
  }
  // ---- dequantise the tile's rows into LDS: two threads per row, each a contiguous half row; the row index first, then
  //      every code / scale / min load of the thread in one batch (two dependent round trips in all)
  {
    const int r = tid >> 1, half = tid & 1;
    const int row = m0 + r;
    const int wpt = K / 16;                                           // int4 words per thread (K/8 per row, 2 threads)
    uint16_t* dst = xs + r * ldx + half * (K / 2);
    if (row < a.rows) {
      const int64_t src = a.row_index ? max(a.row_index[row], 0) : row;
      const int32_t* pw = a.packed + src * a.packed_stride + half * wpt;
      const int64_t sb = src * a.scale_stride;
      auto dequant4 = [&](const uint4 w4, int w0) {                   // 4 words starting at word w0 of the half row
        const uint32_t wd[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int g = ((half * wpt + w0 + u) * 8) / a.group_size;
          const float sc = load_param(a.scale, sb + g, a.scale_dtype), mn = load_param(a.mn, sb + g, a.scale_dtype);
          uint32_t p[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q0 = (float)((wd[u] >> (8 * e)) & 15u), q1 = (float)((wd[u] >> (8 * e + 4)) & 15u);
            p[e] = pack2_bf16(add_rn(mul_rn(q0, sc), mn), add_rn(mul_rn(q1, sc), mn));
          }
          *reinterpret_cast<uint4*>(dst + (w0 + u) * 8) = make_uint4(p[0], p[1], p[2], p[3]);
        }
      };
      if constexpr (KT > 0) {
        constexpr int NQ = KT / 64;                                   // 16-byte code loads per thread
        uint4 cw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) cw[q] = *reinterpret_cast<const uint4*>(pw + q * 4);
        if (a.scale_dtype == SVK_DTYPE_BF16 && a.group_size == 32 && (a.scale_stride % 4) == 0 &&
            (reinterpret_cast<uintptr_t>(a.scale) % 8) == 0 && (reinterpret_cast<uintptr_t>(a.mn) % 8) == 0) {
          // group 32 = 4 words = one 16-byte code load: the half row's NQ scales / mins are one short vector each
          // (behind the dtype switch of load_param every parameter load was a round trip of its own: 12 us)
          static_assert(NQ == 4, "KT == 256");
          const uint2 sv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.scale) + sb + half * NQ);
          const uint2 mv = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.mn) + sb + half * NQ);
          const float scs[4] = {bf16_lo(sv.x), bf16_hi(sv.x), bf16_lo(sv.y), bf16_hi(sv.y)};
          const float mns[4] = {bf16_lo(mv.x), bf16_hi(mv.x), bf16_lo(mv.y), bf16_hi(mv.y)};
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const uint32_t wd[4] = {cw[q].x, cw[q].y, cw[q].z, cw[q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              uint32_t p[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float q0 = (float)((wd[u] >> (8 * e)) & 15u), q1 = (float)((wd[u] >> (8 * e + 4)) & 15u);
                p[e] = pack2_bf16(add_rn(mul_rn(q0, scs[q]), mns[q]), add_rn(mul_rn(q1, scs[q]), mns[q]));
              }
              *reinterpret_cast<uint4*>(dst + (q * 4 + u) * 8) = make_uint4(p[0], p[1], p[2], p[3]);
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < NQ; ++q) dequant4(cw[q], q * 4);
        }
      } else {
        const bool vec = (wpt % 4) == 0 && (a.packed_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a.packed) % 16) == 0;
        const int nq = vec ? wpt / 4 : 0;
        for (int q = 0; q < nq; ++q) dequant4(*reinterpret_cast<const uint4*>(pw + q * 4), q * 4);
        for (int u = nq * 4; u < wpt; ++u) {                           // unaligned rows / K not a multiple of 64
          const uint32_t wd = (uint32_t)pw[u];
          const int g = ((half * wpt + u) * 8) / a.group_size;
          const float sc = load_param(a.scale, sb + g, a.scale_dtype), mn = load_param(a.mn, sb + g, a.scale_dtype);
          uint32_t p[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q0 = (float)((wd >> (8 * e)) & 15u), q1 = (float)((wd >> (8 * e + 4)) & 15u);
            p[e] = pack2_bf16(add_rn(mul_rn(q0, sc), mn), add_rn(mul_rn(q1, sc), mn));
          }
          *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(p[0], p[1], p[2], p[3]);
        }
      }
    } else {
      for (int u = 0; u < wpt; ++u) *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  // ---- C^T tile of this wave: features n0 + wn*64 .. +64 (MFMA rows), tokens m0 + wm*64 .. +64 (MFMA columns)
  f32x4_t acc[kNI][4];                                                // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < kNI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const uint16_t* xrow = xs + (wm * 64 + fr) * ldx + kc * 8;
  if constexpr (KT > 0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xrow + j * 16 * ldx + ks * 32));
#pragma unroll
      for (int i = 0; i < kNI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[ks][i]), bfr[j], acc[i][j], 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      bf16x8_t af[kNI], bfr[4];
#pragma unroll
      for (int i = 0; i < kNI; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wrow[i] + k0));
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xrow + j * 16 * ldx + k0));
#pragma unroll
      for (int i = 0; i < kNI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- epilogue: lane (col = token fr of tile j, rows = features (lane>>4)*4 + r of tile i).  The bf16 results go
  //      through LDS (the A tile is dead by now) so that the global stores are full 128-byte row segments
  __syncthreads();
  constexpr int ldy = kBN + 8;                                        // halfwords per staged output row
  uint16_t* ys = xs;
#pragma unroll
  for (int i = 0; i < kNI; ++i) {
    const int nl = wn * (kBN / 2) + i * 16 + kc * 4;                  // feature within the tile
    const int n = n0 + nl;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.n) bias[r] = __builtin_bit_cast(float, (uint32_t)a.bias[n + r] << 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = bf16_round(acc[i][j][r] + bias[r]);                 // the Linear's bf16 output
        if (GELU) {
          // aten/src/ATen/native/cuda/ActivationGeluKernel.cu (approximate='none'), computed in fp32
          v = mul_rn(mul_rn(v, 0.5f), add_rn(1.0f, erf_fast(mul_rn(v, 0.70710678118654752440f))));
        }
        y[r] = v;
      }
      *reinterpret_cast<uint2*>(ys + (wm * 64 + j * 16 + fr) * ldy + nl) = make_uint2(pack2_bf16(y[0], y[1]), pack2_bf16(y[2], y[3]));
    }
  }
  __syncthreads();
  const bool vec_out = (a.out_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a.out) % 16) == 0;
  for (int c = tid; c < kBM * (kBN / 8); c += 256) {
    const int r = c / (kBN / 8), seg = c % (kBN / 8);
    const int row = m0 + r, n = n0 + seg * 8;
    if (row >= a.rows || n >= a.n) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(ys + r * ldy + seg * 8);
    uint16_t* o = a.out + (int64_t)row * a.out_stride + n;
    if (vec_out && n + 8 <= a.n) {
      *reinterpret_cast<uint4*>(o) = v;
    } else {
      const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
      for (int e = 0; e < 8 && n + e < a.n; ++e) o[e] = (uint16_t)(wd[e >> 1] >> ((e & 1) * 16));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K = 256, group 32, bf16 scales (the published compressors): 64 rows x 256 features per workgroup.  The kernel above
// gives every 64-feature tile its own workgroup, so a launch of [3 layers, 2048 rows, 2048 features] is 1536 short
// workgroups in three rounds, each paying the row-index -> codes round trips and dequantising the same 128 latent rows
// 32 times over (39 us for 6.4 GFLOP).  Here the A tile is dequantised once per 256 features, the weights of the next
// 64-feature sub-tile are fetched under the MFMAs / GELU epilogue of the current one, and the 768 workgroups of that
// launch are resident together (43 KiB of LDS, three per CU).
// ------------------------------------------------------------------------------------------------
// A tile in LDS: rows of 512 + 32 bytes.  The 16 lanes of every `ds_read_b128` lane group of an operand read ({0-3, 12-15,
// 20-27}, ...: row fr = lane & 15, 16-byte column kc = lane >> 4) then sit on 16 different bank quads; at 512 + 16 bytes
// one pair of every group shared a quad, which doubles the cycles of the read (54 % of this kernel's LDS cycles were
// conflict cycles; 19 % now, what is left are the 16-byte stores of the dequantisation).  53 248 B per workgroup: still
// three per CU, which is what keeps the 768 workgroups of a 2048-row launch resident together.
constexpr int kM2 = 64, kSub = 64, kNSub = 4, kLdx2 = 256 + 16, kLdy2 = kSub + 8;

template <bool GELU, bool TABLE>
__global__ void __launch_bounds__(256) dequant_linear_act_k256_kernel(const SvkDequantLinearArgs a_in, const SvkDequantLinearBatch lb) {
  extern __shared__ __attribute__((aligned(16))) uint16_t xs[];      // [kM2][kLdx2] bf16 | 2 x [kM2][kLdy2] bf16 | GELU table
  SvkDequantLinearArgs a = a_in;
  if (gridDim.z > 1) {
    const int64_t z = blockIdx.z;
    a.packed += z * lb.packed_stride_batch;
    a.scale = reinterpret_cast<const uint16_t*>(a.scale) + z * lb.scale_stride_batch;
    a.mn = reinterpret_cast<const uint16_t*>(a.mn) + z * lb.scale_stride_batch;
    a.weight += z * lb.weight_stride_batch;
    if (a.bias != nullptr) a.bias += z * lb.bias_stride_batch;
    a.out += z * lb.out_stride_batch;
  }
  constexpr int K = 256, KS = K / 32;
  const int m0 = blockIdx.y * kM2, nb0 = blockIdx.x * (kSub * kNSub);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, kc = lane >> 4;
  uint16_t* ys = xs + kM2 * kLdx2;
  [[maybe_unused]] uint16_t* gtab = ys + 2 * kM2 * kLdy2;
  auto load_w = [&](int ns, uint4 (&wf)[KS]) {
    const int n = min(nb0 + ns * kSub + w * 16 + fr, a.n - 1);            // clamped: features past N are never stored
    const uint16_t* wr = a.weight + (int64_t)n * a.weight_stride + kc * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(wr + ks * 32);
  };
  uint4 wa[KS], wb[KS];
  load_w(0, wa);
  // ---- dequantise 64 rows: four threads per row, 64 codes (two groups) each
  {
    const int r = tid >> 2, qd = tid & 3;
    const int row = m0 + r;
    uint16_t* dst = xs + r * kLdx2 + qd * 64;
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
    uint32_t sv = 0, mv = 0;
    if (row < a.rows) {
      const int64_t src = a.row_index ? max(a.row_index[row], 0) : row;
      const int32_t* pw = a.packed + src * a.packed_stride + qd * 8;
      c0 = *reinterpret_cast<const uint4*>(pw);
      c1 = *reinterpret_cast<const uint4*>(pw + 4);
      sv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.scale) + src * a.scale_stride + qd * 2);
      mv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.mn) + src * a.scale_stride + qd * 2);
    }
    // the GELU table is built while the row index -> codes round trips are in flight
    if constexpr (GELU && TABLE) gelu_table_build(gtab, tid, 256);
    if (row < a.rows) {
      const float scs[2] = {bf16_lo(sv), bf16_hi(sv)}, mns[2] = {bf16_lo(mv), bf16_hi(mv)};
      const uint32_t wd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float sc = scs[u >> 2], mn = mns[u >> 2];
        uint32_t p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q0 = (float)((wd[u] >> (8 * e)) & 15u), q1 = (float)((wd[u] >> (8 * e + 4)) & 15u);
          p[e] = pack2_bf16(add_rn(mul_rn(q0, sc), mn), add_rn(mul_rn(q1, sc), mn));
        }
        *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(p[0], p[1], p[2], p[3]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  const uint16_t* xrow = xs + fr * kLdx2 + kc * 8;
  const bool vec_out = (a.out_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a.out) % 16) == 0;
  auto sub_tile = [&](int ns, const uint4 (&wf)[KS]) {
    const int n0 = nb0 + ns * kSub;
    if (n0 >= a.n) return;                                     // (uniform over the workgroup)
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xrow + j * 16 * kLdx2 + ks * 32));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[ks]), bfr[j], acc[j], 0, 0, 0);
    }
    // lane (token fr of tile j, features w*16 + kc*4 + r): bias, bf16 rounding of the Linear output, GELU, bf16
    const int nl = w * 16 + kc * 4;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + nl + r < a.n) bias[r] = __builtin_bit_cast(float, (uint32_t)a.bias[n0 + nl + r] << 16);
    }
    uint16_t* yb = ys + (ns & 1) * (kM2 * kLdy2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // (bf16 rounding of the Linear output by v_cvt_pk_bf16_f32 and a shift back: see the row-walking kernel below)
      const uint32_t r01 = pack2_bf16(acc[j][0] + bias[0], acc[j][1] + bias[1]);
      const uint32_t r23 = pack2_bf16(acc[j][2] + bias[2], acc[j][3] + bias[3]);
      if constexpr (GELU && TABLE) {
        *reinterpret_cast<uint2*>(yb + (j * 16 + fr) * kLdy2 + nl) = make_uint2(gelu_table_pair(gtab, r01), gelu_table_pair(gtab, r23));
      } else {
        float y[4] = {bf16_lo(r01), bf16_hi(r01), bf16_lo(r23), bf16_hi(r23)};
        if (GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = gelu_erf(y[r]);
        }
        *reinterpret_cast<uint2*>(yb + (j * 16 + fr) * kLdy2 + nl) = make_uint2(pack2_bf16(y[0], y[1]), pack2_bf16(y[2], y[3]));
      }
    }
    __syncthreads();
    // 64 rows x 128 B: two 16-byte segments per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + u * 256;
      const int r = c >> 3, seg = c & 7;
      const int row = m0 + r, n = n0 + seg * 8;
      if (row >= a.rows || n >= a.n) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(yb + r * kLdy2 + seg * 8);
      uint16_t* o = a.out + (int64_t)row * a.out_stride + n;
      if (vec_out && n + 8 <= a.n) {
        *reinterpret_cast<uint4*>(o) = v;
      } else {
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 8 && n + e < a.n; ++e) o[e] = (uint16_t)(wd[e >> 1] >> ((e & 1) * 16));
      }
    }
  };
  // weights of sub-tile s+1 in flight under sub-tile s (two register sets, statically named)
  load_w(1, wb);
  sub_tile(0, wa);
  load_w(2, wa);
  sub_tile(1, wb);
  load_w(3, wb);
  sub_tile(2, wa);
  sub_tile(3, wb);
}

// ------------------------------------------------------------------------------------------------
// The same tile for launches with many row tiles (B >= 2 DeltaKV batches: 8192 rows x 2 layers).  Above, every
// workgroup fetches its 256 features' weights (128 KB) for ONE 64-row tile: 2048 workgroups x 128 KB = 268 MB of L2 reads
// per launch against 67 MB of output - the launch ran at the L2's delivery rate (69 us at 2 x 8192 rows, 250 TFLOP/s).
// Here a workgroup keeps the weights of all four sub-tiles in registers (128 VGPRs) and walks `tiles` row tiles
// (blockIdx.y, + gridDim.y, ...), the codes / scales of the next tile in flight under the current one: weight traffic
// divided by `tiles`, same arithmetic element for element (bit-identical outputs).
// ------------------------------------------------------------------------------------------------
template <bool GELU, bool TABLE>
__global__ void __launch_bounds__(256) dequant_linear_act_k256_rows_kernel(const SvkDequantLinearArgs a_in, const SvkDequantLinearBatch lb,
                                                                           int row_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint16_t xs[];      // [kM2][kLdx2] bf16 | 2 x [kM2][kLdy2] bf16 | GELU table
  SvkDequantLinearArgs a = a_in;
  if (gridDim.z > 1) {
    const int64_t z = blockIdx.z;
    a.packed += z * lb.packed_stride_batch;
    a.scale = reinterpret_cast<const uint16_t*>(a.scale) + z * lb.scale_stride_batch;
    a.mn = reinterpret_cast<const uint16_t*>(a.mn) + z * lb.scale_stride_batch;
    a.weight += z * lb.weight_stride_batch;
    if (a.bias != nullptr) a.bias += z * lb.bias_stride_batch;
    a.out += z * lb.out_stride_batch;
  }
  constexpr int K = 256, KS = K / 32;
  const int nb0 = blockIdx.x * (kSub * kNSub);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, kc = lane >> 4;
  uint16_t* ys = xs + kM2 * kLdx2;
  [[maybe_unused]] uint16_t* gtab = ys + 2 * kM2 * kLdy2;
  if constexpr (GELU && TABLE) gelu_table_build(gtab, threadIdx.x, 256);       // (the first tile's barrier publishes it)
  // weights of the workgroup's 256 features: lane (fr, kc) of wave w holds row nb0 + ns*64 + w*16 + fr, K chunks kc*8 + ks*32
  uint4 wf[kNSub][KS];
#pragma unroll
  for (int ns = 0; ns < kNSub; ++ns) {
    const int n = min(nb0 + ns * kSub + w * 16 + fr, a.n - 1);            // clamped: features past N are never stored
    const uint16_t* wr = a.weight + (int64_t)n * a.weight_stride + kc * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ns][ks] = *reinterpret_cast<const uint4*>(wr + ks * 32);
  }
  float bias[kNSub][4];
#pragma unroll
  for (int ns = 0; ns < kNSub; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nb0 + ns * kSub + w * 16 + kc * 4 + r;
      bias[ns][r] = (a.bias != nullptr && n < a.n) ? __builtin_bit_cast(float, (uint32_t)a.bias[n] << 16) : 0.f;
    }
  // codes / scales of one 64-row tile: four threads per row, 64 codes (two groups) each
  const int r_own = tid >> 2, qd = tid & 3;
  uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
  uint32_t sv = 0, mv = 0;
  bool live = false;
  auto fetch = [&](int tile) {
    const int row = tile * kM2 + r_own;
    live = tile < row_tiles && row < a.rows;
    if (live) {
      const int64_t src = a.row_index ? max(a.row_index[row], 0) : row;
      const int32_t* pw = a.packed + src * a.packed_stride + qd * 8;
      c0 = *reinterpret_cast<const uint4*>(pw);
      c1 = *reinterpret_cast<const uint4*>(pw + 4);
      sv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.scale) + src * a.scale_stride + qd * 2);
      mv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.mn) + src * a.scale_stride + qd * 2);
    }
  };
  const uint16_t* xrow = xs + fr * kLdx2 + kc * 8;
  const bool vec_out = (a.out_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a.out) % 16) == 0;
  fetch(blockIdx.y);
  for (int tile = blockIdx.y; tile < row_tiles; tile += gridDim.y) {
    const int m0 = tile * kM2;
    __syncthreads();                                   // the previous tile's products have read xs, its stores have read ys
    {
      uint16_t* dst = xs + r_own * kLdx2 + qd * 64;
      if (live) {
        const float scs[2] = {bf16_lo(sv), bf16_hi(sv)}, mns[2] = {bf16_lo(mv), bf16_hi(mv)};
        const uint32_t wd[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float sc = scs[u >> 2], mn = mns[u >> 2];
          uint32_t p[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q0 = (float)((wd[u] >> (8 * e)) & 15u), q1 = (float)((wd[u] >> (8 * e + 4)) & 15u);
            p[e] = pack2_bf16(add_rn(mul_rn(q0, sc), mn), add_rn(mul_rn(q1, sc), mn));
          }
          *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(p[0], p[1], p[2], p[3]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(dst + u * 8) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    __syncthreads();
    fetch(tile + gridDim.y);                           // next tile's codes travel under this tile's products and epilogues
#pragma unroll
    for (int ns = 0; ns < kNSub; ++ns) {
      const int n0 = nb0 + ns * kSub;
      if (n0 >= a.n) break;                            // (uniform over the workgroup)
      f32x4_t acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8_t bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xrow + j * 16 * kLdx2 + ks * 32));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[ns][ks]), bfr[j], acc[j], 0, 0, 0);
      }
      const int nl = w * 16 + kc * 4;
      uint16_t* yb = ys + (ns & 1) * (kM2 * kLdy2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // bf16 rounding of the Linear output by v_cvt_pk_bf16_f32 (RNE, two values per instruction) and a shift back,
        // instead of the integer form of bf16_round (eight instructions per value; same bits for finite values)
        const uint32_t r01 = pack2_bf16(acc[j][0] + bias[ns][0], acc[j][1] + bias[ns][1]);
        const uint32_t r23 = pack2_bf16(acc[j][2] + bias[ns][2], acc[j][3] + bias[ns][3]);
        if constexpr (GELU && TABLE) {
          *reinterpret_cast<uint2*>(yb + (j * 16 + fr) * kLdy2 + nl) = make_uint2(gelu_table_pair(gtab, r01), gelu_table_pair(gtab, r23));
        } else {
          float y[4] = {bf16_lo(r01), bf16_hi(r01), bf16_lo(r23), bf16_hi(r23)};
          if (GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = gelu_erf(y[r]);
          }
          *reinterpret_cast<uint2*>(yb + (j * 16 + fr) * kLdy2 + nl) = make_uint2(pack2_bf16(y[0], y[1]), pack2_bf16(y[2], y[3]));
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = tid + u * 256;
        const int r = c >> 3, seg = c & 7;
        const int row = m0 + r, n = n0 + seg * 8;
        if (row >= a.rows || n >= a.n) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(yb + r * kLdy2 + seg * 8);
        uint16_t* o = a.out + (int64_t)row * a.out_stride + n;
        if (vec_out && n + 8 <= a.n) {
          *reinterpret_cast<uint4*>(o) = v;
        } else {
          const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 8 && n + e < a.n; ++e) o[e] = (uint16_t)(wd[e >> 1] >> ((e & 1) * 16));
        }
      }
    }
  }
}

}  // namespace
}  // namespace svk

namespace svk {
namespace {
int launch_dequant_linear_act(const SvkDequantLinearArgs* a, const SvkDequantLinearBatch& lb, hipStream_t s, const char* who) {
  SVK_REQUIRE(a != nullptr && a->packed != nullptr && a->scale != nullptr && a->mn != nullptr && a->weight != nullptr && a->out != nullptr,
              SVK_ERR_VALUE, "%s: null args", who);
  SVK_REQUIRE(a->k > 0 && a->k % 32 == 0 && a->k <= 512, SVK_ERR_LAYOUT, "%s: K = %d must be a multiple of 32, <= 512", who, a->k);
  SVK_REQUIRE(a->group_size > 0 && a->group_size % 8 == 0 && a->k % a->group_size == 0, SVK_ERR_VALUE,
              "dequantization requires output_dim divisible by group_size, got output_dim=%d, group_size=%d.", a->k, a->group_size);
  SVK_REQUIRE(a->n > 0 && (a->weight_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a->weight) % 16) == 0, SVK_ERR_LAYOUT,
              "%s: weight rows must be 16-byte aligned", who);
  SVK_REQUIRE(a->activation == 0 || a->activation == 1, SVK_ERR_VALUE, "%s: activation %d (0 none, 1 erf-GELU)", who, a->activation);
  SVK_REQUIRE(lb.n_batch >= 1 && lb.n_batch <= 65535, SVK_ERR_VALUE, "%s: n_batch %d out of range", who, lb.n_batch);
  SVK_REQUIRE(lb.n_batch == 1 || (lb.weight_stride_batch % 8 == 0 && lb.packed_stride_batch % 4 == 0), SVK_ERR_LAYOUT,
              "%s: per-layer strides must keep 16-byte alignment", who);
  if (a->rows <= 0) return SVK_OK;
  // the staged A tile, reused for the transposed output tile of the epilogue (K = 32 needs the larger of the two)
  const size_t shm = sizeof(uint16_t) * kBM * (size_t)((a->k > kBN ? a->k : kBN) + 8);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dequant_linear_act_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dequant_linear_act_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dequant_linear_act_kernel<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dequant_linear_act_kernel<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    attr_set = true;
  }
  const dim3 grid((a->n + kBN - 1) / kBN, (a->rows + kBM - 1) / kBM, lb.n_batch), block(256);
  if (a->k == 256 && a->group_size == 32 && a->scale_dtype == SVK_DTYPE_BF16 && (a->packed_stride % 4) == 0 &&
      (reinterpret_cast<uintptr_t>(a->packed) % 16) == 0 && (a->scale_stride % 2) == 0 &&
      (reinterpret_cast<uintptr_t>(a->scale) % 4) == 0 && (reinterpret_cast<uintptr_t>(a->mn) % 4) == 0 &&
      (lb.n_batch == 1 || lb.scale_stride_batch % 2 == 0)) {
    const dim3 grid2((a->n + kSub * kNSub - 1) / (kSub * kNSub), (a->rows + kM2 - 1) / kM2, lb.n_batch);
    const size_t shm2 = sizeof(uint16_t) * (kM2 * kLdx2 + 2 * kM2 * kLdy2);
    // many row tiles: workgroups that keep their weights and walk several row tiles, two workgroups per CU
    static const bool rows_form = getenv("SVK_DQL_ROWS") == nullptr || atoi(getenv("SVK_DQL_ROWS")) != 0;
    const int row_tiles = (int)grid2.y;
    const int groups = max(1, 512 / (int)(grid2.x * grid2.z));
    if (rows_form && row_tiles >= 2 * groups) {
      const dim3 grid3(grid2.x, groups, grid2.z);
      // (SVK_DQL_GELU_TABLE=0: the arithmetic GELU, the A/B reference of the table form - same bits)
      static const bool gelu_table = getenv("SVK_DQL_GELU_TABLE") == nullptr || atoi(getenv("SVK_DQL_GELU_TABLE")) != 0;
      if (a->activation == 1 && gelu_table)
        hipLaunchKernelGGL((dequant_linear_act_k256_rows_kernel<true, true>), grid3, block, shm2 + sizeof(uint16_t) * kGeluN, s, *a, lb, row_tiles);
      else if (a->activation == 1) hipLaunchKernelGGL((dequant_linear_act_k256_rows_kernel<true, false>), grid3, block, shm2, s, *a, lb, row_tiles);
      else hipLaunchKernelGGL((dequant_linear_act_k256_rows_kernel<false, false>), grid3, block, shm2, s, *a, lb, row_tiles);
      return check_launch(who);
    }
    // (one row tile per workgroup = 64 elements per thread: the table form - built under the code loads' round trips,
    //  TABLE = true - measures 20.9 against 20.5 us at 2 x 2048 rows; the arithmetic form stays)
    if (a->activation == 1) hipLaunchKernelGGL((dequant_linear_act_k256_kernel<true, false>), grid2, block, shm2, s, *a, lb);
    else hipLaunchKernelGGL((dequant_linear_act_k256_kernel<false, false>), grid2, block, shm2, s, *a, lb);
    return check_launch(who);
  }
  if (a->k == 256 && (a->packed_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a->packed) % 16) == 0) {
    // the latent width of the published compressors: fully unrolled
    if (a->activation == 1) hipLaunchKernelGGL((dequant_linear_act_kernel<true, 256>), grid, block, shm, s, *a, lb);
    else hipLaunchKernelGGL((dequant_linear_act_kernel<false, 256>), grid, block, shm, s, *a, lb);
  } else {
    if (a->activation == 1) hipLaunchKernelGGL((dequant_linear_act_kernel<true, 0>), grid, block, shm, s, *a, lb);
    else hipLaunchKernelGGL((dequant_linear_act_kernel<false, 0>), grid, block, shm, s, *a, lb);
  }
  return check_launch(who);
}
}  // namespace
}  // namespace svk

extern "C" int svk_dequant_linear_act(const SvkDequantLinearArgs* a, svk_stream_t stream) {
  SvkDequantLinearBatch one = {};
  one.n_batch = 1;
  return svk::launch_dequant_linear_act(a, one, static_cast<hipStream_t>(stream), "svk_dequant_linear_act");
}

extern "C" int svk_dequant_linear_act_batched(const SvkDequantLinearArgs* first, const SvkDequantLinearBatch* b, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(b != nullptr, SVK_ERR_VALUE, "svk_dequant_linear_act_batched: null batch description");
  return launch_dequant_linear_act(first, *b, static_cast<hipStream_t>(stream), "svk_dequant_linear_act_batched");
}
