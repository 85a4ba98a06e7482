// Chunked-prefill causal attention over the paged slot table (gfx950).
//
// Same operand plumbing as decode stage 1 v3, with the roles of the MFMA rows changed from "query heads of one token"
// to "32 consecutive query tokens of one head":
//   workgroup = (32 query tokens, one KV head, one sequence); wave w = query head kvh*G + w, so the G waves of a
//   workgroup walk the same key tiles and share every K/V line in the vector L1;
//   Q.K^T : A = the wave's 32 queries (two 16-row tiles, fragments kept in registers), B = 16 key rows per column
//           group loaded straight from HBM/L2 in the B-operand layout;
//   softmax: base-2 online softmax per query row (DPP row reductions across the 16 key columns of a group);
//   P.V   : P (bf16) goes through a 2.5 KiB per-wave LDS tile [query][key]; V is read as 16-byte segments in the
//           B-operand token order and MFMA i takes head dim n*8+i as its column (byte permutes), so a lane's
//           accumulator is 8 consecutive head dims of 4 query rows -> 16-byte bf16 output stores.

#include <type_traits>

#include "svk_common.hpp"

namespace svk {
namespace {

constexpr int kQTile = 32;      // query tokens per wave
constexpr int kKTile = 32;      // keys per iteration
constexpr int kPRowP = 40;      // P tile row stride (bf16), 16-byte aligned rows

typedef __attribute__((ext_vector_type(2))) __bf16 pa_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float pa_f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16_pair(float lo, float hi) {
  const pa_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pa_bf16x2_t));      // v_cvt_pk_bf16_f32 (RNE)
}

template <int D, bool OFF32>
__global__ void __launch_bounds__(512) context_attention_kernel(const SvkContextAttentionArgs a) {
  constexpr int NC = D / 32, DW = D / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = a.num_q_heads / a.num_kv_heads;
  const int b = blockIdx.z, kvh = blockIdx.y;
  const int head = kvh * G + w;
  const int n = lane & 15, kc = lane >> 4;
  const int dg = n % DW;
  const int pc = a.b_prompt_cache_len[b];
  const int q_len = a.b_seq_len[b] - pc;                     // queries of this chunk
  const int m0 = blockIdx.x * kQTile;
  if (m0 >= q_len) return;
  const int start_loc = a.b_start_loc[b];
  const int kv_end = min(m0 + kQTile + pc, q_len + pc);       // keys visible to the last query of the block
  // per-wave LDS: P tile [32][kPRowP] bf16 | 32 slot ids
  uint16_t* Pl = reinterpret_cast<uint16_t*>(lds_raw + (size_t)w * (kQTile * kPRowP * 2 + 256));
  int* slot_lds = reinterpret_cast<int*>(Pl + kQTile * kPRowP);
  const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;

  // Q fragments: lane (m = n, kc) of tile t holds Q[m0 + t*16 + n][head][c*32 + kc*8 .. +8]
  bf16x8_t qa[2][NC];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qi = m0 + t * 16 + n;
    const uint16_t* qp = a.q + (int64_t)(start_loc + min(qi, q_len - 1)) * a.q_stride_t + (int64_t)head * a.q_stride_h + kc * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) qa[t][c] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qp + c * 32));
  }
  const float sm_scale = rsqrtf((float)D) * 1.4426950408889634f;
  const float mask_raw = -1.0e8f / sm_scale;
  float m[2][4], l[2][4];
  f32x4_t acc[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[t][r] = -INFINITY; l[t][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // Row addressing.  OFF32: the K (V) tensor spans < 4 GiB, so a row address is the tensor base + a 32-bit byte
  // offset: one v_mul_lo + one add per row instead of 64-bit multiply-adds (10 row addresses per key tile).
  const char* const kt = reinterpret_cast<const char*>(a.k_cache);
  const char* const vt = reinterpret_cast<const char*>(a.v_cache);
  const int64_t slot_bytes = a.kv_slot_stride * 2;
  const int64_t k_lane_bytes = ((int64_t)kvh * a.kv_head_stride + kc * 8) * 2;
  const int64_t v_lane_bytes = ((int64_t)kvh * a.kv_head_stride + dg * 8) * 2;
  auto k_ptr = [&](int slot) -> const uint16_t* {
    if (OFF32) return reinterpret_cast<const uint16_t*>(kt + (size_t)((uint32_t)slot * (uint32_t)slot_bytes + (uint32_t)k_lane_bytes));
    return reinterpret_cast<const uint16_t*>(kt + (int64_t)slot * slot_bytes + k_lane_bytes);
  };
  auto v_ptr = [&](int slot) -> const uint16_t* {
    if (OFF32) return reinterpret_cast<const uint16_t*>(vt + (size_t)((uint32_t)slot * (uint32_t)slot_bytes + (uint32_t)v_lane_bytes));
    return reinterpret_cast<const uint16_t*>(vt + (int64_t)slot * slot_bytes + v_lane_bytes);
  };
  uint32_t* Pl32 = reinterpret_cast<uint32_t*>(Pl);
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // slot ids of one key tile: lanes 0..31 read row[k0 + lane], clamped to the last visible key (always a legal row)
  auto fetch_slots = [&](int k0) -> int { return row[min(k0 + (lane & 31), kv_end - 1)]; };
  // The k index j of the second product maps to key column (j & 1) * 16 + (j >> 1), so the two probabilities a lane
  // holds for one query row (columns n and 16 + n) are neighbours in the P tile: one packed 32-bit LDS store per row.
  const int vcol0 = kc * 4;                          // V row of k index kc*8 + e: column (e & 1) * 16 + kc*4 + (e >> 1)

  // ---- prologue: tile 0 slot ids -> LDS, tile 1 ids parked in a register, K(0) in flight
  if (lane < kKTile) slot_lds[lane] = fetch_slots(0);
  int s_next = fetch_slots(kKTile);
  wave_sync();
  uint4 kr[2][NC];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const uint16_t* kp = k_ptr(slot_lds[g * 16 + n]);
#pragma unroll
    for (int c = 0; c < NC; ++c) kr[g][c] = *reinterpret_cast<const uint4*>(kp + c * 32);
  }
  int k0 = 0, buf = 0;
  // one key tile; the last one is peeled (compile-time flag) so the K(i+1) re-arm is straight-line code
  auto tile = [&](auto has_next_c) {
    constexpr bool has_next = decltype(has_next_c)::value;
    const int* cur = slot_lds + buf * 32;
    int* nxt = slot_lds + (buf ^ 1) * 32;
    // ---- V(i) loads; publish tile i+1's ids, fetch tile i+2's
    uint4 vr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      vr[e] = *reinterpret_cast<const uint4*>(v_ptr(cur[(e & 1) * 16 + vcol0 + (e >> 1)]));
    if (lane < kKTile) nxt[lane] = s_next;
    s_next = fetch_slots(k0 + 2 * kKTile);
    // ---- S = Q K^T for both query tiles, then the K registers are dead: re-arm them with K(i+1)
    f32x4_t s[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        s[t][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[t][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[t][c], __builtin_bit_cast(bf16x8_t, kr[g][c]), s[t][g], 0, 0, 0);
      }
    wave_sync();
    if constexpr (has_next) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint16_t* kp = k_ptr(nxt[g * 16 + n]);
#pragma unroll
        for (int c = 0; c < NC; ++c) kr[g][c] = *reinterpret_cast<const uint4*>(kp + c * 32);
      }
    }
    // ---- mask + base-2 online softmax (rows = queries m0 + t*16 + kc*4 + r, columns = keys k0 + g*16 + n)
    const bool diag = k0 + kKTile > m0 + pc;          // only tiles touching the diagonal / the end need the mask
    bool rescale = false;
    float alpha[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = m0 + t * 16 + kc * 4 + r;
        // the running maximum lives in the raw-logit domain (sm_scale > 0 keeps the order): p = exp2(s*scale - m*scale)
        // is one fma per element instead of a multiply and a subtract; masked logits sit at -1e8 / scale like the
        // reference's -1e8 after scaling
        float x0 = s[t][0][r], x1 = s[t][1][r];
        if (diag) {
          const int key = k0 + n;
          if (!(key <= qrow + pc && key < kv_end)) x0 = mask_raw;
          if (!(key + 16 <= qrow + pc && key + 16 < kv_end)) x1 = mask_raw;
        }
        const float nm = vmax(m[t][r], row16_allmax(vmax(x0, x1)));
        const float nms = nm * sm_scale;
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x0, sm_scale, -nms));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x1, sm_scale, -nms));
        float al = 1.0f;
        if (nm != m[t][r]) {                                // the rescale factor only when the row maximum moved
          al = __builtin_amdgcn_exp2f(m[t][r] * sm_scale - nms);
          rescale = true;
        }
        alpha[t][r] = al;
        l[t][r] = l[t][r] * al + (p0 + p1);               // lane-partial row sum: reduced across the 16 columns once, in the epilogue
        m[t][r] = nm;
        Pl32[(t * 16 + kc * 4 + r) * (kPRowP / 2) + n] = pack_bf16_pair(p0, p1);
      }
    if (__any(rescale)) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][i][r] *= alpha[t][r];
    }
    wave_sync();
    // ---- O += P V
    {
      bf16x8_t pf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        pf[t] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Pl + (t * 16 + n) * kPRowP + kc * 8));
      const uint32_t* vv = reinterpret_cast<const uint32_t*>(vr);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t vf[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          vf[e2] = __builtin_amdgcn_perm(vv[(2 * e2 + 1) * 4 + i / 2], vv[(2 * e2) * 4 + i / 2], (i & 1) ? 0x07060302u : 0x05040100u);
        const bf16x8_t vb = __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3]));
        acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[0], vb, acc[0][i], 0, 0, 0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[1], vb, acc[1][i], 0, 0, 0);
      }
    }
    wave_sync();
    k0 += kKTile;
    buf ^= 1;
  };
  while (k0 + kKTile < kv_end) tile(std::true_type{});
  tile(std::false_type{});
  // ---- epilogue: lane (n, kc) owns query rows t*16 + kc*4 + r and head dims dg*8 .. +8
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) l[t][r] = row16_allsum(l[t][r]);       // all 64 lanes take part in the DPP reduction
  if (n < DW) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = m0 + t * 16 + kc * 4 + r;
        if (qrow < q_len) {
          const float inv = 1.0f / l[t][r];
          uint32_t ow[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            ow[e2] = f32_to_bf16_bits(acc[t][2 * e2][r] * inv) | (f32_to_bf16_bits(acc[t][2 * e2 + 1][r] * inv) << 16);
          *reinterpret_cast<uint4*>(a.o + (int64_t)(start_loc + qrow) * a.o_stride_t + (int64_t)head * a.o_stride_h + dg * 8) =
              make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
      }
  }
}

}  // namespace
}  // namespace svk

extern "C" int svk_context_attention_fwd(const SvkContextAttentionArgs* a, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(a != nullptr, SVK_ERR_VALUE, "svk_context_attention_fwd: null args");
  SVK_REQUIRE(a->head_dim == 64 || a->head_dim == 128, SVK_ERR_LAYOUT, "svk_context_attention_fwd: head_dim %d unsupported (64, 128)", a->head_dim);
  SVK_REQUIRE(a->num_kv_heads >= 1 && a->num_q_heads % a->num_kv_heads == 0, SVK_ERR_LAYOUT,
              "svk_context_attention_fwd: q heads %d not divisible by kv heads %d", a->num_q_heads, a->num_kv_heads);
  const int G = a->num_q_heads / a->num_kv_heads;
  SVK_REQUIRE(G >= 1 && G <= 8, SVK_ERR_LAYOUT, "svk_context_attention_fwd: GQA group size %d unsupported (1..8)", G);
  SVK_REQUIRE((a->q_stride_t % 8) == 0 && (a->q_stride_h % 8) == 0 && (a->o_stride_t % 8) == 0 && (a->o_stride_h % 8) == 0 &&
                  (a->kv_slot_stride % 8) == 0 && (a->kv_head_stride % 8) == 0,
              SVK_ERR_LAYOUT, "svk_context_attention_fwd: q/k/v/o strides must keep 16-byte alignment");
  if (a->batch <= 0 || a->max_input_len <= 0) return SVK_OK;
  dim3 grid((a->max_input_len + kQTile - 1) / kQTile, a->num_kv_heads, a->batch), block(64 * G);
  const size_t shm = (size_t)G * (kQTile * kPRowP * 2 + 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool off32 = a->kv_num_slots > 0 && (a->kv_num_slots * a->kv_slot_stride * 2) < (int64_t)0xffffffffll;
  if (a->head_dim == 128) {
    if (off32) hipLaunchKernelGGL((context_attention_kernel<128, true>), grid, block, shm, s, *a);
    else hipLaunchKernelGGL((context_attention_kernel<128, false>), grid, block, shm, s, *a);
  } else {
    if (off32) hipLaunchKernelGGL((context_attention_kernel<64, true>), grid, block, shm, s, *a);
    else hipLaunchKernelGGL((context_attention_kernel<64, false>), grid, block, shm, s, *a);
  }
  return check_launch("svk_context_attention_fwd");
}
