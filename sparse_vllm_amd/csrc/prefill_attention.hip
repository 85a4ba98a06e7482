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

#include <stdlib.h>
#include <type_traits>

#include "svk_common.hpp"
#include "lds_dma.hpp"

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
  const int m0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * kQTile;   // longest key ranges first (see the v2 kernel)
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


// ------------------------------------------------------------------------------------------------
// Second generation: the key / value tile is staged ONCE per workgroup in LDS and shared by the G query-head waves
// (the kernel above lets every wave fetch the whole tile itself: G x the L1 requests, one tile of prefetch).
//   * tile = 64 keys; K rows land in LDS with their 16-byte chunks XOR-swizzled by the key index, so the A-operand
//     reads of Q.K^T (lane = key row, stride 256 B) are conflict-free; V rows are stored as they are;
//   * Q.K^T is computed SWAPPED on v_mfma_f32_32x32x16_bf16: A = K (32 keys x 16 dims), B = Q^T, so that a lane's 16
//     accumulator registers are 16 keys of ONE query row: the row maximum / sum are in-lane reductions plus one
//     exchange with the lane that holds the other 16 keys (lane ^ 32), no DPP ladders, and P never leaves the
//     registers: the 8 accumulator registers [8s, 8s+8) ARE the A operand of P.V step s once packed to bf16 (the k
//     index of that product is free as long as V is read in the same key order);
//   * P.V: B = V with MFMA column n <-> head dim (n & 15) * 8 + 2i + (n >> 4) (i = which of the four 32-column MFMAs),
//     so a lane reads eight 16-byte row segments from LDS and picks one bf16 of each with a byte permute (lane-dependent
//     selector) - no transposed copy of V;
//   * the per-row rescale factor lives in "lane = query row" layout while the output accumulator has its query rows in
//     registers: it crosses through a 128-byte per-wave LDS row only when some row maximum moved.
// ------------------------------------------------------------------------------------------------
constexpr int kKV2 = 64;                 // keys per tile
constexpr int kRowB = 256;               // bytes of one K or V head row (D = 128)


// Developer build (make EXTRA=-DSVK_PA_TIMING, then tools/pa_timing.py): per-wave s_memrealtime sums of the five phases of
// a tile, written behind the output rows when bit 30 of max_input_len is set.  Not part of the product build.
#ifdef SVK_PA_TIMING
constexpr int kPaLenMask = 0xfffffff;
#define PA_TIMING_BEGIN()                                        \
  const bool timing = (a.max_input_len >> 30) & 1;               \
  long long T[6] = {0, 0, 0, 0, 0, 0};                           \
  const long long t_entry = __builtin_amdgcn_s_memrealtime();    \
  long long t_last = t_entry
#define PA_STAMP(i)                                              \
  do {                                                           \
    const long long now_ = __builtin_amdgcn_s_memrealtime();     \
    T[i] += now_ - t_last;                                       \
    t_last = now_;                                               \
  } while (0)
#define PA_SETTLE(x, y) asm volatile("s_nop 0" ::"v"(x), "v"(y))   /* results complete before the stamp */
#define PA_TIMING_END()                                                                                                  \
  if (timing && lane == 0) {                                                                                             \
    long long* dbg = reinterpret_cast<long long*>(a.o + (int64_t)(a.max_input_len & kPaLenMask) * a.o_stride_t);         \
    long long* e = dbg + ((int64_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8 + w) * 8;                                   \
    e[0] = T[0]; e[1] = T[1]; e[2] = T[2]; e[3] = T[3]; e[4] = T[4];                                                     \
    e[5] = __builtin_amdgcn_s_memrealtime() - t_entry; e[6] = ntiles;                                                    \
  }
#else
constexpr int kPaLenMask = -1;
#define PA_TIMING_BEGIN()
#define PA_STAMP(i)
#define PA_SETTLE(x, y)
#define PA_TIMING_END()
#endif

typedef __attribute__((ext_vector_type(4))) short pa_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short pa_s16x8_t;

template <bool OFF32>
__global__ void __launch_bounds__(512) context_attention_kernel_v2(const SvkContextAttentionArgs a) {
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = a.num_q_heads / a.num_kv_heads;
  const int b = blockIdx.z;
  // (query tile, KV head) of this workgroup.  Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), each
  // XCD with its own L2: when the grid allows it, a KV head is served by 8 / Hkv XCDs only, so an L2 streams one head's
  // K/V rows instead of all of them.
  int qt = blockIdx.x, kvh = blockIdx.y;
  {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    if ((ny == 1 || ny == 2 || ny == 4 || ny == 8) && ((nx * ny) & 7) == 0) {
      const unsigned L = blockIdx.x + nx * blockIdx.y;
      const unsigned xcd = L & 7, j = L >> 3;
      kvh = xcd % ny;
      qt = j * (8 / ny) + xcd / ny;
    }
  }
  // G compute waves (wave = query head) + for G <= 7 one helper wave that issues all of the workgroup's DMA: it sits on
  // the SIMD that holds a single compute wave, and the compute waves' instruction streams lose the ~0.3 us per tile of
  // slot-id reads, address arithmetic and DMA issue
  const bool has_helper = (int)(blockDim.x >> 6) > G;
  const bool helper = has_helper && w == G;
  const int head = kvh * G + min(w, G - 1);
  const int lq = lane & 31, half = lane >> 5;
  const int pc = a.b_prompt_cache_len[b];
  const int q_len = a.b_seq_len[b] - pc;
  // query tiles in DESCENDING order of their key count (causal rows: the last tile of the chunk has the longest key
  // range): the longest workgroups start first and the short ones fill the tail of the launch
  const int m0 = ((int)gridDim.x - 1 - qt) * kQTile;
  if (m0 >= q_len) return;
  PA_TIMING_BEGIN();
  const int start_loc = a.b_start_loc[b];
  const int kv_end = min(m0 + kQTile + pc, q_len + pc);
  // LDS: two tile buffers of (K | V) (the epilogue reuses this part as per-wave output staging, G x 8 KiB) |
  //      2 x 64 slot ids | per-wave 32-float rows (rescale factors / 1 / row sum)
  constexpr int kBuf = 2 * kKV2 * kRowB;             // one (K | V) tile
  constexpr int kRing = 3;                             // K|V tiles in LDS: t-1 (its last P V), t, t+1
  const int aux0 = max(kRing * kBuf, G * kQTile * D * 2);
  int* slot_lds = reinterpret_cast<int*>(lds_raw + aux0);
  float* fac = reinterpret_cast<float*>(slot_lds + 2 * kKV2) + w * 32;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw);
  const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;

  // Q^T fragments (B operand): lane (q = lq, half) holds Q[m0 + q][head][ds*16 + half*8 .. +8]
  bf16x8_t qb[D / 16];
  {
    const uint16_t* qp = a.q + (int64_t)(start_loc + min(m0 + lq, q_len - 1)) * a.q_stride_t + (int64_t)head * a.q_stride_h + half * 8;
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) qb[ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qp + ds * 16));
  }
  const float sm_scale = rsqrtf((float)D) * 1.4426950408889634f;
  const float mask_raw = -1.0e8f / sm_scale;
  float m_run = -INFINITY, l_part = 0.f;            // of query row lq (both halves keep the same maximum)
  f32x16_t o[4];                                    // o[i][r]: query row (r&3) + 8(r>>2) + 4 half, head dim 32 i + lq
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  const char* const kt = reinterpret_cast<const char*>(a.k_cache);
  const char* const vt = reinterpret_cast<const char*>(a.v_cache);
  const int64_t slot_bytes = a.kv_slot_stride * 2;
  const int64_t head_bytes = (int64_t)kvh * a.kv_head_stride * 2;
  // transpose-read addresses into the V half of buffer 0 (LDS byte addresses): lane (group member q = lane & 15) points at
  // key (q >> 2) + 4 half of a block of 8 keys, dims 32 i + 16 (lq >> 4) + 4 (q & 3) .. + 3, through the V swizzle
  uint32_t vbase[4];
  {
    const uint32_t q16 = lane & 15;
    const uint32_t v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + kKV2 * kRowB +
                        ((q16 >> 2) + 4 * half) * kRowB + (lq >> 4) * 32 + (q16 & 3) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) vbase[i] = v0 + ((i ^ (q16 >> 2)) << 6);
  }
  const int qrow = m0 + lq;

  // ---- pipeline: while tile t is computed out of buffer t & 1, the DMA of tile t+1 fills the other buffer and the slot
  //      ids of tile t+2 travel to a register; ONE wait (vmcnt(0): this wave's DMA pieces and id load) + ONE workgroup
  //      barrier per tile.  The 32 DMA instructions of a tile (16 x 4 K rows, 16 x 4 V rows) are dealt round-robin to the
  //      G waves.  K chunks are swizzled on the SOURCE address (position p of row `key` holds chunk p ^ (key & 15)).
  const int ntiles = (kv_end + kKV2 - 1) / kKV2;
  // A tile is 32 DMA instructions (16 x 4 K rows, 16 x 4 V rows) = 8 groups of four that share one M0 write (group g <
  // 4: K rows 16 g .. 16 g + 15, else V rows).  Everything about a lane's source address except the slot id is a
  // constant of (K or V, position in the group): an instruction costs one ds_read of the id and one multiply-add - the
  // first version computed the addresses per instruction (25 instructions each; at the ~5 cycles per instruction one wave
  // gets, 0.55 us of every 2.5 us tile).  With 7 waves on 4 SIMDs wave 3 has a SIMD to itself (waves go to the SIMDs
  // cyclically) and takes two groups, the others one.  (Placement is a speed heuristic only.)
  const int widx = G == 7 ? (w == 3 ? 0 : (w < 3 ? w + 1 : w)) : w;
  const uint32_t lane_row = lane >> 4, lane_pos = lane & 15;
  uint32_t kc[4], vc;                                  // byte offset inside a slot (+ the instruction-offset bias)
#pragma unroll
  for (int i = 0; i < 4; ++i) kc[i] = (uint32_t)head_bytes + ((lane_pos ^ ((i << 2) + lane_row)) << 4) + 3072 - 1024 * i;
  vc = (uint32_t)head_bytes + ((lane_pos ^ (lane_row << 2)) << 4) + 3072;   // V: position p of row `key` holds chunk p ^ 4 (key & 3)
  const uint32_t slot_b32 = (uint32_t)slot_bytes;
  auto issue_group = [&](int t, int g, auto is_v_c) {
    constexpr bool is_v = decltype(is_v_c)::value;
    const int* ids = slot_lds + (t & 1) * kKV2 + (g & 3) * 16 + lane_row;
    int slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) slot[i] = ids[4 * i];
    const uint32_t dst = lds0 + (t % kRing) * kBuf + g * 4096;
    if constexpr (OFF32) {
      uint32_t voff[4];
      // (slot < 2^24: a slot is >= 256 bytes and the tensor < 4 GiB)
#pragma unroll
      for (int i = 0; i < 4; ++i) voff[i] = __umul24((uint32_t)slot[i], slot_b32) + (is_v ? vc - 1024 * i : kc[i]);
      pa_dma4x16_off32(voff, (is_v ? vt : kt) - 3072, dst);
    } else {
      const char* src[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) src[i] = (is_v ? vt : kt) + ((int64_t)slot[i] * slot_bytes + (int64_t)(is_v ? vc - 1024 * i : kc[i]) - 3072);
      pa_dma4x16(src, dst);
    }
  };
  const int g_first = has_helper ? (helper ? 0 : 8) : widx, g_step = has_helper ? 1 : G;
  auto issue_tile = [&](int t) {
    for (int g = g_first; g < 8; g += g_step) {
      if (g < 4) issue_group(t, g, std::false_type{}); else issue_group(t, g, std::true_type{});
    }
  };
  // slot ids of tile t: 64 x 4 bytes by LDS-DMA as well (wave 0), straight into the id buffer t & 1 - no register in flight
  const int ids_wave = has_helper ? G : 0;
  auto issue_ids = [&](int t) {
    if (w == ids_wave) {
      const int32_t* p = row + min(t * kKV2 + lane, kv_end - 1);
      uint32_t keep;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)(slot_lds + (t & 1) * kKV2));
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    }
  };
  // K fragment addresses (A operand of S^T = K Q^T): lane (key row lq of a 32-key block, k-chunk ds*2 + half) through the
  // K swizzle; LDS byte addresses relative to the block's first row
  uint32_t kaddr[D / 16];
#pragma unroll
  for (int ds = 0; ds < D / 16; ++ds)
    kaddr[ds] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + lq * kRowB + (((ds * 2 + half) ^ (lq & 15)) << 4);

  issue_ids(0);
  issue_ids(1);
  // (step 0 multiplies the V rows of "tile -1, block 1" - ring slot 2 - by P = 0: they must be finite)
  for (int i = threadIdx.x; i < 32 * kRowB / 16; i += blockDim.x)
    *reinterpret_cast<uint4*>(lds_raw + 2 * kBuf + kKV2 * kRowB + 32 * kRowB + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  issue_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_ids(2);
#pragma unroll
  for (int ds = 0; ds < D / 16; ++ds) asm volatile("" : "+v"(qb[ds]));

  // ---- software pipeline over 32-key blocks j = 2 t + kb.  Step j holds three stages of three different blocks:
  //        C(j-1): O += P(j-1) V(j-1)        8 MFMAs, V by LDS transpose reads
  //        B(j)  : online softmax of S(j)    the vector work (maximum, exp2, row sum, bf16 packing)
  //        A(j+1): S(j+1) = K(j+1) Q^T       8 MFMAs, K fragments by ds_read_b128
  //      so the vector unit and the matrix unit of the SIMD are busy at the same time inside ONE wave (with the stages of
  //      a block back to back, the softmax waits for its QK and the P V for its softmax: measured 1.5 us per tile for a
  //      wave that has a SIMD to itself, the sum of its MFMA and VALU times).  The rescale of O by 2^(m_old - m_new) of
  //      block j (only when some row's maximum moved) sits between C(j-1) and C(j).  Tile t+1 must have landed before
  //      A(2t+2) in step 2t+1 and tile t-1 is read last by C(2t-1) in step 2t: the workgroup barrier of tile t sits
  //      between those two steps and the DMA of tile t+2 is issued right behind it - a ring of three tile buffers.
  if (helper) {
    // one barrier per tile like the compute waves: behind barrier t, tile t+1 has landed (this wave's DMA drained) and
    // tile t-1 is dead, so tile t+2 and the ids of tile t+3 go out
    for (int t = 0; t < ntiles; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < ntiles) issue_tile(t + 2);
      if (t + 3 < ntiles) issue_ids(t + 3);
    }
    __syncthreads();
    return;
  }
  uint32_t ob_prev = 2 * kBuf, ob_cur = 0, ob_next = kBuf;          // ring offsets of tiles t-1, t, t+1
  f32x16_t s_a, s_b;                                                 // S of the even / odd block of a tile
  {
    pa_u32x4_t ka[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) ka[ds] = *reinterpret_cast<const __attribute__((address_space(3))) pa_u32x4_t*>(kaddr[ds]);
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) asm volatile("" : "+v"(ka[ds]));
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds)
      s_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[ds]), qb[ds], ds == 0 ? f32x16_t{} : s_a, 0, 0, 0);
  }
  uint32_t pp_a[8], pp_b[8];                                         // P of the even / odd block as bf16 pairs (A operand of P V)
#pragma unroll
  for (int e = 0; e < 8; ++e) pp_b[e] = 0u;                          // (step 0 multiplies V of block 0 by P = 0)
  // one step: kb = which block of tile t the softmax works on; s_in / p_prev are consumed, s_out / p_out produced
  auto step = [&](int t, auto kb_c, f32x16_t& s_in, f32x16_t& s_out, const uint32_t (&p_prev)[8], uint32_t (&p_out)[8])
                  __attribute__((always_inline)) {
    constexpr int kb = decltype(kb_c)::value;
    const int k0 = t * kKV2;
    // K of block j+1: the other block of this tile, or the first one of the next tile; V of block j-1 likewise
    const uint32_t k_off_next = kb == 0 ? ob_cur + 32 * kRowB : ob_next;
    const uint32_t v_off_prev = kb == 0 ? ob_prev + 32 * kRowB : ob_cur;
    // operands of the two matrix stages (K first: its products form the dependent chain)
    pa_u32x4_t ka[D / 16];
#pragma unroll
    for (int ds = 0; ds < D / 16; ++ds) ka[ds] = *reinterpret_cast<const __attribute__((address_space(3))) pa_u32x4_t*>(kaddr[ds] + k_off_next);
    pa_s16x4_t vr[2][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto* vp = reinterpret_cast<__attribute__((address_space(3))) pa_s16x4_t*>(vbase[i] + v_off_prev);
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int eh = 0; eh < 2; ++eh)
          vr[sp][i][eh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<__attribute__((address_space(3))) pa_s16x4_t*>(vp) + (sp * 16 + eh * 8) * kRowB / 8);
    }
    if (k0 + kKV2 > m0 + pc || k0 + kKV2 > kv_end) {
      // only the tiles on the diagonal / at the end of the row: branch-free selects (register r = key k0 + kb*32 +
      // (r&3) + 8(r>>2) + 4 half)
      const int lim = min(qrow + pc, kv_end - 1) - k0 - kb * 32 - 4 * half;        // the key offset (inside the block) must be <= lim
#pragma unroll
      for (int r = 0; r < 16; ++r) s_in[r] = ((r & 3) + 8 * (r >> 2) <= lim) ? s_in[r] : mask_raw;
    }
    // B(j): row maximum as a tree of three-input maxima
    float mx = vmax(vmax3(vmax3(s_in[0], s_in[1], s_in[2]), vmax3(s_in[3], s_in[4], s_in[5]), s_in[15]),
                    vmax3(vmax3(s_in[6], s_in[7], s_in[8]), vmax3(s_in[9], s_in[10], s_in[11]), vmax3(s_in[12], s_in[13], s_in[14])));
    mx = vmax(mx, lane_xor32(mx));
    // The reference point of the row's exponents follows the row maximum lazily: it moves only when the maximum grew by
    // more than 2^8 in the exponent domain (the probabilities of a row are then at most 256 instead of 1 until the next
    // move: no precision is lost in fp32 / bf16, sums and outputs scale together).  With the exact maximum as the
    // reference, almost every block of a workgroup's first tiles rescales O (64 multiplies + an LDS round trip).
    const float nm_true = vmax(m_run, mx);
    const bool moved = (nm_true - m_run) * sm_scale > 8.0f;
    const float nm = moved ? nm_true : m_run;
    const float nms = nm * sm_scale;
    const float al = moved ? __builtin_amdgcn_exp2f(m_run * sm_scale - nms) : 1.0f;
    m_run = nm;
    // A(j+1) and C(j-1) alternate (the A products form a dependent chain, the C products of one step are independent), each
    // followed by the softmax of one S register (multiply-add, exp2, row sum, every other time a bf16 pack); scheduling
    // barriers keep exactly this order - left to itself hipcc issues the sixteen products back to back behind the vector
    // work.  C step sp uses the keys of registers [8 sp, 8 sp + 8): key 16 sp + 8(e>>2) + 4 half + (e&3) of the block.  Its
    // B operand (lane = head dim 32 i + lq, 8 keys) comes out of the row-major V tile by the LDS transpose read: each
    // 16-lane group reads a [4 keys][16 dims] block, a lane supplying the address of 4 consecutive dims of one key.
    float psum = 0.f;
    {
      const bf16x8_t pa0 = __builtin_bit_cast(bf16x8_t, make_uint4(p_prev[0], p_prev[1], p_prev[2], p_prev[3]));
      const bf16x8_t pa1 = __builtin_bit_cast(bf16x8_t, make_uint4(p_prev[4], p_prev[5], p_prev[6], p_prev[7]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if ((g & 1) == 0) {
          s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[g >> 1]), qb[g >> 1], g == 0 ? f32x16_t{} : s_out, 0, 0, 0);
        } else {
          const int sp = g >> 3, i = (g >> 1) & 3;
          const pa_s16x8_t vb = __builtin_shufflevector(vr[sp][i][0], vr[sp][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp ? pa1 : pa0, __builtin_bit_cast(bf16x8_t, vb), o[i], 0, 0, 0);
        }
        s_in[g] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_in[g], sm_scale, -nms));
        psum += s_in[g];
        if (g & 1) p_out[g >> 1] = pack_bf16_pair(s_in[g - 1], s_in[g]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_part = l_part * al + psum;
    // (pin the softmax results to this block: left alone, the compiler sinks the exp2 work below the barrier branch)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(p_out[e]));
    asm volatile("" : "+v"(l_part));
    PA_SETTLE(o[0][0], s_out[15]);
    PA_STAMP(1);
    if (__any(moved)) {
      // rescale factors: "lane = query row" -> "register = query row" through the wave's LDS row
      if (half == 0) fac[lq] = al;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float4 f4 = *reinterpret_cast<const float4*>(fac + 8 * jj + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i][4 * jj + 0] *= f4.x; o[i][4 * jj + 1] *= f4.y; o[i][4 * jj + 2] *= f4.z; o[i][4 * jj + 3] *= f4.w;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    PA_STAMP(2);
  };
  for (int t = 0; t < ntiles; ++t) {
    PA_STAMP(5);
    step(t, std::integral_constant<int, 0>{}, s_a, s_b, pp_b, pp_a);
    // tile t+1 has landed for every wave after this pair, and nobody reads tile t-1 any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PA_STAMP(3);
    __syncthreads();
    PA_STAMP(4);
    if (t + 2 < ntiles) issue_tile(t + 2);
    if (t + 3 < ntiles) issue_ids(t + 3);            // into the id buffer of tile t+1, whose DMA was issued one barrier ago
    PA_STAMP(0);
    step(t, std::integral_constant<int, 1>{}, s_b, s_a, pp_a, pp_b);
    const uint32_t x = ob_prev;
    ob_prev = ob_cur; ob_cur = ob_next; ob_next = x;
  }
  const uint32_t v_off_prev = ob_prev + 32 * kRowB;                  // V of the last block
  const uint32_t (&pp)[8] = pp_b;
  // C of the last block
  {
    pa_s16x4_t vr[2][4][2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int eh = 0; eh < 2; ++eh)
          vr[sp][i][eh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) pa_s16x4_t*>(
              vbase[i] + v_off_prev + (uint32_t)((sp * 16 + eh * 8) * kRowB)));
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const bf16x8_t pa = __builtin_bit_cast(bf16x8_t, make_uint4(pp[4 * sp], pp[4 * sp + 1], pp[4 * sp + 2], pp[4 * sp + 3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const pa_s16x8_t vb = __builtin_shufflevector(vr[sp][i][0], vr[sp][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8_t, vb), o[i], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                   // (the epilogue stages through the tile buffers)
  PA_TIMING_END();
  // ---- epilogue: row sums across the two halves, 1/l into register-row layout, outputs through LDS as whole rows
  l_part += lane_xor32(l_part);
  if (a.score_row_stats != nullptr && half == 0) {
    // the score window's rows leave their final softmax statistics behind for the token-score pass (svk.h): the base-2
    // exponent constant of the row, m * c + log2 l (any reference point m gives the same sum)
    const int qs = a.score_q_start[b];
    const int r = pc + qrow - qs;
    if (qs >= 0 && r >= 0 && r < a.score_wpad && qrow < q_len)
      a.score_row_stats[((int64_t)(b * a.num_kv_heads + kvh) * G + w) * a.score_wpad + r] =
          m_run * sm_scale + __builtin_amdgcn_logf(l_part);
  }
  if (a.score_clear != nullptr && kvh == 0 && m0 + kQTile >= q_len) {
    // one workgroup per sequence (the tile that holds the chunk's last rows, KV head 0) zeroes the sequence's score row
    float* dst = a.score_clear + (int64_t)b * a.score_clear_stride;
    for (int c = threadIdx.x; c < a.score_clear_cols; c += G * 64) dst[c] = 0.f;
  }
  if (half == 0) fac[lq] = 1.0f / l_part;            // (the loop's last barrier retired the tiles: staging may reuse them)
  uint16_t* ost = reinterpret_cast<uint16_t*>(lds_raw + (size_t)w * (kQTile * D * 2));     // [32 rows][128] bf16 per wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 f4 = *reinterpret_cast<const float4*>(fac + 8 * j + 4 * half);
    const float f[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r3 = 0; r3 < 4; ++r3) {
        const int qr = 8 * j + 4 * half + r3;
        ost[qr * D + 32 * i + lq] = (uint16_t)f32_to_bf16_bits(o[i][4 * j + r3] * f[r3]);
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 32 rows x 256 B: lane -> (row = it*4 + lane/16, 16-byte piece lane%16)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int qr = it * 4 + (lane >> 4);
    if (m0 + qr < q_len)
      *reinterpret_cast<uint4*>(a.o + (int64_t)(start_loc + m0 + qr) * a.o_stride_t + (int64_t)head * a.o_stride_h + (lane & 15) * 8) =
          *reinterpret_cast<const uint4*>(ost + qr * D + (lane & 15) * 8);
  }
}


// ------------------------------------------------------------------------------------------------
// The reference's score-collecting forms of context_attention_fwd (context_flashattention_nopad.py:82-240).  Both are
// sums of RAW logits over query rows, so they factor through the query rows' suffix sums:
//   sum_{r in Q, r >= r0} q[r, h] . k[t] = (sum_{r in Q, r >= r0} q[r, h]) . k[t]
// Kernel A writes, for every 128-row query block Q (the reference's BLOCK_M), the within-block suffix sums of the bf16
// query rows in fp32; kernel B walks the blocks that see key t and takes one 128-dim dot per (head, block).
// ------------------------------------------------------------------------------------------------
constexpr int kScoreBlockM = 128;

__global__ void __launch_bounds__(256) ctx_q_block_suffix_kernel(const SvkContextAttentionArgs a) {
  const int b = blockIdx.z, qb = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int cols = a.num_q_heads * a.head_dim;
  const int chunk = a.b_seq_len[b] - a.b_prompt_cache_len[b];
  const int r0 = qb * kScoreBlockM, r1 = min(r0 + kScoreBlockM, chunk);
  if (col >= cols || r0 >= r1) return;
  const int h = col / a.head_dim, d = col - h * a.head_dim;
  const int64_t t0 = a.b_start_loc[b];
  float run = 0.f;
  for (int r = r1 - 1; r >= r0; --r) {
    run += bf16_lo((uint32_t)a.q[(t0 + r) * a.q_stride_t + (int64_t)h * a.q_stride_h + d]);
    a.score_workspace[(t0 + r) * cols + col] = run;
  }
}

template <int DIM>
__global__ void __launch_bounds__(64) ctx_attn_score_kernel(const SvkContextAttentionArgs a) {
  __shared__ __attribute__((aligned(16))) float ksh[8 * 256];           // k[t] of every KV head, fp32
  const int b = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
  const int pc = a.b_prompt_cache_len[b];
  const int len = a.b_seq_len[b];
  const int chunk = len - pc;
  if (t >= len || chunk <= 0) return;
  const int D = a.head_dim, Hkv = a.num_kv_heads, Hq = a.num_q_heads, G = Hq / Hkv;
  const int64_t slot = a.req_to_tokens[(int64_t)a.b_req_idx[b] * a.req_stride + t];
  for (int i = lane; i < Hkv * D; i += 64) {
    const int kh = i / D, d = i - kh * D;
    ksh[i] = bf16_lo((uint32_t)a.k_cache[slot * a.kv_slot_stride + (int64_t)kh * a.kv_head_stride + d]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int h = lane;
  const int64_t t0 = a.b_start_loc[b];
  const int cols = Hq * D;
  // query block Q (rows [Q*128, ..)) visits key t iff t < min(Q*128 + 128, chunk) + pc; its first row that sees t
  const int rmin = max(t - pc, 0);
  const int nblk = (chunk + kScoreBlockM - 1) / kScoreBlockM;
  float total = 0.f, best = -INFINITY;
  if (h < Hq) {
    const float* kv = ksh + (h / G) * D;
    for (int qb = rmin / kScoreBlockM; qb < nblk; ++qb) {
      const int r0 = max(qb * kScoreBlockM, rmin);
      const float* qs = a.score_workspace + (t0 + r0) * cols + (int64_t)h * D;
      float acc = 0.f;
      for (int d = 0; d < D; d += 4) {
        const float4 x = *reinterpret_cast<const float4*>(qs + d), y = *reinterpret_cast<const float4*>(kv + d);
        acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
      }
      total += acc;
      best = fmaxf(best, acc / (float)chunk);
    }
  }
  if (DIM == 3) {
    if (h < Hq) a.attn_score[(int64_t)b * a.attn_score_stride_b + (int64_t)h * a.attn_score_stride_h + t] += total;
  } else {
    best = wave_allmax(best);
    if (lane == 0) {
      float* dst = a.attn_score + (int64_t)b * a.attn_score_stride_b + t;
      *dst = fmaxf(*dst, best);
    }
  }
}

int launch_attn_scores(const SvkContextAttentionArgs& a, hipStream_t s) {
  const int cols = a.num_q_heads * a.head_dim;
  const int nblk = ((a.max_input_len & kPaLenMask) + kScoreBlockM - 1) / kScoreBlockM;
  hipLaunchKernelGGL(ctx_q_block_suffix_kernel, dim3((cols + 255) / 256, nblk, a.batch), dim3(256), 0, s, a);
  // (keys per sequence: up to the longest context; rows past a sequence's length return at once)
  const dim3 grid((unsigned)a.attn_score_cols, a.batch);
  if (a.attn_score_dim == 3) hipLaunchKernelGGL(ctx_attn_score_kernel<3>, grid, dim3(64), 0, s, a);
  else hipLaunchKernelGGL(ctx_attn_score_kernel<2>, grid, dim3(64), 0, s, a);
  return check_launch("svk_context_attention_fwd (scores)");
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
  if (a->attn_score != nullptr) {
    SVK_REQUIRE(a->attn_score_dim == 2 || a->attn_score_dim == 3, SVK_ERR_VALUE,
                "svk_context_attention_fwd: attn_score must be rank 2 or 3, got %d", a->attn_score_dim);
    SVK_REQUIRE(a->score_workspace != nullptr && a->attn_score_cols > 0, SVK_ERR_VALUE,
                "svk_context_attention_fwd: attn_score needs score_workspace and attn_score_cols");
    SVK_REQUIRE(a->num_q_heads <= 64 && a->num_kv_heads <= 8 && a->head_dim <= 256 && a->head_dim % 4 == 0, SVK_ERR_LAYOUT,
                "svk_context_attention_fwd: score collection supports <= 64 query heads, <= 8 KV heads");
  }
  SVK_REQUIRE((a->score_row_stats == nullptr && a->score_clear == nullptr) || a->head_dim == 128, SVK_ERR_LAYOUT,
              "svk_context_attention_fwd: score statistics are produced by the head_dim 128 kernel only");
  SVK_REQUIRE(a->score_row_stats == nullptr || (a->score_q_start != nullptr && a->score_wpad > 0), SVK_ERR_VALUE,
              "svk_context_attention_fwd: score_row_stats needs score_q_start and score_wpad");
  dim3 grid(((a->max_input_len & kPaLenMask) + kQTile - 1) / kQTile, a->num_kv_heads, a->batch), block(64 * G);
  const size_t shm = (size_t)G * (kQTile * kPRowP * 2 + 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool off32 = a->kv_num_slots > 0 && (a->kv_num_slots * a->kv_slot_stride * 2) < (int64_t)0xffffffffll - 8192;
  if (a->head_dim == 128) {   // LDS-shared K/V tiles; head_dim 64 keeps the first kernel (every wave fetches its own tile)
    const size_t tiles = 3 * 2 * kKV2 * kRowB, stage = (size_t)G * kQTile * 128 * 2;
    const size_t shm2 = (tiles > stage ? tiles : stage) + 2 * kKV2 * sizeof(int) + (size_t)G * 32 * sizeof(float);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(context_attention_kernel_v2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(context_attention_kernel_v2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    const dim3 block2(64 * (G + (G <= 7 ? 1 : 0)));          // GQA groups <= 7: an eighth wave only issues the tile DMA
    if (off32) hipLaunchKernelGGL((context_attention_kernel_v2<true>), grid, block2, shm2, s, *a);
    else hipLaunchKernelGGL((context_attention_kernel_v2<false>), grid, block2, shm2, s, *a);
    if (a->attn_score != nullptr) return launch_attn_scores(*a, s);
    return check_launch("svk_context_attention_fwd");
  }
  if (off32) hipLaunchKernelGGL((context_attention_kernel<64, true>), grid, block, shm, s, *a);
  else hipLaunchKernelGGL((context_attention_kernel<64, false>), grid, block, shm, s, *a);
  if (a->attn_score != nullptr) return launch_attn_scores(*a, s);
  return check_launch("svk_context_attention_fwd");
}

extern "C" int64_t svk_context_attention_score_workspace_bytes(int64_t tokens, int32_t num_q_heads, int32_t head_dim) {
  if (tokens <= 0 || num_q_heads <= 0 || head_dim <= 0) return 0;
  return tokens * (int64_t)num_q_heads * head_dim * (int64_t)sizeof(float);
}
