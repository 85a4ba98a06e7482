// DeltaKV look-ahead reconstruction, second half in ONE launch (gfx950): the second Linear of compress_up
//     delta[n, :] = bf16(hidden[n, :] . W2^T + b2)                     (utils/compressor.py:69-73)
// and the reconstruction of the selected tokens from it
//     K = delta_K + mean_f(father K), optional RMS k-norm, RoPE(out_pos);  V = delta_V + mean_f(father V)
//                                                                      (kernels/triton/deltakv_kernels.py:2732-2907)
// for the layers of a look-ahead sub-batch.  The library GEMM + reconstruct pair this replaces writes delta to HBM and
// reads it back; here a workgroup owns 128 tokens x one head of K or V (128 output features = the rotate-half partners of
// a head stay together), multiplies on the matrix cores and finishes the rows out of LDS.
//
//   * operands: hidden rows (tokens) and W2 rows (features) are both K-contiguous, so both tiles of a stage - 128 rows x
//     64 bf16 = 16 KiB each - travel HBM/L2 -> LDS by `global_load_lds_dwordx4` (no registers in flight), four stages
//     deep; 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7) of its 128-byte LDS row (swizzled on the SOURCE
//     address): a `ds_read_b128` is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...) over 64 banks, i.e. two
//     128-byte rows per bank sweep - within every group the 8 even and the 8 odd rows then sit at 8 different positions
//     (c ^ (r & 7), the textbook swizzle, puts rows 12 and 20 of a group on the same banks: two-way conflicts on every read).
//   * product: swapped, C^T = W2 . hidden^T on v_mfma_f32_32x32x16_bf16 (A = 32 features x 16, B = 16 x 32 tokens), so a
//     lane's accumulator registers are 4 consecutive FEATURES of one token: bias add, bf16 rounding and an 8-byte LDS
//     store per group give the delta tile row-major [token][feature].
//   * epilogue: the lane mapping and element arithmetic of deltakv_reconstruct_vec_kernel (8 lanes per token, elements
//     p..p+7 and p+64..p+71), delta read from LDS; the plan's per-token indices (out slot / position / fathers) are
//     requested before the main loop.
// One wave = a 64 x 64 quarter of the tile (2 x 2 MFMA blocks, 64 accumulator registers).

#include <cstdlib>

#include "lds_dma.hpp"
#include "svk_common.hpp"

// Developer build (make EXTRA=-DSVK_UR_TIMING, tools/ur_timing.py): s_memtime stamps of every wave of the first 64
// workgroups inside main-loop steps 8..15 of up_recon_kernel: 0 step start, 1 reads + MFMAs of k-substeps 0, 1 issued,
// 2 tile t+1 waited for, 3 behind the barrier, 4 DMA issued, 5 step end.
#ifdef SVK_UR_TIMING
__device__ unsigned long long g_ur_stamps[64 * 8 * 8 * 6];
#define SVK_UR_STAMP(t_, i_)                                                                                           \
  do {                                                                                                                 \
    if ((t_) >= 8 && (t_) < 16 && blockIdx.x < 64u && blockIdx.y == 0u && (threadIdx.x & 63) == 0)                      \
      g_ur_stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + ((t_) - 8)) * 6 + (i_)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
extern "C" int svk_debug_up_recon_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ur_stamps), sizeof(g_ur_stamps));
}
#else
#define SVK_UR_STAMP(t_, i_)
#endif

namespace svk {
namespace {

constexpr int kUrWTileBytes = 128 * 128;             // weight tile of a stage: 128 features x 64 bf16
constexpr int kUrDeltaRow = 272;                     // bytes per token row of the delta tile (256 + 16: 16-byte aligned rows)

struct UpReconParams {
  SvkDeltakvUpReconArgs u;
  SvkDeltakvReconstructArgs r;
  SvkDeltakvReconstructBatch lb;
  int m_tiles;
  int debug_same_k;      // developer (SVK_UP_RECON_SAME_K=1): every stage re-reads the first 64 hidden features (cache-resident operands)
};

typedef __attribute__((address_space(3))) const pa_u32x4_t* lds_u32x4_ptr;
typedef __attribute__((ext_vector_type(2))) __bf16 ur_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float ur_f32x2_t;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const ur_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ur_bf16x2_t));     // v_cvt_pk_bf16_f32 (RNE)
}

__device__ __forceinline__ bf16x8_t lds_frag(uint32_t addr) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<lds_u32x4_ptr>(addr));
}

// two 16-byte-per-lane LDS-DMA loads behind one M0 write (lds_dma.hpp pa_dma4x16_off32 for the conventions)
__device__ __forceinline__ void ur_dma2x16_off32(uint32_t v0, uint32_t v1, const char* base, uint32_t lds_addr) {
  uint32_t keep;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(lds_addr) : "memory");
}

// TM = tokens per workgroup, WAVES = its waves:
//   (128, 4)  a wave = 64 tokens x 64 features; ring of 4 stages of 32 KiB
//   (256, 8)  the same wave tile, two waves per SIMD; ring of 3 stages of 48 KiB, a quarter less operand traffic per product
//   (256, 4)  a wave = 64 tokens x all 128 features (128 accumulator registers): 6 operand reads per 8 MFMAs instead of
//             4 per 4.  Measured equal per step to (256, 8) (1.18 us per 64-deep step of a 256 x 128 tile) and slower in
//             the epilogue (one wave per SIMD): 110 against 107 us at 2 x 8192 tokens - developer form only
// KF = fathers per token held in registers (k_fathers <= KF).
// SLOTS = ring slots of a stage each; the DMA runs DIST = SLOTS - 1 tiles ahead (SLOTS 2: also 2 - tile t + 2 goes into tile
// t's own slot behind the mid-step barrier, when every wave holds the rest of tile t in registers).
template <int TM, int WAVES, int KF, int SLOTS = (TM == 128 ? 4 : 3)>
__global__ void __launch_bounds__(WAVES * 64, SLOTS == 2 ? 2 : 1) up_recon_kernel(const UpReconParams P) {
  constexpr int D = 128, HD2 = 64;
  constexpr int NT = WAVES * 64;                       // threads
  constexpr int STAGES = SLOTS;
  constexpr int DIST = SLOTS == 2 ? 2 : SLOTS - 1;
  constexpr int H_TILE = TM * 128;                     // bytes of the hidden tile of a stage
  constexpr int STAGE = H_TILE + kUrWTileBytes;
  constexpr int TOKB = 2;                              // 32-token MFMA blocks of a wave
  constexpr int FEATB = (TM / 64) * 4 / WAVES;         // 32-feature MFMA blocks of a wave: 2, or 4 (all features)
  constexpr int HG = (TM / 32) / WAVES;                // hidden DMA groups (32 rows, 4 instructions) of a wave per stage
  constexpr int WI = 16 / WAVES;                       // weight DMA instructions (8 rows each) of a wave per stage
  constexpr int PER_TILE = 4 * HG + WI;                // DMA instructions of one wave per stage
  constexpr int AHEAD = DIST - 2;                      // tiles still in flight behind the one a step waits for
  constexpr int RPP = NT / 8;                          // token rows per epilogue pass (8 lanes per token)
  constexpr int NP = TM / RPP;                         // epilogue passes
  static_assert(FEATB == 2 || FEATB == 4, "wave tile");
  static_assert(2 * PER_TILE <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = FEATB == 4 ? w : w >> 1, wn = FEATB == 4 ? 0 : w & 1;
  // workgroup -> (m tile, n tile, layer).  Workgroups are dealt to the 8 XCDs round-robin: the n tiles of one m tile (they
  // read the same hidden rows) stay on one XCD's L2.
  const int n_tiles = 2 * P.r.num_kv_heads;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int n_tile = idx % n_tiles, m_tile = (idx / n_tiles) * 8 + xcd;
  if (m_tile >= P.m_tiles) return;
  const int64_t z = blockIdx.y;
  SvkDeltakvReconstructArgs a = P.r;
  if (a.father_table != nullptr) a.father_table += z * P.lb.father_table_stride_batch;
  a.k_cache += z * P.lb.kv_cache_stride_batch;
  a.v_cache += z * P.lb.kv_cache_stride_batch;
  if (a.k_norm_weight != nullptr) a.k_norm_weight += z * P.lb.k_norm_stride_batch;
  if (a.out_k_cache != nullptr) { a.out_k_cache += z * P.lb.out_cache_stride_batch; a.out_v_cache += z * P.lb.out_cache_stride_batch; }
  const int m0 = m_tile * TM;
  const bool is_v = n_tile >= a.num_kv_heads;
  const int h = is_v ? n_tile - a.num_kv_heads : n_tile;

  // ---- operand DMA: wave w brings hidden rows 32 w .. 32 w + 31 of a stage (4 instructions of 8 rows) and its share of
  //      the 128 weight rows (TM 128: rows 32 w ..; TM 256: rows 16 w ..)
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw);
  const char* hbase = reinterpret_cast<const char*>(P.u.hidden + z * P.u.hidden_stride_batch) - 3072;
  const char* wbase = reinterpret_cast<const char*>(P.u.weight + z * P.u.weight_stride_batch + (int64_t)n_tile * 128 * P.u.weight_stride) - 3072;
  uint32_t hoff[HG][4], woff[WI];
  {
    const uint32_t lrow = lane >> 3;
#pragma unroll
    for (int g = 0; g < HG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (w * HG + g) * 32 + i * 8 + (int)lrow;
        const uint32_t chunk = (lane & 7) ^ (((uint32_t)row >> 1) & 7u);      // position p of row r holds chunk p ^ ((r >> 1) & 7)
        const int tok = min(m0 + row, a.n - 1);
        hoff[g][i] = (uint32_t)tok * (uint32_t)(P.u.hidden_stride * 2) + (chunk << 4) + 3072u - 1024u * i;
      }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = w * (8 * WI) + i * 8 + (int)lrow;
      const uint32_t chunk = (lane & 7) ^ (((uint32_t)row >> 1) & 7u);
      woff[i] = (uint32_t)row * (uint32_t)(P.u.weight_stride * 2) + (chunk << 4) + 3072u - 1024u * i;
    }
  }
  auto issue = [&](int t) {
    const uint32_t tk = P.debug_same_k ? 0u : (uint32_t)t;
    const uint32_t sb = lds0 + (uint32_t)(t % STAGES) * STAGE;
#pragma unroll
    for (int g = 0; g < HG; ++g) {
      uint32_t hv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[i] = hoff[g][i] + tk * 128u;
      pa_dma4x16_off32(hv, hbase, sb + (uint32_t)(w * HG + g) * 4096u);
    }
    if constexpr (WI == 4) {
      uint32_t wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[i] = woff[i] + tk * 128u;
      pa_dma4x16_off32(wv, wbase, sb + H_TILE + (uint32_t)w * 4096u);
    } else {
      ur_dma2x16_off32(woff[0] + tk * 128u, woff[1] + tk * 128u, wbase, sb + H_TILE + (uint32_t)w * 2048u);
    }
  };
  const int KT = P.u.k / 64;
#pragma unroll
  for (int t = 0; t < DIST; ++t)
    if (t < KT) issue(t);

  // ---- the plan's per-token indices of the epilogue (4 passes of RPP tokens, 8 lanes per token) and the lane's bias
  //      values, requested behind the first tiles' DMA and in rounds (entry -> father row): all loads of a round go out
  //      together, the compiler's wait for them is pinned behind the round, not at their first use
  const int tq = threadIdx.x >> 3, p = (threadIdx.x & 7) * 8;
  int out_slot[NP], out_pos[NP], fidx[NP], fs[NP][KF];
  bool active[NP];
  int nc[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int n = m0 + ps * RPP + tq;
    active[ps] = n < a.n;
    nc[ps] = min(n, a.n - 1);
    out_slot[ps] = a.out_slots[nc[ps]];
    out_pos[ps] = a.out_pos[nc[ps]];
    fidx[ps] = nc[ps];
  }
  if (a.father_table != nullptr) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) fidx[ps] = a.father_index[nc[ps]];
  }
  uint2 bw[FEATB][4];
#pragma unroll
  for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) bw[fi][rq] = make_uint2(0u, 0u);
  if (P.u.bias != nullptr) {
    const uint16_t* bias = P.u.bias + z * P.u.bias_stride_batch + n_tile * 128 + wn * (FEATB * 32) + (lane >> 5) * 4;
#pragma unroll
    for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) bw[fi][rq] = *reinterpret_cast<const uint2*>(bias + fi * 32 + rq * 8);
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) asm volatile("" : "+v"(out_slot[ps]), "+v"(out_pos[ps]), "+v"(fidx[ps]));
#pragma unroll
  for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) asm volatile("" : "+v"(bw[fi][rq].x), "+v"(bw[fi][rq].y));
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    if (!active[ps]) { out_slot[ps] = -1; out_pos[ps] = -1; }
  }
  {
    const bool tab = a.father_table != nullptr;
    const int32_t* fbase = tab ? a.father_table : a.father_slots;
    const int64_t fstride = tab ? a.father_table_stride : a.father_stride;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int32_t* fathers = fbase + (int64_t)max(fidx[ps], 0) * fstride;
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) fs[ps][kk] = fathers[min(kk, a.k_fathers - 1)];
    }
    if (tab) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int kk = 0; kk < KF; ++kk) fs[ps][kk] = max(fs[ps][kk], 0);
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int ps = 0; ps < NP; ++ps)
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) asm volatile("" : "+v"(fs[ps][kk]));

  // ---- operand reads: lane (row l & 31 of a 32-row block, 16-byte chunk 2 s + (l >> 5)) through the swizzle ((row >> 1) & 7 = (l >> 1) & 7)
  uint32_t fa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fa[s] = (uint32_t)(lane & 31) * 128u + ((uint32_t)((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
  const uint32_t h_rows = (uint32_t)wm * 64u * 128u, w_rows = H_TILE + (uint32_t)wn * (FEATB * 32u) * 128u;
  f32x16_t acc[FEATB][TOKB];
#pragma unroll
  for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
    for (int ti = 0; ti < TOKB; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fi][ti][r] = 0.f;

  // ---- main loop, software-pipelined around ONE barrier in the middle of a step: the operand fragments of k-substeps 0, 1
  //      of tile t are in registers when the step starts; their MFMAs cover the reads of substeps 2, 3; then tile t + 1 is
  //      waited for, the barrier retires tile t - 1's ring slot for the DMA of tile t + STAGES - 1, and the MFMAs of
  //      substeps 2, 3 cover the reads of tile t + 1's substeps 0, 1.
  constexpr int NF = FEATB + TOKB;                     // operand fragments of a k-substep: features first, then tokens
  auto read4 = [&](uint32_t sb, int s, bf16x8_t (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < FEATB; ++i) f[i] = lds_frag(sb + w_rows + i * (32 * 128) + fa[s]);
#pragma unroll
    for (int i = 0; i < TOKB; ++i) f[FEATB + i] = lds_frag(sb + h_rows + i * (32 * 128) + fa[s]);
  };
  auto mfma4 = [&](const bf16x8_t (&f)[NF]) {
#pragma unroll
    for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
      for (int ti = 0; ti < TOKB; ++ti)
        acc[fi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[fi], f[FEATB + ti], acc[fi][ti], 0, 0, 0);
  };
  bf16x8_t f0[NF], f1[NF], g0[NF], g1[NF];
  // tile 0 has landed: the prologue put STAGES - 1 tiles in flight
  if (DIST == 3 && KT >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
  else if (KT >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read4(lds0, 0, f0);
  read4(lds0, 1, f1);
  for (int t = 0; t < KT; ++t) {
    const uint32_t sb = lds0 + (uint32_t)(t % STAGES) * STAGE;
    const uint32_t sbn = lds0 + (uint32_t)((t + 1) % STAGES) * STAGE;
    SVK_UR_STAMP(t, 0);
    read4(sb, 2, g0);
    mfma4(f0);
    read4(sb, 3, g1);
    mfma4(f1);
    SVK_UR_STAMP(t, 1);
    if (t + 1 < KT) {
      // tile t + 1 has landed; still in flight behind it: AHEAD tiles
      if (AHEAD == 1 && t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SVK_UR_STAMP(t, 2);
    __syncthreads();                                  // everyone's tile t + 1 is in LDS, everyone has read all of tile t
    SVK_UR_STAMP(t, 3);
    // (dealing these 8 instructions over the step, two at a time between the MFMA groups, moves their 400-550 cycles
    //  from here into the MFMA phases and leaves the step as long: tools/ur_timing.py, 1716 -> 1700 cycles)
    if (t + DIST < KT) issue(t + DIST);
    SVK_UR_STAMP(t, 4);
    if (t + 1 < KT) read4(sbn, 0, f0);
    mfma4(g0);
    if (t + 1 < KT) read4(sbn, 1, f1);
    mfma4(g1);
    SVK_UR_STAMP(t, 5);
  }
  // (the barrier of the last step is behind every operand read: the ring becomes the delta tile)

  // ---- the epilogue's gathers (father rows, cos | sin rows): a rolling window of two passes in flight
  auto up8 = [](const uint4& v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
  };
  auto pk8 = [](const float (&f)[8]) {
    return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
  };
  const uint16_t* cache = is_v ? a.v_cache : a.k_cache;
  const float* cs_tab = reinterpret_cast<const float*>(a.cos_sin);
  bool act[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) act[ps] = active[ps] && out_slot[ps] >= 0 && out_pos[ps] >= 0;
  auto gather = [&](int ps, uint4 (&y1)[KF], uint4 (&y2)[KF], float4 (&cs)[4]) {
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) {
      const int f = (act[ps] && kk < a.k_fathers) ? fs[ps][kk] : 0;
      const int64_t base = (int64_t)f * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      y1[kk] = *reinterpret_cast<const uint4*>(cache + base);
      y2[kk] = *reinterpret_cast<const uint4*>(cache + base + HD2);
    }
    const float* cp = cs_tab + (int64_t)(act[ps] ? out_pos[ps] : 0) * a.cos_stride + p;
    cs[0] = *reinterpret_cast<const float4*>(cp);
    cs[1] = *reinterpret_cast<const float4*>(cp + 4);
    cs[2] = *reinterpret_cast<const float4*>(cp + HD2);
    cs[3] = *reinterpret_cast<const float4*>(cp + HD2 + 4);
  };
  uint4 ya1[KF], ya2[KF], yb1[KF], yb2[KF];
  float4 csa[4], csb[4];
  gather(0, ya1, ya2, csa);
  gather(1, yb1, yb2, csb);

  // ---- delta tile: C^T[feature][token] -> bf16(acc + bias) at [token][feature]
#pragma unroll
  for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
    for (int ti = 0; ti < TOKB; ++ti) {
      const int tok = wm * 64 + ti * 32 + (lane & 31);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int f0c = wn * (FEATB * 32) + fi * 32 + rq * 8 + (lane >> 5) * 4;
        const float b0 = bf16_lo(bw[fi][rq].x), b1 = bf16_hi(bw[fi][rq].x), b2 = bf16_lo(bw[fi][rq].y), b3 = bf16_hi(bw[fi][rq].y);
        const uint2 o = make_uint2(pack2_bf16(acc[fi][ti][rq * 4] + b0, acc[fi][ti][rq * 4 + 1] + b1),
                                   pack2_bf16(acc[fi][ti][rq * 4 + 2] + b2, acc[fi][ti][rq * 4 + 3] + b3));
        *reinterpret_cast<uint2*>(lds_raw + tok * kUrDeltaRow + f0c * 2) = o;
      }
    }
  __syncthreads();

  // ---- reconstruction of the tile's tokens for head h of K (or V)
  const float inv = 1.0f / (float)a.k_fathers;
  auto finish = [&](int ps, const uint4 (&y1)[KF], const uint4 (&y2)[KF], const float4 (&cs)[4]) {
    const int row = ps * RPP + tq;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) {
      if (kk < a.k_fathers) {
        float f1[8], f2[8];
        up8(y1[kk], f1);
        up8(y2[kk], f2);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] += f1[i]; s2[i] += f2[i]; }
      }
    }
    float d1[8], d2[8], k1[8], k2[8];
    up8(*reinterpret_cast<const uint4*>(lds_raw + row * kUrDeltaRow + p * 2), d1);
    up8(*reinterpret_cast<const uint4*>(lds_raw + row * kUrDeltaRow + (p + HD2) * 2), d2);
#pragma unroll
    for (int i = 0; i < 8; ++i) { k1[i] = d1[i] + s1[i] * inv; k2[i] = d2[i] + s2[i] * inv; }
    if (!act[ps]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) k1[i] = k2[i] = 0.f;
    }
    if (!is_v && a.k_norm_weight != nullptr) {       // (uniform: every lane of the wave takes the shuffles)
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += k1[i] * k1[i] + k2[i] * k2[i];
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      const float rstd = rsqrtf(ss / (float)D + a.k_norm_eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) { k1[i] = k1[i] * rstd * a.k_norm_weight[p + i]; k2[i] = k2[i] * rstd * a.k_norm_weight[p + HD2 + i]; }
    }
    if (!act[ps]) return;
    float o1[8], o2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { o1[i] = k1[i]; o2[i] = k2[i]; }
    if (!is_v) {
      const float c[8] = {cs[0].x, cs[0].y, cs[0].z, cs[0].w, cs[1].x, cs[1].y, cs[1].z, cs[1].w};
      const float sn[8] = {cs[2].x, cs[2].y, cs[2].z, cs[2].w, cs[3].x, cs[3].y, cs[3].z, cs[3].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) { o1[i] = k1[i] * c[i] - k2[i] * sn[i]; o2[i] = k2[i] * c[i] + k1[i] * sn[i]; }
    }
    uint16_t* dst = is_v ? a.v_cache : a.k_cache;
    int64_t ob = (int64_t)out_slot[ps] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
    if (a.out_k_cache != nullptr) {                  // straight into the entry's row of the attention view
      const int n = m0 + row;
      const int64_t vrow = (int64_t)(n / a.out_entries_per_row) * a.out_view_width + a.out_view_offset + n % a.out_entries_per_row;
      dst = is_v ? a.out_v_cache : a.out_k_cache;
      ob = vrow * a.out_slot_stride + (int64_t)h * a.out_head_stride + p;
    }
    *reinterpret_cast<uint4*>(dst + ob) = pk8(o1);
    *reinterpret_cast<uint4*>(dst + ob + HD2) = pk8(o2);
  };
#pragma unroll
  for (int ps = 0; ps < NP; ps += 2) {
    finish(ps, ya1, ya2, csa);
    if (ps + 2 < NP) gather(ps + 2, ya1, ya2, csa);
    finish(ps + 1, yb1, yb2, csb);
    if (ps + 3 < NP) gather(ps + 3, yb1, yb2, csb);
  }
}

// ---- the 256 x 256 form: 256 tokens x TWO heads (256 features) per workgroup, 8 waves (2 x 4: a wave = 128 tokens x 64
// features, 128 accumulator registers, 6 operand reads per 8 MFMAs).  What the smaller tiles are short of is operand
// delivery - ~25 bytes per clock and CU arrive in LDS whatever the pipeline depth, i.e. a 128 x 128 tile (32 KiB per
// 64-deep step for 512 MFMA cycles per SIMD) keeps the matrix cores 40 % busy; this tile moves 64 KiB per step for 2048.
// (That was the idea.  Measured: 2.3 us per step = 43 % of the step's 2048 MFMA cycles even with cache-resident operands,
// 122 against 107 us for the (256, 8) form at 2 x 8192 tokens - developer form, SVK_UP_RECON_TM=512.)
// Two ring slots of 64 KiB: the DMA of tile t + 2 goes into tile t's own slot once every wave holds the fragments of its
// last two k-substeps in registers (the mid-step barrier).  The plan's indices are read behind the main loop (the
// registers belong to the accumulators before), the whole delta tile [256][256] goes to LDS at once (135 KiB) and the
// two heads are finished one after the other.
constexpr int kUr2DeltaRow = 528;                    // bytes per token row of the 256-feature delta tile (512 + 16)
constexpr int kUr2Lds = 256 * kUr2DeltaRow;          // 135168 >= the operand ring (2 x 64 KiB)

template <int KF>
__global__ void __launch_bounds__(512) up_recon256_kernel(const UpReconParams P) {
  constexpr int D = 128, HD2 = 64, TM = 256;
  constexpr int H_TILE = 256 * 128, STAGE = 2 * H_TILE;
  constexpr int TOKB = 4, FEATB = 2, NF = FEATB + TOKB;
  constexpr int RPP = 64, NP = 4;                      // epilogue: 4 passes of 64 tokens, 8 lanes per token
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int n_tiles = P.r.num_kv_heads;                // pairs of heads: K pairs first, then V pairs
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int n_tile = idx % n_tiles, m_tile = (idx / n_tiles) * 8 + xcd;
  if (m_tile >= P.m_tiles) return;
  const int64_t z = blockIdx.y;
  SvkDeltakvReconstructArgs a = P.r;
  if (a.father_table != nullptr) a.father_table += z * P.lb.father_table_stride_batch;
  a.k_cache += z * P.lb.kv_cache_stride_batch;
  a.v_cache += z * P.lb.kv_cache_stride_batch;
  if (a.k_norm_weight != nullptr) a.k_norm_weight += z * P.lb.k_norm_stride_batch;
  if (a.out_k_cache != nullptr) { a.out_k_cache += z * P.lb.out_cache_stride_batch; a.out_v_cache += z * P.lb.out_cache_stride_batch; }
  const int m0 = m_tile * TM;
  const bool is_v = 2 * n_tile >= a.num_kv_heads;
  const int h0 = (2 * n_tile) % a.num_kv_heads;

  // ---- operand DMA: wave w brings rows 32 w .. 32 w + 31 of both tiles of a stage (4 + 4 instructions of 8 rows)
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw);
  const char* hbase = reinterpret_cast<const char*>(P.u.hidden + z * P.u.hidden_stride_batch) - 3072;
  const char* wbase = reinterpret_cast<const char*>(P.u.weight + z * P.u.weight_stride_batch + (int64_t)n_tile * 256 * P.u.weight_stride) - 3072;
  uint32_t hoff[4], woff[4];
  {
    const uint32_t lrow = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = w * 32 + i * 8 + (int)lrow;
      const uint32_t chunk = (lane & 7) ^ (((uint32_t)row >> 1) & 7u);      // position p of row r holds chunk p ^ ((r >> 1) & 7)
      const int tok = min(m0 + row, a.n - 1);
      hoff[i] = (uint32_t)tok * (uint32_t)(P.u.hidden_stride * 2) + (chunk << 4) + 3072u - 1024u * i;
      woff[i] = (uint32_t)row * (uint32_t)(P.u.weight_stride * 2) + (chunk << 4) + 3072u - 1024u * i;
    }
  }
  auto issue = [&](int t) {
    const uint32_t tk = P.debug_same_k ? 0u : (uint32_t)t;
    const uint32_t sb = lds0 + (uint32_t)(t & 1) * STAGE + (uint32_t)w * 4096u;
    uint32_t hv[4], wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { hv[i] = hoff[i] + tk * 128u; wv[i] = woff[i] + tk * 128u; }
    pa_dma4x16_off32(hv, hbase, sb);
    pa_dma4x16_off32(wv, wbase, sb + H_TILE);
  };
  const int KT = P.u.k / 64;
  issue(0);
  if (KT > 1) issue(1);

  uint32_t fa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fa[s] = (uint32_t)(lane & 31) * 128u + ((uint32_t)((2 * s + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
  const uint32_t h_rows = (uint32_t)wm * 128u * 128u, w_rows = H_TILE + (uint32_t)wn * 64u * 128u;
  f32x16_t acc[FEATB][TOKB];
#pragma unroll
  for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
    for (int ti = 0; ti < TOKB; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fi][ti][r] = 0.f;
  auto read6 = [&](uint32_t sb, int s, bf16x8_t (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < FEATB; ++i) f[i] = lds_frag(sb + w_rows + i * (32 * 128) + fa[s]);
#pragma unroll
    for (int i = 0; i < TOKB; ++i) f[FEATB + i] = lds_frag(sb + h_rows + i * (32 * 128) + fa[s]);
  };
  auto mfma8 = [&](const bf16x8_t (&f)[NF]) {
#pragma unroll
    for (int ti = 0; ti < TOKB; ++ti)
#pragma unroll
      for (int fi = 0; fi < FEATB; ++fi)
        acc[fi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[fi], f[FEATB + ti], acc[fi][ti], 0, 0, 0);
  };
  bf16x8_t f0[NF], f1[NF], g0[NF], g1[NF];
  if (KT > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read6(lds0, 0, f0);
  read6(lds0, 1, f1);
  for (int t = 0; t < KT; ++t) {
    const uint32_t sb = lds0 + (uint32_t)(t & 1) * STAGE, sbn = lds0 + (uint32_t)((t + 1) & 1) * STAGE;
    read6(sb, 2, g0);
    mfma8(f0);
    read6(sb, 3, g1);
    mfma8(f1);
    if (t + 1 < KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t + 1 (the only one in flight) has landed
    __syncthreads();                                  // ... for everyone, and everyone holds the rest of tile t in registers
    if (t + 2 < KT) issue(t + 2);                    // into tile t's own slot
    if (t + 1 < KT) read6(sbn, 0, f0);
    mfma8(g0);
    if (t + 1 < KT) read6(sbn, 1, f1);
    mfma8(g1);
  }
  __syncthreads();                                    // (the last step's barrier is in front of no DMA; this one retires the ring)

  // ---- delta tile [256 tokens][256 features]: bf16(acc + bias)
  {
    uint2 bw[FEATB][4];
#pragma unroll
    for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) bw[fi][rq] = make_uint2(0u, 0u);
    if (P.u.bias != nullptr) {
      const uint16_t* bias = P.u.bias + z * P.u.bias_stride_batch + n_tile * 256 + wn * 64 + (lane >> 5) * 4;
#pragma unroll
      for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) bw[fi][rq] = *reinterpret_cast<const uint2*>(bias + fi * 32 + rq * 8);
    }
#pragma unroll
    for (int fi = 0; fi < FEATB; ++fi)
#pragma unroll
      for (int ti = 0; ti < TOKB; ++ti) {
        const int tok = wm * 128 + ti * 32 + (lane & 31);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int f0c = wn * 64 + fi * 32 + rq * 8 + (lane >> 5) * 4;
          const float b0 = bf16_lo(bw[fi][rq].x), b1 = bf16_hi(bw[fi][rq].x), b2 = bf16_lo(bw[fi][rq].y), b3 = bf16_hi(bw[fi][rq].y);
          const uint2 o = make_uint2(pack2_bf16(acc[fi][ti][rq * 4] + b0, acc[fi][ti][rq * 4 + 1] + b1),
                                     pack2_bf16(acc[fi][ti][rq * 4 + 2] + b2, acc[fi][ti][rq * 4 + 3] + b3));
          *reinterpret_cast<uint2*>(lds_raw + tok * kUr2DeltaRow + f0c * 2) = o;
        }
      }
  }

  // ---- the plan's per-token indices (4 passes of 64 tokens, 8 lanes per token), two rounds: entry -> father row
  const int tq = threadIdx.x >> 3, p = (threadIdx.x & 7) * 8;
  int out_slot[NP], out_pos[NP], fidx[NP], fs[NP][KF];
  bool act[NP];
  {
    int nc[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int n = m0 + ps * RPP + tq;
      act[ps] = n < a.n;
      nc[ps] = min(n, a.n - 1);
      out_slot[ps] = a.out_slots[nc[ps]];
      out_pos[ps] = a.out_pos[nc[ps]];
      fidx[ps] = nc[ps];
    }
    if (a.father_table != nullptr) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) fidx[ps] = a.father_index[nc[ps]];
    }
    const bool tab = a.father_table != nullptr;
    const int32_t* fbase = tab ? a.father_table : a.father_slots;
    const int64_t fstride = tab ? a.father_table_stride : a.father_stride;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int32_t* fathers = fbase + (int64_t)max(fidx[ps], 0) * fstride;
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) fs[ps][kk] = fathers[min(kk, a.k_fathers - 1)];
    }
    if (tab) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int kk = 0; kk < KF; ++kk) fs[ps][kk] = max(fs[ps][kk], 0);
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) act[ps] = act[ps] && out_slot[ps] >= 0 && out_pos[ps] >= 0;
  }
  __syncthreads();                                    // the delta tile is complete

  // ---- reconstruction, one head after the other
  auto up8 = [](const uint4& v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
  };
  auto pk8 = [](const float (&f)[8]) {
    return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
  };
  const uint16_t* cache = is_v ? a.v_cache : a.k_cache;
  const float* cs_tab = reinterpret_cast<const float*>(a.cos_sin);
  const float inv = 1.0f / (float)a.k_fathers;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = h0 + hh;
    auto gather = [&](int ps, uint4 (&y1)[KF], uint4 (&y2)[KF], float4 (&cs)[4]) {
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) {
        const int f = (act[ps] && kk < a.k_fathers) ? fs[ps][kk] : 0;
        const int64_t base = (int64_t)f * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
        y1[kk] = *reinterpret_cast<const uint4*>(cache + base);
        y2[kk] = *reinterpret_cast<const uint4*>(cache + base + HD2);
      }
      const float* cp = cs_tab + (int64_t)(act[ps] ? out_pos[ps] : 0) * a.cos_stride + p;
      cs[0] = *reinterpret_cast<const float4*>(cp);
      cs[1] = *reinterpret_cast<const float4*>(cp + 4);
      cs[2] = *reinterpret_cast<const float4*>(cp + HD2);
      cs[3] = *reinterpret_cast<const float4*>(cp + HD2 + 4);
    };
    auto finish = [&](int ps, const uint4 (&y1)[KF], const uint4 (&y2)[KF], const float4 (&cs)[4]) {
      const int row = ps * RPP + tq;
      float s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KF; ++kk) {
        if (kk < a.k_fathers) {
          float f1[8], f2[8];
          up8(y1[kk], f1);
          up8(y2[kk], f2);
#pragma unroll
          for (int i = 0; i < 8; ++i) { s1[i] += f1[i]; s2[i] += f2[i]; }
        }
      }
      float d1[8], d2[8], k1[8], k2[8];
      up8(*reinterpret_cast<const uint4*>(lds_raw + row * kUr2DeltaRow + (hh * 128 + p) * 2), d1);
      up8(*reinterpret_cast<const uint4*>(lds_raw + row * kUr2DeltaRow + (hh * 128 + p + HD2) * 2), d2);
#pragma unroll
      for (int i = 0; i < 8; ++i) { k1[i] = d1[i] + s1[i] * inv; k2[i] = d2[i] + s2[i] * inv; }
      if (!act[ps]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) k1[i] = k2[i] = 0.f;
      }
      if (!is_v && a.k_norm_weight != nullptr) {     // (uniform: every lane of the wave takes the shuffles)
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += k1[i] * k1[i] + k2[i] * k2[i];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        const float rstd = rsqrtf(ss / (float)D + a.k_norm_eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) { k1[i] = k1[i] * rstd * a.k_norm_weight[p + i]; k2[i] = k2[i] * rstd * a.k_norm_weight[p + HD2 + i]; }
      }
      if (!act[ps]) return;
      float o1[8], o2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { o1[i] = k1[i]; o2[i] = k2[i]; }
      if (!is_v) {
        const float c[8] = {cs[0].x, cs[0].y, cs[0].z, cs[0].w, cs[1].x, cs[1].y, cs[1].z, cs[1].w};
        const float sn[8] = {cs[2].x, cs[2].y, cs[2].z, cs[2].w, cs[3].x, cs[3].y, cs[3].z, cs[3].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { o1[i] = k1[i] * c[i] - k2[i] * sn[i]; o2[i] = k2[i] * c[i] + k1[i] * sn[i]; }
      }
      uint16_t* dst = is_v ? a.v_cache : a.k_cache;
      int64_t ob = (int64_t)out_slot[ps] * a.kv_slot_stride + (int64_t)h * a.kv_head_stride + p;
      if (a.out_k_cache != nullptr) {                // straight into the entry's row of the attention view
        const int n = m0 + row;
        const int64_t vrow = (int64_t)(n / a.out_entries_per_row) * a.out_view_width + a.out_view_offset + n % a.out_entries_per_row;
        dst = is_v ? a.out_v_cache : a.out_k_cache;
        ob = vrow * a.out_slot_stride + (int64_t)h * a.out_head_stride + p;
      }
      *reinterpret_cast<uint4*>(dst + ob) = pk8(o1);
      *reinterpret_cast<uint4*>(dst + ob + HD2) = pk8(o2);
    };
    uint4 ya1[KF], ya2[KF], yb1[KF], yb2[KF];
    float4 csa[4], csb[4];
    gather(0, ya1, ya2, csa);
    gather(1, yb1, yb2, csb);
#pragma unroll
    for (int ps = 0; ps < NP; ps += 2) {
      finish(ps, ya1, ya2, csa);
      if (ps + 2 < NP) gather(ps + 2, ya1, ya2, csa);
      finish(ps + 1, yb1, yb2, csb);
      if (ps + 3 < NP) gather(ps + 3, yb1, yb2, csb);
    }
  }
}

}  // namespace
}  // namespace svk

extern "C" int svk_deltakv_up_reconstruct(const SvkDeltakvUpReconArgs* u, const SvkDeltakvReconstructArgs* first,
                                          const SvkDeltakvReconstructBatch* b, svk_stream_t stream) {
  using namespace svk;
  SVK_REQUIRE(u != nullptr && first != nullptr && b != nullptr, SVK_ERR_VALUE, "svk_deltakv_up_reconstruct: null args");
  SVK_REQUIRE(u->hidden != nullptr && u->weight != nullptr, SVK_ERR_VALUE, "svk_deltakv_up_reconstruct: hidden / weight missing");
  SVK_REQUIRE(b->n_batch >= 1 && b->n_batch <= 65535, SVK_ERR_VALUE, "svk_deltakv_up_reconstruct: bad batch description");
  SVK_REQUIRE(first->head_dim == 128 && first->num_kv_heads >= 1 && first->num_kv_heads <= 8, SVK_ERR_LAYOUT,
              "svk_deltakv_up_reconstruct: head_dim 128 and 1..8 KV heads, got %d / %d", first->head_dim, first->num_kv_heads);
  SVK_REQUIRE(first->raw_k_cache != 0 && first->store_raw_k == 0 && first->cos_dtype == SVK_DTYPE_F32, SVK_ERR_LAYOUT,
              "svk_deltakv_up_reconstruct: un-rotated father keys, rotated output and fp32 cos|sin (the static decode path)");
  SVK_REQUIRE(first->k_fathers >= 1 && first->k_fathers <= 4, SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: 1..4 fathers per token, got %d",
              first->k_fathers);
  SVK_REQUIRE(u->k >= 64 && u->k % 64 == 0, SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: hidden features must be a multiple of 64, got %d", u->k);
  SVK_REQUIRE(u->hidden_stride % 8 == 0 && u->weight_stride % 8 == 0 && u->hidden_stride_batch % 8 == 0 && u->weight_stride_batch % 8 == 0 &&
                  reinterpret_cast<uintptr_t>(u->hidden) % 16 == 0 && reinterpret_cast<uintptr_t>(u->weight) % 16 == 0,
              SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: hidden / weight rows must keep 16-byte alignment");
  SVK_REQUIRE(u->bias == nullptr || (reinterpret_cast<uintptr_t>(u->bias) % 8 == 0 && u->bias_stride_batch % 4 == 0), SVK_ERR_LAYOUT,
              "svk_deltakv_up_reconstruct: bias rows must keep 8-byte alignment");
  SVK_REQUIRE(first->kv_slot_stride % 8 == 0 && first->kv_head_stride % 8 == 0 && first->cos_stride % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(first->cos_sin) % 16 == 0,
              SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: cache / cos|sin rows must keep 16-byte alignment");
  if (first->out_k_cache != nullptr)
    SVK_REQUIRE(first->out_v_cache != nullptr && first->out_slot_stride % 8 == 0 && first->out_head_stride % 8 == 0 &&
                    first->out_entries_per_row > 0 && (b->n_batch == 1 || b->out_cache_stride_batch % 8 == 0),
                SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: bad view destination");
  if (first->n <= 0) return SVK_OK;
  // 32-bit DMA offsets: a layer's hidden rows and a head's weight rows stay below 4 GiB
  SVK_REQUIRE((int64_t)first->n * u->hidden_stride * 2 + 4096 < (int64_t)1 << 32 && (int64_t)128 * u->weight_stride * 2 + 4096 < (int64_t)1 << 32,
              SVK_ERR_LAYOUT, "svk_deltakv_up_reconstruct: operand rows beyond 32-bit offsets");
  UpReconParams p;
  p.u = *u;
  p.r = *first;
  p.lb = *b;
  p.debug_same_k = getenv("SVK_UP_RECON_SAME_K") != nullptr && atoi(getenv("SVK_UP_RECON_SAME_K")) != 0;
  const int n_tiles = 2 * first->num_kv_heads;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // more than one round of 128-token tiles: the two-slot form, two workgroups per CU (SVK_UP_RECON_TM = 128 / 1282 / 256 /
  // 2564 / 512 forces the (128, 4) form with four / two ring slots, the (256, 8) / (256, 4) form, the two-head 256 x 256 tile)
  const char* env = getenv("SVK_UP_RECON_TM");
  const int forced = env ? atoi(env) : 0;
  const int tiles128 = ((first->n + 127) / 128) * n_tiles * b->n_batch;
  int form = forced ? forced : (tiles128 > 256 ? 1282 : 128);
  if (form == 512 && first->num_kv_heads % 2 != 0) form = 256;       // the two-head tile pairs heads
  const int tm = (form == 128 || form == 1282) ? 128 : 256;
  p.m_tiles = (first->n + tm - 1) / tm;
  if (form == 512) {
    const dim3 grid2(8u * (unsigned)first->num_kv_heads * (unsigned)((p.m_tiles + 7) / 8), (unsigned)b->n_batch);
    static const bool attr2_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_recon256_kernel<4>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kUr2Lds) == hipSuccess;
    (void)attr2_ok;
    hipLaunchKernelGGL((up_recon256_kernel<4>), grid2, dim3(512), (size_t)kUr2Lds, s, p);
    return check_launch("svk_deltakv_up_reconstruct");
  }
  const dim3 grid(8u * (unsigned)n_tiles * (unsigned)((p.m_tiles + 7) / 8), (unsigned)b->n_batch);
  const size_t shm = form == 128 ? (size_t)4 * (128 * 128 + kUrWTileBytes)
                                 : (form == 1282 ? (size_t)2 * (128 * 128 + kUrWTileBytes) : (size_t)3 * (256 * 128 + kUrWTileBytes));
#define SVK_UR_LAUNCH(TM_, W_, S_)                                                                                     \
  do {                                                                                                                 \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_recon_kernel<TM_, W_, 4, S_>),   \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess; \
    (void)attr_ok;                                                                                                     \
    hipLaunchKernelGGL((up_recon_kernel<TM_, W_, 4, S_>), grid, dim3(W_ * 64), shm, s, p);                             \
  } while (0)
  if (form == 128) SVK_UR_LAUNCH(128, 4, 4);
  else if (form == 1282) SVK_UR_LAUNCH(128, 4, 2);
  else if (form == 256) SVK_UR_LAUNCH(256, 8, 3);
  else SVK_UR_LAUNCH(256, 4, 3);
#undef SVK_UR_LAUNCH
  return check_launch("svk_deltakv_up_reconstruct");
}
