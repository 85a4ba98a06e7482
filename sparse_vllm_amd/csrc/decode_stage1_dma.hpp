// Stage 1 of the split-KV GQA decode, fourth generation ("v4"): the K tile travels HBM -> LDS by LDS-DMA.
// Included by decode_attention.hip (inside namespace svk::{anonymous}); head_dim 128 only.
//
// Why: in v3 the K rows are loaded straight into the MFMA B-operand layout (lane = (token, 8-element chunk)), i.e.
// 16 rows x 64 B per wave-instruction.  tools/probe_gather.hip shows that pattern capped at 5.8 TB/s (and worse with
// the non-temporal policy because the two halves of a 128-B line are fetched by different instructions), and with one
// wave per KV head and one workgroup per CU the wave holds only K(i+1) + V(i) = 16 KiB in flight: one memory latency
// per tile.  tools/probe_kdma.hip: the same gather with K brought in as whole 256-B head rows by
// `global_load_lds_dwordx4` (4 rows x 256 B per wave-instruction, lane-linear in LDS) and V in registers reaches
// 6.9-7.1 TB/s once ~128 KiB per CU are in flight.
//
// Structure (one workgroup = (batch lane, block_seq block), one wave per KV head - unchanged):
//   * K ring in LDS: S stages x 32 tokens x 256 B per wave, filled by LDS-DMA S-1 tiles ahead.  The DMA needs no
//     registers, so the depth is bounded by LDS only.  DMA instruction k of a tile carries tokens {rq*8 + k}
//     (rq = lane/16): a lane's 8 instructions read 8 consecutive slot ids (two ds_read_b128).  LDS row 4k+rq,
//     16-byte position p holds chunk p ^ n (n = the token's MFMA column) - the swizzle is applied on the SOURCE
//     address (the DMA destination is lane-linear), so the B-operand ds_read_b128 (row stride 256 B) is
//     conflict-free for every 16-lane service group.
//   * V: registers, VB tiles deep (v3's layout: lane (n, jq) = head dims n*8.. of tokens jq*8+e), non-temporal.
//   * every vector-memory instruction of the loop is inline asm with hand-counted `s_waitcnt vmcnt(N)`: hipcc
//     drains vmcnt(0) in front of any ds_read / load use while an LDS-DMA it knows of is in flight, which would
//     serialise the ring.  Loads return in order, so "K(i) has landed" = at most the (known) younger ops are
//     still outstanding.  Compiler-issued stores inside the loop (token scores) only make a wait stricter.
//   * the block's slot ids and old token scores are staged in LDS once (<= kV4MaxRange tokens per workgroup); the
//     score combine stays v3's (one owner thread per token column, no atomics), reading the old value from LDS.
//   * Q fragments live in registers (the 32 K registers of v3 are gone), P.V and the softmax are v3's.

constexpr int kV4MaxRange = 2112;                  // tokens per workgroup (block_seq) the LDS staging is sized for
constexpr int kV4StageBytes = kTileTokens * 256;   // one K stage of one wave (32 tokens x 256 B)
constexpr int kV4ScoreChunk = 128;                 // tokens between two score-combine barriers (LDS: two workgroups per CU)

struct Stage1V4Lds {
  // byte layout of the dynamic LDS: slot ids | old scores | score partials | per-wave P tiles | per-wave K rings
  int slot_off, old_off, spart_off, p_off, ring_off, total;
  __host__ __device__ Stage1V4Lds(int range32, int hkv, int jq, int stages, bool headmax) {
    slot_off = 0;
    old_off = slot_off + range32 * 4;
    spart_off = old_off + (headmax ? range32 * 4 : 0);
    p_off = spart_off + (headmax ? kV4ScoreChunk * hkv * jq * 4 : 0);
    ring_off = (p_off + hkv * kPFloats * 4 + 255) & ~255;
    total = ring_off + hkv * stages * kV4StageBytes;
  }
};

__device__ __forceinline__ void v4_wait_vm(int n) {     // n: wave-uniform count of ops that may stay outstanding
  // s_waitcnt takes an immediate: dispatch on n / 8 (every count this kernel needs is a multiple of 8; a smaller
  // immediate is always safe)
  switch (n >> 3) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
  }
}

// 8 LDS-DMA instructions (1 KiB each) of one K stage.  M0 = LDS byte address of the instruction's 1 KiB piece; it is
// compiler-reserved, so it is saved and restored inside the statement.  `s_nop 0`: SALU write of M0 -> LDS-DMA read.
template <bool NT, bool OFF32, typename A>
__device__ __forceinline__ void v4_dma8(const A (&src)[8], const char* base, uint32_t lds_addr) {
  uint32_t keep;
#define SVK_DMA8(POL_, SADDR_)                                                                                    \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %10\n\ts_nop 0\n\t"                                            \
               "global_load_lds_dwordx4 %1, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %2, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %3, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %4, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %5, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %6, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %7, " SADDR_ POL_ "\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"            \
               "global_load_lds_dwordx4 %8, " SADDR_ POL_ "\n\ts_mov_b32 m0, %0"                                  \
               : "=&s"(keep)                                                                                      \
               : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]),       \
                 "v"(src[7]), "s"(base), "s"(lds_addr)                                                            \
               : "memory")
  if constexpr (OFF32) {
    if constexpr (NT) SVK_DMA8(" nt", "%9"); else SVK_DMA8("", "%9");
  } else {
    if constexpr (NT) SVK_DMA8(" nt", "off"); else SVK_DMA8("", "off");
  }
#undef SVK_DMA8
}

// 8 non-temporal 16-byte register loads (one V tile of this lane).  Early-clobber outputs: the statement writes
// them before it has read every address.  The data is NOT there when the statement ends - see v4_wait_vm.
template <bool OFF32, typename A>
__device__ __forceinline__ void v4_ldv8(u32x4_t (&v)[8], const A (&src)[8], const char* base) {
#define SVK_LDV8(SADDR_)                                                                                          \
  asm volatile("global_load_dwordx4 %0, %8, " SADDR_ " nt\n\tglobal_load_dwordx4 %1, %9, " SADDR_ " nt\n\t"       \
               "global_load_dwordx4 %2, %10, " SADDR_ " nt\n\tglobal_load_dwordx4 %3, %11, " SADDR_ " nt\n\t"     \
               "global_load_dwordx4 %4, %12, " SADDR_ " nt\n\tglobal_load_dwordx4 %5, %13, " SADDR_ " nt\n\t"     \
               "global_load_dwordx4 %6, %14, " SADDR_ " nt\n\tglobal_load_dwordx4 %7, %15, " SADDR_ " nt"         \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])  \
               : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]),       \
                 "v"(src[7]), "s"(base)                                                                           \
               : "memory")
  if constexpr (OFF32) SVK_LDV8("%16"); else SVK_LDV8("off");
#undef SVK_LDV8
}

template <int G, int MODE, int S, int VB, bool NTK, bool OFF32>
__global__ void __launch_bounds__(512)
decode_stage1_kernel_v4(const SvkFlashDecodeStage1Args a) {
  constexpr int D = 128;
  using C = Stage1Cfg<D, G>;
  constexpr int NC = C::NC, JQ = C::JQ;
  constexpr int DW = D / 8;                       // 16-byte segments per head row
  constexpr int score_mode = MODE;
  // VB = 3 needs more than the 256 VGPRs two waves per SIMD leave; a spill (or an AGPR copy) of a register that an
  // in-flight asm load is about to write would be silent corruption, so only VB = 2 is instantiated.
  static_assert(VB == 2 && S >= 2 && S <= 5, "ring depths");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hkv = a.num_kv_heads;
  const int b = blockIdx.y;
  const int blk = blockIdx.x;
  const int n = lane & 15;
  const int jq = lane >> 4;

  const int len = a.b_seqlen[b];
  const int start = blk * a.block_seq;
  const int end = min(len, start + a.block_seq);

  float* mid_o = a.mid_o + (int64_t)b * a.mid_o_stride_b + (int64_t)blk * a.mid_o_stride_s;
  float* mid_lse = a.mid_lse + (int64_t)b * a.mid_lse_stride_b + blk;
  if (end <= start) {
    for (int h = 0; h < G; ++h) {
      float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h;
      for (int d = lane; d < D; d += 64) o[d] = 0.f;
      if (lane == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = -INFINITY;
    }
    return;
  }
  const int ntiles = (end - start + kTileTokens - 1) / kTileTokens;
  const int range32 = ((a.block_seq + kTileTokens - 1) / kTileTokens) * kTileTokens;
  const Stage1V4Lds L(range32, Hkv, JQ, S, score_mode == SVK_SCORE_HEADMAX);
  int* slot_lds = reinterpret_cast<int*>(lds_raw + L.slot_off);
  float* old_lds = reinterpret_cast<float*>(lds_raw + L.old_off);
  float* spart = reinterpret_cast<float*>(lds_raw + L.spart_off);
  float* Pw = reinterpret_cast<float*>(lds_raw + L.p_off) + w * kPFloats;
  uint16_t* Pl = reinterpret_cast<uint16_t*>(Pw);                 // [16 heads][kPRow] bf16, rows >= G stay zero
  const char* ring = lds_raw + L.ring_off + w * (S * kV4StageBytes);
  const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)ring);
  const int SP = Hkv * JQ;

  if (a.new_k != nullptr && end == len) {
    // fused store_kvcache (v3's): the workgroup that owns the lane's newest token writes its K/V rows first
    const int ns = a.slot_mapping[b];
    if (ns >= 0 && lane < 2 * DW) {
      const bool is_v = lane >= DW;
      const int seg = lane % DW;
      const uint16_t* src = (is_v ? a.new_v : a.new_k) + (int64_t)b * a.new_stride_b + (int64_t)w * a.new_stride_h + seg * 8;
      uint16_t* dst = const_cast<uint16_t*>(is_v ? a.v_cache : a.k_cache) + (int64_t)ns * a.kv_slot_stride +
                      (int64_t)w * a.kv_head_stride + seg * 8;
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }

  // ---- stage the block's slot ids (clamped to the last valid token: every later load is legal) and old scores.
  //      All of a thread's loads are issued before the first LDS write (a plain `for` is compiled into one load +
  //      vmcnt(0) per trip: five dependent round trips at block_seq 1056).
  {
    const int32_t* row = a.req_to_tokens + (int64_t)a.b_req_idx[b] * a.req_stride;
    const float* dst = a.attn_score + (int64_t)b * a.score_stride_b + start;
    constexpr int PER = 5;
    const int nslots = ntiles * kTileTokens, nold = score_mode == SVK_SCORE_HEADMAX ? end - start : 0;
    for (int base = 0; base < nslots; base += PER * (int)blockDim.x) {
      int sid[PER];
      float old[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = base + u * (int)blockDim.x + (int)threadIdx.x;
        sid[u] = row[min(start + idx, end - 1)];
        if constexpr (score_mode == SVK_SCORE_HEADMAX) old[u] = dst[min(idx, nold - 1)];
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int idx = base + u * (int)blockDim.x + (int)threadIdx.x;
        if (idx < nslots) slot_lds[idx] = sid[u];
        if constexpr (score_mode == SVK_SCORE_HEADMAX) {
          if (idx < nold) old_lds[idx] = old[u];
        }
      }
    }
    for (int i = lane; i < kPFloats; i += 64) Pw[i] = 0.f;
  }
  // Q fragments (A operand: lane (head n, jq) holds q[head][c*32 + jq*8 .. +8]); rows >= G are zero
  bf16x8_t qa[NC];
  {
    const uint16_t* qp = a.q + (int64_t)b * a.q_stride_b + (int64_t)(w * G + n) * a.q_stride_h + jq * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 t = make_uint4(0, 0, 0, 0);
      if (n < G) t = *reinterpret_cast<const uint4*>(qp + c * 32);
      qa[c] = __builtin_bit_cast(bf16x8_t, t);
    }
  }
  // the compiler's wait for its own Q loads goes here, in front of the hand-counted loads (at the first MFMA it would
  // be a vmcnt(0) that also drains the whole prologue of the ring)
#pragma unroll
  for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(qa[c]));
  __syncthreads();

  const char* const kt = reinterpret_cast<const char*>(a.k_cache);
  const char* const vt = reinterpret_cast<const char*>(a.v_cache);
  const int64_t slot_bytes = a.kv_slot_stride * 2;
  const int64_t head_bytes = (int64_t)w * a.kv_head_stride * 2;
  const float sm_scale = rsqrtf((float)D);
  using addr_t = std::conditional_t<OFF32, uint32_t, const char*>;

  // K DMA of tile t into stage t % S.  Lane (rq = jq, pc = n): instruction k carries token rq*8+k, whose MFMA column is
  // n_tok = (rq&1)*8 + k; the lane fetches chunk pc ^ n_tok of that row into LDS row 4k+rq, position pc.
  auto issue_k = [&](int t) {
    const int4 s0 = *reinterpret_cast<const int4*>(slot_lds + t * kTileTokens + jq * 8);
    const int4 s1 = *reinterpret_cast<const int4*>(slot_lds + t * kTileTokens + jq * 8 + 4);
    const int sl[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    addr_t src[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int chunk = n ^ (((jq & 1) << 3) + k);
      if constexpr (OFF32) src[k] = (uint32_t)sl[k] * (uint32_t)slot_bytes + (uint32_t)(head_bytes + chunk * 16);
      else src[k] = kt + (int64_t)sl[k] * slot_bytes + head_bytes + chunk * 16;
    }
    const uint32_t dst = ring_addr + (uint32_t)(t % S) * kV4StageBytes;
    v4_dma8<NTK, OFF32>(src, kt, __builtin_amdgcn_readfirstlane(dst));
  };
  // V loads of tile t: lane (n, jq) takes head dims n*8.. of tokens jq*8+e
  auto issue_v = [&](int t, u32x4_t (&v)[8]) {
    const int4 s0 = *reinterpret_cast<const int4*>(slot_lds + t * kTileTokens + jq * 8);
    const int4 s1 = *reinterpret_cast<const int4*>(slot_lds + t * kTileTokens + jq * 8 + 4);
    const int sl[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    addr_t src[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (OFF32) src[e] = (uint32_t)sl[e] * (uint32_t)slot_bytes + (uint32_t)(head_bytes + n * 16);
      else src[e] = vt + (int64_t)sl[e] * slot_bytes + head_bytes + n * 16;
    }
    v4_ldv8<OFF32>(v, src, vt);
  };
  // Issue slot j of the schedule (j may be negative: the prologue runs the same schedule): V(j+VB-1) then K(j+S-1).
  // `younger_*` below count exactly these.
  auto v_issued = [&](int j) { const int t = j + VB - 1; return t >= 0 && t < ntiles; };
  auto k_issued = [&](int j) { const int t = j + S - 1; return t >= 0 && t < ntiles; };

  u32x4_t vbuf[VB][8];
  constexpr int PRO = (S > VB ? S : VB) - 1;
  // prologue (static V buffer index: tile t lives in vbuf[t % VB], and t = j + VB - 1 with j in [-PRO, -1])
#pragma unroll
  for (int j = -PRO; j < 0; ++j) {
    if (j + VB - 1 >= 0 && j + VB - 1 < ntiles) issue_v(j + VB - 1, vbuf[(j + VB - 1 + VB) % VB]);
    if (k_issued(j)) issue_k(j + S - 1);
  }

  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  f32x4_t acc[8];                                  // acc[i][r]: head jq*4+r, head dim n*8+i
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // LDS read addresses of the K fragments: row R(g, n) = 4*(n&7) + 2g + (n>>3), position (c*4+jq) ^ n
  int kaddr[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) kaddr[c] = (4 * (n & 7) + (n >> 3)) * 256 + (((c * 4 + jq) ^ n) << 4);

  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  int c0 = 0;          // first token (relative to start) of the current score chunk
  int i = 0;           // tile index
  auto tile = [&](auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;            // vbuf[CUR] holds V(i)
    constexpr int NXT = (CUR + VB - 1) % VB;               // buffer of V(i + VB - 1) (free since P.V of tile i-1)
    const int t0 = i * kTileTokens;                        // relative to start
    const bool full = start + t0 + kTileTokens <= end;

    // ---- issue V(i+VB-1), K(i+S-1)
    if (v_issued(i)) issue_v(i + VB - 1, vbuf[NXT]);
    if (k_issued(i)) issue_k(i + S - 1);

    // ---- K(i) has landed when only the ops issued after it are outstanding: schedule slots i-S+2 .. i
    {
      int younger = 0;
#pragma unroll
      for (int d = 0; d < S - 1; ++d) younger += (v_issued(i - d) ? 8 : 0) + (k_issued(i - d) ? 8 : 0);
      v4_wait_vm(younger);
    }
    f32x4_t s[2];
    {
      const char* st = ring + (i % S) * kV4StageBytes;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint4 kf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) kf[c] = *reinterpret_cast<const uint4*>(st + g * 512 + kaddr[c]);
        s[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
          s[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], __builtin_bit_cast(bf16x8_t, kf[c]), s[g], 0, 0, 0);
      }
    }

    bool tv[2];
    tv[0] = full || (start + t0 + n < end);
    tv[1] = full || (start + t0 + 16 + n < end);

    if constexpr (score_mode == SVK_SCORE_PERHEAD) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int h = jq * 4 + r;
          if (h < G && tv[g])
            a.attn_score[(int64_t)b * a.score_stride_b + (int64_t)(w * G + h) * a.score_stride_h + start + t0 + g * 16 + n] = s[g][r];
        }
    } else if constexpr (score_mode == SVK_SCORE_HEADMAX) {
      if (jq < JQ) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float pm = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jq * 4 + r < G) pm = fmaxf(pm, s[g][r]);
          spart[(t0 - c0 + g * 16 + n) * SP + w * JQ + jq] = tv[g] ? pm : -INFINITY;
        }
      }
    }

    float p[2][4];
    float alpha[4];
    bool rescale = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool hv = (jq * 4 + r < G);
      const float x0 = (hv && tv[0]) ? s[0][r] * sm_scale : -INFINITY;
      const float x1 = (hv && tv[1]) ? s[1][r] * sm_scale : -INFINITY;
      const float tmax = row16_allmax(fmaxf(x0, x1));
      const float nm = fmaxf(m[r], tmax);
      if (hv) {
        alpha[r] = __expf(m[r] - nm);
        p[0][r] = __expf(x0 - nm);
        p[1][r] = __expf(x1 - nm);
        rescale |= (nm != m[r]);
      } else {
        alpha[r] = 1.f; p[0][r] = 0.f; p[1][r] = 0.f;
      }
      l[r] = l[r] * alpha[r] + row16_allsum(p[0][r] + p[1][r]);
      m[r] = hv ? nm : m[r];
    }
    if (jq < JQ) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pl[(jq * 4 + r) * kPRow + g * 16 + n] = (uint16_t)f32_to_bf16_bits(p[g][r]);
    }
    if (__any(rescale)) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] *= alpha[r];
    }
    wave_sync();

    // ---- V(i) has landed when only the ops issued after it are outstanding: K of schedule slot i-VB+1 and
    //      everything of slots i-VB+2 .. i
    {
      int younger = k_issued(i - VB + 1) ? 8 : 0;
#pragma unroll
      for (int d = 0; d < VB - 1; ++d) younger += (v_issued(i - d) ? 8 : 0) + (k_issued(i - d) ? 8 : 0);
      v4_wait_vm(younger);
    }
    {
      u32x4_t(&vr)[8] = vbuf[CUR];
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(vr[e]));        // no use of V above the wait
      const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Pl + n * kPRow + jq * 8));
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        uint32_t vf[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          vf[e2] = __builtin_amdgcn_perm(vr[2 * e2 + 1][q / 2], vr[2 * e2][q / 2], (q & 1) ? 0x07060302u : 0x05040100u);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(vf[0], vf[1], vf[2], vf[3])), acc[q], 0, 0, 0);
      }
    }
    wave_sync();

    // ---- end of a score chunk (or of the block): one owner thread per token column
    if (score_mode == SVK_SCORE_HEADMAX && (i == ntiles - 1 || (t0 + kTileTokens - c0) == kV4ScoreChunk)) {
      const int c1 = min(end - start, t0 + kTileTokens);
      __syncthreads();
      float* const dst = a.attn_score + (int64_t)b * a.score_stride_b + start + c0;     // wave-uniform base
      for (uint32_t t = threadIdx.x; t < (uint32_t)(c1 - c0); t += blockDim.x) {
        float mx = old_lds[c0 + t];
        for (int j = 0; j < SP; ++j) mx = fmaxf(mx, spart[t * SP + j]);
        dst[t] = mx;
      }
      __syncthreads();
      c0 = t0 + kTileTokens;
    }
    ++i;
  };
  // VB tiles per trip so that every V buffer index is a compile-time constant
  while (i < ntiles) {
    tile(std::integral_constant<int, 0>{});
    if (i >= ntiles) break;
    tile(std::integral_constant<int, 1>{});
  }

  // ---- epilogue (v3's): lane (n, jq) owns heads jq*4+r and head dims n*8 .. +8
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int n_e = lane_e & 15, jq_e = lane_e >> 4;
  if (jq_e < JQ) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = jq_e * 4 + r;
      if (h < G) {
        if (n_e == 0) mid_lse[(int64_t)(w * G + h) * a.mid_lse_stride_h] = m[r] + __logf(l[r]);
        float* o = mid_o + (int64_t)(w * G + h) * a.mid_o_stride_h + n_e * 8;
        *reinterpret_cast<float4*>(o) = make_float4(acc[0][r] / l[r], acc[1][r] / l[r], acc[2][r] / l[r], acc[3][r] / l[r]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4][r] / l[r], acc[5][r] / l[r], acc[6][r] / l[r], acc[7][r] / l[r]);
      }
    }
  }
}
