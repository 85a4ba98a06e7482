// Decode stage 1 over KIVI-int4 blocks, the wide launch: 128-token tiles whose whole-block tiles take 16-byte loads
// through a per-wave LDS buffer (head_dim 128, <= 4 KV heads, fp32 key parameters, block_seq a multiple of 128).
// The kernel is kivi_stage1_tile128_kernel<.., WIDE = true> of decode_kivi_tile.hpp; DESIGN.md 4.8.

// Developer build (make EXTRA=-DSVK_KV_TIMING, then tools/kv_timing.py): s_memrealtime stamps (100 MHz) of wave 0 of every
// workgroup of the wide KIVI kernel: 0 entry, 1 range known, 2 before the tile loop, 3 first K tile in LDS, 4 first tile
// done, 5 tile loop done, 6 partials written.
#include "svk_common.hpp"
#ifdef SVK_KV_TIMING
__device__ unsigned long long g_kv_stamps[4096 * 8];
#define SVK_KV_STAMP(i)                                                                                                \
  do {                                                                                                                 \
    if (threadIdx.x == 0) {                                                                                            \
      const unsigned wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                                        \
      if (wg_ < 4096u) g_kv_stamps[wg_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                                   \
    }                                                                                                                  \
  } while (0)
extern "C" int svk_debug_kivi_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kv_stamps), sizeof(g_kv_stamps));
}
#else
#define SVK_KV_STAMP(i)
#endif

#include "decode_kivi_tile.hpp"

namespace svk {

int launch_kivi_wide(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  constexpr int D = 128;
  const int G = a.num_q_heads / a.num_kv_heads;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  dim3 grid(nblk + a.extra_partials, a.batch), block(64 * a.num_kv_heads);
  // 72.75 KiB per workgroup of 4 KV heads: two workgroups per CU
  const size_t shm_w = (size_t)a.num_kv_heads * (16 * 136 * 2 + 2 * (D / 32) * 128 * 2 + kXBufBytes);
  switch (G) {
#define SVK_CASE(G_)                                                                                          \
  case G_: {                                                                                                  \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&kivi_stage1_tile128_kernel<D, G_, true, true>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (16 * 136 * 2 + 2 * (D / 32) * 128 * 2 + kXBufBytes)) == hipSuccess; \
    (void)attr_ok;                                                                                            \
    hipLaunchKernelGGL((kivi_stage1_tile128_kernel<D, G_, true, true>), grid, block, shm_w, s, a);           \
    break;                                                                                                    \
  }
    SVK_ALL_G_CASES
#undef SVK_CASE
    default:
      set_error("svk_kivi_decode_stage1: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
  return check_launch("svk_kivi_decode_stage1");
}

}  // namespace svk
