// Decode stage 1 over KIVI-int4 blocks, group_size 32, any head shape: 128-token tiles, both products on the matrix cores,
// codes fetched by 4-byte loads straight into operand registers (kivi_stage1_tile128_kernel<.., WIDE = false> of
// decode_kivi_tile.hpp).  Serves what the wide launch does not: head_dim 64, bf16 key parameters, > 4 KV heads, block_seq
// not a multiple of 128.

#include "decode_kivi_tile.hpp"

namespace svk {

template <int D>
static int launch_narrow(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  const int G = a.num_q_heads / a.num_kv_heads;
  const int nblk = (a.max_len_in_batch + a.block_seq - 1) / a.block_seq;
  dim3 grid(nblk, a.batch), block(64 * a.num_kv_heads);
  const size_t shm_t = (size_t)a.num_kv_heads * (16 * 136 * 2 + 2 * (D / 32) * 128 * 2);
  switch (G) {
#define SVK_CASE(G_)                                                                                          \
  case G_:                                                                                                    \
    if (a.key_param_dtype == SVK_DTYPE_F32) hipLaunchKernelGGL((kivi_stage1_tile128_kernel<D, G_, true>), grid, block, shm_t, s, a); \
    else hipLaunchKernelGGL((kivi_stage1_tile128_kernel<D, G_, false>), grid, block, shm_t, s, a);            \
    break;
    SVK_ALL_G_CASES
#undef SVK_CASE
    default:
      set_error("svk_kivi_decode_stage1: GQA group size %d unsupported (1..8)", G);
      return SVK_ERR_LAYOUT;
  }
  return check_launch("svk_kivi_decode_stage1");
}

int launch_kivi_narrow(const SvkKiviDecodeStage1Args& a, hipStream_t s) {
  return a.head_dim == 128 ? launch_narrow<128>(a, s) : launch_narrow<64>(a, s);
}

}  // namespace svk
