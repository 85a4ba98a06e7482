/*
 * svk.h - C ABI of the MI355X (gfx950) sparse paged-attention hot path.
 *
 * This is the drop-in boundary of the rebuild of CURRENTF/Sparse-vLLM's sparse
 * decode/prefill-selection path.  The reference has no FFI of its own (its device
 * code is Triton called from Python); every entry point below replaces one
 * reference operator and cites it (paths relative to the reference's
 * src/sparsevllm/).  The Python host side (sparse_vllm_amd/) binds these with
 * ctypes and re-exposes the reference's own names and argument order; a
 * maintainer of the reference binds them the same way.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless a field is documented "host".
 *  - Every call only enqueues work on `stream` (a hipStream_t passed as void*):
 *    no allocation, no synchronisation, no host read-back -> hipGraph-capture safe.
 *  - Tensors are dense in their last dimension; other strides are given in
 *    ELEMENTS where a *_stride field exists, otherwise the layout is contiguous.
 *  - bf16 tensors are `uint16_t` bit patterns (torch.bfloat16 storage).
 *  - Return value: SVK_OK or a negative SvkStatus; svk_last_error() returns a
 *    thread-local message.  The host wrapper maps SVK_ERR_VALUE -> ValueError,
 *    SVK_ERR_LAYOUT -> AssertionError, SVK_ERR_STATE/SVK_ERR_LAUNCH -> RuntimeError
 *    (the reference's own exception classes for the same conditions).
 */
#ifndef SVK_H_
#define SVK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* svk_stream_t; /* hipStream_t */

typedef enum SvkStatus {
  SVK_OK = 0,
  SVK_ERR_VALUE = -1,   /* bad argument value/shape            -> ValueError     */
  SVK_ERR_LAYOUT = -2,  /* unsupported layout / head config    -> AssertionError */
  SVK_ERR_STATE = -3,   /* invariant of the cache state broken -> RuntimeError   */
  SVK_ERR_LAUNCH = -4   /* HIP launch failure                  -> RuntimeError   */
} SvkStatus;

#define SVK_ABI_VERSION 20

int svk_abi_version(void);
const char* svk_last_error(void);
/* Developer-build switches compiled in (bit 0: -DSVK_PA_TIMING, bit 1: -DSVK_QV_TIMING, bit 2: -DSVK_KV_TIMING); 0 in a
 * product build. */
int svk_build_flags(void);

/* ------------------------------------------------------------------------------------
 * KV payload
 * ---------------------------------------------------------------------------------- */

/* store_kvcache: cache[slot_mapping[i]] = kv[i], rows with slot == -1 skipped.
 * Replaces kernels/triton/store_kvcache.py:10-71 (store_kvcache_kernel / store_kvcache),
 * called from engine/cache_manager/base.py:674-694 (_store_layer_kv). */
typedef struct SvkStoreKvcacheArgs {
  const uint16_t* key;          /* [n_tokens, Hkv*D] bf16, row stride key_stride   */
  const uint16_t* value;        /* [n_tokens, Hkv*D] bf16, row stride value_stride */
  uint16_t* k_cache;            /* [slots, Hkv*D] bf16 contiguous                  */
  uint16_t* v_cache;            /* [slots, Hkv*D] bf16 contiguous                  */
  const int32_t* slot_mapping;  /* [n_tokens]                                      */
  int64_t key_stride;
  int64_t value_stride;
  int32_t n_tokens;
  int32_t row_elems;            /* Hkv*D, multiple of 8                            */
} SvkStoreKvcacheArgs;
int svk_store_kvcache(const SvkStoreKvcacheArgs* a, svk_stream_t stream);

/* Device-side guard for a decode step's slot table (MI355X addition): the attention launches trust the table exactly as
 * the reference's Triton kernels do, and the reference's debugging aid for a corrupted table
 * (layers/attention_backend.py:397-439, SVLLM_DEBUG_DECODE_BOUNDS) is a HOST check that synchronises and cannot run
 * under stream capture.  This launch checks the same three things on the device - request rows inside the table, visible
 * length inside its width, every visible slot (or page slot) inside the KV pool - never synchronises, may be captured
 * into the step's graph, and records the FIRST violation of any launch since `status` was last cleared:
 *   status[0] = 0 (clean) or SVK_SLOT_CHECK_*, status[1..5] = batch lane, table row, position, slot, context length.
 * The caller reads `status` whenever it next synchronises anyway (the reference's `torch._assert_async` discipline). */
enum { SVK_SLOT_CHECK_ROW = 1, SVK_SLOT_CHECK_WIDTH = 2, SVK_SLOT_CHECK_SLOT = 3 };
typedef struct SvkCheckSlotTableArgs {
  const int32_t* slot_table;    /* [num_rows, width] (table_stride): token slots, or page slots when slot_page_size > 1 */
  const int32_t* req_indices;   /* [batch] */
  const int32_t* context_lens;  /* [batch] */
  int32_t* status;              /* [8] int32, caller-owned, zero = clean */
  int64_t table_stride;
  int32_t batch, num_rows, width;
  int32_t slot_cap;             /* slots (or pages) of the KV pool */
  int32_t slot_page_size;       /* 0 / 1: token slots */
} SvkCheckSlotTableArgs;
int svk_check_slot_table(const SvkCheckSlotTableArgs* a, svk_stream_t stream);

/* copy_slots: cache[dst[i]] = cache[src[i]] for K and V through a caller-owned
 * workspace (gather all, then scatter all => safe when src and dst sets overlap).
 * Replaces the K/V move of H2OCacheManager._compact_final_prefill_dense_batch,
 * engine/cache_manager/h2o.py:1296-1329 (index_select -> workspace -> index_copy_)
 * and ExplicitKVStorage.copy_slots, engine/cache_manager/storage/explicit_kv.py:150-203. */
typedef struct SvkCopySlotsArgs {
  uint16_t* k_cache;
  uint16_t* v_cache;
  const int64_t* src_slots;  /* [n] */
  const int64_t* dst_slots;  /* [n] */
  uint16_t* workspace;       /* [2, n, row_elems] bf16 */
  int32_t n;
  int32_t row_elems;
} SvkCopySlotsArgs;
int svk_copy_slots(const SvkCopySlotsArgs* a, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Decode attention (split-KV, GQA) with fused token scores
 * ---------------------------------------------------------------------------------- */

#define SVK_SCORE_NONE 0     /* flash_decode_stage1                                   */
#define SVK_SCORE_HEADMAX 2  /* 2-D attn_score [B, W]: max over q heads of raw q.k    */
#define SVK_SCORE_PERHEAD 3  /* 3-D attn_score [B, Hq, W]: raw q.k per head           */

/* Stage 1.  Replaces kernels/triton/gqa_flash_decoding_stage1.py:
 *   flash_decode_stage1            :329-394 (kernel :6-121)
 *   flash_decode_stage1_with_score :397-445 (kernels :124-208 3-D, :211-295 2-D)
 * Same math: per (batch, kv head, block_seq block) online-softmax partials
 * mid_o = acc / sum_exp, mid_lse = max + log(sum_exp); empty blocks write (0, -inf);
 * scores are raw logits (before 1/sqrt(D)), 2-D scores are max-combined with what the
 * buffer already holds (the reference pre-fills -1e20 and uses atomic_max).
 * P is rounded to bf16 before P.V like `exp_logic.to(v.dtype)` (:280). */
typedef struct SvkFlashDecodeStage1Args {
  const uint16_t* q;             /* [B, Hq, D] bf16                                      */
  const uint16_t* k_cache;       /* [slots, Hkv, D] bf16, slot stride kv_slot_stride     */
  const uint16_t* v_cache;       /* same layout as k_cache                               */
  const int32_t* req_to_tokens;  /* [rows, req_stride] slot table                        */
  const int32_t* b_req_idx;      /* [B] row of the slot table per batch lane             */
  const int32_t* b_seqlen;       /* [B] valid tokens per lane                            */
  float* mid_o;                  /* [B, Hq, nblk, D] f32, strides below                  */
  float* mid_lse;                /* [B, Hq, nblk] f32                                    */
  float* attn_score;             /* NULL, [B, W] or [B, Hq, W] f32                       */
  int64_t q_stride_b, q_stride_h;
  int64_t kv_slot_stride;        /* elements between consecutive slots (>= Hkv*D)        */
  int64_t kv_head_stride;        /* elements between kv heads of one slot (>= D)         */
  int64_t kv_num_slots;          /* slots in k_cache/v_cache (0 = unknown); enables the  */
                                 /* 32-bit row-offset fast path when the tensor < 4 GiB  */
  int64_t req_stride;
  int64_t mid_o_stride_b, mid_o_stride_h, mid_o_stride_s;
  int64_t mid_lse_stride_b, mid_lse_stride_h;
  int64_t score_stride_b, score_stride_h;
  int32_t batch;
  int32_t num_q_heads;
  int32_t num_kv_heads;
  int32_t head_dim;              /* 64 or 128                                            */
  int32_t max_len_in_batch;      /* grid covers ceil(max_len/block_seq) blocks           */
  int32_t block_seq;             /* multiple of 16                                       */
  int32_t score_mode;            /* SVK_SCORE_*                                          */
  /* Optional fused store_kvcache (kernels/triton/store_kvcache.py:33-71 riding in the attention launch): when
   * new_k != NULL, the workgroup that owns lane b's newest token (position b_seqlen[b]-1) first writes
   * new_k[b], new_v[b] ([B, Hkv, D] bf16) to cache slot slot_mapping[b] (-1 = skip, like store_kvcache) and only
   * then reads the row - the separate svk_store_kvcache launch of the decode step disappears.  All NULL/0 = off. */
  const uint16_t* new_k;
  const uint16_t* new_v;
  const int32_t* slot_mapping;   /* [B] */
  int64_t new_stride_b, new_stride_h;
  /* Optional direct output: when the grid has ONE block per sequence (max_len_in_batch <= block_seq) the stage-2 merge
   * (flash_decoding_stage2.py:8-46) of that single partial is the identity followed by the bf16 rounding, so with
   * direct_o != NULL the kernel writes bf16(acc / l) to direct_o[b, head, :] itself - bit-identical to stage 1 +
   * stage 2 - and leaves mid_o untouched (mid_lse is still written); a row without tokens gets zeros.  The stage-2
   * launch of the layer disappears.  Stage-1 variant 3 only. */
  uint16_t* direct_o;            /* NULL or [B, Hq, D] bf16 */
  int64_t direct_stride_b, direct_stride_h;
  /* 2-D (head-max) scores only: 1 = store the launch's score of every position < b_seqlen[b] instead of max-combining it
   * with what attn_score holds - the caller then needs no -1e20 pre-fill of the buffer (every position below the length is
   * written exactly once per launch, by one owner thread) and masks the positions at or beyond the length itself
   * (SvkH2oDecodeScoreArgs.mask_by_len).  0 = the reference's max-combine contract. */
  int32_t score_overwrite;
  /* MI355X: > 0 = req_to_tokens is a table of PAGE slots and position t of a row lives in token slot
   * req_to_tokens[row, t / slot_page_size] * slot_page_size + t % slot_page_size (paged caches whose token slots are
   * page_slot * page_size + offset, e.g. Quest: the decode view is then 1 / page_size of the entries).  0 = token slots,
   * the reference's contract.  Must be a power of two. */
  int32_t slot_page_size;
  /* MI355X (ABI 20), unscored launches only: the ROTATED form of the fused store, for an attention view that is a rotated
   * copy of a pre-RoPE cache (DeltaKV sparse layers: the newest row of deltakv_materialize_sparse_view,
   * kernels/triton/deltakv_kernels.py:3588-3693, and the raw store of save_raw_kv_if_needed,
   * deltakv_less_memory.py:1269-1281, riding in the attention launch).  With new_cos_sin != NULL (and new_k, new_v,
   * slot_mapping as above) the workgroup that owns position b_seqlen[b]-1 writes
   *   raw_k_cache / raw_v_cache[slot_mapping[b]]            = new_k[b], new_v[b]        (the pre-RoPE rows), and
   *   k_cache[req_to_tokens[row, b_seqlen[b]-1]]            = bf16(RoPE(k_norm(new_k[b]))) at position
   *                                                           max(new_slot_to_pos[slot_mapping[b]], 0),
   *   v_cache[same slot]                                    = new_v[b]
   * before it reads the view's row; slot_mapping[b] < 0 or >= raw_num_slots writes nothing.  Same bits as the view launch
   * with new_k / new_v / new_slots.  slot_page_size must be 0. */
  const void* new_cos_sin;         /* NULL = plain fused store; [max_pos, D] cos | sin halves (new_cos_dtype) */
  const int32_t* new_slot_to_pos;  /* [raw_num_slots]                                       */
  const int32_t* new_row_lens;     /* NULL, or [B]: the absolute length of lane b's row INCLUDING the new token - the
                                    * position is then max(new_row_lens[b] - 1, 0), which is what
                                    * new_slot_to_pos[slot_mapping[b]] holds once the step's allocation has recorded the
                                    * token, read without the dependent trip through slot_mapping                        */
  const float* new_k_norm_weight;  /* NULL or [D] f32                                       */
  uint16_t* raw_k_cache;           /* [raw_num_slots, Hkv, D] bf16 (raw_slot_stride / raw_head_stride) */
  uint16_t* raw_v_cache;
  int64_t raw_slot_stride, raw_head_stride, new_cos_stride;
  int32_t raw_num_slots, new_cos_dtype;
  float new_k_norm_eps;
  int32_t _pad_rot;
} SvkFlashDecodeStage1Args;
int svk_flash_decode_stage1(const SvkFlashDecodeStage1Args* a, svk_stream_t stream);

/* Stage 2: LSE-weighted merge of the partials into O (bf16).
 * Replaces kernels/triton/flash_decoding_stage2.py:49-81 (kernel :8-46). */
typedef struct SvkFlashDecodeStage2Args {
  const float* mid_o;
  const float* mid_lse;
  const int32_t* b_seqlen;
  uint16_t* o;                   /* [B, Hq, D] bf16 */
  int64_t mid_o_stride_b, mid_o_stride_h, mid_o_stride_s;
  int64_t mid_lse_stride_b, mid_lse_stride_h;
  int64_t o_stride_b, o_stride_h;
  int32_t batch, num_q_heads, head_dim, block_seq;
  int32_t extra_partials;        /* partials merged beyond ceil(len / block_seq) (svk_kivi_decode_stage1), normally 0 */
  int32_t max_partials;          /* partials THIS launch may merge per row (ceil(max_len_in_batch / block_seq) +
                                  * extra_partials); picks the merge's thread geometry.  0 = unknown: the lse row stride
                                  * (the workspace's capacity, which a grow-only workspace inflates) decides instead. */
  /* Optional two-level merge for launches with many partials per row (max_partials > 256: KIVI full layers at 256 k
   * tokens, one-row launches in 32-token blocks): with split_ws != NULL and split_ws_bytes >=
   * svk_flash_decode_stage2_split_workspace_bytes(batch, num_q_heads, head_dim, max_partials), workgroup (b, h, s) merges
   * 32 partials into a second-level partial in the workspace and the last workgroup of a (b, h) to arrive (a ticket in the
   * workspace) merges those in index order and writes O - ceil(max_partials / 32) workgroups per (row, head) instead of
   * one.  split_ws: scratch, 256-byte aligned, contents irrelevant.  split_tickets: [batch * num_q_heads] int32 that are
   * ZERO before the first launch; every launch leaves them at zero again.  One launch at a time per workspace / ticket
   * array.  Either pointer NULL, or the scratch too small = one level. */
  void* split_ws;
  int64_t split_ws_bytes;
  int32_t* split_tickets;
} SvkFlashDecodeStage2Args;
int svk_flash_decode_stage2(const SvkFlashDecodeStage2Args* a, svk_stream_t stream);
/* bytes of split_ws for a launch shape; 0 when the two-level form does not apply (max_partials <= 256 or > 1024). */
int64_t svk_flash_decode_stage2_split_workspace_bytes(int32_t batch, int32_t num_q_heads, int32_t head_dim, int32_t max_partials);

/* ------------------------------------------------------------------------------------
 * H2O scores
 * ---------------------------------------------------------------------------------- */

/* Fill the per-step raw-score scratch with -1e20.
 * Replaces SparseController._get_h2o_decode_score_buffer `view.fill_(-1e20)`,
 * engine/sparse_controller.py:427-461. */
int svk_fill_f32(float* dst, int64_t n, float value, svk_stream_t stream);

/* In-place `x *= scale; softmax(x, dim=-1)` over the full row width (padding -1e20
 * underflows to exactly 0), optionally fused with the cumulative-score update
 *   cum[row(b), t] = cum[row(b), t] + x[b, t]   for t < len(b)  (new token: 0 + x)
 * Replaces SparseController.on_layer_attention_end, engine/sparse_controller.py:762-767,
 * and H2OCacheManager.update_decode_attention_scores{,_all_layers},
 * engine/cache_manager/h2o.py:897-1038 (pad(prev,1) + normalized[:kv_len]). */
typedef struct SvkH2oDecodeScoreArgs {
  float* attn_score;        /* [B, width] raw head-max logits in, probabilities out      */
  float* cum_score;         /* NULL or [rows, cum_stride] persistent cumulative scores   */
  const int32_t* b_req_idx; /* [B] (used when cum_score != NULL)                         */
  const int32_t* b_seqlen;  /* [B] current physical length incl. the new token          */
  const int32_t* b_new_slot;/* NULL or [B] this step's slot_mapping: lanes holding -1 are the padded hipGraph lanes
                             * of prepare_decode_static (h2o.py:419-424, they mirror lane 0's row); they are normalised
                             * but never accumulated - the reference adds only normalized[:, :len(seqs)]
                             * (sparse_controller.py:1226-1282)                                                    */
  int64_t score_stride_b;
  int64_t cum_stride;
  float scale;              /* head_dim ** -0.5                                          */
  int32_t batch;
  int32_t width;
  int32_t mask_by_len;      /* 1: positions >= b_seqlen[b] count as -1e20 whatever the buffer holds there (the step's raw
                             * scores were stored with score_overwrite, no pre-fill); needs b_seqlen                 */
  int32_t _pad0;
} SvkH2oDecodeScoreArgs;
int svk_h2o_decode_score_update(const SvkH2oDecodeScoreArgs* a, svk_stream_t stream);
/* The same for `n_layers` layers in ONE launch (engine/cache_manager/h2o.py:957-1038
 * `update_decode_attention_scores_all_layers`: the reference, too, folds the step's scores into the cumulative
 * rows of all layers at once): `first` describes layer 0; layer l uses attn_score + l * score_stride_layer,
 * cum_score + l * cum_stride_layer, b_new_slot + l * new_slot_stride_layer, b_req_idx + l * req_stride_layer and
 * b_seqlen + l * seqlen_stride_layer (element strides; 0 = shared by all layers).  Issued once per decode step after
 * the layer loop: 28 latency-bound row launches become one bandwidth-bound launch. */
int svk_h2o_decode_score_update_layers(const SvkH2oDecodeScoreArgs* first, int32_t n_layers, int64_t score_stride_layer,
                                       int64_t cum_stride_layer, int64_t new_slot_stride_layer, int64_t req_stride_layer,
                                       int64_t seqlen_stride_layer, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * H2O selection + slot-table compaction
 * ---------------------------------------------------------------------------------- */

/* keep = sort(top-`heavy` of scores[:recent_start] by (score desc, index asc)
 *             ++ [recent_start, kv_len)),  heavy = budget - recent_count.
 * Bit-exact restatement of H2OCacheManager.select_h2o_indices{,_batch},
 * engine/cache_manager/h2o.py:478-563 (stable descending argsort => ties keep the
 * lower index).  Rows with kv_len <= budget return arange(kv_len) (keep_len = kv_len).
 * NaN scores are not supported (the reference asserts finiteness upstream). */
typedef struct SvkH2oSelectArgs {
  const float* scores;   /* [rows, score_stride]                                   */
  int64_t* keep;         /* [rows, keep_stride] int64 (torch.long), ascending     */
  int64_t score_stride;
  int64_t keep_stride;
  int32_t rows;
  int32_t kv_len;
  int32_t budget;
  int32_t recent_count;  /* host: min(max(1, int(budget*ratio)), budget, kv_len)  */
} SvkH2oSelectArgs;
int svk_h2o_select_indices(const SvkH2oSelectArgs* a, svk_stream_t stream);

/* keep = [0, prefix) ++ (prefix + top-`topk` of scores[prefix : kv_len - suffix] by (score desc,
 * index asc)) ++ [kv_len - suffix, kv_len), ascending, int64; exactly prefix + topk + suffix entries
 * (requires topk <= kv_len - suffix - prefix).
 * Replaces SparseController._snapkv_select_indices{,_batch}, engine/sparse_controller.py:1670-1747
 * (sink ++ topk(middle) ++ recent; the reference's `topk` order/tie choice is unspecified and
 * free_part_slots sorts the indices, so parity is on the index SET with lowest-index tie-break),
 * and the DeltaKV compressed-position top-k, sparse_controller.py:1813-1822. */
typedef struct SvkSelectTopkArgs {
  const float* scores;
  int64_t* keep;
  int64_t score_stride, keep_stride;
  int32_t rows, kv_len;
  int32_t prefix, topk, suffix;
} SvkSelectTopkArgs;
int svk_select_prefix_topk_suffix(const SvkSelectTopkArgs* a, svk_stream_t stream);

/* Slot-table compaction for a batch of (layer, row) pairs of one uniform length:
 *   new_row = old_row[keep]; dropped slots (ascending position) are appended to the
 *   layer's free stack at free_base[layer] + lane * (cur_len - keep_len);
 *   row[keep_len:cur_len] = 0; optional per-row f32 payload (H2O cumulative scores)
 *   is gathered with the same keep.
 * Replaces SnapKVCacheManager.free_part_slots{,_batch,_batch_layers},
 * engine/cache_manager/snapkv.py:1528-1803, and the `kept_scores = scores.gather(...)`
 * of H2OCacheManager._evict_decode_rows, engine/cache_manager/h2o.py:1584-1602.
 * `keep` must be ascending and in [0, cur_len) (reference: keep_indices_sorted=True).
 * Host-side pointers (free_ptr bookkeeping) stay with the caller exactly as in the
 * reference (`_num_free_slots` is a Python list there). */
typedef struct SvkCompactRowsArgs {
  int32_t* slot_table;        /* [L, rows, table_stride_row]                         */
  int32_t* free_stack;        /* [L, stack_stride]                                   */
  float* row_payload;         /* NULL or [L, rows, payload_stride_row] f32           */
  const int64_t* keep;        /* [n_layers, n_lanes, keep_len] ascending             */
  const int32_t* layer_ids;   /* [n_layers] kv-layer index into the tensors above    */
  const int32_t* row_ids;     /* [n_layers, n_lanes] physical row per (layer, lane)  */
  const int64_t* free_base;   /* [n_layers] free-stack write offset per layer        */
  int64_t table_stride_layer, table_stride_row;
  int64_t stack_stride;
  int64_t payload_stride_layer, payload_stride_row;
  int32_t n_layers, n_lanes;
  int32_t cur_len, keep_len;
} SvkCompactRowsArgs;
int svk_compact_rows(const SvkCompactRowsArgs* a, svk_stream_t stream);

/* Device-resident decode bookkeeping of the H2O path (the reference keeps row lengths and
 * free-stack pointers in host numpy / Python lists and uploads the step's metadata, h2o.py:256-476, and decides the
 * burst on the host, h2o.py:1498-1625).  Row lengths `row_len[l, row]` and stack pointers `free_ptr[l]` live on the
 * device; a decode step is then two calls that need no host value of the step and can sit in one hipGraph together with
 * the layer loop:
 *   svk_h2o_device_step_begin  the allocation of svk_decode_alloc_slots from the device state: lane b of layer l takes
 *                              free_stack[l, free_ptr[l] - B + b], appends it to its row, fills slot_mapping /
 *                              context_lens / req_indices (padded graph lanes: slot -1, lane 0's metadata), then
 *                              row_len += 1 and free_ptr[l] -= B;
 *   svk_h2o_device_burst       every (layer, lane) whose row has reached `trigger_len` runs svk_h2o_select_indices on
 *                              its cumulative score row and svk_compact_rows on its slot-table / score row (dropped
 *                              slots go to the free stack in lane order, exactly where the host-driven burst puts
 *                              them), then row_len = budget and free_ptr[l] += dropped; rows below the trigger return
 *                              at once.  Without `tickets` the burst is three launches (select, compact, commit); with
 *                              `tickets` it is ONE launch (see the field).
 * Preconditions of svk_h2o_device_burst (the caller's, not checked on the device):
 *   - `row_ids` are DISTINCT rows: a row listed twice would be counted twice in the layer's fired-row total and
 *     compacted by two workgroups at once;
 *   - with `tickets`: tickets[l] == 0 on entry for every layer.  The launch leaves them at 0 when it completes; a launch
 *     that is aborted mid-burst (device fault, stream destroyed) can leave a non-zero ticket behind, and that layer would
 *     then never commit again - re-zero the tickets whenever the device state is rebuilt after a failed step
 *     (SnapKVCacheManager._device_state_upload does);
 *   - recent_count >= 1 in every select mode (the kernels index the recent range's first position).
 * Rows are uniform across layers (H2O's invariant, h2o.py:256-271).
 * The same two calls serve StreamingLLM (round 4): select_mode SVK_DEVICE_SELECT_WINDOW keeps the sink
 * [0, budget - recent_count) and the recent_count newest positions of a row that reached trigger_len = 2 * (sink + recent)
 * (SparseController._streamingllm_decode_eviction, sparse_controller.py:1558-1668 -> free_prefix_recent_slots_batch_layers,
 * snapkv.py:1805-1896); `scores` may then be NULL (no payload rows to compact).
 * select_mode SVK_DEVICE_SELECT_SNAPKV: SnapKV's decode re-eviction (SparseController._snapkv_decode_eviction,
 * sparse_controller.py:1104-1223): a row that reached trigger_len = 2 x decode_keep keeps its prefix_count sink tokens, the
 * top (budget - prefix_count - recent_count) middle tokens by THIS STEP's head-max raw scores and its recent_count newest
 * tokens; `scores` is then the step's scratch [L, graph lanes, width] indexed by batch lane (score_stride_row = lane stride),
 * nothing of it is compacted. */
#define SVK_DEVICE_SELECT_H2O 0
#define SVK_DEVICE_SELECT_WINDOW 1
#define SVK_DEVICE_SELECT_SNAPKV 2
typedef struct SvkH2oDeviceStepArgs {
  int32_t* slot_table;         /* [L, rows, table_stride_row]                         */
  int32_t* free_stack;         /* [L, stack_stride]                                   */
  float* scores;               /* [L, rows, score_stride_row] cumulative H2O scores   */
  int32_t* row_len;            /* [L, rows_total] device-resident row lengths         */
  int64_t* free_ptr;           /* [L] device-resident free-stack pointers             */
  const int32_t* row_ids;      /* [batch] physical row of every decode lane           */
  int32_t* slot_mapping;       /* [L, out_stride] out (begin)                         */
  int32_t* context_lens;       /* [L, out_stride] out (begin)                         */
  int32_t* req_indices;        /* [L, out_stride] out (begin)                         */
  int64_t* keep;               /* [L, batch, budget] scratch (burst)                  */
  int64_t table_stride_layer, table_stride_row, stack_stride;
  int64_t score_stride_layer, score_stride_row, out_stride;
  int32_t n_layers, rows_total, batch, graph_batch;
  int32_t budget, recent_count, trigger_len;
  int32_t select_mode;         /* SVK_DEVICE_SELECT_H2O (0) | _WINDOW (1) | _SNAPKV (2)       */
  int32_t prefix_count, _pad;  /* SNAPKV: sink tokens always kept                            */
  int32_t* tickets;            /* NULL (burst = three launches) or [L] int32, zero before the first launch: the burst runs as
                                * ONE launch - the workgroups of the rows that fired take a ticket when they are done, the
                                * last of them commits the layer and resets its ticket; all others return at once        */
} SvkH2oDeviceStepArgs;
int svk_h2o_device_step_begin(const SvkH2oDeviceStepArgs* a, svk_stream_t stream);
int svk_h2o_device_burst(const SvkH2oDeviceStepArgs* a, svk_stream_t stream);

/* Decode slot allocation for all layers at once: lane b of layer l takes
 * free_stack[l, free_ptr - B + b], writes it at slot_table[l, row[b], cur_len[b]] and
 * into slot_mapping[l, b]; context_lens[l, b] = cur_len[b] + 1; req_indices[l, b] = row[b].
 * Replaces the device half of H2OCacheManager.prepare_decode_static,
 * engine/cache_manager/h2o.py:386-437 (and SnapKVCacheManager.prepare_decode_static). */
typedef struct SvkDecodeAllocArgs {
  int32_t* slot_table;
  const int32_t* free_stack;
  const int32_t* layer_ids;   /* [n_layers]                                         */
  const int32_t* row_ids;     /* [B], or [n_layers, B] with meta_stride_layer = B   */
  const int32_t* cur_lens;    /* [B] length before the append (same layout)         */
  const int64_t* free_ptrs;   /* NULL (every layer pops at free_ptr) or [n_layers]: the per-layer stack pointers of
                               * the reference's non-uniform branch (snapkv.py:2656-2673)                  */
  int32_t* slot_mapping;      /* [n_layers, out_stride]  (lanes >= B get -1)        */
  int32_t* context_lens;      /* [n_layers, out_stride]                             */
  int32_t* req_indices;       /* [n_layers, out_stride]                             */
  int64_t table_stride_layer, table_stride_row;
  int64_t stack_stride;
  int64_t out_stride;
  int64_t meta_stride_layer;  /* 0: row_ids / cur_lens shared by all layers          */
  int64_t free_ptr;           /* common stack pointer before the pop                */
  int32_t n_layers, batch, graph_batch;
} SvkDecodeAllocArgs;
int svk_decode_alloc_slots(const SvkDecodeAllocArgs* a, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Prefill token scores (H2O accumulation / SnapKV selection)
 * ---------------------------------------------------------------------------------- */

#define SVK_PREFILL_SCORE_PROBABILITY 0
#define SVK_PREFILL_SCORE_LOGITS 1

/* attn_score[i, t] for score range i over the candidate keys t of its sequence:
 *   probability: max_h (1/max(q_end-q_start,1)) * sum_{p in window} softmax_t(q_{p,h}.k_t / sqrt(D))
 *                (softmax over the valid = candidate & causal keys of each query), others 0
 *   logits:      max_{p,h} q_{p,h}.k_t (raw), others -inf
 * The whole [n_ranges, score_cols] buffer is overwritten (fill + scores), like the reference wrapper.
 * Replaces kernels/triton/prefill_score.py: prefill_score_fwd :432-670 (kernels :70-429),
 * PrefillScoreWorkspace :6-67.  Q.K^T runs on the matrix cores (v_mfma_f32_16x16x32_bf16). */
typedef struct SvkPrefillScoreArgs {
  const uint16_t* q;                  /* [tokens, Hq, D] bf16 (this step's chunk queries)          */
  const uint16_t* k_cache;            /* [slots, Hkv, D] bf16                                      */
  float* attn_score;                  /* [n_ranges, score_stride] f32 out                          */
  const int32_t* b_req_idx;           /* [batch] slot-table row per sequence                       */
  const int32_t* b_start_loc;         /* [batch] first row of the sequence's chunk in q            */
  const int32_t* b_seq_len;           /* [batch] context length incl. the chunk                    */
  const int32_t* b_prompt_cache_len;  /* [batch] tokens before the chunk                           */
  const int32_t* req_to_tokens;       /* [rows, req_stride]                                        */
  const int32_t* score_q_start;       /* [n_ranges] absolute query window start                    */
  const int32_t* score_q_end;         /* [n_ranges]                                                */
  const int32_t* batch_indices;       /* NULL (range i <-> sequence i) or [n_ranges]               */
  float* workspace;                   /* probability mode: svk_prefill_score_workspace_bytes()     */
  int64_t q_stride_t, q_stride_h;
  int64_t kv_slot_stride, kv_head_stride;
  int64_t req_stride;
  int64_t score_stride;
  int32_t n_ranges;
  int32_t num_q_heads, num_kv_heads, head_dim;
  int32_t max_query_len;              /* longest window; probability mode: <= 128                  */
  int32_t score_cols;                 /* columns of attn_score = longest candidate range           */
  int32_t candidate_start, num_recent_tokens;
  int32_t score_mode;                 /* SVK_PREFILL_SCORE_*                                       */
  /* MI355X: softmax statistics of the window's query rows that svk_context_attention_fwd of the same chunk left behind
   * (`score_row_stats` there): probability mode then runs its final pass only - Q.K^T once instead of twice, one launch
   * instead of three; attn_score must have been cleared (`score_clear` there).  Only when every causal key is a
   * candidate (candidate_start = 0, num_recent_tokens = 0: H2O), range i <-> sequence i, head_dim 128.  NULL = compute
   * the statistics here (the reference's three-launch form). */
  const float* row_stats;
} SvkPrefillScoreArgs;
/* columns of one (sequence, head) in `row_stats`: the window padded as the scoring kernel tiles it */
int32_t svk_prefill_score_window_pad(int32_t num_q_heads, int32_t num_kv_heads, int32_t max_query_len);
int64_t svk_prefill_score_workspace_bytes(int32_t n_ranges, int32_t num_q_heads, int32_t num_kv_heads,
                                          int32_t max_query_len, int32_t score_cols);
int svk_prefill_score(const SvkPrefillScoreArgs* a, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Quest: query-aware page top-k
 * ---------------------------------------------------------------------------------- */

/* Per-page, per-KV-head elementwise max / min of the page's post-RoPE keys, for whole pages:
 *   metadata[0, l, p] = max_t K[l, p*page + t],  metadata[1, l, p] = min_t ...
 * for every KV layer l < n_layers and every listed physical page p.  Exact (bf16 compares).
 * Replaces QuestCacheManager.on_kv_stored / on_forward_end,
 * engine/cache_manager/quest.py:1607-1685, :1718-1771 (aminmax + index_copy_ per layer). */
typedef struct SvkQuestPageMinmaxArgs {
  const uint16_t* k_cache;     /* [L, slots, row_elems] bf16 (K half of kv_cache)          */
  uint16_t* metadata;          /* [2, L, pages, row_elems] bf16: 0 = max, 1 = min          */
  const int64_t* page_slots;   /* [n_pages] physical page ids                              */
  int64_t k_layer_stride;      /* elements between layers of k_cache                       */
  int64_t meta_kind_stride;    /* elements between the max and the min tensors             */
  int64_t meta_layer_stride;   /* elements between layers of metadata                      */
  int32_t n_pages, n_layers;
  int32_t page_size;           /* tokens per page (16)                                     */
  int32_t row_elems;           /* Hkv*D, multiple of 8                                     */
} SvkQuestPageMinmaxArgs;
int svk_quest_page_minmax(const SvkQuestPageMinmaxArgs* a, svk_stream_t stream);

/* Upper-bound page scores of the previous (complete) pages of every batch lane:
 *   s[b,p] = max_{q head h} bf16( bf16(q_h^+ . max_p) + bf16(q_h^- . min_p) ),  -inf for p >= num_pages-1
 * (bf16 roundings exactly where torch.bmm / `+=` on bf16 tensors round).
 * Replaces QuestCacheManager._score_pages_batched + the metadata gather and mask of
 * _build_decode_view_static, engine/cache_manager/quest.py:1773-1802, :1868-1885. */
typedef struct SvkQuestScorePagesArgs {
  const uint16_t* q;            /* [B, Hq, D] bf16                                         */
  const uint16_t* page_max;     /* [pages, Hkv, D] bf16 (metadata[0, layer])               */
  const uint16_t* page_min;     /* [pages, Hkv, D] bf16 (metadata[1, layer])               */
  const int32_t* page_table;    /* [rows, page_table_stride] physical page per logical page, -1 = none */
  const int32_t* req_indices;   /* [B]                                                     */
  const int32_t* context_lens;  /* [B]                                                     */
  float* page_scores;           /* [B, score_stride] f32 out (bf16-valued), first n_prev columns */
  int64_t q_stride_b, q_stride_h;
  int64_t page_table_stride;
  int64_t score_stride;
  int32_t batch, num_q_heads, num_kv_heads, head_dim;
  int32_t page_size;
  int32_t n_prev;               /* max_pages - 1 columns to score                          */
} SvkQuestScorePagesArgs;
int svk_quest_score_pages(const SvkQuestScorePagesArgs* a, svk_stream_t stream);

/* Packed decode view: top-`prev_budget` previous pages by (score desc, page index asc) in
 * ascending page order, then the last page; token slots = page_slot*page_size + [0, page_size);
 * lens = prev_budget*page_size + last_page_len.  Rows that are short (len <= token_budget or
 * num_pages <= page_budget_base) copy their dense prefix instead unless is_long_text.
 * `topk(sorted=False)` leaves the order and the choice among boundary ties unspecified in the
 * reference; this kernel makes the deterministic lowest-index choice.
 * Replaces _build_decode_view_static, engine/cache_manager/quest.py:1886-1913. */
typedef struct SvkQuestBuildViewArgs {
  const float* page_scores;     /* [B, score_stride]                                       */
  const int32_t* page_table;    /* [rows, page_table_stride]                               */
  const int32_t* token_table;   /* [rows, token_table_stride] dense slot table             */
  const int32_t* req_indices;   /* [B]                                                     */
  const int32_t* context_lens;  /* [B]                                                     */
  int32_t* packed_slots;        /* [B, packed_stride] out                                  */
  int32_t* local_lens;          /* [B] out                                                 */
  int32_t* local_req;           /* [B] out = arange(B)                                     */
  int64_t score_stride, page_table_stride, token_table_stride, packed_stride;
  int32_t batch;
  int32_t page_size;
  int32_t n_prev;               /* scored columns = max_pages - 1                          */
  int32_t prev_budget;          /* previous pages kept                                     */
  int32_t token_budget;
  int32_t page_budget_base;
  int32_t max_keep;             /* columns of packed_slots that are defined                */
  int32_t is_long_text;
  /* MI355X (optional): 1 = packed_slots receives PAGE slots (prev_budget selected pages + the last page; dense rows:
   * their first ceil(max_keep / page_size) pages) for svk_flash_decode_stage1's `slot_page_size` addressing - 1/page_size
   * of the view's entries.  0 = the reference-shaped token-slot view. */
  int32_t emit_page_slots;
  int32_t _pad0;
} SvkQuestBuildViewArgs;
int svk_quest_build_view(const SvkQuestBuildViewArgs* a, svk_stream_t stream);

/* Quest decode slot allocation: lane b appends one token to row req_indices[b]; lanes whose
 * cur_len % page_size == 0 take new_page_slots[b] (host-popped from the page stack) as their
 * next page.  Writes page table, token table, slot_mapping, context_lens (= cur+1), req rows.
 * Replaces QuestCacheManager._allocate_batch / prepare_decode_static device half,
 * engine/cache_manager/quest.py:1279-1360, :1542-1605. */
typedef struct SvkQuestDecodeAllocArgs {
  int32_t* page_table;
  int32_t* token_table;
  const int32_t* row_ids;        /* [B]                                                    */
  const int32_t* cur_lens;       /* [B]                                                    */
  const int32_t* new_page_slots; /* [B] page for lanes starting a new page, else ignored   */
  int32_t* slot_mapping;         /* [graph_batch] (lanes >= B get -1)                      */
  int32_t* context_lens;         /* [graph_batch]                                          */
  int32_t* req_indices;          /* [graph_batch]                                          */
  int64_t page_table_stride, token_table_stride;
  int32_t batch, graph_batch, page_size;
} SvkQuestDecodeAllocArgs;
int svk_quest_decode_alloc(const SvkQuestDecodeAllocArgs* a, svk_stream_t stream);

/* Device-resident Quest decode bookkeeping (the reference pops pages from a host stack and
 * uploads rows / lengths / new pages every step, quest.py:1279-1360, :1542-1605, and scans for completed pages on the
 * host after the step, :1718-1771).  Row lengths `row_len[rows]`, the LIFO page stack `free_pages[pages]` and its pointer
 * `free_page_ptr[1]` live on the device:
 *   svk_quest_device_step_begin  svk_quest_decode_alloc from the device state: the k-th lane (in lane order) whose row
 *                                starts a new page takes free_pages[ptr - 1 - k] (the order of `_pop_pages`,
 *                                `[ptr-n:ptr][::-1]`), then row_len += 1 and ptr -= n;
 *   svk_quest_device_step_end    predicated svk_quest_page_minmax: every lane whose row has just completed a page
 *                                (row_len % page_size == 0) refreshes that page's min / max rows on all layers
 *                                (`on_forward_end`); other lanes return at once.
 * Neither needs a host value of the step; both can sit in the step's hipGraph. */
typedef struct SvkQuestDeviceStepArgs {
  int32_t* page_table;           /* [rows, page_table_stride]                              */
  int32_t* token_table;          /* [rows, token_table_stride]                             */
  int32_t* row_len;              /* [rows] device-resident row lengths                     */
  int32_t* free_pages;           /* [pages] device-resident LIFO page stack                */
  int32_t* free_page_ptr;        /* [1]                                                    */
  const int32_t* row_ids;        /* [B]                                                    */
  int32_t* slot_mapping;         /* [graph_batch] (lanes >= B get -1)                      */
  int32_t* context_lens;         /* [graph_batch]                                          */
  int32_t* req_indices;          /* [graph_batch]                                          */
  const uint16_t* k_cache;       /* end: [L, slots, Hkv * D] bf16 keys (layer 0)           */
  uint16_t* metadata;            /* end: max rows of layer 0; min rows at + meta_kind_stride */
  int64_t page_table_stride, token_table_stride;
  int64_t k_layer_stride, meta_kind_stride, meta_layer_stride;
  int32_t batch, graph_batch, page_size, n_layers, row_elems, _pad;
} SvkQuestDeviceStepArgs;
int svk_quest_device_step_begin(const SvkQuestDeviceStepArgs* a, svk_stream_t stream);
int svk_quest_device_step_end(const SvkQuestDeviceStepArgs* a, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DeltaKV: compressed-KV decode
 * ---------------------------------------------------------------------------------- */

#define SVK_DTYPE_F32 0
#define SVK_DTYPE_BF16 1
#define SVK_DTYPE_F16 2

/* Slot bookkeeping of one DeltaKV decode step in ONE launch (engine/cache_manager/deltakv_base.py:2038-2154
 * `prepare_decode_static`: the reference issues four indexed scatters and five buffer fills, each with its own host
 * upload): every real lane b < batch gets one new raw slot in the full-layer pool and one in the sparse-layer pool at
 * position cur_len of its row; the graph-stable metadata buffers are written for all graph_batch lanes (padded lanes
 * mirror lane 0 with slot -1, like h2o.py:419-424).  `meta` is the step's host data as one upload:
 * [5, meta_stride] int32 = row, cur_len, full_slot, sparse_slot, compressed_len of lanes [0, batch). */
typedef struct SvkDeltakvDecodeAllocArgs {
  const int32_t* meta;
  int64_t meta_stride;
  int32_t* full_slots_map;        /* [rows, full_map_stride]   <- full_slot at [row, cur_len]   */
  int64_t full_map_stride;
  int32_t* full_slot_to_pos;      /* [full slots]              <- cur_len at [full_slot]        */
  int32_t* sparse_raw_slots_map;  /* [rows, sparse_map_stride] <- sparse_slot at [row, cur_len] */
  int64_t sparse_map_stride;
  int32_t* sparse_slot_to_pos;    /* [sparse slots]            <- cur_len at [sparse_slot]      */
  int32_t* context_lens;          /* [graph_batch] cur_len + 1                                   */
  int32_t* req_indices;           /* [graph_batch] row                                           */
  int32_t* slot_mapping;          /* [graph_batch] full_slot   (-1 on padded lanes)              */
  int32_t* sparse_slot_mapping;   /* [graph_batch] sparse_slot (-1 on padded lanes)              */
  int32_t* compressed_lens;       /* [graph_batch]                                               */
  int32_t batch;
  int32_t graph_batch;
} SvkDeltakvDecodeAllocArgs;
int svk_deltakv_decode_alloc(const SvkDeltakvDecodeAllocArgs* a, svk_stream_t stream);

/* MI355X: the same step with the bookkeeping RESIDENT on the device: row lengths, compressed lengths and
 * the two LIFO free stacks with their pointers live in HBM, and the step pops `batch` slots from each stack exactly as the
 * host's `pop(batch)` does (lane b takes stack[ptr - batch + b]; the pointer drops by `batch`), writes the maps and the
 * graph-stable buffers like svk_deltakv_decode_alloc and advances row_len of its rows.  No host data: the launch is a
 * node of the step's hipGraph.  One workgroup; rows must be distinct and both stacks must hold at least `batch` entries
 * (the host, which keeps mirrors by the same arithmetic, checks that before it takes this form of the step) - the
 * launch itself does not test the pointers.  The host re-uploads the state after anything else touched it (compression,
 * admission, free_seq). */
typedef struct SvkDeltakvDeviceStepArgs {
  const int32_t* rows;            /* [batch] row of lane b                                        */
  int32_t* row_len;               /* [rows] tokens held (read, then + 1 for the step's rows)      */
  const int32_t* compressed_len;  /* [rows]                                                       */
  const int32_t* full_stack;      /* full-layer free stack (entries [0, *full_ptr) are free)      */
  int32_t* full_ptr;              /* [1]                                                          */
  const int32_t* sparse_stack;    /* sparse-layer raw free stack                                  */
  int32_t* sparse_ptr;            /* [1]                                                          */
  int32_t* full_slots_map;  int64_t full_map_stride;
  int32_t* full_slot_to_pos;
  int32_t* sparse_raw_slots_map;  int64_t sparse_map_stride;
  int32_t* sparse_slot_to_pos;
  int32_t* context_lens;          /* [graph_batch] the five graph-stable buffers of svk_deltakv_decode_alloc */
  int32_t* req_indices;
  int32_t* slot_mapping;
  int32_t* sparse_slot_mapping;
  int32_t* compressed_lens;
  int32_t batch, graph_batch;
} SvkDeltakvDeviceStepArgs;
int svk_deltakv_device_step_begin(const SvkDeltakvDeviceStepArgs* a, svk_stream_t stream);

/* Sparse-layer view + reconstruct work list of one decode step:
 *   row b = [sink raw slots | for j < min(clen,K): temp slot if the selected compressed position has a
 *            latent else its raw slot | up to max_buffer recent raw slots], padding = first sink slot;
 *   recon_{pos,latent,out_slot}[b*K + j] = work item or -1;  new_context_lens = sink + min(clen,K) + buf.
 * Bit-exact restatement of deltakv_static_decode_plan, kernels/triton/deltakv_kernels.py:3854-3942
 * (kernel :3695-3851). */
typedef struct SvkDeltakvPlanArgs {
  const int32_t* raw_slots_map;      /* [rows, max_positions]                     */
  const int32_t* latent_slots_map;   /* [rows, max_positions]                     */
  const int32_t* active_compressed;  /* [B, K] relative compressed positions      */
  const int32_t* req_indices;        /* [B]                                       */
  const int32_t* context_lens;       /* [B]                                       */
  const int32_t* compressed_lens;    /* [B]                                       */
  const int32_t* temp_slots;         /* [B, K]                                    */
  int32_t* active_slots_out;         /* [B, sink+K+max_buffer]                    */
  int32_t* active_pos_out;           /* [B, sink+K+max_buffer]                    */
  int32_t* new_context_lens_out;     /* [B]                                       */
  int32_t* recon_pos_out;            /* [B*K]                                     */
  int32_t* recon_latent_out;         /* [B*K]                                     */
  int32_t* recon_out_slot_out;       /* [B*K]                                     */
  int64_t raw_stride, latent_stride, active_stride, temp_stride, out_stride, pos_stride;
  int32_t batch, k_max, sink, max_buffer, max_positions;
} SvkDeltakvPlanArgs;
int svk_deltakv_static_decode_plan(const SvkDeltakvPlanArgs* a, svk_stream_t stream);

/* Reconstruct compressed tokens into temp slots:
 *   K = delta_K + mean_f(de-RoPE(father K)), optional RMS k-norm, RoPE(out_pos);  V = delta_V + mean_f(father V)
 * delta is either dense [N, 2*Hkv*D] (delta_bits = 0; entry valid iff out_slot >= 0 and out_pos >= 0) or a
 * packed int2/int4/int8 residual + per-group scale/min looked up by latent_slots (entry valid iff latent >= 0).
 * Replaces deltakv_reconstruct_writeback_grouped_heads, deltakv_kernels.py:2909-3012 (kernel :2732-2907) and
 * deltakv_less_memory_reconstruct_writeback_quantized / _int4, :3344-3485 (kernel :3173-3341). */
typedef struct SvkDeltakvReconstructArgs {
  const void* delta;               /* dense: [N, 2*Hkv*D] (delta_dtype); packed: int32 [latents, 2*Hkv*D*bits/32] */
  const void* scale;               /* packed: [latents, groups] (scale_dtype)                                    */
  const void* mn;                  /* packed: [latents, groups]                                                  */
  const int32_t* latent_slots;     /* packed: [N]                                                                */
  const int32_t* father_slots;     /* [N, K]                                                                     */
  const int32_t* slot_to_pos;      /* [slots]                                                                    */
  const int32_t* out_slots;        /* [N]                                                                        */
  const int32_t* out_pos;          /* [N]                                                                        */
  const void* cos_sin;             /* [max_pos, D]: cos | sin halves (cos_dtype)                                 */
  uint16_t* k_cache;               /* [slots, Hkv, D] bf16, read (fathers) and written (out slots)               */
  uint16_t* v_cache;
  const float* k_norm_weight;      /* NULL or [D] f32                                                            */
  int64_t delta_stride, scale_stride, father_stride, cos_stride, kv_slot_stride, kv_head_stride;
  float k_norm_eps;
  int32_t n, k_fathers, num_kv_heads, head_dim;
  int32_t delta_bits;              /* 0 dense, else 2 / 4 / 8                                                    */
  int32_t group_size;              /* packed: features per scale/min group                                       */
  int32_t delta_dtype, scale_dtype, cos_dtype;   /* SVK_DTYPE_*                                                  */
  int32_t raw_k_cache;             /* fathers hold un-rotated K                                                  */
  int32_t store_raw_k;             /* write un-rotated K                                                         */
  /* fused father lookup (static decode): when father_table != NULL, father_slots is ignored and the fathers of
   * entry n are max(father_table[max(father_index[n], 0), :], 0) - the reference's
   * `deltakv_latent_to_full_slots[l, recon_latent.clamp_min(0)].clamp_min(0)` (deltakv_less_memory.py:4054-4058). */
  const int32_t* father_table;     /* NULL or [latents, K]                                                       */
  const int32_t* father_index;     /* [N] latent slot per entry (may be -1)                                      */
  int64_t father_table_stride;
  /* MI355X (dense bf16 delta, 16-byte form): when out_k_cache != NULL the reconstructed rows are written THERE instead
   * of into k_cache / v_cache - entry n goes to row (n / out_entries_per_row) * out_view_width + out_view_offset +
   * n % out_entries_per_row, i.e. straight into its place in the layer's attention view (the [sink | K selected |
   * tail] layout of the static decode plan; svk_deltakv_materialize_sparse_view then skips those rows: skip_temp).
   * out_slots[n] < 0 still marks an entry that reconstructs nothing. */
  uint16_t* out_k_cache;           /* NULL or [rows, Hkv, D] bf16 (out_slot_stride / out_head_stride)            */
  uint16_t* out_v_cache;
  int64_t out_slot_stride, out_head_stride;
  int32_t out_view_width, out_view_offset, out_entries_per_row, _pad1;
} SvkDeltakvReconstructArgs;
int svk_deltakv_reconstruct_writeback(const SvkDeltakvReconstructArgs* a, svk_stream_t stream);
/* The dense-delta decode form for `n_batch` layers in one launch (same plan: out_slots / out_pos / father_index shared;
 * delta, the father table, the K / V caches and the k-norm weight advance by their per-layer element strides). */
typedef struct SvkDeltakvReconstructBatch {
  int32_t n_batch;
  int64_t delta_stride_batch, father_table_stride_batch, kv_cache_stride_batch, k_norm_stride_batch;
  int64_t out_cache_stride_batch;  /* elements between the layers' out_k_cache / out_v_cache                     */
} SvkDeltakvReconstructBatch;
int svk_deltakv_reconstruct_writeback_batched(const SvkDeltakvReconstructArgs* first, const SvkDeltakvReconstructBatch* b,
                                              svk_stream_t stream);

/* MI355X: the second Linear of compress_up and the reconstruction above in ONE launch for `b->n_batch` layers that share a
 * plan - delta[n, :] = bf16(hidden[n, :] . weight^T + bias) (utils/compressor.py:69-73, the reference's F.linear) never
 * leaves the chip: a workgroup multiplies 128 tokens x one head of K or V on the matrix cores and finishes those rows as
 * svk_deltakv_reconstruct_writeback_batched does (same element arithmetic; `first->delta*` are ignored).  Static decode
 * form only: head_dim 128, un-rotated father keys (raw_k_cache), rotated output, fp32 cos|sin, 1..4 fathers per token.
 * Against the library GEMM + svk_deltakv_reconstruct_writeback_batched the results differ by the fp32 summation order
 * of the product only. */
typedef struct SvkDeltakvUpReconArgs {
  const uint16_t* hidden;   /* [n_batch][N][hidden_stride] bf16: act(Linear1) rows (svk_dequant_linear_act_batched)   */
  const uint16_t* weight;   /* [n_batch][2*Hkv*D][weight_stride] bf16, input features contiguous                     */
  const uint16_t* bias;     /* NULL or [n_batch][2*Hkv*D] bf16                                                      */
  int64_t hidden_stride, hidden_stride_batch, weight_stride, weight_stride_batch, bias_stride_batch;   /* elements   */
  int32_t k, _pad0;         /* input features of the Linear: a multiple of 64                                       */
} SvkDeltakvUpReconArgs;
int svk_deltakv_up_reconstruct(const SvkDeltakvUpReconArgs* u, const SvkDeltakvReconstructArgs* first,
                               const SvkDeltakvReconstructBatch* b, svk_stream_t stream);

/* out[r, f] = code(r, f) * scale[r, f/group] + mn[r, f/group]; codes are `bits`-wide fields packed
 * LSB-first into int32.  Replaces triton_dequantize_2d_int4_grouped (kernels/triton/quant.py:160-216) and
 * unpack_tensor + unpack_quantized_to_16bit (:304-349). */
typedef struct SvkDequantGroupedArgs {
  const int32_t* packed;  /* [rows, features*bits/32] */
  const void* scale;      /* [rows, features/group]   */
  const void* mn;
  void* out;              /* [rows, features]         */
  int64_t packed_stride, scale_stride, out_stride;
  int32_t rows, features, bits, group_size;
  int32_t scale_dtype, out_dtype;
  const int32_t* row_index; /* NULL or [rows]: output row r reads source row max(row_index[r], 0) - the fused
                               `cache[l, recon_latent.clamp_min(0)]` gather of _load_residual */
} SvkDequantGroupedArgs;
int svk_dequantize_grouped(const SvkDequantGroupedArgs* a, svk_stream_t stream);

/* Residual load, fused: out = act(bf16(dequant_int4(packed[row_index]) . weight^T + bias)), bf16 [rows, N].
 * One MFMA launch for the int4 latent dequantisation (kernels/triton/quant.py:160-216), the first nn.Linear and the
 * nn.GELU of the `compress_up` module (utils/compressor.py:69-73) in `_load_residual`
 * (engine/cache_manager/deltakv_less_memory.py:2841-2848).  Rounding points as in the separate ops: q*scale and +min
 * rounded separately, bf16 dequant output, fp32 accumulate, bf16 Linear output, erf-GELU evaluated in fp32. */
typedef struct SvkDequantLinearArgs {
  const int32_t* packed;    /* [latent_rows, K/8] int4 codes, LSB first            */
  const void* scale;        /* [latent_rows, K/group_size] (scale_dtype)           */
  const void* mn;
  const int32_t* row_index; /* NULL or [rows]: row r reads latent row max(row_index[r], 0) */
  const uint16_t* weight;   /* [N, K] bf16 (nn.Linear weight), row stride weight_stride */
  const uint16_t* bias;     /* NULL or [N] bf16                                    */
  uint16_t* out;            /* [rows, N] bf16                                      */
  int64_t packed_stride, scale_stride, weight_stride, out_stride;
  int32_t rows, k, n, group_size;
  int32_t scale_dtype;
  int32_t activation;       /* 0 = none, 1 = erf-GELU                              */
} SvkDequantLinearArgs;
int svk_dequant_linear_act(const SvkDequantLinearArgs* a, svk_stream_t stream);
/* The same for `n_batch` layers in one launch: `first` describes layer 0, layer z adds z * (the element stride of the
 * field's tensor between consecutive layers); row_index is shared.  (DeltaKV sparse layers of one observation group:
 * their residual loads depend on the group's plan only, deltakv_less_memory.py:2841-2848 per layer.) */
typedef struct SvkDequantLinearBatch {
  int32_t n_batch;
  int64_t packed_stride_batch, scale_stride_batch, weight_stride_batch, bias_stride_batch, out_stride_batch;
} SvkDequantLinearBatch;
int svk_dequant_linear_act_batched(const SvkDequantLinearArgs* first, const SvkDequantLinearBatch* b, svk_stream_t stream);

/* Attention-facing contiguous copy of a DeltaKV sparse layer's active slots: entry n = b*width + w takes slot
 * active_slots[b, w] (clamped into [0, num_slots)); V is copied; K is copied when postrope_mask[slot] != 0,
 * otherwise (optional RMS k-norm, then) rotated at slot_to_pos[slot] (clamped at 0).  Every entry is written.
 * Replaces deltakv_materialize_sparse_view, kernels/triton/deltakv_kernels.py:3489-3585 (kernel :3588-3693);
 * caller get_layer_compute_view, engine/cache_manager/deltakv_less_memory.py:1344-1402. */
typedef struct SvkDeltakvMaterializeArgs {
  const int32_t* active_slots;     /* [batch, width] (active_stride)                          */
  const int32_t* slot_to_pos;      /* [num_slots]                                             */
  const uint8_t* postrope_mask;    /* NULL or [num_slots] bool                                */
  uint16_t* k_cache;               /* [num_slots, Hkv, D] bf16 (kv_slot_stride/kv_head_stride); written only with new_slots */
  uint16_t* v_cache;
  uint16_t* out_k;                 /* [>= batch*width, Hkv, D] bf16 (out_slot_stride/out_head_stride) */
  uint16_t* out_v;
  const void* cos_sin;             /* [max_pos, D]: cos | sin halves (cos_dtype)              */
  const float* k_norm_weight;      /* NULL or [D] f32                                         */
  int64_t active_stride, kv_slot_stride, kv_head_stride, out_slot_stride, out_head_stride, cos_stride;
  float k_norm_eps;
  int32_t batch, width, num_slots, num_kv_heads, head_dim, cos_dtype;
  /* static-decode alternative to postrope_mask: entry (b, w) with temp_offset <= w < temp_offset + temp_count is
   * post-RoPE iff its slot equals temp_slots[b, w - temp_offset] (the reconstruct scratch of this step) */
  const int32_t* temp_slots;       /* NULL or [batch, temp_count] (temp_stride)               */
  int64_t temp_stride;
  int32_t temp_offset, temp_count;
  /* MI355X: this step's raw store (save_raw_kv_if_needed -> store_kvcache, deltakv_less_memory.py:1269-1281) riding
   * in the same launch.  With new_slots != NULL row b's new token (new_k/new_v[b], pre-RoPE key) is written to
   * cache slot new_slots[b] (skipped when < 0), and a view entry (b, w) whose slot equals new_slots[b] takes its
   * data from new_k/new_v[b] instead of the cache, so the result equals store-then-materialise. */
  const uint16_t* new_k;           /* NULL or [batch, Hkv, D] bf16 (new_token_stride/new_head_stride) */
  const uint16_t* new_v;
  const int32_t* new_slots;        /* NULL or [batch]                                         */
  int64_t new_token_stride, new_head_stride;
  /* MI355X: 1 = the entries recognised as this step's reconstruct scratch (temp_slots) are NOT copied: the
   * reconstruction wrote them into out_k / out_v itself (SvkDeltakvReconstructArgs.out_k_cache) */
  int32_t skip_temp;
  /* MI355X (ABI 20): 1 = the entry whose slot equals new_slots[b] is NOT written either (new_slots != NULL, new_k / new_v
   * NULL): the layer's attention launch stores that row itself, rotated (SvkFlashDecodeStage1Args.new_cos_sin). */
  int32_t skip_new;
  /* MI355X (ABI 20): layer_count > 1 = the same slot table materialised for layer_count consecutive layers in one
   * launch: layer i reads k_cache / v_cache + i * kv_layer_stride, writes out_k / out_v + i * out_layer_stride and
   * normalises with k_norm_weight + i * k_norm_layer_stride (elements).  Carries no store (new_k NULL).  0 / 1 = one layer. */
  int64_t kv_layer_stride, out_layer_stride, k_norm_layer_stride;
  int32_t layer_count, _pad0;
} SvkDeltakvMaterializeArgs;
int svk_deltakv_materialize_sparse_view(const SvkDeltakvMaterializeArgs* a, svk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Chunked-prefill causal attention over the paged slot table
 * ---------------------------------------------------------------------------------- */

/* o[t, h, :] = softmax_j(q[t, h] . k[j] * D^-0.5) v[j] over the keys j <= prompt_cache_len[b] + (t - start_loc[b]) of
 * sequence b, keys / values read through req_to_tokens[b_req_idx[b], j] (the chunk's own K/V are already stored).
 * base-2 online softmax with the reference's constants (sm_scale = D^-0.5 * log2(e), masked logits = -1e8), P rounded
 * to bf16 before P.V, fp32 accumulation.
 * Replaces context_attention_fwd (attn_score=None), kernels/triton/context_flashattention_nopad.py:10-78, 242-276
 * (call sites layers/attention_backend.py:140, operators/prefill_attention.py:621-625). */
typedef struct SvkContextAttentionArgs {
  const uint16_t* q;             /* [tokens, Hq, D] bf16 (q_stride_t / q_stride_h)        */
  const uint16_t* k_cache;       /* [slots, Hkv, D] bf16 (kv_slot_stride / kv_head_stride) */
  const uint16_t* v_cache;
  uint16_t* o;                   /* [tokens, Hq, D] bf16 (o_stride_t / o_stride_h)        */
  const int32_t* b_req_idx;      /* [B]                                                   */
  const int32_t* b_start_loc;    /* [B] first query token of the sequence in q            */
  const int32_t* b_seq_len;      /* [B] total length incl. the cached prefix              */
  const int32_t* b_prompt_cache_len; /* [B]                                               */
  const int32_t* req_to_tokens;  /* [rows, req_stride]                                    */
  int64_t q_stride_t, q_stride_h, kv_slot_stride, kv_head_stride, o_stride_t, o_stride_h, req_stride;
  int32_t batch, num_q_heads, num_kv_heads, head_dim, max_input_len;
  int64_t kv_num_slots;          /* slots in k_cache / v_cache (0 = unknown): < 4 GiB tensors get 32-bit row offsets */
  /* MI355X, optional (NULL = off; head_dim 128 kernel only): the attention already owns what prefill_score_fwd's first
   * two launches compute - the final softmax statistics of every query row - so the rows of a score window
   * [score_q_start[b], b_seq_len[b]) leave them behind for svk_prefill_score(row_stats=...):
   *   score_row_stats[((b * Hkv + kvh) * G + g) * score_wpad + (pos - score_q_start[b])] = m * D^-0.5 * log2 e + log2 l
   * (base-2 domain: p = 2^(s * D^-0.5 * log2 e - stat)), pos = absolute position of the query row in the sequence.
   * `score_clear` [batch, score_clear_stride] f32: the first score_clear_cols columns of every row are zeroed (the
   * state the final scoring pass publishes into by atomic max). */
  float* score_row_stats;
  const int32_t* score_q_start;  /* [B] absolute position of the window's first query row (< 0: no window)          */
  float* score_clear;
  int64_t score_clear_stride;
  int32_t score_wpad, score_clear_cols;
  /* The reference's score-collecting forms (context_flashattention_nopad.py:82-240, `attn_score=` of
   * context_attention_fwd; what its OmniKV / DeltaKV observation layers receive in prefill), optional (NULL = off):
   *   attn_score_dim 3: attn_score[b, h, t] += sum over the chunk's query rows r with cache_len + r >= t of q[r, h] . k[t]
   *                     (raw logits, _fwd_kernel_with_score :127-132)
   *   attn_score_dim 2: attn_score[b, t] = max(attn_score[b, t], max over heads h and over the 128-row query blocks Q of
   *                     (sum over r in Q, cache_len + r >= t of q[r, h] . k[t]) / chunk_len)   (_with_score_2d :205-212)
   * for t < b_seq_len[b].  Computed beside the attention launch from per-block suffix sums of the query rows
   * (`score_workspace`, svk_context_attention_score_workspace_bytes(tokens, Hq, D) bytes), fp32. */
  float* attn_score;
  float* score_workspace;
  int64_t attn_score_stride_b, attn_score_stride_h;
  int32_t attn_score_dim;        /* 2 or 3                                                                          */
  int32_t attn_score_cols;       /* columns of attn_score (>= the longest b_seq_len)                               */
} SvkContextAttentionArgs;
int svk_context_attention_fwd(const SvkContextAttentionArgs* a, svk_stream_t stream);
int64_t svk_context_attention_score_workspace_bytes(int64_t tokens, int32_t num_q_heads, int32_t head_dim);

/* ---- DeltaKV compression side -------------------------------------------------------------- */

/* Grouped quantise + pack of residual rows: per group scale = (max - min) / (2^bits - 1) stored in the data dtype,
 * q = round_half_even(clamp((x - min) / (scale + 1e-6), 0, 2^bits - 1)) with the reference kernel's rounding points
 * (max/min/scale in fp32, `x - min` in the data dtype, quotient in fp32), codes LSB-first in int32.  Output row r goes
 * to row dst_rows[r] (or r when NULL) of code / scale / mn - the `cache[l, latent_slots] = ...` stores fused in.
 * Replaces triton_quantize_and_pack_2d_int4_grouped, kernels/triton/quant.py:29-117 (caller _store_residual,
 * engine/cache_manager/deltakv_less_memory.py:2157-2179). */
typedef struct SvkQuantPackArgs {
  const void* data;           /* [rows, features] (data_dtype, data_stride)      */
  const int32_t* dst_rows;    /* NULL or [rows]                                  */
  int32_t* code;              /* [*, features*bits/32] (code_stride)             */
  void* scale;                /* [*, features/group] (data_dtype, scale_stride)  */
  void* mn;
  int64_t data_stride, code_stride, scale_stride;
  int32_t rows, features, bits, group_size, data_dtype;
} SvkQuantPackArgs;
int svk_quantize_pack_grouped(const SvkQuantPackArgs* a, svk_stream_t stream);

/* KIVI-int4 block store: for block j the G = group_size tokens at cache rows raw_slots[j, 0..G) are quantised
 * per channel (K: one scale/min per (head, dim) over the block's tokens) and per token (V: one scale/min per
 * (token, group of G dims)) with torch's bf16 arithmetic (every op rounds), and written to block block_slots[j].
 * Replaces _store_full_layer_kivi_blocks, engine/cache_manager/deltakv_less_memory.py:1741-1780
 * (triton_quantize_and_pack_along_last_dim, kernels/triton/quant.py:264-301) including the gather at :3544-3545. */
typedef struct SvkKiviStoreArgs {
  const uint16_t* k_cache;      /* [slots, Hkv, D] bf16 (kv_slot_stride / kv_head_stride) */
  const uint16_t* v_cache;
  const int32_t* raw_slots;     /* [blocks, G]                                        */
  const int32_t* block_slots;   /* [blocks]                                           */
  int32_t* key_packed;          /* [*, Hkv, D, G/8]                                   */
  void* key_scales;             /* [*, Hkv, D] f32 or bf16 (key_param_dtype)          */
  void* key_mins;
  int32_t* value_packed;        /* [*, Hkv, G, D/8]                                   */
  uint16_t* value_scales;       /* [*, Hkv, G, D/G] bf16                              */
  uint16_t* value_mins;
  int64_t kv_slot_stride, kv_head_stride;
  int32_t blocks, num_kv_heads, head_dim, group_size, key_param_dtype;
} SvkKiviStoreArgs;
int svk_kivi_store_blocks(const SvkKiviStoreArgs* a, svk_stream_t stream);

/* Father assignment: per row r the k best columns of scores[r, :m] (best first, lower column on ties), where column
 * c >= m0 is admissible only if new_center_rel[c - m0] <= row_offset + r (a token only sees the centres of its own
 * block at or before itself).  Replaces the mask + `scores.topk(k_eff, sorted=False)` of _cluster_compress,
 * engine/cache_manager/deltakv_less_memory.py:2780-2787.  k <= 8. */
typedef struct SvkClusterTopkArgs {
  const void* scores;             /* [rows, m] (score_dtype, score_stride)   */
  const int32_t* new_center_rel;  /* [m - m0]                                */
  int32_t* topk;                  /* [rows, k] (topk_stride)                 */
  int64_t score_stride, topk_stride;
  int32_t rows, m, m0, k, row_offset, score_dtype;
} SvkClusterTopkArgs;
int svk_cluster_topk(const SvkClusterTopkArgs* a, svk_stream_t stream);

/* MI355X: the ranking product, the causal mask and the top-k of `_cluster_compress` in one MFMA launch that never
 * materialises the [rows, m] score matrix - what `_deltakv_l2_topk_block_kernel` + `deltakv_l2_topk_blockwise`
 * (kernels/triton/deltakv_kernels.py:3945-4134) and the candidate merge (deltakv_base.py:3352-3358) do for the
 * reference, with the runtime's bf16 rounding points (`_metric_l2`, deltakv_base.py:2168-2190):
 *   score[r, c] = bf16(2 * bf16(sum_d token[r, d] * centre[c, d]) - bf16(sum_d bf16(centre[c, d]^2)))
 * centre c = concat(K[center_slots[c]], V[center_slots[c]]) read from the layer caches (slot rows contiguous: Hkv*D
 * values), masked to -inf when c >= m0 and new_center_rel[c - m0] > row_offset + r; topk[r, :k] = columns by (score
 * descending, column ascending) - the order of svk_cluster_topk, which this launch replaces together with the library
 * GEMM in front of it.  Hkv*D a multiple of 32, at most 512 (SVK_ERR_LAYOUT beyond: the caller keeps the GEMM).
 * `workspace`: svk_cluster_l2_topk_workspace_bytes(rows, m, k) bytes, 16-byte aligned (centre norms + per-split
 * candidates); not retained after the launch has run. */
typedef struct SvkClusterL2TopkArgs {
  const uint16_t* tokens;         /* [rows, 2*Hkv*D] bf16 (token_stride)     */
  const uint16_t* k_cache;        /* [slots, Hkv*D] bf16 (kv_slot_stride)    */
  const uint16_t* v_cache;
  const int32_t* center_slots;    /* [m]                                     */
  const int32_t* new_center_rel;  /* [m - m0]                                */
  int32_t* topk;                  /* [rows, k] (topk_stride)                 */
  void* workspace;
  int64_t workspace_bytes;
  int64_t token_stride, kv_slot_stride, topk_stride;
  int32_t rows, m, m0, k, row_offset, half_dim;   /* half_dim = Hkv*D */
} SvkClusterL2TopkArgs;
int64_t svk_cluster_l2_topk_workspace_bytes(int32_t rows, int32_t m, int32_t k);
int svk_cluster_l2_topk(const SvkClusterL2TopkArgs* a, svk_stream_t stream);

/* base[r] = mean over the k father rows of concat(K[slot], V[slot]) (fp32 accumulate, bf16 out): the
 * `all_centers.gather(...).mean(dim=2)` of _cluster_compress (:2788-2789) / batch_gather_mean
 * (kernels/triton/deltakv_kernels.py:2268-2301), reading the centre rows straight from the layer's cache.
 * father = center_slots[topk[r, j]]. */
typedef struct SvkGatherMeanArgs {
  const uint16_t* k_cache;        /* [slots, Hkv, D] bf16 */
  const uint16_t* v_cache;
  const int32_t* center_slots;    /* [m] slot of every centre column          */
  const int32_t* topk;            /* [rows, k] column indices (topk_stride)   */
  uint16_t* base;                 /* [rows, 2*Hkv*D] bf16 (base_stride)       */
  int32_t* father_slots;          /* NULL or [rows, k_out]: center_slots[topk], padded with the first father */
  int64_t kv_slot_stride, kv_head_stride, topk_stride, base_stride, father_stride;
  int32_t rows, k, k_out, num_kv_heads, head_dim;
} SvkGatherMeanArgs;
int svk_gather_mean_fathers(const SvkGatherMeanArgs* a, svk_stream_t stream);

/* Decode stage 1 over a full-attention layer whose older tokens are KIVI-int4 blocks and whose newest
 * tokens are raw bf16 rows; optional 3-D raw scores (observation layers).  Same partial format as
 * svk_flash_decode_stage1.  Per token t of row r: raw_slots_map[r,t] >= 0 -> raw row, else block
 * b = kivi_block_slots_map[r,t], local = t - kivi_block_start_pos[b] in [0, group_size):
 *   K = code_K[b,h,d,local]*key_scales[b,h,d] + key_mins[b,h,d];  V = code_V[b,h,local,d]*vs[b,h,local,d/G] + vm
 * (4-bit codes, 8 per int32, LSB first), both rounded to bf16 like `.to(q.dtype)`.
 * Replaces full_layer_kivi_flash_decode_stage1, kernels/triton/deltakv_kernels.py:973-1142 (kernel :675-929). */
typedef struct SvkKiviDecodeStage1Args {
  const uint16_t* q;                 /* [B, Hq, D] bf16                                  */
  const uint16_t* raw_k;             /* [slots, Hkv, D] bf16                             */
  const uint16_t* raw_v;
  const int32_t* raw_slots_map;      /* [rows, map_stride]                               */
  const int32_t* kivi_block_slots_map; /* [rows, map_stride]                             */
  const int32_t* kivi_block_start_pos; /* [blocks]                                       */
  const int32_t* key_packed;         /* [blocks, Hkv, D, G/8]                            */
  const void* key_scales;            /* [blocks, Hkv, D] f32 or bf16 (key_param_dtype)   */
  const void* key_mins;
  const int32_t* value_packed;       /* [blocks, Hkv, G, D/8]                            */
  const uint16_t* value_scales;      /* [blocks, Hkv, G, D/G] bf16                       */
  const uint16_t* value_mins;
  const int32_t* req_indices;        /* [B]                                              */
  const int32_t* context_lens;       /* [B]                                              */
  float* mid_o;                      /* [B, Hq, nblk, D]                                 */
  float* mid_lse;                    /* [B, Hq, nblk]                                    */
  float* attn_score;                 /* NULL or [B, Hq, W] raw logits                    */
  int64_t q_stride_b, q_stride_h, raw_slot_stride, raw_head_stride, map_stride;
  int64_t mid_o_stride_b, mid_o_stride_h, mid_o_stride_s, mid_lse_stride_b, mid_lse_stride_h;
  int64_t score_stride_b, score_stride_h;
  int32_t batch, num_q_heads, num_kv_heads, head_dim, max_len_in_batch, block_seq, group_size;
  int32_t key_param_dtype;           /* SVK_DTYPE_F32 (reference manager) or SVK_DTYPE_BF16 */
  /* MI355X: 0, or svk_kivi_decode_stage1_extra_partials(args) when mid_o / mid_lse have that many partial slots more
   * than ceil(max_len_in_batch / block_seq).  The launch then gives the latency-bound pieces of every row (raw sink
   * tokens, raw residual tail, the ragged quantised piece in front of it) to extra workgroups that write the partials
   * nblk_row .. nblk_row + extra - 1 (nblk_row = ceil(len / block_seq)); partials of regular blocks at or past nblk_row
   * are then NOT written, and stage 2 must be told (SvkFlashDecodeStage2Args.extra_partials). */
  int32_t extra_partials;
  /* MI355X: this step's raw store of the layer (store_kvcache of the B new rows into raw_k / raw_v at new_slots,
   * engine/cache_manager/base.py:629-694 `_store_layer_kv`) riding in the launch, like SvkFlashDecodeStage1Args.new_k:
   * the workgroup that owns position len - 1 of row b writes new_k/new_v[b] to raw slot new_slots[b] (skipped when < 0)
   * before it reads the row.  Only launches svk_kivi_decode_stage1_extra_partials() reports > 0 for (the wide kernel)
   * carry it; NULL = the caller stores.  The result equals store-then-launch. */
  const uint16_t* new_k;             /* NULL or [B, Hkv, D] bf16 (new_stride_b / new_stride_h) */
  const uint16_t* new_v;
  const int32_t* new_slots;          /* [B]                                              */
  int64_t new_stride_b, new_stride_h;
} SvkKiviDecodeStage1Args;
int svk_kivi_decode_stage1(const SvkKiviDecodeStage1Args* a, svk_stream_t stream);
/* extra partial slots the launch described by `a` can use (its extra_partials field is ignored): 3 or 0 */
int32_t svk_kivi_decode_stage1_extra_partials(const SvkKiviDecodeStage1Args* a);

/* Observation-layer token scores for the query-aware top-k:
 *   s[b, t] = max_h softmax_{t in [start, start+len_b)} (raw[b,h,t] * scale), cast to `round_dtype`,
 *   everything outside the candidate range = fill_value.
 * Replaces SparseController._decode_softmax_token_scores, engine/sparse_controller.py:255-299. */
typedef struct SvkDeltakvTokenScoresArgs {
  const float* raw_scores;        /* [B, H, L] f32 (3-D output of stage 1)        */
  const int32_t* candidate_lens;  /* [B]                                          */
  float* token_scores;            /* [B, L] f32 out                               */
  float* workspace;               /* [B, H, svk_deltakv_token_scores_chunks(L), 2] f32 partial max / sum + a ticket word per
                                     (row, head): ZERO before the first launch, zero again after every launch (ABI 19) */
  int64_t raw_stride_b, raw_stride_h, out_stride;
  float scale, fill_value;
  int32_t batch, num_heads, length, candidate_start;
  int32_t round_dtype;            /* SVK_DTYPE_*: F32 = no rounding               */
} SvkDeltakvTokenScoresArgs;
int svk_deltakv_token_scores(const SvkDeltakvTokenScoresArgs* a, svk_stream_t stream);
/* statistics slots per (row, head) the workspace must hold for score rows of `length` elements: one per 4096-element
 * chunk, one for their combination and one for the ticket of the chunk that combines them (the last to arrive) */
int svk_deltakv_token_scores_chunks(int32_t length);

/* idx[r, :k] = indices of the k largest of scores[r, :n] ordered by (score desc, index asc) -
 * `topk(k, sorted=True)` with a deterministic tie rule (the reference adds a position key to get one,
 * sparse_controller.py:1797-1811).  Entries at index >= valid_len[r] compare as `masked_value`.
 * Replaces the DeltaKV branch of _update_dynamic_omnikv_indices, sparse_controller.py:1790-1822. k <= 4096.
 * Long rows are split over several workgroups through `workspace` (svk_topk_sorted_workspace_bytes(); may be NULL
 * when that returns 0, and a NULL workspace always selects the single-workgroup path).  ABI 19: the first
 * rows x 16 KiB of a non-NULL workspace (the level-1 histograms of the two-level plan) must be ZERO when the launch
 * starts and are zero again when it has run - zero-fill a workspace once and reuse it call after call (one launch
 * at a time per workspace); the rest of the workspace needs no initialisation. */
typedef struct SvkTopkSortedArgs {
  const float* scores;        /* [rows, score_stride]            */
  const int32_t* valid_len;   /* NULL or [rows]                  */
  int32_t* indices;           /* [rows, index_stride] int32 out  */
  int64_t score_stride, index_stride;
  float masked_value;
  int32_t rows, n, k;
} SvkTopkSortedArgs;
int64_t svk_topk_sorted_workspace_bytes(int32_t rows, int32_t n, int32_t k);
int svk_topk_sorted_desc(const SvkTopkSortedArgs* a, void* workspace, svk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SVK_H_ */
