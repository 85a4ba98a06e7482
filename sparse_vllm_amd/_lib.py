"""ctypes binding of the C-ABI HIP library (include/svk.h -> libsvk.so).

There is NO fallback: if the library is missing or a symbol is absent, importing the
kernels fails loudly.  PyTorch is used only for device memory and streams; the
structs below carry raw device pointers and sizes.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvk.so")

SVK_OK = 0
SVK_ERR_VALUE = -1
SVK_ERR_LAYOUT = -2
SVK_ERR_STATE = -3
SVK_ERR_LAUNCH = -4

SVK_SCORE_NONE = 0
SVK_SCORE_HEADMAX = 2
SVK_SCORE_PERHEAD = 3

SVK_ABI_VERSION = 20

SVK_PREFILL_SCORE_PROBABILITY = 0
SVK_PREFILL_SCORE_LOGITS = 1

SVK_DTYPE_F32, SVK_DTYPE_BF16, SVK_DTYPE_F16 = 0, 1, 2

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f32 = C.c_float


class SvkStoreKvcacheArgs(C.Structure):
    _fields_ = [("key", _p), ("value", _p), ("k_cache", _p), ("v_cache", _p), ("slot_mapping", _p),
                ("key_stride", _i64), ("value_stride", _i64), ("n_tokens", _i32), ("row_elems", _i32)]


class SvkCopySlotsArgs(C.Structure):
    _fields_ = [("k_cache", _p), ("v_cache", _p), ("src_slots", _p), ("dst_slots", _p), ("workspace", _p),
                ("n", _i32), ("row_elems", _i32)]


class SvkFlashDecodeStage1Args(C.Structure):
    _fields_ = [("q", _p), ("k_cache", _p), ("v_cache", _p), ("req_to_tokens", _p), ("b_req_idx", _p),
                ("b_seqlen", _p), ("mid_o", _p), ("mid_lse", _p), ("attn_score", _p),
                ("q_stride_b", _i64), ("q_stride_h", _i64), ("kv_slot_stride", _i64), ("kv_head_stride", _i64),
                ("kv_num_slots", _i64), ("req_stride", _i64), ("mid_o_stride_b", _i64), ("mid_o_stride_h", _i64), ("mid_o_stride_s", _i64),
                ("mid_lse_stride_b", _i64), ("mid_lse_stride_h", _i64), ("score_stride_b", _i64),
                ("score_stride_h", _i64),
                ("batch", _i32), ("num_q_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
                ("max_len_in_batch", _i32), ("block_seq", _i32), ("score_mode", _i32),
                ("new_k", _p), ("new_v", _p), ("slot_mapping", _p), ("new_stride_b", _i64), ("new_stride_h", _i64),
                ("direct_o", _p), ("direct_stride_b", _i64), ("direct_stride_h", _i64), ("score_overwrite", _i32), ("slot_page_size", _i32),
                ("new_cos_sin", _p), ("new_slot_to_pos", _p), ("new_row_lens", _p), ("new_k_norm_weight", _p), ("raw_k_cache", _p), ("raw_v_cache", _p),
                ("raw_slot_stride", _i64), ("raw_head_stride", _i64), ("new_cos_stride", _i64),
                ("raw_num_slots", _i32), ("new_cos_dtype", _i32), ("new_k_norm_eps", _f32), ("_pad_rot", _i32)]


class SvkFlashDecodeStage2Args(C.Structure):
    _fields_ = [("mid_o", _p), ("mid_lse", _p), ("b_seqlen", _p), ("o", _p),
                ("mid_o_stride_b", _i64), ("mid_o_stride_h", _i64), ("mid_o_stride_s", _i64),
                ("mid_lse_stride_b", _i64), ("mid_lse_stride_h", _i64), ("o_stride_b", _i64), ("o_stride_h", _i64),
                ("batch", _i32), ("num_q_heads", _i32), ("head_dim", _i32), ("block_seq", _i32), ("extra_partials", _i32),
                ("max_partials", _i32), ("split_ws", _p), ("split_ws_bytes", _i64), ("split_tickets", _p)]


class SvkH2oDecodeScoreArgs(C.Structure):
    _fields_ = [("attn_score", _p), ("cum_score", _p), ("b_req_idx", _p), ("b_seqlen", _p), ("b_new_slot", _p),
                ("score_stride_b", _i64), ("cum_stride", _i64), ("scale", _f32), ("batch", _i32), ("width", _i32),
                ("mask_by_len", _i32), ("_pad0", _i32)]


class SvkH2oSelectArgs(C.Structure):
    _fields_ = [("scores", _p), ("keep", _p), ("score_stride", _i64), ("keep_stride", _i64),
                ("rows", _i32), ("kv_len", _i32), ("budget", _i32), ("recent_count", _i32)]


class SvkSelectTopkArgs(C.Structure):
    _fields_ = [("scores", _p), ("keep", _p), ("score_stride", _i64), ("keep_stride", _i64),
                ("rows", _i32), ("kv_len", _i32), ("prefix", _i32), ("topk", _i32), ("suffix", _i32)]


class SvkCompactRowsArgs(C.Structure):
    _fields_ = [("slot_table", _p), ("free_stack", _p), ("row_payload", _p), ("keep", _p), ("layer_ids", _p),
                ("row_ids", _p), ("free_base", _p),
                ("table_stride_layer", _i64), ("table_stride_row", _i64), ("stack_stride", _i64),
                ("payload_stride_layer", _i64), ("payload_stride_row", _i64),
                ("n_layers", _i32), ("n_lanes", _i32), ("cur_len", _i32), ("keep_len", _i32)]


class SvkDecodeAllocArgs(C.Structure):
    _fields_ = [("slot_table", _p), ("free_stack", _p), ("layer_ids", _p), ("row_ids", _p), ("cur_lens", _p),
                ("free_ptrs", _p), ("slot_mapping", _p), ("context_lens", _p), ("req_indices", _p),
                ("table_stride_layer", _i64), ("table_stride_row", _i64), ("stack_stride", _i64),
                ("out_stride", _i64), ("meta_stride_layer", _i64), ("free_ptr", _i64),
                ("n_layers", _i32), ("batch", _i32), ("graph_batch", _i32)]


class SvkH2oDeviceStepArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("slot_table", "free_stack", "scores", "row_len", "free_ptr", "row_ids", "slot_mapping",
                                  "context_lens", "req_indices", "keep")] + \
               [(n, _i64) for n in ("table_stride_layer", "table_stride_row", "stack_stride", "score_stride_layer",
                                    "score_stride_row", "out_stride")] + \
               [(n, _i32) for n in ("n_layers", "rows_total", "batch", "graph_batch", "budget", "recent_count",
                                    "trigger_len", "select_mode", "prefix_count", "_pad")] + [("tickets", _p)]


class SvkQuestDeviceStepArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("page_table", "token_table", "row_len", "free_pages", "free_page_ptr", "row_ids", "slot_mapping",
                                  "context_lens", "req_indices", "k_cache", "metadata")] + \
               [(n, _i64) for n in ("page_table_stride", "token_table_stride", "k_layer_stride", "meta_kind_stride",
                                    "meta_layer_stride")] + \
               [(n, _i32) for n in ("batch", "graph_batch", "page_size", "n_layers", "row_elems", "_pad")]


class SvkQuestPageMinmaxArgs(C.Structure):
    _fields_ = [("k_cache", _p), ("metadata", _p), ("page_slots", _p),
                ("k_layer_stride", _i64), ("meta_kind_stride", _i64), ("meta_layer_stride", _i64),
                ("n_pages", _i32), ("n_layers", _i32), ("page_size", _i32), ("row_elems", _i32)]


class SvkQuestScorePagesArgs(C.Structure):
    _fields_ = [("q", _p), ("page_max", _p), ("page_min", _p), ("page_table", _p), ("req_indices", _p),
                ("context_lens", _p), ("page_scores", _p),
                ("q_stride_b", _i64), ("q_stride_h", _i64), ("page_table_stride", _i64), ("score_stride", _i64),
                ("batch", _i32), ("num_q_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
                ("page_size", _i32), ("n_prev", _i32)]


class SvkQuestBuildViewArgs(C.Structure):
    _fields_ = [("page_scores", _p), ("page_table", _p), ("token_table", _p), ("req_indices", _p),
                ("context_lens", _p), ("packed_slots", _p), ("local_lens", _p), ("local_req", _p),
                ("score_stride", _i64), ("page_table_stride", _i64), ("token_table_stride", _i64),
                ("packed_stride", _i64),
                ("batch", _i32), ("page_size", _i32), ("n_prev", _i32), ("prev_budget", _i32),
                ("token_budget", _i32), ("page_budget_base", _i32), ("max_keep", _i32), ("is_long_text", _i32),
                ("emit_page_slots", _i32), ("_pad0", _i32)]


class SvkQuestDecodeAllocArgs(C.Structure):
    _fields_ = [("page_table", _p), ("token_table", _p), ("row_ids", _p), ("cur_lens", _p), ("new_page_slots", _p),
                ("slot_mapping", _p), ("context_lens", _p), ("req_indices", _p),
                ("page_table_stride", _i64), ("token_table_stride", _i64),
                ("batch", _i32), ("graph_batch", _i32), ("page_size", _i32)]


class SvkPrefillScoreArgs(C.Structure):
    _fields_ = [("q", _p), ("k_cache", _p), ("attn_score", _p), ("b_req_idx", _p), ("b_start_loc", _p),
                ("b_seq_len", _p), ("b_prompt_cache_len", _p), ("req_to_tokens", _p), ("score_q_start", _p),
                ("score_q_end", _p), ("batch_indices", _p), ("workspace", _p),
                ("q_stride_t", _i64), ("q_stride_h", _i64), ("kv_slot_stride", _i64), ("kv_head_stride", _i64),
                ("req_stride", _i64), ("score_stride", _i64),
                ("n_ranges", _i32), ("num_q_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
                ("max_query_len", _i32), ("score_cols", _i32), ("candidate_start", _i32),
                ("num_recent_tokens", _i32), ("score_mode", _i32), ("_pad", _i32), ("row_stats", _p)]


class SvkDequantLinearBatch(C.Structure):
    _fields_ = [("n_batch", C.c_int32), ("packed_stride_batch", C.c_int64), ("scale_stride_batch", C.c_int64),
                ("weight_stride_batch", C.c_int64), ("bias_stride_batch", C.c_int64), ("out_stride_batch", C.c_int64)]


class SvkDeltakvReconstructBatch(C.Structure):
    _fields_ = [("n_batch", C.c_int32), ("delta_stride_batch", C.c_int64), ("father_table_stride_batch", C.c_int64),
                ("kv_cache_stride_batch", C.c_int64), ("k_norm_stride_batch", C.c_int64), ("out_cache_stride_batch", C.c_int64)]


class SvkDeltakvUpReconArgs(C.Structure):
    _fields_ = [("hidden", _p), ("weight", _p), ("bias", _p), ("hidden_stride", C.c_int64), ("hidden_stride_batch", C.c_int64),
                ("weight_stride", C.c_int64), ("weight_stride_batch", C.c_int64), ("bias_stride_batch", C.c_int64),
                ("k", C.c_int32), ("_pad0", C.c_int32)]


class SvkDeltakvDecodeAllocArgs(C.Structure):
    _fields_ = [("meta", _p), ("meta_stride", C.c_int64), ("full_slots_map", _p), ("full_map_stride", C.c_int64),
                ("full_slot_to_pos", _p), ("sparse_raw_slots_map", _p), ("sparse_map_stride", C.c_int64),
                ("sparse_slot_to_pos", _p), ("context_lens", _p), ("req_indices", _p), ("slot_mapping", _p),
                ("sparse_slot_mapping", _p), ("compressed_lens", _p), ("batch", C.c_int32), ("graph_batch", C.c_int32)]


class SvkDeltakvDeviceStepArgs(C.Structure):
    _fields_ = [("rows", _p), ("row_len", _p), ("compressed_len", _p), ("full_stack", _p), ("full_ptr", _p),
                ("sparse_stack", _p), ("sparse_ptr", _p), ("full_slots_map", _p), ("full_map_stride", C.c_int64),
                ("full_slot_to_pos", _p), ("sparse_raw_slots_map", _p), ("sparse_map_stride", C.c_int64),
                ("sparse_slot_to_pos", _p), ("context_lens", _p), ("req_indices", _p), ("slot_mapping", _p),
                ("sparse_slot_mapping", _p), ("compressed_lens", _p), ("batch", C.c_int32), ("graph_batch", C.c_int32)]


class SvkDeltakvPlanArgs(C.Structure):
    _fields_ = [("raw_slots_map", _p), ("latent_slots_map", _p), ("active_compressed", _p), ("req_indices", _p),
                ("context_lens", _p), ("compressed_lens", _p), ("temp_slots", _p), ("active_slots_out", _p),
                ("active_pos_out", _p), ("new_context_lens_out", _p), ("recon_pos_out", _p), ("recon_latent_out", _p),
                ("recon_out_slot_out", _p),
                ("raw_stride", _i64), ("latent_stride", _i64), ("active_stride", _i64), ("temp_stride", _i64),
                ("out_stride", _i64), ("pos_stride", _i64),
                ("batch", _i32), ("k_max", _i32), ("sink", _i32), ("max_buffer", _i32), ("max_positions", _i32)]


class SvkDeltakvReconstructArgs(C.Structure):
    _fields_ = [("delta", _p), ("scale", _p), ("mn", _p), ("latent_slots", _p), ("father_slots", _p),
                ("slot_to_pos", _p), ("out_slots", _p), ("out_pos", _p), ("cos_sin", _p), ("k_cache", _p),
                ("v_cache", _p), ("k_norm_weight", _p),
                ("delta_stride", _i64), ("scale_stride", _i64), ("father_stride", _i64), ("cos_stride", _i64),
                ("kv_slot_stride", _i64), ("kv_head_stride", _i64),
                ("k_norm_eps", _f32), ("n", _i32), ("k_fathers", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
                ("delta_bits", _i32), ("group_size", _i32), ("delta_dtype", _i32), ("scale_dtype", _i32),
                ("cos_dtype", _i32), ("raw_k_cache", _i32), ("store_raw_k", _i32),
                ("father_table", _p), ("father_index", _p), ("father_table_stride", _i64),
                ("out_k_cache", _p), ("out_v_cache", _p), ("out_slot_stride", _i64), ("out_head_stride", _i64),
                ("out_view_width", _i32), ("out_view_offset", _i32), ("out_entries_per_row", _i32), ("_pad1", _i32)]


class SvkDequantGroupedArgs(C.Structure):
    _fields_ = [("packed", _p), ("scale", _p), ("mn", _p), ("out", _p),
                ("packed_stride", _i64), ("scale_stride", _i64), ("out_stride", _i64),
                ("rows", _i32), ("features", _i32), ("bits", _i32), ("group_size", _i32),
                ("scale_dtype", _i32), ("out_dtype", _i32), ("row_index", _p)]


class SvkDequantLinearArgs(C.Structure):
    _fields_ = [("packed", _p), ("scale", _p), ("mn", _p), ("row_index", _p), ("weight", _p), ("bias", _p), ("out", _p),
                ("packed_stride", _i64), ("scale_stride", _i64), ("weight_stride", _i64), ("out_stride", _i64),
                ("rows", _i32), ("k", _i32), ("n", _i32), ("group_size", _i32), ("scale_dtype", _i32),
                ("activation", _i32)]


class SvkDeltakvTokenScoresArgs(C.Structure):
    _fields_ = [("raw_scores", _p), ("candidate_lens", _p), ("token_scores", _p), ("workspace", _p),
                ("raw_stride_b", _i64), ("raw_stride_h", _i64), ("out_stride", _i64),
                ("scale", _f32), ("fill_value", _f32),
                ("batch", _i32), ("num_heads", _i32), ("length", _i32), ("candidate_start", _i32),
                ("round_dtype", _i32)]


class SvkTopkSortedArgs(C.Structure):
    _fields_ = [("scores", _p), ("valid_len", _p), ("indices", _p), ("score_stride", _i64), ("index_stride", _i64),
                ("masked_value", _f32), ("rows", _i32), ("n", _i32), ("k", _i32)]


class SvkDeltakvMaterializeArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("active_slots", "slot_to_pos", "postrope_mask", "k_cache", "v_cache", "out_k", "out_v",
                                  "cos_sin", "k_norm_weight")] + \
               [(n, _i64) for n in ("active_stride", "kv_slot_stride", "kv_head_stride", "out_slot_stride",
                                    "out_head_stride", "cos_stride")] + \
               [("k_norm_eps", _f32)] + \
               [(n, _i32) for n in ("batch", "width", "num_slots", "num_kv_heads", "head_dim", "cos_dtype")] + \
               [("temp_slots", _p), ("temp_stride", _i64), ("temp_offset", _i32), ("temp_count", _i32)] + \
               [("new_k", _p), ("new_v", _p), ("new_slots", _p), ("new_token_stride", _i64), ("new_head_stride", _i64)] + \
               [("skip_temp", _i32), ("skip_new", _i32), ("kv_layer_stride", _i64), ("out_layer_stride", _i64),
                ("k_norm_layer_stride", _i64), ("layer_count", _i32), ("_pad0", _i32)]


class SvkContextAttentionArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("q", "k_cache", "v_cache", "o", "b_req_idx", "b_start_loc", "b_seq_len",
                                  "b_prompt_cache_len", "req_to_tokens")] + \
               [(n, _i64) for n in ("q_stride_t", "q_stride_h", "kv_slot_stride", "kv_head_stride", "o_stride_t", "o_stride_h",
                                    "req_stride")] + \
               [(n, _i32) for n in ("batch", "num_q_heads", "num_kv_heads", "head_dim", "max_input_len")] + \
               [("_pad", _i32), ("kv_num_slots", _i64), ("score_row_stats", _p), ("score_q_start", _p), ("score_clear", _p),
                ("score_clear_stride", _i64), ("score_wpad", _i32), ("score_clear_cols", _i32),
                ("attn_score", _p), ("score_workspace", _p), ("attn_score_stride_b", _i64), ("attn_score_stride_h", _i64),
                ("attn_score_dim", _i32), ("attn_score_cols", _i32)]


class SvkQuantPackArgs(C.Structure):
    _fields_ = [("data", _p), ("dst_rows", _p), ("code", _p), ("scale", _p), ("mn", _p),
                ("data_stride", _i64), ("code_stride", _i64), ("scale_stride", _i64),
                ("rows", _i32), ("features", _i32), ("bits", _i32), ("group_size", _i32), ("data_dtype", _i32)]


class SvkKiviStoreArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("k_cache", "v_cache", "raw_slots", "block_slots", "key_packed", "key_scales", "key_mins",
                                  "value_packed", "value_scales", "value_mins")] + \
               [("kv_slot_stride", _i64), ("kv_head_stride", _i64)] + \
               [(n, _i32) for n in ("blocks", "num_kv_heads", "head_dim", "group_size", "key_param_dtype")]


class SvkClusterTopkArgs(C.Structure):
    _fields_ = [("scores", _p), ("new_center_rel", _p), ("topk", _p), ("score_stride", _i64), ("topk_stride", _i64)] + \
               [(n, _i32) for n in ("rows", "m", "m0", "k", "row_offset", "score_dtype")]


class SvkCheckSlotTableArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("slot_table", "req_indices", "context_lens", "status")] + [("table_stride", _i64)] + \
               [(n, _i32) for n in ("batch", "num_rows", "width", "slot_cap", "slot_page_size")]


class SvkClusterL2TopkArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("tokens", "k_cache", "v_cache", "center_slots", "new_center_rel", "topk", "workspace")] + \
               [(n, _i64) for n in ("workspace_bytes", "token_stride", "kv_slot_stride", "topk_stride")] + \
               [(n, _i32) for n in ("rows", "m", "m0", "k", "row_offset", "half_dim")]


class SvkGatherMeanArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("k_cache", "v_cache", "center_slots", "topk", "base", "father_slots")] + \
               [(n, _i64) for n in ("kv_slot_stride", "kv_head_stride", "topk_stride", "base_stride", "father_stride")] + \
               [(n, _i32) for n in ("rows", "k", "k_out", "num_kv_heads", "head_dim")]


class SvkKiviDecodeStage1Args(C.Structure):
    _fields_ = [(n, _p) for n in ("q", "raw_k", "raw_v", "raw_slots_map", "kivi_block_slots_map", "kivi_block_start_pos",
                                  "key_packed", "key_scales", "key_mins", "value_packed", "value_scales", "value_mins",
                                  "req_indices", "context_lens", "mid_o", "mid_lse", "attn_score")] + \
               [(n, _i64) for n in ("q_stride_b", "q_stride_h", "raw_slot_stride", "raw_head_stride", "map_stride",
                                    "mid_o_stride_b", "mid_o_stride_h", "mid_o_stride_s", "mid_lse_stride_b",
                                    "mid_lse_stride_h", "score_stride_b", "score_stride_h")] + \
               [(n, _i32) for n in ("batch", "num_q_heads", "num_kv_heads", "head_dim", "max_len_in_batch", "block_seq",
                                    "group_size", "key_param_dtype", "extra_partials")] + \
               [("new_k", _p), ("new_v", _p), ("new_slots", _p), ("new_stride_b", _i64), ("new_stride_h", _i64)]


# symbol -> (argtypes) ; every entry point declared in include/svk.h
ENTRY_POINTS = {
    "svk_abi_version": ([], C.c_int),
    "svk_last_error": ([], C.c_char_p),
    "svk_build_flags": ([], C.c_int),
    "svk_store_kvcache": ([C.POINTER(SvkStoreKvcacheArgs), _p], C.c_int),
    "svk_copy_slots": ([C.POINTER(SvkCopySlotsArgs), _p], C.c_int),
    "svk_flash_decode_stage1": ([C.POINTER(SvkFlashDecodeStage1Args), _p], C.c_int),
    "svk_flash_decode_stage2": ([C.POINTER(SvkFlashDecodeStage2Args), _p], C.c_int),
    "svk_flash_decode_stage2_split_workspace_bytes": ([C.c_int32, C.c_int32, C.c_int32, C.c_int32], C.c_int64),
    "svk_kivi_decode_stage1_extra_partials": ([C.POINTER(SvkKiviDecodeStage1Args)], C.c_int32),
    "svk_fill_f32": ([_p, _i64, _f32, _p], C.c_int),
    "svk_h2o_decode_score_update": ([C.POINTER(SvkH2oDecodeScoreArgs), _p], C.c_int),
    "svk_h2o_decode_score_update_layers": ([C.POINTER(SvkH2oDecodeScoreArgs), C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _p], C.c_int),
    "svk_h2o_select_indices": ([C.POINTER(SvkH2oSelectArgs), _p], C.c_int),
    "svk_select_prefix_topk_suffix": ([C.POINTER(SvkSelectTopkArgs), _p], C.c_int),
    "svk_compact_rows": ([C.POINTER(SvkCompactRowsArgs), _p], C.c_int),
    "svk_decode_alloc_slots": ([C.POINTER(SvkDecodeAllocArgs), _p], C.c_int),
    "svk_h2o_device_step_begin": ([C.POINTER(SvkH2oDeviceStepArgs), _p], C.c_int),
    "svk_h2o_device_burst": ([C.POINTER(SvkH2oDeviceStepArgs), _p], C.c_int),
    "svk_prefill_score_workspace_bytes": ([_i32, _i32, _i32, _i32, _i32], C.c_int64),
    "svk_prefill_score_window_pad": ([_i32, _i32, _i32], C.c_int32),
    "svk_prefill_score": ([C.POINTER(SvkPrefillScoreArgs), _p], C.c_int),
    "svk_deltakv_decode_alloc": ([C.POINTER(SvkDeltakvDecodeAllocArgs), _p], C.c_int),
    "svk_deltakv_device_step_begin": ([C.POINTER(SvkDeltakvDeviceStepArgs), _p], C.c_int),
    "svk_deltakv_static_decode_plan": ([C.POINTER(SvkDeltakvPlanArgs), _p], C.c_int),
    "svk_deltakv_reconstruct_writeback": ([C.POINTER(SvkDeltakvReconstructArgs), _p], C.c_int),
    "svk_deltakv_reconstruct_writeback_batched": ([C.POINTER(SvkDeltakvReconstructArgs), C.POINTER(SvkDeltakvReconstructBatch), _p], C.c_int),
    "svk_deltakv_up_reconstruct": ([C.POINTER(SvkDeltakvUpReconArgs), C.POINTER(SvkDeltakvReconstructArgs),
                                    C.POINTER(SvkDeltakvReconstructBatch), _p], C.c_int),
    "svk_dequantize_grouped": ([C.POINTER(SvkDequantGroupedArgs), _p], C.c_int),
    "svk_dequant_linear_act": ([C.POINTER(SvkDequantLinearArgs), _p], C.c_int),
    "svk_dequant_linear_act_batched": ([C.POINTER(SvkDequantLinearArgs), C.POINTER(SvkDequantLinearBatch), _p], C.c_int),
    "svk_deltakv_token_scores": ([C.POINTER(SvkDeltakvTokenScoresArgs), _p], C.c_int),
    "svk_deltakv_token_scores_chunks": ([_i32], C.c_int),
    "svk_topk_sorted_workspace_bytes": ([_i32, _i32, _i32], C.c_int64),
    "svk_topk_sorted_desc": ([C.POINTER(SvkTopkSortedArgs), _p, _p], C.c_int),
    "svk_deltakv_materialize_sparse_view": ([C.POINTER(SvkDeltakvMaterializeArgs), _p], C.c_int),
    "svk_context_attention_fwd": ([C.POINTER(SvkContextAttentionArgs), _p], C.c_int),
    "svk_context_attention_score_workspace_bytes": ([_i64, _i32, _i32], C.c_int64),
    "svk_quantize_pack_grouped": ([C.POINTER(SvkQuantPackArgs), _p], C.c_int),
    "svk_kivi_store_blocks": ([C.POINTER(SvkKiviStoreArgs), _p], C.c_int),
    "svk_cluster_topk": ([C.POINTER(SvkClusterTopkArgs), _p], C.c_int),
    "svk_check_slot_table": ([C.POINTER(SvkCheckSlotTableArgs), _p], C.c_int),
    "svk_cluster_l2_topk": ([C.POINTER(SvkClusterL2TopkArgs), _p], C.c_int),
    "svk_cluster_l2_topk_workspace_bytes": ([_i32, _i32, _i32], C.c_int64),
    "svk_gather_mean_fathers": ([C.POINTER(SvkGatherMeanArgs), _p], C.c_int),
    "svk_kivi_decode_stage1": ([C.POINTER(SvkKiviDecodeStage1Args), _p], C.c_int),
    "svk_quest_page_minmax": ([C.POINTER(SvkQuestPageMinmaxArgs), _p], C.c_int),
    "svk_quest_score_pages": ([C.POINTER(SvkQuestScorePagesArgs), _p], C.c_int),
    "svk_quest_build_view": ([C.POINTER(SvkQuestBuildViewArgs), _p], C.c_int),
    "svk_quest_decode_alloc": ([C.POINTER(SvkQuestDecodeAllocArgs), _p], C.c_int),
    "svk_quest_device_step_begin": ([C.POINTER(SvkQuestDeviceStepArgs), _p], C.c_int),
    "svk_quest_device_step_end": ([C.POINTER(SvkQuestDeviceStepArgs), _p], C.c_int),
}

_lib = None


class SvkLibraryError(RuntimeError):
    pass


def load():
    """Load libsvk.so (once).  Raises SvkLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvkLibraryError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sparse_vllm_amd/csrc`. There is no CPU fallback for the sparse attention hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in ENTRY_POINTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise SvkLibraryError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.svk_abi_version() != SVK_ABI_VERSION:
        raise SvkLibraryError(f"libsvk.so ABI {lib.svk_abi_version()} != binding {SVK_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, lib=None):
    """Map a C status to the reference's exception classes (include/svk.h header)."""
    if status == SVK_OK:
        return
    lib = lib or load()
    msg = (lib.svk_last_error() or b"").decode("utf-8", "replace")
    if status == SVK_ERR_VALUE:
        raise ValueError(msg)
    if status == SVK_ERR_LAYOUT:
        raise AssertionError(msg)
    raise RuntimeError(msg)


def current_stream_handle() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> int | None:
    return None if t is None else int(t.data_ptr())
