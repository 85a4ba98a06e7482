"""Attention layer of the sparse path (mirror of layers/attention.py:75-257 and of
layers/attention_backend.py:113-153 `run_prefill`, :217-349 `run_decode`).

Per-layer call order is the reference's:
    prefill (:88-161): sparse_controller.get_prefill_selection -> cache_manager.before_prefill_layer_attention
        -> cache_manager.build_prefill_compute_view (ExplicitKVPayload or TypeError) -> [fake | run_prefill]
        -> cache_manager.collect_prefill_attention_score(layer, q, view, b_start_loc=, chunk_lens=)
        -> cache_manager.record_prefill_query(layer, q, view, b_start_loc=, chunk_lens=)
    decode (:162-250): get_decode_selection -> build_decode_compute_view -> get_decode_block_seq -> run_decode
        -> record_decode_query
    both: sparse_controller.on_layer_attention_end -> cache_manager.on_layer_attention_end
        -> release_layer_temp_slots (finally)
"""

from __future__ import annotations

import os

import torch

from ..engine.cache_manager.base import DecodeComputeView, ExplicitKVPayload, PrefillComputeView
from ..kernels import flash_decode_stage1, flash_decode_stage1_with_score, flash_decode_stage2
from ..kernels.context_flashattention_nopad import context_attention_fwd
from ..kernels.gqa_flash_decoding_stage1 import direct_out_supported
from ..kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1
from ..utils.context import get_context
from ..utils.profiler import profiler


def get_decode_workspace(context, batch_size: int, num_heads: int, num_blocks: int, head_dim: int,
                         device: torch.device) -> tuple[torch.Tensor, torch.Tensor]:
    """layers/attention.py:15-51: grow-only fp32 partial buffers kept on the context."""
    mid_o = context.decode_mid_o
    if (mid_o is None or mid_o.device != device or mid_o.shape[0] < batch_size or mid_o.shape[1] < num_heads
            or mid_o.shape[2] < num_blocks or mid_o.shape[3] < head_dim):
        mid_o = torch.empty((batch_size, num_heads, num_blocks, head_dim), dtype=torch.float32, device=device)
        context.decode_mid_o = mid_o
    mid_lse = context.decode_mid_o_logexpsum
    if (mid_lse is None or mid_lse.device != device or mid_lse.shape[0] < batch_size or mid_lse.shape[1] < num_heads
            or mid_lse.shape[2] < num_blocks):
        mid_lse = torch.empty((batch_size, num_heads, num_blocks), dtype=torch.float32, device=device)
        context.decode_mid_o_logexpsum = mid_lse
    return (mid_o[:batch_size, :num_heads, :num_blocks, :head_dim], mid_lse[:batch_size, :num_heads, :num_blocks])


def _env_truthy(name: str) -> bool:
    return os.environ.get(name, "").lower() in {"1", "true", "yes", "on"}


def _fake_prefill_attention_enabled() -> bool:
    """layers/attention_backend.py:25-47: SPARSEVLLM_FAKE_{PREFILL_,}ATTENTION, refused unless
    SPARSEVLLM_ALLOW_FAKE_ATTENTION=1 (a fake output invalidates correctness and benchmark results)."""
    enabled = _env_truthy("SPARSEVLLM_FAKE_PREFILL_ATTENTION") or _env_truthy("SPARSEVLLM_FAKE_ATTENTION")
    if enabled and not _env_truthy("SPARSEVLLM_ALLOW_FAKE_ATTENTION"):
        raise RuntimeError(
            "Sparse-vLLM fake attention was requested, but it is disabled by default because it "
            "invalidates correctness and benchmark results. Set SPARSEVLLM_ALLOW_FAKE_ATTENTION=1 "
            "only for explicit fake-attention tests or profiling.")
    return enabled


def _fake_decode_attention_enabled() -> bool:
    """layers/attention_backend.py:50-55: SPARSEVLLM_FAKE_{DECODE_,}ATTENTION, same refusal as the prefill switch."""
    enabled = _env_truthy("SPARSEVLLM_FAKE_DECODE_ATTENTION") or _env_truthy("SPARSEVLLM_FAKE_ATTENTION")
    if enabled and not _env_truthy("SPARSEVLLM_ALLOW_FAKE_ATTENTION"):
        raise RuntimeError(
            "Sparse-vLLM fake attention was requested, but it is disabled by default because it "
            "invalidates correctness and benchmark results. Set SPARSEVLLM_ALLOW_FAKE_ATTENTION=1 "
            "only for explicit fake-attention tests or profiling.")
    return enabled


def _fake_attention_output(q: torch.Tensor) -> torch.Tensor:
    """layers/attention_backend.py:58-69."""
    mode = os.environ.get("SPARSEVLLM_FAKE_ATTENTION_MODE", "zero").strip().lower()
    if mode in {"zero", "zeros"}:
        return torch.zeros_like(q)
    if mode == "copy":
        return q.clone()
    if mode == "empty":
        return torch.empty_like(q)
    raise ValueError("SPARSEVLLM_FAKE_ATTENTION_MODE must be one of 'zero', 'copy', or 'empty', "
                     f"got {mode!r}.")


def _require_explicit_payload(view, *, operation: str) -> ExplicitKVPayload:
    payload = view.payload
    if not isinstance(payload, ExplicitKVPayload):
        raise TypeError(f"{operation} requires ExplicitKVPayload, got {type(payload).__name__}.")
    return payload


class HipAttentionBackend:
    """layers/attention_backend.py `TritonAttentionBackend.run_prefill / run_decode` on libsvk."""

    name = "hip"

    def maybe_run_fake_prefill(self, q: torch.Tensor, view: PrefillComputeView, *, chunk_lens: torch.Tensor,
                               max_input_len: int) -> torch.Tensor | None:
        """layers/attention_backend.py:97-111 (tests of the hook order without a GPU)."""
        if not _fake_prefill_attention_enabled():
            return None
        if view.meta.attn_score is not None:
            view.meta.attn_score.zero_()
        return _fake_attention_output(q)

    def debug_check_prefill_bounds(self, q: torch.Tensor, view: PrefillComputeView, *, chunk_lens: torch.Tensor) -> None:
        """layers/attention_backend.py:155-215: host-side bound check of the prompt view, SVLLM_DEBUG_PREFILL_BOUNDS=1."""
        if os.environ.get("SVLLM_DEBUG_PREFILL_BOUNDS", "0") != "1":
            return
        if q.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        meta = view.meta
        if int(chunk_lens.sum().item()) != int(q.shape[0]):
            raise RuntimeError(f"prefill chunk_lens sum {int(chunk_lens.sum().item())} != q tokens {int(q.shape[0])}")
        if meta.active_slots.dim() == 2 and meta.context_lens.numel() > 0:
            if int(meta.context_lens.max().item()) > int(meta.active_slots.shape[1]):
                raise RuntimeError("prefill context length exceeds active slot table width: "
                                   f"context_lens_max={int(meta.context_lens.max().item())} "
                                   f"slot_table_len={int(meta.active_slots.shape[1])}")
            rows = int(meta.active_slots.shape[0])
            req = meta.req_indices
            if int(req.min().item()) < 0 or int(req.max().item()) >= rows:
                raise RuntimeError(f"prefill req_indices out of range for {rows} slot-table rows")

    def run_prefill(self, q: torch.Tensor, view: PrefillComputeView, *, b_start_loc: torch.Tensor,
                    chunk_lens: torch.Tensor, max_input_len: int) -> torch.Tensor:
        """layers/attention_backend.py:113-153: causal attention of the chunk's queries over [cached prefix | chunk]
        through the view's slot table."""
        payload = _require_explicit_payload(view, operation="HIP prefill")
        meta = view.meta
        b_seq_len = meta.context_lens
        if b_seq_len.numel() != chunk_lens.numel():
            layer_idx = getattr(get_context(), "now_layer_idx", None)
            raise RuntimeError(
                "prefill context_lens/chunk_lens batch mismatch: "
                f"layer={layer_idx} context_lens_shape={tuple(b_seq_len.shape)} "
                f"chunk_lens_shape={tuple(chunk_lens.shape)} q_shape={tuple(q.shape)} "
                f"req_indices_shape={tuple(meta.req_indices.shape)} "
                f"active_slots_shape={tuple(meta.active_slots.shape)}")
        md = payload.metadata or {}
        # b_seq_len - chunk_lens, derived once per chunk by the caller when it can (one small launch per layer otherwise)
        b_prompt_cache_len = md.get("b_prompt_cache_len")
        if b_prompt_cache_len is None:
            ctx = get_context()
            key = (b_seq_len.data_ptr(), b_seq_len._version, chunk_lens.data_ptr(), chunk_lens._version)
            cached = getattr(ctx, "_prefill_cache_len", None)
            if cached is None or cached[0] != key:
                cached = ctx._prefill_cache_len = (key, b_seq_len - chunk_lens)
            b_prompt_cache_len = cached[1]
        self.debug_check_prefill_bounds(q, view, chunk_lens=chunk_lens)
        if _fake_prefill_attention_enabled():
            if meta.attn_score is not None:
                meta.attn_score.zero_()
            return _fake_attention_output(q)
        o = torch.empty_like(q)
        # the kernel's grid is ceil(max_input_len / tile) query tiles per sequence: the reference passes the longest
        # CONTEXT (an upper bound, surplus programs exit); the host-known longest CHUNK is the tight bound
        max_chunk = get_context().max_chunk_len
        grid_len = int(max_input_len) if max_chunk is None else min(int(max_input_len), int(max_chunk))
        # MI355X: a manager whose prefill token scores use the attention's own softmax statistics (H2O, probability
        # mode) put a request into the view's payload; the launch leaves the statistics behind and says so
        stats = md.get("score_stats")
        with profiler.record("prefill_attention"):
            context_attention_fwd(q, payload.k_cache, payload.v_cache, o, meta.req_indices, b_start_loc, b_seq_len,
                                  b_prompt_cache_len, grid_len, meta.active_slots, attn_score=meta.attn_score,
                                  score_stats=None if stats is None else stats["request"])
        if stats is not None:
            stats["written_for"] = (q.data_ptr(), q._version, tuple(q.shape))
        return o

    def _debug_check_decode_bounds(self, view: DecodeComputeView) -> None:
        """layers/attention_backend.py:397-439 (SVLLM_DEBUG_DECODE_BOUNDS=1, never under stream capture): request rows
        inside the slot table, the longest context inside its width, every visible slot id inside the KV pool.  The HIP
        kernels trust the slot table exactly as the reference's Triton kernels do; this host check is the reference's
        debugging aid for a corrupted table, with its messages."""
        mode = os.environ.get("SVLLM_DEBUG_DECODE_BOUNDS", "0")
        if mode == "device":
            # MI355X: the same three checks as a launch of the step (no synchronisation, capture-safe); the record is read
            # by `raise_if_slot_check_failed` wherever the caller synchronises anyway (SparseDecodeDriver: after the step)
            payload = _require_explicit_payload(view, operation="Decode bounds check")
            meta = view.meta
            if payload.backend in {"dense", "flash_attn_contiguous"} and meta.active_slots.dim() == 2 and meta.active_slots.is_cuda:
                from ..kernels.store_kvcache import check_slot_table_async
                page = int((payload.metadata or {}).get("slot_page_size", 0) or 0)
                cap = int(payload.k_cache.shape[0]) if page <= 1 else int(payload.k_cache.shape[0]) // page
                check_slot_table_async(meta.active_slots, meta.req_indices, meta.context_lens, slot_cap=cap, slot_page_size=page)
            return
        if mode != "1":
            return
        payload = _require_explicit_payload(view, operation="Decode bounds check")
        meta = view.meta
        if payload.backend not in {"dense", "flash_attn_contiguous"}:
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        if meta.active_slots.dim() != 2:
            raise RuntimeError(f"debug slot bounds check expects 2D active_slots, got shape={tuple(meta.active_slots.shape)}")
        rows = meta.req_indices.to(torch.long)
        row_min = int(rows.min().item()) if rows.numel() > 0 else 0
        row_max = int(rows.max().item()) if rows.numel() > 0 else -1
        num_rows, width = int(meta.active_slots.shape[0]), int(meta.active_slots.shape[1])
        if row_min < 0 or row_max >= num_rows:
            raise RuntimeError(f"decode req row index out of bounds: row_min={row_min} row_max={row_max} num_rows={num_rows}")
        visible_len = int(meta.context_lens.max().item()) if meta.context_lens.numel() > 0 else 0
        if visible_len > width:
            raise RuntimeError("decode visible length exceeds Req_to_tokens width: "
                               f"visible_len={visible_len} req_to_tokens_width={width}")
        page = int((payload.metadata or {}).get("slot_page_size", 0) or 0)
        visible = meta.active_slots.index_select(0, rows)[:, : visible_len if page <= 1 else -(-visible_len // page)]
        pos = torch.arange(visible.shape[1], device=visible.device)[None, :]
        lens = meta.context_lens[:, None] if page <= 1 else (meta.context_lens[:, None] + page - 1) // page
        slot_cap = int(payload.k_cache.shape[0]) if page <= 1 else int(payload.k_cache.shape[0]) // page
        bad = ((visible < 0) | (visible >= slot_cap)) & (pos < lens)
        if bool(bad.any().item()):
            b, p_ = (int(x) for x in bad.nonzero(as_tuple=False)[0])
            raise RuntimeError("decode physical slot out of bounds before attention: "
                               f"batch={b} req_row={int(rows[b].item())} pos={p_} slot={int(visible[b, p_].item())} "
                               f"slot_cap={slot_cap} context_len={int(meta.context_lens[b].item())}")

    def run_decode(self, q: torch.Tensor, view: DecodeComputeView, *, mid_o, mid_o_logexpsum, max_len_in_batch: int,
                   block_seq: int, num_heads: int, num_kv_heads: int, gqa_block_n: int = 16,
                   gqa_num_warps: int = 2, new_kv=None, score_overwrite: bool = False) -> torch.Tensor:
        """layers/attention_backend.py:217-349.  `new_kv` / `score_overwrite` are MI355X extras the attention layer
        passes only when they apply (this step's K/V rows riding in the stage-1 launch; scores stored instead of
        max-merged)."""
        payload = _require_explicit_payload(view, operation="HIP decode")
        meta = view.meta
        if _fake_decode_attention_enabled():
            if meta.attn_score is not None:
                meta.attn_score.zero_()
            return _fake_attention_output(q)
        kind = "full" if int(max_len_in_batch) > 8192 else "sparse"
        if payload.backend == "full_layer_kivi":
            # layers/attention_backend.py:236-247, :351-395
            md = payload.metadata
            if md is None:
                raise RuntimeError("full_layer_kivi decode view is missing metadata.")
            nblk = (int(max_len_in_batch) + int(block_seq) - 1) // int(block_seq)
            with profiler.record("decode_attention_stage1_kivi"):
                extra = full_layer_kivi_flash_decode_stage1(
                    q=q, raw_k=payload.k_cache, raw_v=payload.v_cache, raw_slots_map=meta.active_slots,
                    kivi_block_slots_map=md["kivi_block_slots_map"], kivi_block_start_pos=md["kivi_block_start_pos"],
                    key_packed=md["key_packed"], key_scales=md["key_scales"], key_mins=md["key_mins"],
                    value_packed=md["value_packed"], value_scales=md["value_scales"], value_mins=md["value_mins"],
                    req_indices=meta.req_indices, context_lens=meta.context_lens, max_len_in_batch=max_len_in_batch,
                    mid_out=mid_o, mid_out_logsumexp=mid_o_logexpsum, group_size=int(md["group_size"]),
                    block_seq=block_seq, block_n=int(md.get("block_n", 16)), num_warps=int(md.get("num_warps", 2)),
                    num_stages=int(md.get("num_stages", 3)), attn_score=meta.attn_score,
                    extra_partial_slots=max(0, int(mid_o.shape[2]) - nblk), new_kv=new_kv)
            o = torch.empty_like(q)
            flash_decode_stage2(mid_o, mid_o_logexpsum, meta.context_lens, o, block_seq, extra_partials=extra)
            return o
        self._debug_check_decode_bounds(view)
        if new_kv is not None and payload.backend not in ("dense", "full_layer_kivi"):
            raise RuntimeError("the fused decode store is only wired into the dense stage-1 launch")
        # MI355X: when one block covers every row of the launch the split-KV merge has nothing to merge and stage 1
        # writes the output itself (bit-identical, include/svk.h `direct_o`): no stage-2 launch for this layer
        direct = payload.backend == "dense" and direct_out_supported(max_len_in_batch, block_seq)
        o = torch.empty_like(q)
        direct_out = o if direct else None
        slot_page_size = int((payload.metadata or {}).get("slot_page_size", 0))     # a page-slot table (Quest view), scored or not
        # MI355X: a view whose newest row the manager left to this launch (DeltaKV sparse layers: raw rows to the pre-RoPE
        # cache, the rotated row into the view; include/svk.h `new_cos_sin`)
        rotated = (payload.metadata or {}).get("rotated_store")
        rotated_args = None
        if rotated is not None:
            if new_kv is not None or meta.attn_score is not None:
                raise RuntimeError("a rotated store rides in an unscored launch that carries no other store")
            new_kv, rotated_args = rotated["new_kv"], rotated["args"]
        with profiler.record(f"decode_attention_stage1_{kind}"):
            if meta.attn_score is not None:
                flash_decode_stage1_with_score(q, payload.k_cache, payload.v_cache, meta.active_slots, meta.req_indices,
                                               meta.context_lens, max_len_in_batch, mid_o, mid_o_logexpsum,
                                               meta.attn_score, block_seq, new_kv=new_kv, direct_out=direct_out,
                                               score_overwrite=score_overwrite, slot_page_size=slot_page_size)
            else:
                flash_decode_stage1(q, payload.k_cache, payload.v_cache, meta.active_slots, meta.req_indices,
                                    meta.context_lens, max_len_in_batch, mid_o, mid_o_logexpsum, block_seq,
                                    gqa_block_n, gqa_num_warps, new_kv=new_kv, direct_out=direct_out,
                                    slot_page_size=slot_page_size, rotated_store=rotated_args)
        if not direct:
            with profiler.record(f"decode_attention_stage2_{kind}"):
                flash_decode_stage2(mid_o, mid_o_logexpsum, meta.context_lens, o, block_seq)
        return o


class Attention(torch.nn.Module):
    def __init__(self, num_heads, head_dim, scale, num_kv_heads, *, decode_launch_op=None):
        super().__init__()
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads = num_heads, head_dim, scale, num_kv_heads
        self.attention_backend = HipAttentionBackend()
        self.decode_launch_op = decode_launch_op

    def forward(self, q: torch.Tensor, k: torch.Tensor | None = None, v: torch.Tensor | None = None):
        context = get_context()
        cache_manager = context.cache_manager
        sparse_controller = context.sparse_controller
        layer_idx = context.now_layer_idx
        if context.is_prefill:
            return self._forward_prefill(context, cache_manager, sparse_controller, layer_idx, q, k, v)
        return self._forward_decode(context, cache_manager, sparse_controller, layer_idx, q)

    def _forward_decode(self, context, cache_manager, sparse_controller, layer_idx, q):
        """layers/attention.py:162-250 hook for hook.  This step's K/V rows were handed to
        `cache_manager.save_rope_kv_if_needed` by the caller (models/qwen2.py:126-131); a manager that lets the store ride
        in the stage-1 launch kept them back and hands them over through `take_deferred_decode_store`."""
        temp_slots = None
        try:
            batch_size = q.shape[0]
            selection = sparse_controller.get_decode_selection(layer_idx, q)
            decode_view = cache_manager.build_decode_compute_view(
                layer_idx, q, selection, num_heads=self.num_heads, num_kv_heads=self.num_kv_heads)
            if not isinstance(decode_view.payload, ExplicitKVPayload):
                raise TypeError("Attention decode requires ExplicitKVPayload, got "
                                f"{type(decode_view.payload).__name__}.")
            decode_meta = decode_view.meta
            temp_slots = decode_meta.temp_slots

            max_context_len = decode_meta.max_context_len
            static_cap = getattr(cache_manager, "_decode_static_max_context_len", None)
            if static_cap is not None:
                max_context_len = max(int(max_context_len) if max_context_len is not None else 0, int(static_cap))
            if max_context_len is None:
                raise RuntimeError(f"static decode requires max_context_len, got None at layer={layer_idx}")
            max_len_in_batch = int(max_context_len)
            if decode_meta.active_slots.dim() == 2:
                slot_table_len = int(decode_meta.active_slots.shape[1])
                if (os.environ.get("SVLLM_DEBUG_DECODE_BOUNDS", "0") == "1"
                        and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())):
                    actual_max_len = int(decode_meta.context_lens.max().item()) if decode_meta.context_lens.numel() > 0 else 0
                    if actual_max_len > slot_table_len:
                        raise RuntimeError("decode context length exceeds active slot table width: "
                                           f"layer={layer_idx} context_lens_max={actual_max_len} "
                                           f"slot_table_len={slot_table_len}")
                if max_len_in_batch > slot_table_len:
                    max_len_in_batch = slot_table_len
                if max_len_in_batch <= 0:
                    raise RuntimeError(f"decode requires a positive context length, got {max_len_in_batch} at layer={layer_idx}")
            block_seq = cache_manager.get_decode_block_seq(layer_idx, 256)
            kivi = decode_view.payload.backend == "full_layer_kivi"
            if self.decode_launch_op is None or kivi:
                # KIVI layers keep the manager's full_layer_kivi_decode_block_seq (128-token tiles, >= 2 workgroups per CU)
                gqa_block_n, gqa_num_warps = 16, 2
            else:
                extra = {"batch_size": batch_size} if getattr(self.decode_launch_op, "accepts_batch_size", False) else {}
                block_seq, gqa_block_n, gqa_num_warps = self.decode_launch_op.launch_config(
                    block_seq=block_seq, max_context_len=max_len_in_batch,
                    requires_attention_scores=decode_meta.attn_score is not None, **extra)
            num_seq_blocks = (max_len_in_batch + block_seq - 1) // block_seq
            if kivi:
                num_seq_blocks += 3           # room for the wide KIVI launch's extra partials (raw / ragged pieces of a row)
            mid_o, mid_lse = get_decode_workspace(context, batch_size, self.num_heads, num_seq_blocks, self.head_dim,
                                                  q.device)
            # MI355X extras, passed only when they apply (a backend with the reference's signature is called with the
            # reference's arguments)
            extras = {}
            take = getattr(cache_manager, "take_deferred_decode_store", None)
            new_kv = take(layer_idx) if take is not None else None
            if new_kv is not None:
                if decode_view.payload.backend in ("dense", "full_layer_kivi"):
                    extras["new_kv"] = new_kv
                else:
                    cache_manager.store_deferred_decode_rows(layer_idx, new_kv)
            if getattr(sparse_controller, "decode_scores_overwrite", False):
                extras["score_overwrite"] = True
            o = self.attention_backend.run_decode(
                q, decode_view, mid_o=mid_o, mid_o_logexpsum=mid_lse, max_len_in_batch=max_len_in_batch,
                block_seq=block_seq, num_heads=self.num_heads, num_kv_heads=self.num_kv_heads,
                gqa_block_n=gqa_block_n, gqa_num_warps=gqa_num_warps, **extras)
            cache_manager.record_decode_query(layer_idx, q)
            sparse_controller.on_layer_attention_end(layer_idx)
            cache_manager.on_layer_attention_end(layer_idx)
            return o
        finally:
            if temp_slots is not None and temp_slots.numel() > 0:
                cache_manager.release_layer_temp_slots(layer_idx, temp_slots)

    def _forward_prefill(self, context, cache_manager, sparse_controller, layer_idx, q, k, v):
        """layers/attention.py:88-161 hook for hook (the chunk's K/V were stored by the caller, models/qwen2.py:126-131)."""
        temp_slots = None
        try:
            selection = sparse_controller.get_prefill_selection(layer_idx)
            cache_manager.before_prefill_layer_attention(layer_idx, selection)
            prefill_view = cache_manager.build_prefill_compute_view(layer_idx, k, v, selection)
            if not isinstance(prefill_view.payload, ExplicitKVPayload):
                raise TypeError("Attention prefill requires ExplicitKVPayload, got "
                                f"{type(prefill_view.payload).__name__}.")
            prefill_meta = prefill_view.meta
            temp_slots = prefill_meta.temp_slots

            if context.cu_seqlens_q is None or context.cu_seqlens_q.numel() <= 1:
                return torch.empty_like(q)

            # derived once per chunk, not once per layer (two small launches per layer otherwise): keyed on the storage
            # and version of cu_seqlens_q
            cu = context.cu_seqlens_q
            key = (cu.data_ptr(), cu._version, int(cu.numel()))
            cached = getattr(context, "_prefill_derived", None)
            if cached is None or cached[0] != key:
                b_start_loc = cu[:-1]
                chunk_lens = cu[1:] - cu[:-1]
                if cu.dtype != torch.int32:
                    b_start_loc, chunk_lens = b_start_loc.to(torch.int32), chunk_lens.to(torch.int32)
                cached = context._prefill_derived = (key, b_start_loc, chunk_lens)
            _, b_start_loc, chunk_lens = cached
            max_context_len = prefill_meta.max_context_len
            if max_context_len is not None:
                max_input_len = int(max_context_len)
            elif q.is_cuda and torch.cuda.is_current_stream_capturing():
                max_input_len = int(prefill_meta.active_slots.shape[1])
            else:
                max_input_len = int(prefill_meta.context_lens.max().item())

            fake_output = self.attention_backend.maybe_run_fake_prefill(
                q, prefill_view, chunk_lens=chunk_lens, max_input_len=max_input_len)
            if fake_output is not None:
                o = fake_output
            else:
                o = self.attention_backend.run_prefill(q, prefill_view, b_start_loc=b_start_loc, chunk_lens=chunk_lens,
                                                       max_input_len=max_input_len)
            cache_manager.collect_prefill_attention_score(layer_idx, q, prefill_view, b_start_loc=b_start_loc,
                                                          chunk_lens=chunk_lens)
            cache_manager.record_prefill_query(layer_idx, q, prefill_view, b_start_loc=b_start_loc, chunk_lens=chunk_lens)
            sparse_controller.on_layer_attention_end(layer_idx)
            cache_manager.on_layer_attention_end(layer_idx)
            return o
        finally:
            if temp_slots is not None and temp_slots.numel() > 0:
                cache_manager.release_layer_temp_slots(layer_idx, temp_slots)
