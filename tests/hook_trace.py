"""Recording stand-ins for the per-layer hook contract of `Attention.forward` (prefill branch; decode branch: Part C).

Used twice with the SAME code: by tests/golden/gen_fixtures.py (`attention_hooks` group) against the reference's
`sparsevllm.layers.attention.Attention` with the reference's dataclasses, and by tests/test_attention_hooks.py against
`sparse_vllm_amd.layers.attention.Attention` with this build's dataclasses.  The trace (hook names, the shapes and values
of their arguments, the result or the exception) must be identical.  Contains no reference code: `types` is the namespace
that provides SparseSelection / AttentionViewMeta / ExplicitKVPayload / PrefillComputeView.
"""

from __future__ import annotations

import os

import torch


class MlaLatentPayload:
    """A payload that is not an ExplicitKVPayload (the class name appears in the TypeError text)."""


def _shape(x):
    return None if x is None else [int(v) for v in x.shape]


def _ints(x):
    return None if x is None else [int(v) for v in x.tolist()]


class RecordingController:
    def __init__(self, trace, types, *, context_lens, req_indices, max_context_len, attn_score=None):
        self.trace, self.types = trace, types
        self.context_lens, self.req_indices, self.max_context_len, self.attn_score = (
            context_lens, req_indices, max_context_len, attn_score)

    def get_prefill_selection(self, layer_idx):
        self.trace.append(["sparse_controller.get_prefill_selection", {"layer": int(layer_idx)}])
        return self.types.SparseSelection(kind="full", req_indices=self.req_indices, context_lens=self.context_lens,
                                          max_context_len=self.max_context_len, attn_score=self.attn_score)

    def on_layer_attention_end(self, layer_idx):
        self.trace.append(["sparse_controller.on_layer_attention_end", {"layer": int(layer_idx)}])


class RecordingManager:
    def __init__(self, trace, types, *, slots, explicit=True, temp_slots=None, fail_in_collect=False):
        self.trace, self.types, self.slots = trace, types, slots
        self.explicit, self.temp_slots, self.fail_in_collect = explicit, temp_slots, fail_in_collect
        self.k_cache = torch.zeros((8, 1, 4))
        self.view = None

    def before_prefill_layer_attention(self, layer_idx, selection):
        self.trace.append(["cache_manager.before_prefill_layer_attention",
                           {"layer": int(layer_idx), "kind": selection.kind, "max_context_len": selection.max_context_len}])

    def build_prefill_compute_view(self, layer_idx, k, v, selection):
        self.trace.append(["cache_manager.build_prefill_compute_view",
                           {"layer": int(layer_idx), "k": _shape(k), "v": _shape(v), "kind": selection.kind}])
        t = self.types
        meta = t.AttentionViewMeta(active_slots=self.slots, req_indices=selection.req_indices,
                                   context_lens=selection.context_lens, max_context_len=selection.max_context_len,
                                   attn_score=selection.attn_score, temp_slots=self.temp_slots)
        payload = t.ExplicitKVPayload(k_cache=self.k_cache, v_cache=self.k_cache) if self.explicit else MlaLatentPayload()
        self.view = t.PrefillComputeView(meta=meta, payload=payload)
        return self.view

    def _after(self, name, layer_idx, q, view, b_start_loc, chunk_lens):
        self.trace.append([name, {"layer": int(layer_idx), "q": _shape(q), "same_view": view is self.view,
                                  "b_start_loc": _ints(b_start_loc), "chunk_lens": _ints(chunk_lens)}])

    def collect_prefill_attention_score(self, layer_idx, q, view, *, b_start_loc, chunk_lens):
        self._after("cache_manager.collect_prefill_attention_score", layer_idx, q, view, b_start_loc, chunk_lens)
        if self.fail_in_collect:
            raise RuntimeError("collect failed on purpose")

    def record_prefill_query(self, layer_idx, q, view, *, b_start_loc, chunk_lens):
        self._after("cache_manager.record_prefill_query", layer_idx, q, view, b_start_loc, chunk_lens)

    def on_layer_attention_end(self, layer_idx):
        self.trace.append(["cache_manager.on_layer_attention_end", {"layer": int(layer_idx)}])

    def release_layer_temp_slots(self, layer_idx, temp_slots):
        self.trace.append(["cache_manager.release_layer_temp_slots", {"layer": int(layer_idx), "temp_slots": _ints(temp_slots)}])


CASES = [
    # name, fake-attention env, manager kwargs, controller kwargs, cu_seqlens_q
    dict(name="plain", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0, 5, 8], context_lens=[9, 3], max_context_len=9),
    dict(name="temp_slots_released", env={"SPARSEVLLM_FAKE_PREFILL_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1",
                                          "SPARSEVLLM_FAKE_ATTENTION_MODE": "copy"},
         cu=[0, 4], context_lens=[4], max_context_len=4, temp_slots=[7, 2, 5]),
    dict(name="empty_temp_slots_not_released", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0, 4], context_lens=[6], max_context_len=6, temp_slots=[]),
    dict(name="max_len_from_context_lens", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0, 2, 8], context_lens=[11, 6], max_context_len=None),
    dict(name="score_buffer_zeroed_by_fake", env={"SPARSEVLLM_FAKE_ATTENTION": "true", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "yes"},
         cu=[0, 3], context_lens=[3], max_context_len=3, attn_score=True),
    dict(name="not_explicit_payload", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0, 3], context_lens=[3], max_context_len=3, explicit=False, temp_slots=[1]),
    dict(name="no_queries", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=None, context_lens=[3], max_context_len=3, temp_slots=[4]),
    dict(name="single_boundary_cu", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0], context_lens=[3], max_context_len=3),
    dict(name="collect_raises", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"},
         cu=[0, 3], context_lens=[3], max_context_len=3, temp_slots=[9], fail_in_collect=True),
    dict(name="fake_not_allowed", env={"SPARSEVLLM_FAKE_ATTENTION": "1"},
         cu=[0, 3], context_lens=[3], max_context_len=3, temp_slots=[3]),
    dict(name="bad_fake_mode", env={"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1",
                                    "SPARSEVLLM_FAKE_ATTENTION_MODE": "ones"},
         cu=[0, 3], context_lens=[3], max_context_len=3),
]

_ENV_KEYS = ("SPARSEVLLM_FAKE_ATTENTION", "SPARSEVLLM_FAKE_PREFILL_ATTENTION", "SPARSEVLLM_FAKE_DECODE_ATTENTION",
             "SPARSEVLLM_ALLOW_FAKE_ATTENTION", "SPARSEVLLM_FAKE_ATTENTION_MODE")


def run_case(case, *, attention_cls, types, install_context):
    """-> {"trace": [...], "result": ...}.  `install_context(is_prefill, cu, cache_manager, sparse_controller, layer)`
    installs the per-forward context of the implementation under test."""
    saved = {k: os.environ.pop(k, None) for k in _ENV_KEYS}
    os.environ.update(case["env"])
    try:
        trace = []
        Hq, Hkv, D, layer = 4, 2, 4, 3
        cu = None if case["cu"] is None else torch.tensor(case["cu"], dtype=torch.int32)
        n_tok = 0 if cu is None or cu.numel() == 0 else int(cu[-1])
        n_tok = max(n_tok, 1)
        q = (torch.arange(n_tok * Hq * D, dtype=torch.float32).reshape(n_tok, Hq, D) / 8).to(torch.bfloat16)
        k = torch.ones((n_tok, Hkv, D), dtype=torch.bfloat16)
        v = torch.ones((n_tok, Hkv, D), dtype=torch.bfloat16)
        B = len(case["context_lens"])
        score = torch.full((B, 16), 3.0) if case.get("attn_score") else None
        sc = RecordingController(trace, types, context_lens=torch.tensor(case["context_lens"], dtype=torch.int32),
                                 req_indices=torch.arange(B, dtype=torch.int32), max_context_len=case["max_context_len"],
                                 attn_score=score)
        ts = case.get("temp_slots")
        cm = RecordingManager(trace, types, slots=torch.zeros((B, 16), dtype=torch.int32), explicit=case.get("explicit", True),
                              temp_slots=None if ts is None else torch.tensor(ts, dtype=torch.int32),
                              fail_in_collect=case.get("fail_in_collect", False))
        install_context(True, cu, cm, sc, layer)
        attn = attention_cls(Hq, D, D ** -0.5, Hkv)
        backend = attn.attention_backend
        orig = backend.maybe_run_fake_prefill

        def spy(q_, view, *, chunk_lens, max_input_len):
            trace.append(["attention_backend.maybe_run_fake_prefill",
                          {"q": _shape(q_), "same_view": view is cm.view, "chunk_lens": _ints(chunk_lens),
                           "max_input_len": int(max_input_len)}])
            return orig(q_, view, chunk_lens=chunk_lens, max_input_len=max_input_len)

        backend.maybe_run_fake_prefill = spy
        try:
            o = attn(q, k, v)
            result = {"shape": _shape(o), "dtype": str(o.dtype)}
            mode = case["env"].get("SPARSEVLLM_FAKE_ATTENTION_MODE", "zero")
            if cu is not None and cu.numel() > 1:
                result["equals_q" if mode == "copy" else "all_zero"] = bool(
                    torch.equal(o, q) if mode == "copy" else (o == 0).all())
            if score is not None:
                result["score_all_zero"] = bool((score == 0).all())
        except Exception as e:          # exception class and text are part of the contract
            result = {"err": type(e).__name__, "msg": str(e)}
        return {"trace": trace, "result": result}
    finally:
        for k_ in _ENV_KEYS:
            os.environ.pop(k_, None)
            if saved[k_] is not None:
                os.environ[k_] = saved[k_]


# ---------------------------------------------------------------------------------------------------------------------
# Part B: a REAL H2O manager driven through Attention.forward with fake attention; the scoring launch replaced by a
# deterministic stand-in (the reference's own unit tests patch `_run_prefill_score` the same way,
# tests/test_h2o_cache_manager.py:418-448).
# ---------------------------------------------------------------------------------------------------------------------
H2O_FLOW = dict(layers=2, heads=4, kv_heads=2, dim=4, window=4, chunks=[[6, 5], [4, 7]])


def fake_prefill_score_fn(calls):
    """-> a `_run_prefill_score` stand-in that records its arguments and writes exactly representable scores."""
    def fake(q, k_cache, attn_score, meta, b_start_loc, prompt_cache_lens, max_query_len, score_starts, score_ends, **kwargs):
        n = len(calls)
        kw = {k_: (None if v_ is None else (int(v_) if not torch.is_tensor(v_) else _ints(v_))) for k_, v_ in kwargs.items()
              if k_ in ("candidate_start", "num_recent_tokens", "batch_indices")}
        calls.append({"q": _shape(q), "k_cache": _shape(k_cache), "score": _shape(attn_score),
                      "context_lens": _ints(meta.context_lens), "req_indices": _ints(meta.req_indices),
                      "active_slots": _shape(meta.active_slots), "b_start_loc": _ints(b_start_loc),
                      "prompt_cache_lens": _ints(prompt_cache_lens), "max_query_len": int(max_query_len),
                      "score_starts": _ints(score_starts), "score_ends": _ints(score_ends), "kwargs": kw})
        ends = _ints(score_ends)
        for b, end in enumerate(ends):
            t = torch.arange(end, dtype=torch.float32)
            attn_score[b, :end] = ((t * 7 + b * 3 + n * 5) % 11) / 16.0
    return fake


# ---------------------------------------------------------------------------------------------------------------------
# Part C: the DECODE branch of `Attention.forward` (layers/attention.py:162-250 of the reference) on recording stand-ins:
# hook order and arguments, the payload TypeError, SVLLM_DEBUG_DECODE_BOUNDS, what the launch provider is asked, what
# `attention_backend.run_decode` receives, temp-slot release on success and on every failure.
# ---------------------------------------------------------------------------------------------------------------------
class RecordingDecodeController:
    def __init__(self, trace, types, *, context_lens, req_indices, max_context_len, attn_score=None):
        self.trace, self.types = trace, types
        self.context_lens, self.req_indices, self.max_context_len, self.attn_score = (
            context_lens, req_indices, max_context_len, attn_score)

    def get_decode_selection(self, layer_idx, q):
        self.trace.append(["sparse_controller.get_decode_selection", {"layer": int(layer_idx), "q": _shape(q)}])
        return self.types.SparseSelection(kind="full", req_indices=self.req_indices, context_lens=self.context_lens,
                                          max_context_len=self.max_context_len, attn_score=self.attn_score)

    def on_layer_attention_end(self, layer_idx):
        self.trace.append(["sparse_controller.on_layer_attention_end", {"layer": int(layer_idx)}])


class RecordingDecodeManager:
    def __init__(self, trace, types, *, slots, explicit=True, temp_slots=None, static_cap=None, block_seq=64,
                 fail_in_record=False):
        self.trace, self.types, self.slots = trace, types, slots
        self.explicit, self.temp_slots, self.block_seq, self.fail_in_record = explicit, temp_slots, block_seq, fail_in_record
        if static_cap is not None:
            self._decode_static_max_context_len = static_cap
        self.k_cache = torch.zeros((8, 1, 4))
        self.view = None

    def build_decode_compute_view(self, layer_idx, q, selection, *, num_heads, num_kv_heads):
        self.trace.append(["cache_manager.build_decode_compute_view",
                           {"layer": int(layer_idx), "q": _shape(q), "kind": selection.kind, "num_heads": int(num_heads),
                            "num_kv_heads": int(num_kv_heads)}])
        t = self.types
        meta = t.AttentionViewMeta(active_slots=self.slots, req_indices=selection.req_indices,
                                   context_lens=selection.context_lens, max_context_len=selection.max_context_len,
                                   attn_score=selection.attn_score, temp_slots=self.temp_slots)
        payload = t.ExplicitKVPayload(k_cache=self.k_cache, v_cache=self.k_cache) if self.explicit else MlaLatentPayload()
        self.view = t.DecodeComputeView(meta=meta, payload=payload)
        return self.view

    def get_decode_block_seq(self, layer_idx, default):
        self.trace.append(["cache_manager.get_decode_block_seq", {"layer": int(layer_idx), "default": int(default)}])
        return self.block_seq

    def record_decode_query(self, layer_idx, q):
        self.trace.append(["cache_manager.record_decode_query", {"layer": int(layer_idx), "q": _shape(q)}])
        if self.fail_in_record:
            raise RuntimeError("record failed on purpose")

    def on_layer_attention_end(self, layer_idx):
        self.trace.append(["cache_manager.on_layer_attention_end", {"layer": int(layer_idx)}])

    def release_layer_temp_slots(self, layer_idx, temp_slots):
        self.trace.append(["cache_manager.release_layer_temp_slots", {"layer": int(layer_idx), "temp_slots": _ints(temp_slots)}])


class RecordingLaunchOp:
    """Stands where a PreparedDecodeAttentionLaunchOp stands; has the reference's three-keyword `launch_config` only."""

    def __init__(self, trace, answer):
        self.trace, self.answer = trace, answer

    def launch_config(self, *, block_seq, max_context_len, requires_attention_scores):
        self.trace.append(["decode_launch_op.launch_config",
                           {"block_seq": int(block_seq), "max_context_len": int(max_context_len),
                            "requires_attention_scores": bool(requires_attention_scores)}])
        return tuple(self.answer)


_FAKE = {"SPARSEVLLM_FAKE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"}
_FAKE_DECODE = {"SPARSEVLLM_FAKE_DECODE_ATTENTION": "1", "SPARSEVLLM_ALLOW_FAKE_ATTENTION": "1"}

DECODE_CASES = [
    dict(name="plain_no_launch_op", env=_FAKE_DECODE, context_lens=[9, 3], max_context_len=9, width=16),
    dict(name="launch_op_consulted", env=_FAKE, context_lens=[9, 3], max_context_len=9, width=16, launch=[128, 32, 4]),
    dict(name="launch_op_with_scores", env=_FAKE, context_lens=[700, 3], max_context_len=700, width=1024, launch=[96, 16, 4],
         attn_score=True),
    dict(name="static_cap_raises_max_len", env=_FAKE, context_lens=[5, 5], max_context_len=5, width=64, static_cap=40,
         launch=[32, 16, 2]),
    dict(name="static_cap_without_max_len", env=_FAKE, context_lens=[5], max_context_len=None, width=64, static_cap=20),
    dict(name="clamped_to_slot_table_width", env=_FAKE, context_lens=[5, 5], max_context_len=300, width=10, block_seq=4),
    dict(name="no_max_len", env=_FAKE, context_lens=[5], max_context_len=None, width=64, temp_slots=[3, 1]),
    dict(name="zero_width_table", env=_FAKE, context_lens=[5], max_context_len=5, width=0, temp_slots=[6]),
    dict(name="one_dim_slot_table_not_clamped", env=_FAKE, context_lens=[5, 5], max_context_len=100, width=None),
    dict(name="not_explicit_payload", env=_FAKE, context_lens=[3], max_context_len=3, width=8, explicit=False, temp_slots=[1]),
    dict(name="temp_slots_released", env=dict(_FAKE, SPARSEVLLM_FAKE_ATTENTION_MODE="copy"), context_lens=[4], max_context_len=4,
         width=8, temp_slots=[7, 2, 5]),
    dict(name="empty_temp_slots_not_released", env=_FAKE, context_lens=[4], max_context_len=4, width=8, temp_slots=[]),
    dict(name="score_buffer_zeroed_by_fake", env=_FAKE, context_lens=[3, 2], max_context_len=3, width=8, attn_score=True),
    dict(name="debug_bounds_context_exceeds_table", env=dict(_FAKE, SVLLM_DEBUG_DECODE_BOUNDS="1"), context_lens=[4, 11],
         max_context_len=11, width=10, temp_slots=[2]),
    dict(name="debug_bounds_ok", env=dict(_FAKE, SVLLM_DEBUG_DECODE_BOUNDS="1"), context_lens=[4, 9], max_context_len=9, width=10),
    dict(name="debug_bounds_off_same_inputs", env=_FAKE, context_lens=[4, 11], max_context_len=11, width=10),
    dict(name="fake_not_allowed", env={"SPARSEVLLM_FAKE_ATTENTION": "1"}, context_lens=[3], max_context_len=3, width=8,
         temp_slots=[3]),
    dict(name="bad_fake_mode", env=dict(_FAKE, SPARSEVLLM_FAKE_ATTENTION_MODE="ones"), context_lens=[3], max_context_len=3, width=8),
    dict(name="record_query_raises", env=_FAKE, context_lens=[3], max_context_len=3, width=8, temp_slots=[9], fail_in_record=True),
]

_DECODE_ENV_KEYS = _ENV_KEYS + ("SVLLM_DEBUG_DECODE_BOUNDS",)


def run_decode_case(case, *, attention_cls, types, install_context):
    """-> {"trace": [...], "result": ...} for the decode branch (see Part C above)."""
    saved = {k: os.environ.pop(k, None) for k in _DECODE_ENV_KEYS}
    os.environ.update(case["env"])
    try:
        trace = []
        Hq, Hkv, D, layer = 4, 2, 4, 5
        B = len(case["context_lens"])
        q = (torch.arange(B * Hq * D, dtype=torch.float32).reshape(B, Hq, D) / 8).to(torch.bfloat16)
        k = torch.ones((B, Hkv, D), dtype=torch.bfloat16)
        v = torch.ones((B, Hkv, D), dtype=torch.bfloat16)
        score = torch.full((B, 16), 3.0) if case.get("attn_score") else None
        sc = RecordingDecodeController(trace, types, context_lens=torch.tensor(case["context_lens"], dtype=torch.int32),
                                       req_indices=torch.arange(B, dtype=torch.int32), max_context_len=case["max_context_len"],
                                       attn_score=score)
        width = case["width"]
        slots = torch.zeros((B * 4,), dtype=torch.int32) if width is None else torch.zeros((B, width), dtype=torch.int32)
        ts = case.get("temp_slots")
        cm = RecordingDecodeManager(trace, types, slots=slots, explicit=case.get("explicit", True),
                                    temp_slots=None if ts is None else torch.tensor(ts, dtype=torch.int32),
                                    static_cap=case.get("static_cap"), block_seq=case.get("block_seq", 64),
                                    fail_in_record=case.get("fail_in_record", False))
        install_context(False, None, cm, sc, layer)
        launch = case.get("launch")
        attn = attention_cls(Hq, D, D ** -0.5, Hkv, decode_launch_op=None if launch is None else RecordingLaunchOp(trace, launch))
        backend = attn.attention_backend
        orig = backend.run_decode

        def spy(q_, view, **kw):
            rec = {"q": _shape(q_), "same_view": view is cm.view, "kwargs": sorted(kw)}
            for name in ("mid_o", "mid_o_logexpsum"):
                rec[name] = _shape(kw[name])
                rec[name + "_dtype"] = str(kw[name].dtype)
            for name in ("max_len_in_batch", "block_seq", "num_heads", "num_kv_heads", "gqa_block_n", "gqa_num_warps"):
                rec[name] = int(kw[name])
            trace.append(["attention_backend.run_decode", rec])
            return orig(q_, view, **kw)

        backend.run_decode = spy
        try:
            o = attn(q, k, v)
            result = {"shape": _shape(o), "dtype": str(o.dtype)}
            mode = case["env"].get("SPARSEVLLM_FAKE_ATTENTION_MODE", "zero")
            result["equals_q" if mode == "copy" else "all_zero"] = bool(torch.equal(o, q) if mode == "copy" else (o == 0).all())
            if score is not None:
                result["score_all_zero"] = bool((score == 0).all())
        except Exception as e:          # exception class and text are part of the contract
            result = {"err": type(e).__name__, "msg": str(e)}
        return {"trace": trace, "result": result}
    finally:
        for k_ in _DECODE_ENV_KEYS:
            os.environ.pop(k_, None)
            if saved[k_] is not None:
                os.environ[k_] = saved[k_]


# ---- the backend's own slot check (layers/attention_backend.py:397-439), called directly on hand-built views
BOUNDS_CASES = [
    dict(name="ok", rows=[0, 2], lens=[3, 5], table=[[1, 2, 3, 0, 0, 0], [0] * 6, [4, 5, 6, 7, 1, 0]], cap=8),
    dict(name="env_off", env_on=False, rows=[0, 9], lens=[3, 50], table=[[1, 2, 3]], cap=2),
    dict(name="row_too_large", rows=[0, 3], lens=[1, 1], table=[[1], [1], [1]], cap=8),
    dict(name="row_negative", rows=[-1, 0], lens=[1, 1], table=[[1], [1]], cap=8),
    dict(name="visible_len_exceeds_width", rows=[0, 1], lens=[2, 5], table=[[1, 2, 3, 4], [1, 2, 3, 4]], cap=8),
    dict(name="slot_too_large", rows=[1, 0], lens=[2, 3], table=[[1, 2, 8, 99], [3, 4, 99, 99]], cap=8),
    dict(name="slot_negative", rows=[0, 1], lens=[4, 1], table=[[1, 2, 3, -1], [3, -7, -7, -7]], cap=8),
    dict(name="bad_slot_beyond_context_ignored", rows=[0, 1], lens=[4, 1], table=[[1, 2, 3, 4], [3, -7, 99, -1]], cap=8),
    dict(name="one_dim_table", rows=[0], lens=[2], table=[1, 2, 3], cap=8),
    dict(name="other_backend_skipped", backend="full_layer_kivi", rows=[0, 7], lens=[3, 50], table=[[1, 2, 3]], cap=2),
    dict(name="not_explicit_payload", explicit=False, rows=[0], lens=[1], table=[[1]], cap=8),
    dict(name="empty_batch", rows=[], lens=[], table=[[1, 2]], cap=8),
]


def run_bounds_case(case, *, backend, types):
    saved = os.environ.pop("SVLLM_DEBUG_DECODE_BOUNDS", None)
    if case.get("env_on", True):
        os.environ["SVLLM_DEBUG_DECODE_BOUNDS"] = "1"
    try:
        t = types
        meta = t.AttentionViewMeta(active_slots=torch.tensor(case["table"], dtype=torch.int32),
                                   req_indices=torch.tensor(case["rows"], dtype=torch.int32),
                                   context_lens=torch.tensor(case["lens"], dtype=torch.int32), max_context_len=None)
        cache = torch.zeros((case["cap"], 1, 4))
        payload = (t.ExplicitKVPayload(k_cache=cache, v_cache=cache, backend=case.get("backend", "dense"))
                   if case.get("explicit", True) else MlaLatentPayload())
        view = t.DecodeComputeView(meta=meta, payload=payload)
        try:
            return {"returned": backend._debug_check_decode_bounds(view)}
        except Exception as e:
            return {"err": type(e).__name__, "msg": str(e)}
    finally:
        os.environ.pop("SVLLM_DEBUG_DECODE_BOUNDS", None)
        if saved is not None:
            os.environ["SVLLM_DEBUG_DECODE_BOUNDS"] = saved
