"""Recording stand-ins for the per-layer hook contract of `Attention.forward` (prefill branch).

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
