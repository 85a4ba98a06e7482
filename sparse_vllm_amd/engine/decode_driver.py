"""Minimal own counterpart of the reference's decode step loop for the sparse attention path.

`ModelRunner.run` decode branch (engine/model_runner.py:1418-1445) drives, per step,
    cache_manager.prepare_decode_static -> sparse_controller.prepare_forward
    -> [per layer: save_rope_kv_if_needed, Attention.forward] -> sparse_controller.post_forward
around the model's dense layers.  The dense layers (QKV/MLP GEMMs) are NOT part of this
build; this driver feeds the attention path with per-layer synthetic q/k/v of the model's
shape so the path can be exercised, parity-checked and measured exactly in the order the
engine would run it.
"""

from __future__ import annotations

import os

import contextlib
import gc

import numpy as np
import torch

from ..config import Config
from ..layers.attention import Attention
from ..operators.decode_attention import DecodeAttentionLaunchSpec, prepare_decode_attention_launch_op
from ..utils.context import set_context
from .cache_manager.base import CacheManager
from .sequence import Sequence
from .sparse_controller import SparseController



@contextlib.contextmanager
def capture_without_gc():
    """Around a hipGraph capture: collect cyclic garbage first and keep the collector off until the capture has ended.
    A captured graph that is only reachable from dead reference cycles (an old driver, say) is destroyed whenever the
    collector happens to run; `hipGraphDestroy` inside another stream capture is "operation not permitted when stream is
    capturing", raised from a destructor - the process aborts.  (torch.cuda.graph no longer collects on entry.)"""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()

class SparseDecodeDriver:
    def __init__(self, config: Config, *, use_launch_provider: bool = True):
        self.config = config
        self.device = torch.device(config.device)
        self.cache_manager = CacheManager.create(config)
        self.sparse_controller = SparseController(config, self.cache_manager)
        cm = self.cache_manager
        launch_op = None
        if use_launch_provider:
            # models/minimax_m2.py:108 of the reference builds the op the same way, once per attention module
            launch_op = prepare_decode_attention_launch_op(
                DecodeAttentionLaunchSpec(cm.num_heads, cm.num_kv_heads, cm.head_dim, torch.bfloat16,
                                          page_size=int(getattr(cm, "page_size", 1) or 1)),
                device_index=self.device.index or 0)
        self.attn = Attention(cm.num_heads, cm.head_dim, cm.head_dim ** -0.5, cm.num_kv_heads,
                              decode_launch_op=launch_op)
        self.seqs: list[Sequence] = []
        # lanes of the step's static buffers (>= len(seqs)); the extra lanes are the padded lanes of a hipGraph
        # captured for a larger batch (decode_cuda_graph.py:266-303)
        self.graph_batch_size: int | None = None
        # decode batches are all-long or all-short (the scheduler partitions them, model_runner.py:1135-1150); synthetic
        # drivers leave the flag off, `run()` sets it per step
        self.is_long_text = False

    # ------------------------------------------------------------------ one decode step
    def _forward_layers(self, q, k, v, outputs):
        cm, sc = self.cache_manager, self.sparse_controller
        ctx = set_context(False, cache_manager=cm, sparse_controller=sc, is_long_text=bool(self.is_long_text))
        sc.prepare_forward(self.seqs, False)
        for layer_idx in range(cm.num_layers):
            ctx.now_layer_idx = layer_idx
            save_raw = getattr(cm, "save_raw_kv_if_needed", None)
            if save_raw is not None:          # DeltaKV sparse layers keep the pre-RoPE key (models/qwen2.py attention)
                save_raw(layer_idx, k[layer_idx], v[layer_idx])
            # models/qwen2.py:126-131: hand this step's post-RoPE K/V rows to the manager, then call the attention layer.
            # (The manager stores them with a launch of its own, or keeps them back for the layer's stage-1 launch where
            # the store may ride in it: `fused_decode_store_slots`.)
            cm.save_rope_kv_if_needed(layer_idx, k[layer_idx], v[layer_idx])
            o = self.attn(q[layer_idx], k[layer_idx], v[layer_idx])
            on_layer_end = getattr(sc, "on_layer_end", None)
            if on_layer_end is not None:
                on_layer_end(layer_idx, ctx)
            if outputs is not None:
                outputs[layer_idx].copy_(o)
        cm.flush_deferred_decode_store()
        join = getattr(sc, "join_side_streams", None)
        if join is not None:
            join()

    def enable_decode_graph(self):
        """hipGraph replay of the per-step layer loop (the reference's DecodeCudaGraphRunner,
        engine/decode_cuda_graph.py:403-566): lengths/slots are read from device buffers with
        stable addresses, the context capacity is pinned to the method's decode peak
        (h2o.py:241-254), so one captured graph serves every step between bursts and after."""
        cm = self.cache_manager
        cap_fn = getattr(cm, "decode_cuda_graph_context_capacity", None)
        cap = cap_fn(self.seqs)[0] if cap_fn is not None else int(cm.max_model_len)
        cm._decode_static_max_context_len = int(cap)
        self.config.decode_cuda_graph = True
        self._graph = None
        self._graph_key = None
        self._graph_steps_seen = 0
        self.graph_stats = {"eager": 0, "captured": 0, "replayed": 0}     # steps by how they ran (measurement tools read it)

    @torch.no_grad()
    def step(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, outputs: torch.Tensor | None = None, *,
             after_layers=None, append: bool = True):
        """`after_layers()` (measurement hook) runs between the layer loop and post_forward.  `append=False`: the rows'
        sampled tokens are the scheduler's to append (`StepPlanner.postprocess`)."""
        cm, sc = self.cache_manager, self.sparse_controller
        seqs = self.seqs
        cm.drop_deferred_decode_store()          # (only a step that raised leaves rows behind)
        # a manager with device-resident bookkeeping (H2O) hands its allocation and its predicated burst over as launches
        # of the step: with the layer loop they are ONE hipGraph and the step uploads nothing
        dev = hasattr(cm, "device_step_begin")
        kw = {"defer_device_launch": True} if dev else {}
        if self.graph_batch_size is not None:
            cm.prepare_decode_static(seqs, graph_batch_size=int(self.graph_batch_size), **kw)
        else:
            cm.prepare_decode_static(seqs, **kw)
        dev_active = dev and cm._device_step is not None

        def body():
            if dev_active:
                cm.device_step_begin()
            self._forward_layers(q, k, v, outputs)
            if dev_active:
                cm.device_step_burst()

        if not getattr(self.config, "decode_cuda_graph", False):
            body()
        else:
            key = (q.data_ptr(), k.data_ptr(), v.data_ptr(), None if outputs is None else outputs.data_ptr(),
                   tuple(s.seq_id for s in seqs), int(cm.device_step_generation) if dev_active else None,
                   bool(self.is_long_text))
            if self._graph is None or self._graph_key != key:
                if self._graph_steps_seen == 0 or self._graph_key != key:
                    # first step with these buffers runs eagerly (allocates every scratch buffer)
                    body()
                    self.graph_stats["eager"] += 1
                    self._graph_key = key
                    self._graph_steps_seen = 1
                    self._graph = None
                else:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with capture_without_gc(), torch.cuda.graph(g):
                        body()
                    self._graph = g
                    g.replay()
                    self.graph_stats["captured"] += 1
            else:
                if dev_active:
                    cm.device_step_mark_launched()   # the replayed graph carries the burst launches
                else:
                    # host-driven bookkeeping: post_forward reads the step's score buffers through the controller's layer
                    # states, which a prefill step in between has reset - bind them again (the reference's runner calls
                    # prepare_forward in front of every replay, model_runner.py:1447-1481; same persistent buffers)
                    self._bind_states_for_replay()
                self._graph.replay()
                self.graph_stats["replayed"] += 1
        if after_layers is not None:
            after_layers()
        sc.post_forward(seqs, False)
        cm.on_forward_end(seqs, False)
        if append:
            for s in seqs:
                s.append_token(0)

    def _bind_states_for_replay(self):
        cm, sc = self.cache_manager, self.sparse_controller
        set_context(False, cache_manager=cm, sparse_controller=sc, is_long_text=bool(self.is_long_text))
        sc.prepare_forward(self.seqs, False)

    # ------------------------------------------------------------------ one prefill chunk
    @torch.no_grad()
    def prefill_chunk(self, seqs, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, outputs: torch.Tensor | None = None,
                      *, k_raw: torch.Tensor | None = None, advance: bool = True):
        """The sparse side of one chunked-prefill step (ModelRunner.run prefill branch,
        model_runner.py:1447-1481): allocate the chunk, per layer store its K/V and run `Attention.forward`
        (causal attention of the chunk + the manager's prefill hooks, which collect the method's token scores),
        then the post-forward eviction.  `outputs` [L, tokens, Hq, D] receives the attention outputs.
        q [L, tokens, Hq, D], k/v [L, tokens, Hkv, D]; every seq needs `current_chunk_size`.  `k_raw` [L, tokens, Hkv, D]:
        the pre-RoPE keys a model hands to `save_raw_kv_if_needed` (models/qwen2.py:118-125; default: `k`).  `advance=False`:
        the prompts' progress is the scheduler's to move (`StepPlanner.postprocess`, as in the reference)."""
        cm, sc = self.cache_manager, self.sparse_controller
        cm.drop_deferred_decode_store()
        out = cm._prepare_prefill(seqs)
        cu = out[0] if isinstance(out, tuple) else None
        if cu is None:
            lens = [int(s.current_chunk_size) for s in seqs]
            cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))).astype(np.int32), device=self.device)
        ctx = set_context(True, cu_seqlens_q=cu, cache_manager=cm, sparse_controller=sc)
        ctx.max_chunk_len = max(int(s.current_chunk_size) for s in seqs)
        ctx.seqs = seqs
        sc.prepare_forward(seqs, True)
        save_raw = getattr(cm, "save_raw_kv_if_needed", None)
        # a step without a prompt-side attention view in this build (a DeltaKV continuation chunk) runs the store /
        # compression side only
        run_attention = outputs is not None or bool(getattr(cm, "prefill_attention_view_supported", True))
        for layer_idx in range(cm.num_layers):
            ctx.now_layer_idx = layer_idx
            if save_raw is not None:          # DeltaKV sparse layers keep the pre-RoPE key
                save_raw(layer_idx, (k if k_raw is None else k_raw)[layer_idx], v[layer_idx])
            # models/qwen2.py:126-131: the model stores the chunk's K/V, then calls the attention layer, whose prefill
            # branch runs the manager's hooks (selection, compute view, attention, score collection) in the reference's order
            cm.save_rope_kv_if_needed(layer_idx, k[layer_idx], v[layer_idx])
            if run_attention:
                o = self.attn(q[layer_idx], k[layer_idx], v[layer_idx])
                if outputs is not None:
                    outputs[layer_idx].copy_(o)
        sc.post_forward(seqs, True)
        cm.on_forward_end(seqs, True)
        if advance:
            for s in seqs:
                s.num_prefilled_tokens += int(s.current_chunk_size)
        set_context(False, cache_manager=cm, sparse_controller=sc)

    # ------------------------------------------------------------------ the engine loop over a step planner
    def run(self, planner, prompts, step_inputs, *, on_step=None, max_steps: int = 100000) -> list[dict]:
        """Waiting prompts -> planned prefill chunks -> decode, until every sequence has used its generation budget: the
        reference's loop `schedule -> prepare_step -> forward -> post_forward -> postprocess -> free finished`
        (llm_engine.py:1187-1300, model_runner.py:1447-1481, scheduler.py:398-870) over `StepPlanner`, this build's cache
        manager as its memory oracle and this driver's two step forms.  No model: `step_inputs(is_prefill, seqs, n)` ->
        (q [L, n, Hq, D], k, v [L, n, Hkv, D], outputs or None) stands in for the layers around the attention path
        (n = chunk tokens of the step, or decode rows).  `on_step(record)` sees every executed step before the next one
        is planned.  -> the plan: one record per step."""
        cm = self.cache_manager
        for seq in prompts:
            planner.add(seq)
        plan = []
        while not planner.is_finished():
            if len(plan) >= max_steps:
                raise RuntimeError(f"run() did not finish within {max_steps} steps")
            seqs, is_prefill, _ = planner.schedule()
            if not seqs:
                raise RuntimeError("the planner scheduled nothing although sequences are queued")
            n = sum(int(s.current_chunk_size) for s in seqs) if is_prefill else len(seqs)
            q, k, v, outputs = step_inputs(is_prefill, seqs, n)
            if is_prefill:
                self.prefill_chunk(seqs, q, k, v, outputs=outputs, advance=False)
            else:
                self.seqs = list(seqs)
                self.is_long_text = planner._is_long_decode(seqs[0])
                self.step(q, k, v, outputs=outputs, append=False)
            record = dict(prefill=bool(is_prefill), seqs=list(seqs), q=q, k=k, v=v, outputs=outputs,
                          chunks=[int(s.current_chunk_size) if is_prefill else 1 for s in seqs])
            record["finished"] = finished = planner.postprocess(seqs, [0] * len(seqs), is_prefill)
            if on_step is not None:
                on_step(record)                        # (before the finished rows are released: their state is still there)
            for seq in finished:
                cm.free_seq(seq.seq_id)
            plan.append(record)
        self.seqs = []
        return plan

    def row_len(self) -> np.ndarray:
        cm = self.cache_manager
        if hasattr(cm, "page_size") or isinstance(cm.seq_id_to_row, dict) and np.ndim(cm.row_seq_lens) == 1:
            return np.array([cm.row_seq_lens[cm.seq_id_to_row[s.seq_id]] for s in self.seqs])
        return np.array([cm.row_seq_lens[0][cm.seq_id_to_row[0][s.seq_id]] for s in self.seqs])
