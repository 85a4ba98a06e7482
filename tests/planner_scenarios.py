"""Scenarios for the step planner: the capacity arithmetic of the reference's `Scheduler.schedule`
(engine/scheduler.py:398-792), run twice with the SAME code - by tests/golden/gen_fixtures.py (`step_planner`) against the
reference's `Scheduler`, and by tests/test_step_planner.py against `sparse_vllm_amd.engine.step_planner.StepPlanner`.

`ScriptedOracle` is a plain object with the reference's `MemoryOracle` protocol (engine/runtime_state.py:21-54) whose
answers come from tables in the scenario (the recipe of the reference's tests/test_prefill_schedule_policy.py
`FakeMemoryOracle`, own code).  Sequences are plain objects with the attributes the scheduler reads and writes.  After a
prefill step the scenario applies the queue effect of `postprocess` itself (progress += chunk; unfinished -> head of
`waiting`, finished -> tail of `decoding`), after a decode step it appends one token to every scheduled row; both are done
on the `waiting` / `decoding` deques both implementations expose, so no sampled token is involved.  Preemption is outside
the planner: a step where the reference preempts ends the scenario with {"preempt": seq_id} on both sides.
"""

from __future__ import annotations

from collections import deque


class PlanSeq:
    def __init__(self, seq_id, prompt, *, prefilled=0, completion=0, hit=0):
        self.seq_id = int(seq_id)
        self.num_prompt_tokens = int(prompt)
        self.num_prefilled_tokens = int(prefilled)
        self.num_completion_tokens = int(completion)
        self.prefix_cache_hit_len = int(hit)
        self.current_chunk_size = 0
        self.is_recompute_replay = False
        self.is_recompute_decode = False
        self.decode_progress_checkpoint = 0
        self.status = None

    @property
    def num_tokens(self):
        return self.num_prompt_tokens + self.num_completion_tokens

    def start_recompute_replay(self):          # what the reference's preemption does to its victim (scheduler.py:357-359)
        self.is_recompute_replay = True


class ScriptedOracle:
    def __init__(self, free=1_000_000, *, step_free=None, decode_free=None, admission_free=None, budgets=None, margin=0,
                 action="raise", modes=None, keys=None, min_final=0, per_seq_step_free=None, per_seq_decode_free=None,
                 decode_cost=1, full_staging=(), reservation_factor=1, reserved=None, extra_costs=None, logical_cost=None):
        self.free, self.margin, self.action = int(free), int(margin), action
        self.step_free = int(free if step_free is None else step_free)
        self.decode_free = int(free if decode_free is None else decode_free)
        self.admission_free = int(free if admission_free is None else admission_free)
        self.budgets, self.modes, self.keys = budgets, modes or {}, keys or {}
        self.min_final, self.per_seq_step_free = min_final, per_seq_step_free or {}
        self.per_seq_decode_free, self.decode_cost = per_seq_decode_free or {}, decode_cost
        self.full_staging, self.reservation_factor, self.reserved = set(full_staging), reservation_factor, reserved
        self.extra_costs, self.logical_cost = extra_costs or {}, logical_cost
        self.admitted, self.completed = [], []

    @property
    def num_free_slots(self):
        return self.free

    def prefill_batched_tokens_margin(self):
        return self.margin

    def remaining_prefill_tokens(self, seq):
        return int(seq.num_prompt_tokens - max(seq.num_prefilled_tokens, seq.prefix_cache_hit_len))

    def prefill_execution_mode(self, seq):
        return self.modes.get(seq.seq_id, "chunked")

    def prefill_batch_compatibility_key(self, seq):
        return self.keys.get(seq.seq_id)

    def reset_prefill_execution_state(self, seq_id):
        return None

    def complete_prefill_execution(self, seq):
        self.completed.append(seq.seq_id)

    def reserved_prefill_slots(self, waiting, chunk):
        if self.reserved is not None:
            return int(self.reserved)
        return sum(s.num_prompt_tokens - s.num_prefilled_tokens for s in waiting if 0 < s.num_prefilled_tokens < s.num_prompt_tokens)

    def should_schedule_full_prefill(self, seq):
        return seq.seq_id in self.full_staging and seq.num_prefilled_tokens == 0

    def requires_full_prefill_step(self, seq):
        return self.prefill_execution_mode(seq) == "full"

    def requires_long_prefill_offload(self, seq):
        return self.prefill_execution_mode(seq) == "raw_offload"

    def prefill_step_free_slots(self):
        return self.step_free

    def prefill_step_free_slots_for(self, seq):
        return int(self.per_seq_step_free.get(seq.seq_id, self.step_free))

    def min_final_prefill_chunk_size(self, seq):
        return self.min_final[seq.seq_id] if isinstance(self.min_final, dict) else self.min_final

    def prefill_step_reservation_cost(self, seq, tokens):
        return int(tokens) * self.reservation_factor

    def decode_step_free_slots(self):
        return self.decode_free

    def decode_step_free_slots_for(self, seq):
        return int(self.per_seq_decode_free.get(seq.seq_id, self.decode_free))

    def decode_step_reservation_cost(self, seq):
        return self.decode_cost[seq.seq_id] if isinstance(self.decode_cost, dict) else self.decode_cost

    def prompt_admission_free_slots(self):
        return self.admission_free

    def prompt_admission_budgets(self, waiting, chunk):
        if self.budgets is not None:
            return dict(self.budgets)
        return {"slots": max(0, self.admission_free - self.reserved_prefill_slots(waiting, chunk))}

    def prompt_admission_costs(self, seq):
        costs = {"slots": int(seq.num_prompt_tokens - seq.prefix_cache_hit_len)}
        costs.update(self.extra_costs.get(seq.seq_id, {}))
        return costs

    def prompt_logical_reservation_cost(self, seq):
        if self.logical_cost is not None:
            return int(self.logical_cost)
        return int(seq.num_prompt_tokens - seq.prefix_cache_hit_len)

    def prompt_admission_failure_action(self):
        return self.action

    def on_prompt_admitted(self, seq, costs):
        self.admitted.append([seq.seq_id, dict(costs)])

    def refresh_prefix_cache_hit(self, seq):
        return None

    def clear_prefix_cache_hit(self, seq):
        return None

    def free_slot_stats(self):
        return {"free_slots": self.free}

    def debug_live_seq_slots(self):
        return {}


def _cfg(ns, *, method="h2o", chunk=5, max_tokens=10, max_seqs=4, max_decoding=16, sink=1, recent=1, keep=4):
    from types import SimpleNamespace
    return SimpleNamespace(max_num_seqs_in_batch=max_seqs, max_num_batched_tokens=max_tokens, max_decoding_seqs=max_decoding,
                           chunk_prefill_size=chunk, prefill_schedule_policy="all_chunked", eos=-1, eos_token_ids=(),
                           num_sink_tokens=sink, num_recent_tokens=recent, decode_keep_tokens=keep, snapkv_window_size=2,
                           vllm_sparse_method=method)


def _drive(ns, planner, oracle, steps, *, decode_tokens_consume=0):
    """Run `steps` scheduling rounds, applying the queue effects between them.  -> list of per-step records."""
    out = []
    for _ in range(steps):
        try:
            seqs, is_prefill, preempted = planner.schedule()
        except Exception as e:
            victim = getattr(e, "victim", None)
            if victim is not None:                         # the planner's stand-in for the reference's preemption
                out.append({"preempt": int(victim.seq_id)})
            else:
                out.append({"err": type(e).__name__, "msg": str(e)})
            break
        if preempted:
            out.append({"preempt": int(preempted[0].seq_id)})
            break
        rec = {"prefill": bool(is_prefill),
               "seqs": [[s.seq_id, int(s.current_chunk_size) if is_prefill else 1] for s in seqs]}
        if is_prefill:
            for s in seqs:
                s.num_prefilled_tokens += int(s.current_chunk_size)
                if s.num_prefilled_tokens < s.num_prompt_tokens:
                    planner.waiting.appendleft(s)
                else:
                    oracle.complete_prefill_execution(s)
                    planner.decoding.append(s)
        else:
            for s in seqs:
                s.num_completion_tokens += 1
            oracle.decode_free = max(0, oracle.decode_free - decode_tokens_consume * len(seqs))
        rec["waiting"] = [s.seq_id for s in planner.waiting]
        rec["decoding"] = [s.seq_id for s in planner.decoding]
        out.append(rec)
        if not seqs:
            break
    return out


def _scenario(ns, seqs, oracle, *, steps=12, decoding=(), **cfg):
    planner = ns.make(_cfg(ns, **cfg), oracle)
    for s in seqs:
        planner.add(s)
    for s in decoding:
        planner.decoding.append(s)
    trace = _drive(ns, planner, oracle, steps)
    return {"trace": trace, "admitted": oracle.admitted, "completed": oracle.completed}


def run_all(ns):
    S, O = PlanSeq, ScriptedOracle
    out = {}
    # chunked prefill: chunk 5, token budget 10 -> two prompts share a step; unfinished prompts return to the head
    out["chunked_two_prompts"] = _scenario(ns, [S(0, 12), S(1, 7)], O())
    out["chunked_token_budget_splits"] = _scenario(ns, [S(0, 9), S(1, 9), S(2, 3)], O(), chunk=8, max_tokens=12)
    out["max_seqs_in_batch"] = _scenario(ns, [S(i, 2) for i in range(6)], O(), max_seqs=3, max_tokens=100)
    out["margin_leaves_headroom"] = _scenario(ns, [S(0, 4), S(1, 4), S(2, 4)], O(margin=3), chunk=4, max_tokens=10)
    out["step_capacity_limits_chunk"] = _scenario(ns, [S(0, 9), S(1, 9)], O(step_free=7), chunk=5, max_tokens=20, steps=3)
    out["reservation_cost_scales"] = _scenario(ns, [S(0, 4), S(1, 4), S(2, 4)], O(step_free=10, reservation_factor=2), chunk=4,
                                               max_tokens=100, steps=2)
    out["per_seq_capacity"] = _scenario(ns, [S(0, 9), S(1, 9)], O(per_seq_step_free={0: 2}), chunk=5, max_tokens=20, steps=4)
    out["min_final_chunk_shortens"] = _scenario(ns, [S(0, 12)], O(min_final=4), chunk=5, max_tokens=10)
    out["min_final_chunk_per_seq"] = _scenario(ns, [S(0, 11), S(1, 11)], O(min_final={0: 3, 1: 0}), chunk=5, max_tokens=20)
    out["min_final_exact_fit"] = _scenario(ns, [S(0, 10)], O(min_final=5), chunk=5, max_tokens=10)
    # buckets: execution mode + compatibility key, first-seen order; one bucket per step
    out["buckets_by_key"] = _scenario(ns, [S(0, 4), S(1, 4), S(2, 4), S(3, 4)], O(keys={0: "a", 1: "b", 2: "a", 3: "b"}),
                                      chunk=4, max_tokens=100, steps=4)
    out["full_mode_all_or_nothing"] = _scenario(ns, [S(0, 8), S(1, 30), S(2, 6)], O(modes={0: "full", 1: "full", 2: "full"}),
                                                chunk=5, max_tokens=16, steps=6)
    out["full_mode_cannot_fit"] = _scenario(ns, [S(0, 30)], O(modes={0: "full"}), chunk=5, max_tokens=16, steps=2)
    out["raw_offload_runs_alone"] = _scenario(ns, [S(0, 12), S(1, 12)], O(modes={0: "raw_offload", 1: "raw_offload"}),
                                              chunk=5, max_tokens=100, steps=4)
    out["mixed_modes_order"] = _scenario(ns, [S(0, 6), S(1, 6), S(2, 6)], O(modes={0: "chunked", 1: "full", 2: "chunked"}),
                                         chunk=4, max_tokens=100, steps=6)
    out["full_staging_ignores_step_capacity"] = _scenario(ns, [S(0, 6)], O(step_free=3, per_seq_step_free={0: 50}, full_staging=(0,)),
                                                          chunk=8, max_tokens=100, steps=2)
    out["unknown_mode"] = _scenario(ns, [S(0, 6)], O(modes={0: "whole"}), steps=1)
    out["unhashable_key"] = _scenario(ns, [S(0, 6)], O(keys={0: ["x"]}), steps=1)
    # admission
    out["admission_raise"] = _scenario(ns, [S(0, 8), S(1, 8)], O(admission_free=12, budgets={"slots": 12}), chunk=4,
                                       max_tokens=100, steps=2)
    out["admission_defer_then_fit"] = _scenario(ns, [S(0, 8), S(1, 8), S(2, 3)], O(budgets={"slots": 12}, action="defer"),
                                                chunk=8, max_tokens=100, steps=3)
    out["admission_all_deferred"] = _scenario(ns, [S(0, 8)], O(budgets={"slots": 4}, action="defer"), steps=2)
    out["admission_second_budget"] = _scenario(ns, [S(0, 4), S(1, 4)],
                                               O(budgets={"slots": 100, "latent": 5}, action="defer",
                                                 extra_costs={0: {"latent": 3}, 1: {"latent": 3}}), chunk=4, max_tokens=100, steps=3)
    out["admission_logical_mismatch"] = _scenario(ns, [S(0, 8)], O(admission_free=4, budgets={"slots": 100}), steps=1)
    out["reserved_prefill_counts_partials"] = _scenario(ns, [S(0, 20, prefilled=5), S(1, 10)], O(admission_free=22, action="defer"),
                                                        chunk=5, max_tokens=10, steps=3)
    out["prefix_hit_skips_tokens"] = _scenario(ns, [S(0, 12, hit=8)], O(), chunk=5, max_tokens=10, steps=3)
    out["max_decoding_blocks_prefill"] = _scenario(ns, [S(5, 4)], O(), decoding=[S(0, 4, prefilled=4, completion=1),
                                                                                  S(1, 4, prefilled=4, completion=1)],
                                                   max_decoding=2, steps=2)
    out["no_capacity_for_candidate"] = _scenario(ns, [S(0, 6)], O(per_seq_step_free={0: 0}), steps=2)
    # decode
    dec = lambda: [S(i, n, prefilled=n, completion=1) for i, n in enumerate((3, 40, 4, 50, 2))]
    out["decode_short_first"] = _scenario(ns, [], O(), decoding=dec(), steps=2)
    out["decode_all_long"] = _scenario(ns, [], O(), decoding=[S(0, 40, prefilled=40, completion=1), S(1, 50, prefilled=50, completion=2)],
                                       steps=2)
    out["decode_vanilla_never_long"] = _scenario(ns, [], O(), decoding=dec(), method="", steps=1, max_seqs=8)
    out["decode_streamingllm_threshold"] = _scenario(ns, [], O(), decoding=[S(0, 2, prefilled=2, completion=0),
                                                                           S(1, 2, prefilled=2, completion=1)],
                                                     method="streamingllm", steps=1)
    out["decode_max_seqs"] = _scenario(ns, [], O(), decoding=[S(i, 3, prefilled=3, completion=1) for i in range(6)], max_seqs=4, steps=3)
    out["decode_budget_partial_batch"] = _scenario(ns, [], O(decode_free=2), decoding=[S(i, 3, prefilled=3, completion=1) for i in range(4)],
                                                   steps=2)
    out["decode_per_seq_blocked_retry_first"] = _scenario(ns, [], O(per_seq_decode_free={1: 0}),
                                                          decoding=[S(i, 3, prefilled=3, completion=1) for i in range(3)], steps=2)
    out["decode_costs_differ"] = _scenario(ns, [], O(decode_free=5, decode_cost={0: 1, 1: 3, 2: 2, 3: 1}),
                                           decoding=[S(i, 3, prefilled=3, completion=1) for i in range(4)], steps=2)
    out["decode_nothing_fits_preempts"] = _scenario(ns, [], O(decode_free=0), decoding=[S(0, 3, prefilled=3, completion=1),
                                                                                          S(1, 3, prefilled=3, completion=1)], steps=1)
    out["decode_blocked_only_preempts"] = _scenario(ns, [], O(decode_free=4, per_seq_decode_free={0: 0, 1: 0}),
                                                    decoding=[S(0, 3, prefilled=3, completion=1), S(1, 3, prefilled=3, completion=1)], steps=1)
    out["prefill_then_decode_lifecycle"] = _scenario(ns, [S(0, 7), S(1, 3)], O(), chunk=4, max_tokens=8, steps=6)
    out["empty"] = _scenario(ns, [], O(), steps=1)
    return out
