"""Which prefill chunks / decode rows fit into the next step: the capacity arithmetic of the reference's
`Scheduler.schedule` (engine/scheduler.py:398-792) over the cache managers' scheduler hooks (`base.py:1242-1397`,
`h2o.py:73-152`), as a pure-host planner.  SURVEY.md 8(f).4, bounded: NO preemption / recompute replay, no prefix cache,
no token post-processing, no server - where the reference would preempt a decode row the planner raises
`PreemptionRequired` and leaves its queues as they were before the call.

One step is either prefill or decode (the reference does not mix them):

* prefill (scheduler.py:443-644) - waiting prompts are bucketed by `(prefill_execution_mode, prefill_batch_compatibility_key)`
  in first-seen order; the first bucket that yields a chunk is the step.  Per candidate: chunk = min(remaining,
  chunk_prefill_size, max_num_batched_tokens - batched, step capacity) for "chunked" / "raw_offload", all-or-nothing for
  "full"; the manager's `min_final_prefill_chunk_size` shortens a chunk that would leave too small a final one; a fresh
  prompt is admitted only if every budget of `prompt_admission_costs` fits `prompt_admission_budgets` ("defer": stays
  queued; otherwise RuntimeError with the reference's text) and is then charged to the budgets and to the logical
  reservation; every scheduled chunk is charged to the step capacity through `prefill_step_reservation_cost`.
* decode (scheduler.py:650-722) - short rows first (only when every row is long are long rows scheduled); per row
  `decode_step_reservation_cost` against min(step budget, `decode_step_free_slots_for`).

`memory_oracle` is anything with the reference's `MemoryOracle` protocol (engine/runtime_state.py:21-54): a cache
manager of this build qualifies (`CacheManager` carries the hook defaults).
"""

from __future__ import annotations

from collections import deque

PREFILL_EXECUTION_CHUNKED = "chunked"
PREFILL_EXECUTION_FULL = "full"
PREFILL_EXECUTION_RAW_OFFLOAD = "raw_offload"
SUPPORTED_PREFILL_EXECUTION_MODES = frozenset({PREFILL_EXECUTION_CHUNKED, PREFILL_EXECUTION_FULL, PREFILL_EXECUTION_RAW_OFFLOAD})

_STREAMING_NAMES = ("streamingllm", "attention-sink", "attention_sink")


def validate_prefill_execution_mode(mode: str) -> str:
    """engine/prefill.py:22-29."""
    normalized = str(mode)
    if normalized not in SUPPORTED_PREFILL_EXECUTION_MODES:
        supported = ", ".join(sorted(SUPPORTED_PREFILL_EXECUTION_MODES))
        raise ValueError(f"Unsupported prefill execution mode={mode!r}; expected one of {supported}.")
    return normalized


class PreemptionRequired(RuntimeError):
    """The decode set does not fit and the reference would preempt `victim` (scheduler.py:692-701, :724-735).  Preemption
    and recompute replay are outside this build; the planner's queues are unchanged."""

    def __init__(self, victim):
        super().__init__(f"decode step does not fit without preempting seq_id={victim.seq_id} (preemption is outside this build)")
        self.victim = victim


class StepPlanner:
    def __init__(self, config, memory_oracle):
        self.config = config
        self.memory_oracle = memory_oracle
        self.max_num_seqs_in_batch = int(config.max_num_seqs_in_batch)
        self.max_num_batched_tokens = int(config.max_num_batched_tokens)
        self.max_decoding_seqs = int(config.max_decoding_seqs)
        self.chunk_prefill_size = int(config.chunk_prefill_size)
        self.num_sink_tokens = int(config.num_sink_tokens)
        self.num_recent_tokens = int(config.num_recent_tokens)
        self.decode_keep_tokens = int(config.decode_keep_tokens)
        self.waiting: deque = deque()
        self.decoding: deque = deque()
        self._defer_noted: set[int] = set()

    # ------------------------------------------------------------------ queues
    def add(self, seq) -> None:
        self.waiting.append(seq)

    def is_finished(self) -> bool:
        return not self.waiting and not self.decoding

    def after_prefill(self, seqs) -> None:
        """Queue effect of the reference's `postprocess(is_prefill=True)` (scheduler.py:812-826) without the sampled
        token: progress advances by the chunk; an unfinished prompt returns to the HEAD of `waiting`, a finished one joins
        `decoding`."""
        for seq in seqs:
            seq.num_prefilled_tokens += int(seq.current_chunk_size)
            if seq.num_prefilled_tokens < seq.num_prompt_tokens:
                self.waiting.appendleft(seq)
            else:
                self.memory_oracle.complete_prefill_execution(seq)
                self.decoding.append(seq)

    def postprocess(self, seqs, token_ids, is_prefill: bool) -> list:
        """`Scheduler.postprocess` (scheduler.py:794-870) for sequences of this build, without EOS / log-probs / replay:
        prefill - progress and queues as `after_prefill`, a prompt that finished its last chunk takes its first sampled
        token; decode - every row takes its token.  A row whose generation budget is used up leaves `decoding`.
        -> the finished sequences (the engine frees their cache rows, llm_engine.py:1282-1300)."""
        finished = []
        if is_prefill:
            self.after_prefill(seqs)
            for seq, tok in zip(seqs, token_ids):
                if seq.num_prefilled_tokens >= seq.num_prompt_tokens:
                    seq.append_token(tok)
                    if seq.is_finished:
                        self.decoding.remove(seq)
                        finished.append(seq)
            return finished
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            if seq.is_finished:
                if seq in self.decoding:
                    self.decoding.remove(seq)
                finished.append(seq)
        return finished

    @staticmethod
    def _take(queue: deque, idx: int):
        queue.rotate(-idx)
        item = queue.popleft()
        queue.rotate(idx)
        return item

    # ------------------------------------------------------------------ classification
    def _long_text_threshold(self) -> int:
        """scheduler.py:68-74."""
        if self.config.vllm_sparse_method in _STREAMING_NAMES:
            return self.num_sink_tokens + self.num_recent_tokens
        return self.num_sink_tokens + self.decode_keep_tokens + self.num_recent_tokens

    def _is_long_decode(self, seq) -> bool:
        if not self.config.vllm_sparse_method:
            return False
        return int(seq.num_tokens) > int(self._long_text_threshold())

    def _bucket_of(self, seq):
        mode = validate_prefill_execution_mode(self.memory_oracle.prefill_execution_mode(seq))
        key = self.memory_oracle.prefill_batch_compatibility_key(seq)
        try:
            hash(key)
        except TypeError as exc:
            raise TypeError(f"prefill_batch_compatibility_key must be hashable: seq_id={seq.seq_id} key={key!r}.") from exc
        return mode, key

    def _buckets_in_order(self):
        seen = []
        for seq in self.waiting:
            bucket = self._bucket_of(seq)
            if bucket not in seen:
                seen.append(bucket)
        return seen

    # ------------------------------------------------------------------ chunk arithmetic
    def _chunk_tokens(self, mode: str, remaining: int, batched: int, capacity: int) -> int:
        """scheduler.py:244-275."""
        room = self.max_num_batched_tokens - batched
        if mode == PREFILL_EXECUTION_FULL:
            return int(remaining) if remaining <= min(room, capacity) else 0
        return min(remaining, self.chunk_prefill_size, room, capacity)

    def _keep_final_chunk_large_enough(self, seq, remaining: int, proposed: int) -> int:
        """scheduler.py:277-293."""
        min_final = int(self.memory_oracle.min_final_prefill_chunk_size(seq))
        if min_final < 0:
            raise ValueError(f"min_final_prefill_chunk_size must be non-negative, got {min_final} for seq_id={seq.seq_id}.")
        left = int(remaining) - int(proposed)
        if min_final == 0 or left <= 0 or left >= min_final:
            return int(proposed)
        return max(0, int(remaining) - min_final)

    # ------------------------------------------------------------------ the step
    def schedule(self):
        """-> (seqs, is_prefill, preempted) like `Scheduler.schedule`; `preempted` is always [].  A prefill step sets
        `seq.current_chunk_size` on every scheduled sequence."""
        snapshot = getattr(self.memory_oracle, "scheduler_capacity_snapshot", None)
        if callable(snapshot):
            with snapshot():
                return self._plan()
        return self._plan()

    def _plan(self):
        oracle = self.memory_oracle
        free_slots = oracle.num_free_slots
        if self.waiting:
            reserved = int(oracle.reserved_prefill_slots(self.waiting, self.chunk_prefill_size))
            logical_free = max(0, int(oracle.prompt_admission_free_slots()) - reserved)
            step_capacity = int(oracle.prefill_step_free_slots())
            budgets = dict(oracle.prompt_admission_budgets(self.waiting, self.chunk_prefill_size))
            margin = oracle.prefill_batched_tokens_margin()
        else:
            reserved, logical_free, step_capacity, budgets, margin = 0, 0, int(free_slots), {}, 0
        decode_budget = max(0, int(oracle.decode_step_free_slots()))
        notes = {"deferred": None, "atomic": None, "capacity": None}

        chosen, batched_tokens = [], 0
        for mode, key in (self._buckets_in_order() if self.waiting else []):
            if chosen:
                break
            scans = len(self.waiting)
            while scans > 0 and self._bucket_has_room(mode, chosen, step_capacity, batched_tokens, margin):
                idx = next((i for i, s in enumerate(self.waiting) if self._bucket_of(s) == (mode, key)), None)
                if idx is None:
                    break
                seq = self._take(self.waiting, idx)
                scans -= 1
                remaining = oracle.remaining_prefill_tokens(seq)
                own_capacity = int(oracle.prefill_step_free_slots_for(seq))
                if mode != PREFILL_EXECUTION_RAW_OFFLOAD and not bool(oracle.should_schedule_full_prefill(seq)):
                    own_capacity = min(int(step_capacity), own_capacity)
                if remaining <= 0:
                    raise ValueError("a sequence without remaining prefill tokens is in the waiting queue")
                tokens = self._chunk_tokens(mode, remaining, batched_tokens, own_capacity)
                tokens = self._keep_final_chunk_large_enough(seq, remaining, tokens)
                if tokens <= 0:
                    if own_capacity <= 0 and step_capacity > 0 and notes["capacity"] is None:
                        notes["capacity"] = (seq, int(remaining), int(own_capacity), int(step_capacity))
                    if mode == PREFILL_EXECUTION_FULL:
                        notes["atomic"] = (seq, int(remaining), int(min(self.max_num_batched_tokens - batched_tokens, own_capacity)))
                    self.waiting.append(seq)
                    continue
                if seq.num_prefilled_tokens == 0:
                    costs = oracle.prompt_admission_costs(seq)
                    short = next(((n, int(need), int(budgets.get(n, 0) or 0)) for n, need in costs.items()
                                  if int(budgets.get(n, 0) or 0) < int(need)), None)
                    if short is not None:
                        name, need, free = short
                        if oracle.prompt_admission_failure_action() == "defer":
                            if notes["deferred"] is None:
                                notes["deferred"] = (seq, name, need, free)
                            self._defer_noted.add(seq.seq_id)
                            self.waiting.append(seq)
                            continue
                        raise RuntimeError(
                            "Insufficient KV cache slots to admit prompt. "
                            f"cache_manager={type(oracle).__name__} prompt_len={seq.num_prompt_tokens} "
                            f"failed_budget={name} need={need} free={free} budgets={budgets} "
                            f"free_slots={free_slots} reserved_prefill={reserved} logical_free={logical_free}")
                    self._defer_noted.discard(seq.seq_id)
                    for name, need in costs.items():
                        budgets[name] = int(budgets.get(name, 0) or 0) - int(need)
                    oracle.on_prompt_admitted(seq, costs)
                    if int(getattr(seq, "prefix_cache_hit_len", 0) or 0) > 0:
                        seq.num_prefilled_tokens = int(seq.prefix_cache_hit_len)
                    logical_need = oracle.prompt_logical_reservation_cost(seq)
                    if logical_free < logical_need:
                        raise RuntimeError(
                            "Prompt admission budget mismatch after reservation check. "
                            f"cache_manager={type(oracle).__name__} prompt_len={seq.num_prompt_tokens} "
                            f"logical_need={logical_need} logical_free={logical_free} "
                            f"budgets={budgets} costs={costs} free_slots={free_slots} reserved_prefill={reserved}")
                    logical_free -= int(logical_need)
                seq.current_chunk_size = tokens
                batched_tokens += tokens
                step_capacity = max(0, step_capacity - int(oracle.prefill_step_reservation_cost(seq, tokens)))
                chosen.append(seq)
                if mode == PREFILL_EXECUTION_RAW_OFFLOAD:
                    break
        if chosen:
            return chosen, True, []
        return self._plan_decode(decode_budget, notes, free_slots, reserved)

    def _bucket_has_room(self, mode, chosen, step_capacity, batched_tokens, margin) -> bool:
        """scheduler.py:216-242 (without the replay clause)."""
        if not self.waiting or len(self.decoding) >= self.max_decoding_seqs:
            return False
        if mode == PREFILL_EXECUTION_RAW_OFFLOAD:
            return not chosen and step_capacity > 0
        return (step_capacity > 0 and batched_tokens <= self.max_num_batched_tokens - margin
                and len(chosen) < self.max_num_seqs_in_batch)

    def _plan_decode(self, budget: int, notes, free_slots, reserved):
        oracle = self.memory_oracle
        before = list(self.decoding)                      # restored if the step needs a preemption
        want_long = bool(self.decoding) and not any(not self._is_long_decode(s) for s in self.decoding)
        chosen, blocked = [], None
        scans = len(self.decoding)
        while self.decoding and scans > 0 and len(chosen) < self.max_num_seqs_in_batch:
            idx = next((i for i, s in enumerate(self.decoding) if self._is_long_decode(s) == want_long), None)
            if idx is None:
                break
            seq = self._take(self.decoding, idx)
            scans -= 1
            room = min(int(budget), int(oracle.decode_step_free_slots_for(seq)))
            cost = int(oracle.decode_step_reservation_cost(seq))
            if room >= cost:
                budget -= cost
                chosen.append(seq)
                continue
            if budget > 0 or chosen:
                # this row cannot join, others may (or already did): keep it queued and run the partial batch
                blocked = blocked or seq
                self.decoding.append(seq)
                if budget > 0:
                    continue
                break
            self._restore_decoding(before)
            raise PreemptionRequired(seq)
        if not chosen:
            if blocked is not None:
                self._restore_decoding(before)
                raise PreemptionRequired(blocked)
            self._raise_if_stuck(notes, free_slots, reserved)
            return [], False, []
        self.decoding.extendleft(reversed(chosen))
        if blocked is not None:
            # the row that could not join is retried first in the next step (scheduler.py:782-791)
            self.decoding.remove(blocked)
            self.decoding.appendleft(blocked)
        return chosen, False, []

    def _restore_decoding(self, before) -> None:
        self.decoding.clear()
        self.decoding.extend(before)

    def _raise_if_stuck(self, notes, free_slots, reserved) -> None:
        """Nothing runnable and nothing decoding: the reference's three fail-fast diagnoses (scheduler.py:736-779)."""
        if self.decoding:
            return
        oracle, name_ = self.memory_oracle, type(self.memory_oracle).__name__
        if notes["atomic"] is not None:
            seq, need, free = notes["atomic"]
            raise RuntimeError(
                "Prefill candidate requires an atomic prefill step but cannot fit. "
                f"cache_manager={name_} seq_id={seq.seq_id} prompt_len={seq.num_prompt_tokens} "
                f"remaining_prefill_tokens={need} available_step_tokens={free} "
                f"chunk_prefill_size={self.chunk_prefill_size} max_num_batched_tokens={self.max_num_batched_tokens}. "
                "Increase the raw KV budget / max_num_batched_tokens or reduce short-batch size.")
        if notes["capacity"] is not None:
            seq, need, seq_free, global_free = notes["capacity"]
            raise RuntimeError(
                "No prefill candidate can use the remaining cache capacity. "
                f"cache_manager={name_} seq_id={seq.seq_id} prompt_len={seq.num_prompt_tokens} "
                f"remaining_prefill_tokens={need} candidate_step_free={seq_free} global_step_free={global_free} "
                f"free_slots={free_slots} reserved_prefill={reserved} waiting={len(self.waiting)} decoding={len(self.decoding)}. "
                "This usually means the only remaining capacity belongs to another sequence's partial page; "
                "reduce concurrency or free a decode sequence first.")
        if notes["deferred"] is not None:
            seq, name, need, free = notes["deferred"]
            raise RuntimeError(
                "All prompt admissions were deferred and no runnable work remains. "
                f"cache_manager={name_} seq_id={seq.seq_id} prompt_len={seq.num_prompt_tokens} "
                f"failed_budget={name} need={need} free={free} free_slots={free_slots} reserved_prefill={reserved} "
                f"waiting={len(self.waiting)} decoding={len(self.decoding)}. "
                "Reduce batch size/max_num_seqs_in_batch/max_num_batched_tokens, or shorten the prompt / generation budget.")
        del oracle
