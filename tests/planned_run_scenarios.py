"""Prompt sets for the planned-run tests (scheduler capacity hooks + chunked-prefill driver with a
CONSUMER).  Constants only; read twice:

* tests/golden/gen_fixtures.py `planned_run` drives the REFERENCE's `Scheduler` (engine/scheduler.py:398-870) over the
  reference's hand-built `H2OCacheManager` / `QuestCacheManager` on CPU through the engine's step order
  (schedule -> _prepare_prefill | _prepare_decode -> eviction hooks -> postprocess -> free_seq of finished rows) and
  records, per step, what ran, the queues, every row's physical length and the free capacity -> tests/golden/planned_run.json.
  (No attention runs there: H2O's choice of WHICH tokens to drop needs scores, HOW MANY it drops and when does not - the
  generator feeds arbitrary scores; Quest never drops, so its page tables are recorded too.)
* tests/test_gpu_planned_run.py drives `SparseDecodeDriver.run` = `StepPlanner` over this build's cache managers on the GPU
  with the same prompts and must reproduce that plan step for step, while the chained numpy oracle checks slot tables,
  lengths, scores and attention outputs of every executed step.

Both pools are sized so that admission defers at least once and several prompts exceed the prefill budget / one chunk.
"""

H2O = dict(
    method="h2o", layers=2, rows=8, slots=460, max_model_len=512,
    budget=48, interval=16, prefill_budget=128, window=16, recent_ratio=0.5,
    sink=4, recent=16, keep=28,                      # long-text threshold of the decode partition: 48 tokens
    prompts=[300, 217, 40, 150, 90, 260, 48, 131],
    gens=[20, 12, 30, 8, 18, 10, 25, 5],
    planner=dict(max_num_seqs_in_batch=4, max_num_batched_tokens=128, max_decoding_seqs=4, chunk_prefill_size=64),
)

QUEST = dict(
    method="quest", layers=3, skip_layers=1, rows=8, page=16, pages=44, max_model_len=512, token_budget=96,
    sink=4, recent=16, keep=76,                      # threshold 96 tokens
    prompts=[200, 75, 33, 160, 48, 250, 16],
    gens=[14, 30, 9, 20, 26, 6, 12],
    planner=dict(max_num_seqs_in_batch=3, max_num_batched_tokens=160, max_decoding_seqs=4, chunk_prefill_size=64),
)

# StreamingLLM: nothing depends on scores - which slots survive is arithmetic (sink + recent), so the reference run's slot
# tables and free stacks are recorded too and the GPU run must reproduce them bit for bit from the same initial stack
STREAMINGLLM = dict(
    method="streamingllm", layers=2, rows=6, slots=190, max_model_len=512,
    sink=4, recent=24, keep=0,                       # budget 28 tokens, decode re-eviction at 56
    prompts=[100, 20, 61, 45, 130, 28],
    gens=[40, 50, 12, 60, 9, 33],
    planner=dict(max_num_seqs_in_batch=3, max_num_batched_tokens=64, max_decoding_seqs=4, chunk_prefill_size=32),
)

# SnapKV: the whole prompt stays resident until its final chunk (selection from the last `window` queries' scores), decode
# re-evicts at twice the top budget; counts only in the reference run (which tokens survive needs scores)
SNAPKV = dict(
    method="snapkv", layers=2, rows=6, slots=200, max_model_len=512,
    sink=4, recent=8, keep=20, window=8,             # budget 32 tokens, decode re-eviction at 40
    prompts=[100, 20, 61, 45, 130, 33],
    gens=[40, 50, 12, 60, 9, 21],
    planner=dict(max_num_seqs_in_batch=3, max_num_batched_tokens=64, max_decoding_seqs=4, chunk_prefill_size=32),
)

SCENARIOS = {"h2o": H2O, "quest": QUEST, "streamingllm": STREAMINGLLM, "snapkv": SNAPKV}
