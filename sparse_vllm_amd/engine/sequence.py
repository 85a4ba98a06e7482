"""The slice of the reference `Sequence` (engine/sequence.py:90-360) the cache managers read."""

from __future__ import annotations

from dataclasses import dataclass, field
from itertools import count

_ids = count()


@dataclass
class Sequence:
    num_prompt_tokens: int = 0
    seq_id: int = field(default_factory=lambda: next(_ids))
    num_prefilled_tokens: int = 0
    current_chunk_size: int = 0
    num_tokens: int = 0
    last_token: int = 0
    max_tokens: int = 0                  # generation budget (SamplingParams.max_tokens); 0 = unbounded (synthetic rows)
    num_completion_tokens: int = 0
    prefix_cache_hit_len: int = 0        # no prefix cache in this build: always 0 (the scheduler hooks read it)

    def __post_init__(self):
        if self.num_tokens == 0:
            self.num_tokens = self.num_prompt_tokens

    @property
    def is_finished(self) -> bool:
        """engine/scheduler.py:841-843 without EOS: the generation budget is used up."""
        return self.max_tokens > 0 and self.num_completion_tokens >= self.max_tokens

    @property
    def is_last_chunk_prefill(self) -> bool:
        return self.num_prefilled_tokens + self.current_chunk_size >= self.num_prompt_tokens

    @property
    def decode_input_token(self) -> int:
        return int(self.last_token)

    @property
    def decode_input_position(self) -> int:
        return int(self.num_tokens - 1)

    def append_token(self, token_id: int) -> None:
        self.last_token = int(token_id)
        self.num_tokens += 1
        self.num_completion_tokens += 1
