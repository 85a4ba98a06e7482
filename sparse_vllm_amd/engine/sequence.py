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

    def __post_init__(self):
        if self.num_tokens == 0:
            self.num_tokens = self.num_prompt_tokens

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
