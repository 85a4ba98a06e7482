"""Decode-attention launch providers: the reference's operator family (operators/decode_attention.py:13-158) with an
MI355X provider registered next to its default.

Same surface: `DecodeAttentionLaunchSpec(num_query_heads, num_kv_heads, head_dim, activation_dtype, page_size=1)` with the
reference's validation, `DecodeAttentionLaunchProvider.launch_config(*, block_seq, max_context_len,
requires_attention_scores) -> (BLOCK_SEQ, block_n, num_warps)`, `DECODE_ATTENTION_LAUNCH_REGISTRY`,
`DefaultGqaDecodeLaunchProvider` ("default_gqa", priority 10, keeps the caller's BLOCK_SEQ with 16 / 2),
`PreparedDecodeAttentionLaunchOp(spec, provider)` and `prepare_decode_attention_launch_op(spec, *, device_index=None)`.
The reference's H100 provider is not carried over (it refuses every device but "NVIDIA H100 80GB HBM3").

The MI355X provider wants two more facts than the reference's call passes — how many sequences the launch serves and how
many KV heads a workgroup holds.  Both are OPTIONAL keyword arguments of its `launch_config`; the prepared op fills
`num_kv_heads` from its spec and `Attention.forward` passes `batch_size` only to an op that says it takes it
(`accepts_batch_size`), so a provider with the reference's three-argument `launch_config` works here unchanged.
"""

from __future__ import annotations

import inspect
from dataclasses import dataclass

import torch

from .. import platforms
from ..platforms.interface import DeviceCaps, PlatformEnum
from .registry import OpRegistry, OpResolver, SupportResult


@dataclass(frozen=True)
class DecodeAttentionLaunchSpec:
    num_query_heads: int
    num_kv_heads: int
    head_dim: int
    activation_dtype: torch.dtype
    page_size: int = 1

    def __post_init__(self) -> None:
        if self.num_query_heads <= 0 or self.num_kv_heads <= 0:
            raise ValueError("Decode attention head counts must be positive.")
        if self.num_query_heads % self.num_kv_heads:
            raise ValueError("Decode query heads must be divisible by KV heads.")
        if self.head_dim <= 0 or self.page_size <= 0:
            raise ValueError("Decode attention dimensions must be positive.")


class DecodeAttentionLaunchProvider:
    name = ""
    priority = 0

    def launch_config(self, *, block_seq: int, max_context_len: int, requires_attention_scores: bool) -> tuple[int, int, int]:
        """-> (BLOCK_SEQ, block_n, num_warps)"""
        raise NotImplementedError


DECODE_ATTENTION_LAUNCH_REGISTRY: OpRegistry = OpRegistry("decode attention launch")


@DECODE_ATTENTION_LAUNCH_REGISTRY.register
class DefaultGqaDecodeLaunchProvider(DecodeAttentionLaunchProvider):
    """The reference's default (operators/decode_attention.py:107-128): the caller's BLOCK_SEQ, BLOCK_N 16, 2 warps."""
    name = "default_gqa"
    priority = 10

    @classmethod
    def supports(cls, spec: DecodeAttentionLaunchSpec, caps: DeviceCaps) -> SupportResult:
        del spec, caps
        return SupportResult.yes()

    def launch_config(self, *, block_seq: int, max_context_len: int, requires_attention_scores: bool) -> tuple[int, int, int]:
        del max_context_len, requires_attention_scores
        return int(block_seq), 16, 2


@DECODE_ATTENTION_LAUNCH_REGISTRY.register
class Mi355xDecodeLaunchProvider(DecodeAttentionLaunchProvider):
    """BLOCK_SEQ for the HIP stage-1 kernel on MI355X.

    The kernel runs one workgroup (= Hkv waves) per (batch lane, BLOCK_SEQ block); up to three workgroups are
    resident per CU (154 VGPRs -> 3 waves/SIMD).  Measured at L=4224 (tools/kbench.py, stage-1 v3):
      B=64: 1056 -> 83.8 us, 528 -> 85.5, 352 -> 86.4, 272 -> 92.7, 192 -> 101.2, 2112 -> 118.8
      B=32:  528 -> 45.7 us, 272 -> 47.6, 176 -> 46.6        B=16: 352 -> 26.8 us, 272 -> 27.2, 528 -> 33.1
    Fewer, longer blocks amortise the per-block prologue / epilogue and the stage-2 partials as long as every CU
    still owns a workgroup.  So: the largest 16-aligned BLOCK_SEQ that still yields >= one block per CU - and, for the
    1- and 2-KV-head ranks of a tensor-parallel model (a workgroup is then one or two waves), >= one wave per SIMD:
      Hq/Hkv = 7/1, B=256: 4224 -> 178.9 us (3.1 TB/s), 2112 -> 112.6, 1056 -> 97.9 (5.7 TB/s); B=64: 1056 -> 51.3, 272 -> 30.1
      Hq/Hkv = 14/2, B=256: 4224 -> 208.2 us, 2112 -> 186.4 (6.0 TB/s)"""
    name = "mi355x_hip_gqa"
    priority = 100
    RESIDENT_WORKGROUPS = 256       # one workgroup per CU
    RESIDENT_WAVES = 1024           # one wave per SIMD
    MIN_BLOCK_SEQ = 32              # one tile (round 4; 64 before: B=1 H2O 0.378 -> 0.368 ms, StreamingLLM B=1 0.314 -> 0.277 ms,
                                    # a 2184-token DeltaKV view 10.6 -> 9.8 us per stage 1 + merge)

    @classmethod
    def supports(cls, spec: DecodeAttentionLaunchSpec, caps: DeviceCaps) -> SupportResult:
        if caps.platform != PlatformEnum.ROCM:
            return SupportResult.no(f"requires ROCm, got {caps.platform.name}")
        if not str(caps.arch).startswith("gfx950"):
            return SupportResult.no(f"requires gfx950 (MI355X), got arch {caps.arch!r} on {caps.device_name}")
        if spec.activation_dtype != torch.bfloat16:
            return SupportResult.no(f"requires BF16 query/KV tensors, got {spec.activation_dtype}")
        if spec.head_dim not in (64, 128):
            return SupportResult.no(f"requires head_dim 64 or 128, got {spec.head_dim}")
        if not 1 <= spec.num_query_heads // spec.num_kv_heads <= 8:
            return SupportResult.no(f"requires a GQA group of 1..8 query heads per KV head, got {spec.num_query_heads}/{spec.num_kv_heads}")
        if not 1 <= spec.num_kv_heads <= 8:
            return SupportResult.no(f"requires 1..8 KV heads per rank, got {spec.num_kv_heads}")
        if spec.page_size not in (1, 16):
            return SupportResult.no(f"requires token slots or 16-token pages, got page_size={spec.page_size}")
        return SupportResult.yes()

    def launch_config(self, *, block_seq: int, max_context_len: int, requires_attention_scores: bool,
                      batch_size: int | None = None, num_kv_heads: int | None = None) -> tuple[int, int, int]:
        del block_seq, requires_attention_scores
        b = max(1, int(batch_size or 1))
        length = max(1, int(max_context_len))
        hkv = max(1, int(num_kv_heads or 4))                             # waves per workgroup
        groups = max(self.RESIDENT_WORKGROUPS, -(-self.RESIDENT_WAVES // hkv))
        nblk = max(1, -(-groups // b))                                   # blocks per sequence wanted
        bs = -(-length // nblk)                                          # tokens per block
        bs = max(self.MIN_BLOCK_SEQ, -(-bs // 16) * 16)
        if bs < 256:
            # short blocks: whole 32-token tiles (B=4, L=4672, stage 1 + stage 2 in a graph: 80 -> 16.4 us, 96 -> 15.7,
            # 128 -> 15.7, 160 -> 16.8; B=8: 160 -> 20.8, 128 -> 25.2)
            bs = -(-bs // 32) * 32
        return bs, 16, 4


class PreparedDecodeAttentionLaunchOp:
    def __init__(self, spec: DecodeAttentionLaunchSpec, provider: DecodeAttentionLaunchProvider) -> None:
        self.spec = spec
        self.provider = provider
        declared = inspect.signature(provider.launch_config).parameters
        # optional extras a provider may declare beyond the reference's three arguments
        self.accepts_batch_size = "batch_size" in declared
        self._wants_kv_heads = "num_kv_heads" in declared

    @property
    def name(self) -> str:
        return self.provider.name

    def launch_config(self, **kwargs) -> tuple[int, int, int]:
        if self._wants_kv_heads:
            kwargs.setdefault("num_kv_heads", int(self.spec.num_kv_heads))
        return self.provider.launch_config(**kwargs)


def prepare_decode_attention_launch_op(spec: DecodeAttentionLaunchSpec, *,
                                       device_index: int | None = None) -> PreparedDecodeAttentionLaunchOp:
    platform = platforms.current_platform
    if device_index is None:
        device_index = torch.cuda.current_device() if platform.is_cuda_alike() and torch.cuda.is_available() else 0
    caps = platform.get_device_caps(int(device_index))
    provider = OpResolver(DECODE_ATTENTION_LAUNCH_REGISTRY).resolve(spec, caps).provider
    return PreparedDecodeAttentionLaunchOp(spec, provider)
