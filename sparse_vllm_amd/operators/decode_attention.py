"""Decode-attention launch configuration providers (mirror of operators/decode_attention.py:13-158:
DecodeAttentionLaunchSpec :13, provider base :29, default 16/2 :107-128).  The MI355X provider
sizes BLOCK_SEQ so that one launch fills 256 CUs: the HIP stage-1 kernel runs one workgroup
(= Hkv waves) per (batch lane, BLOCK_SEQ block), and wants >= ~1 workgroup per CU."""

from __future__ import annotations


from dataclasses import dataclass

from .registry import DeviceCaps, OpRegistry, OpResolver, PlatformEnum, SupportResult

DECODE_LAUNCH_REGISTRY = OpRegistry("decode_attention_launch")


@dataclass(frozen=True)
class DecodeAttentionLaunchSpec:
    num_heads: int
    num_kv_heads: int
    head_dim: int
    sparse_method: str = ""


class DecodeAttentionLaunchProvider:
    name = "base"
    priority = 0

    def supports(self, spec: DecodeAttentionLaunchSpec, caps: DeviceCaps) -> SupportResult:
        raise NotImplementedError

    def launch_config(self, *, block_seq: int, max_context_len: int, requires_attention_scores: bool,
                      batch_size: int | None = None) -> tuple[int, int, int]:
        """-> (BLOCK_SEQ, block_n, num_warps)"""
        raise NotImplementedError


@DECODE_LAUNCH_REGISTRY.register
class DefaultDecodeLaunchProvider(DecodeAttentionLaunchProvider):
    """Reference default: keep the caller's BLOCK_SEQ, BLOCK_N=16, 2 warps."""
    name = "default"
    priority = 0

    def supports(self, spec, caps):
        return SupportResult.yes()

    def launch_config(self, *, block_seq, max_context_len, requires_attention_scores, batch_size=None):
        return int(block_seq), 16, 2


@DECODE_LAUNCH_REGISTRY.register
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
    name = "mi355x_hip"
    priority = 100
    RESIDENT_WORKGROUPS = 256       # one workgroup per CU
    RESIDENT_WAVES = 1024           # one wave per SIMD
    MIN_BLOCK_SEQ = 32              # one tile (round 4; 64 before: B=1 H2O 0.378 -> 0.368 ms, StreamingLLM B=1 0.314 -> 0.277 ms,
                                    # a 2184-token DeltaKV view 10.6 -> 9.8 us per stage 1 + merge)
    wants_kv_heads = True           # PreparedDecodeAttentionLaunchOp passes the spec's num_kv_heads

    def supports(self, spec, caps):
        if caps.platform != PlatformEnum.ROCM:
            return SupportResult.no("not a ROCm device")
        if not caps.arch.startswith("gfx950"):
            return SupportResult.no(f"arch {caps.arch!r} is not gfx950")
        if spec.head_dim not in (64, 128):
            return SupportResult.no(f"head_dim {spec.head_dim} unsupported")
        if spec.num_heads % spec.num_kv_heads or not 1 <= spec.num_heads // spec.num_kv_heads <= 8:
            return SupportResult.no("GQA group size must be 1..8")
        if not 1 <= spec.num_kv_heads <= 8:
            return SupportResult.no("1..8 KV heads per rank")
        return SupportResult.yes()

    def launch_config(self, *, block_seq, max_context_len, requires_attention_scores, batch_size=None, num_kv_heads=None):
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
    def __init__(self, provider: DecodeAttentionLaunchProvider, spec: DecodeAttentionLaunchSpec | None = None):
        self.provider = provider
        self.spec = spec

    def launch_config(self, **kw):
        if self.spec is not None and getattr(self.provider, "wants_kv_heads", False):
            kw.setdefault("num_kv_heads", int(self.spec.num_kv_heads))
        return self.provider.launch_config(**kw)


def prepare_decode_launch_op(spec: DecodeAttentionLaunchSpec, caps: DeviceCaps) -> PreparedDecodeAttentionLaunchOp:
    return PreparedDecodeAttentionLaunchOp(OpResolver(DECODE_LAUNCH_REGISTRY).resolve(spec, caps), spec)
