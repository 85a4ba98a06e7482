"""bfloat16 helpers for the numpy oracle (TEST INFRASTRUCTURE ONLY)."""

import numpy as np


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bf16 bit pattern (uint16).

    Same rounding as `tensor.to(torch.bfloat16)` / Triton's `.to(tl.bfloat16)`.
    NaN is quieted to 0x7FC0 | sign like torch does.
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    lsb = (u >> np.uint32(16)) & np.uint32(1)
    rounded = (u + np.uint32(0x7FFF) + lsb) >> np.uint32(16)
    nan = np.isnan(x)
    if nan.any():
        rounded = np.where(nan, (u >> np.uint32(16)) | np.uint32(0x0040), rounded)
    return rounded.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bf16 value, returned as float32."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))
