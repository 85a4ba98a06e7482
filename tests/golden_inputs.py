"""Seeded inputs shared by tests/golden/gen_fixtures.py (which runs the REFERENCE on them) and the tests (which run the
oracle and the HIP path on them): large inputs are regenerated from the seed on both sides instead of being committed,
only the reference's OUTPUTS are stored in the fixture.  numpy's PCG64 streams are platform-independent."""

import numpy as np

from oracle import bf16_round


def kivi_long_row_inputs(seed: int = 20261003):
    """One long row of a KIVI-int4 full layer, one KV head of a tensor-parallel Qwen2.5-7B rank (7 query heads, head_dim
    128): sink 8 raw tokens, 257 quantised blocks of 32 (8224 tokens), a raw tail of 27 -> 8259 tokens; block slots and raw
    slots scattered; fp32 per-channel key parameters, bf16-valued per-token value parameters (the manager's formats)."""
    rng = np.random.default_rng(seed)
    Hq, Hkv, D, G = 7, 1, 128, 32
    sink, nblk, tail = 8, 257, 27
    n_blocks, n_raw = nblk + 5, sink + tail + 9
    length = sink + nblk * G + tail
    f = lambda *s, scale=1.0: bf16_round((rng.standard_normal(s) * scale).astype(np.float32))
    u = lambda *s, lo=0.0, hi=1.0: bf16_round((lo + (hi - lo) * rng.random(s)).astype(np.float32))
    d = dict(Hq=Hq, Hkv=Hkv, D=D, G=G, length=length)
    d["q"] = f(1, Hq, D, scale=0.5)
    d["raw_k"], d["raw_v"] = f(n_raw, Hkv, D, scale=0.5), f(n_raw, Hkv, D, scale=0.5)
    d["key_packed"] = rng.integers(-2 ** 31, 2 ** 31 - 1, (n_blocks, Hkv, D, G // 8), dtype=np.int64).astype(np.int32)
    d["value_packed"] = rng.integers(-2 ** 31, 2 ** 31 - 1, (n_blocks, Hkv, G, D // 8), dtype=np.int64).astype(np.int32)
    d["key_scales"], d["key_mins"] = u(n_blocks, Hkv, D, hi=0.08), -u(n_blocks, Hkv, D, hi=0.6)
    d["value_scales"], d["value_mins"] = u(n_blocks, Hkv, G, D // G, hi=0.08), -u(n_blocks, Hkv, G, D // G, hi=0.6)
    raw_map = np.full((2, length + 5), -1, np.int32)
    blk_map = np.full((2, length + 5), -1, np.int32)
    blk_start = np.zeros(n_blocks, np.int32)
    rperm = rng.permutation(n_raw).astype(np.int32)
    bperm = rng.permutation(n_blocks).astype(np.int32)
    row = 1
    raw_map[row, :sink] = rperm[:sink]
    for i in range(nblk):
        blk_map[row, sink + G * i: sink + G * (i + 1)] = bperm[i]
        blk_start[bperm[i]] = sink + G * i
    raw_map[row, sink + nblk * G: length] = rperm[sink: sink + tail]
    d.update(raw_map=raw_map, blk_map=blk_map, blk_start=blk_start, req=np.array([row], np.int32),
             lens=np.array([length], np.int32))
    return d
