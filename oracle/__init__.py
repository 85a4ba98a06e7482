"""CPU oracle for the sparse paged-attention hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy restatement of the
reference algorithms (CURRENTF/Sparse-vLLM, see the file:line citations on each
function).  It is the *checker* for the HIP path and the `cpu_baseline` leg of
`bench.py`; nothing under `sparse_vllm_amd/` may import it, and the product
path must fail loudly when the HIP extension is missing instead of falling back
to anything in here.

Pinning: every function here is checked in `tests/test_oracle_golden.py`
against fixtures under `tests/golden/` that were produced by importing the
reference itself (torch functions run natively on CPU, Triton kernels run under
`TRITON_INTERPRET=1`) with `tests/golden/gen_fixtures.py`, plus the
known-answer vectors of the reference's own unit tests.

Conventions
-----------
* bf16 tensors are carried as float32 numpy arrays whose values are exactly
  bf16-representable (`bf16_round`), mirroring how the reference kernels were
  run under the Triton interpreter.
* Index results are int64 (torch.long) unless the reference stores int32.
"""

from .bf16 import bf16_round, bf16_bits_to_f32, f32_to_bf16_bits  # noqa: F401
