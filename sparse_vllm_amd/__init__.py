"""MI355X-native sparse paged-attention hot path (drop-in for Sparse-vLLM's
`sparse_method=` operator surface).  Kernels live in libsvk.so (HIP, gfx950) behind the
C ABI of include/svk.h; this package is the Python host mirror of the reference's
interfaces for that path."""

__version__ = "0.1.0"
