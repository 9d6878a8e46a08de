"""fastllama_amd -- MI355X-native (gfx950) implementation of fastLLaMa's Q4_0/Q4_1 matmul hot path.

Layout of the package (only what the path needs):

* ``csrc/``   hand-written HIP kernels + the C-ABI (``include/fastllama_hip.h``, ``include/fastllama.h``)
* ``hip.py``  ctypes binding of the C-ABI (loads the in-tree ``libfastllama_hip.so``; no fallback)
* ``ops.py``  host-side mirror of the reference's operator interface (quantize_fns_t / ggml_mul_mat)

Importing the package does not import torch; ``fastllama_amd.ops`` does (device memory + streams only).
"""
from . import hip  # noqa: F401

__all__ = ["hip"]
__version__ = "0.1.0"
