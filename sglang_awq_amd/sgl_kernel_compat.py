"""`sgl_kernel`-shaped module for the AWQ path.

The reference binds its AWQ op by name at import time (awq.py:62-77):
    CUDA / XPU:  from sgl_kernel import awq_dequantize
    HIP:         from ...awq_triton import awq_dequantize_triton as awq_dequantize
`install()` makes `import sgl_kernel; sgl_kernel.awq_dequantize / sgl_kernel.awq_gemm` resolve to the
gfx950 ops: it attaches the two names to an already-imported real `sgl_kernel` (whose ROCm build
registers no AWQ op at all, common_extension_rocm.cc:21-198) or, when no `sgl_kernel` is
importable, publishes this module under that name.  See INTEGRATION.md for the one-line change on
the reference's HIP branch.
"""
import importlib
import sys

from .ops import awq_dequantize, awq_gemm  # noqa: F401  (registers torch.ops.sgl_kernel.* as a side effect)

__all__ = ["awq_dequantize", "awq_gemm", "install"]


def install(force_module: bool = False):
    """Expose awq_dequantize / awq_gemm as attributes of `sgl_kernel`; returns that module."""
    this = sys.modules[__name__]
    target = sys.modules.get("sgl_kernel")
    if target is None and not force_module:
        try:
            target = importlib.import_module("sgl_kernel")
        except Exception:
            target = None
    if target is None or force_module:
        sys.modules["sgl_kernel"] = this
        return this
    target.awq_dequantize = awq_dequantize
    target.awq_gemm = awq_gemm
    return target
