"""MI355X-native AWQ int4 quantized-linear path (drop-in for sgl_kernel.awq_dequantize /
sgl_kernel.awq_gemm and the AWQLinearMethod that calls them).

Importing the package is cheap and never touches the GPU.  `sglang_awq_amd.ops` registers the torch
custom ops and needs the HIP library built (`sglang_awq_amd._lib.build()`); it is imported lazily.
"""
__version__ = "0.3.0"

_LAZY = {
    "awq_dequantize": "ops", "awq_gemm": "ops", "awq_linear": "ops",
    "AWQConfig": "awq", "AWQLinearMethod": "awq", "AWQMoEMethod": "moe",
    "weights_updated": "weight_update",
}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(f"module 'sglang_awq_amd' has no attribute {name!r}")
    import importlib

    return getattr(importlib.import_module(f"{__name__}.{mod}"), name)
