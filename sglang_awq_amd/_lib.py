"""ctypes binding of lib/libawq_hip.so (the C ABI declared in include/awq_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  `build()`
compiles it in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "lib", "libawq_hip.so")

# every symbol include/awq_hip.h declares
EXPORTS = (
    "awq_hip_abi_version",
    "awq_hip_build_info",
    "awq_hip_status_string",
    "awq_dequantize",
    "awq_gemm_workspace_bytes",
    "awq_gemm",
    "awq_gemm_ex",
    "awq_repacked_bytes",
    "awq_repack",
    "awq_gemm_repacked",
    "awq_gemm_repacked_workspace_bytes",
    "awq_gemm_repacked_ws",
)
# include/awq_aux.h (decode-harness helpers, not part of the operator boundary)
AUX_EXPORTS = ("awq_aux_add_rmsnorm", "awq_aux_decode_attention", "awq_aux_decode_attention_workspace_bytes",
               "awq_aux_argmax_advance", "awq_aux_silu_mul",
               "awq_aux_gemv_repacked_fused", "awq_aux_moe_gemv", "awq_aux_moe_gemv_blocks", "awq_aux_moe_align_blocks",
               "awq_aux_moe_align_blocks_n", "awq_aux_moe_gemm_blocks", "awq_aux_moe_sum")
ABI_VERSION = 2

DTYPE_F16, DTYPE_BF16, DTYPE_F32 = 0, 1, 2
GEMM_AUTO, GEMM_GENERIC, GEMM_SKINNY, GEMM_TILED = 0, 1, 2, 3



_lock = threading.Lock()
_lib = None


class AwqHipError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources into lib/libawq_hip.so (make decides what is stale)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise AwqHipError(f"building libawq_hip.so failed:\n{res.stdout}\n{res.stderr}")
    if not os.path.exists(LIB_PATH):
        raise AwqHipError(f"make succeeded but {LIB_PATH} is missing")
    return LIB_PATH


def _bind(L):
    i64, vp, ci, sz = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.awq_hip_abi_version.argtypes = []
    L.awq_hip_abi_version.restype = ci
    L.awq_hip_build_info.argtypes = []
    L.awq_hip_build_info.restype = ctypes.c_char_p
    L.awq_hip_status_string.argtypes = [ci]
    L.awq_hip_status_string.restype = ctypes.c_char_p
    L.awq_dequantize.argtypes = [vp, vp, vp, vp, i64, i64, i64, ci, vp]
    L.awq_dequantize.restype = ci
    L.awq_gemm_workspace_bytes.argtypes = [i64, i64, i64, i64, ci]
    L.awq_gemm_workspace_bytes.restype = sz
    L.awq_gemm.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, ci, ci, vp]
    L.awq_gemm.restype = ci
    L.awq_gemm_ex.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, ci, ci, ci, i64, vp]
    L.awq_gemm_ex.restype = ci
    L.awq_repacked_bytes.argtypes = [i64, i64, i64, ci]
    L.awq_repacked_bytes.restype = sz
    L.awq_repack.argtypes = [vp, vp, vp, vp, i64, i64, i64, ci, vp]
    L.awq_repack.restype = ci
    L.awq_gemm_repacked.argtypes = [vp, i64, vp, vp, vp, i64, i64, i64, i64, ci, vp]
    L.awq_gemm_repacked.restype = ci
    L.awq_gemm_repacked_workspace_bytes.argtypes = [i64, i64, i64, i64, ci]
    L.awq_gemm_repacked_workspace_bytes.restype = ctypes.c_size_t
    L.awq_gemm_repacked_ws.argtypes = [vp, i64, vp, vp, vp, vp, ctypes.c_size_t, i64, i64, i64, i64, ci, vp]
    L.awq_gemm_repacked_ws.restype = ci
    L.awq_aux_add_rmsnorm.argtypes = [vp, vp, vp, vp, i64, i64, ctypes.c_float, vp]
    L.awq_aux_add_rmsnorm.restype = ci
    L.awq_aux_decode_attention.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, ctypes.c_float, ci, vp, sz, vp]
    L.awq_aux_decode_attention_workspace_bytes.argtypes = [i64, i64, i64, ci]
    L.awq_aux_decode_attention_workspace_bytes.restype = sz
    L.awq_aux_decode_attention.restype = ci
    L.awq_aux_gemv_repacked_fused.argtypes = [vp, i64, vp, vp, i64, i64, i64, i64, ci, vp, vp, vp, vp, ctypes.c_float, ci, vp]
    L.awq_aux_gemv_repacked_fused.restype = ci
    L.awq_aux_moe_gemv.argtypes = [vp, i64, ci, vp, i64, i64, vp, vp, vp, i64, i64, i64, i64, ci, ci, vp]
    L.awq_aux_moe_gemv.restype = ci
    L.awq_aux_moe_gemv_blocks.argtypes = [vp, i64, ci, vp, i64, vp, vp, i64, vp, vp, i64, i64, i64, ci, ci, vp]
    L.awq_aux_moe_gemv_blocks.restype = ci
    L.awq_aux_moe_align_blocks.argtypes = [vp, i64, i64, vp, vp, i64, vp]
    L.awq_aux_moe_align_blocks.restype = ci
    L.awq_aux_moe_sum.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp]
    L.awq_aux_moe_sum.restype = ci
    L.awq_aux_moe_align_blocks_n.argtypes = [vp, i64, i64, ci, vp, vp, i64, vp]
    L.awq_aux_moe_align_blocks_n.restype = ci
    L.awq_aux_moe_gemm_blocks.argtypes = [vp, i64, ci, vp, i64, vp, vp, i64, ci, vp, vp, i64, i64, i64, ci, ci, vp]
    L.awq_aux_moe_gemm_blocks.restype = ci
    L.awq_aux_argmax_advance.argtypes = [vp, vp, vp, i64, i64, vp]
    L.awq_aux_argmax_advance.restype = ci
    L.awq_aux_silu_mul.argtypes = [vp, vp, i64, i64, vp]
    L.awq_aux_silu_mul.restype = ci


def load():
    """Load (once) and return the ctypes library; raises AwqHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own libamdhip64.so; if this library (linked against the system ROCm's
    # libamdhip64.so.7) were loaded before it, the process would hold two HIP runtimes and every launch on a torch
    # stream would fail.  Loaded second, the dependency resolves to the runtime torch already brought in.
    import torch  # noqa: F401

    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise AwqHipError(
                    f"{LIB_PATH} not found: the HIP extension is not built. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C sglang_awq_amd/csrc`). "
                    "There is no CPU fallback for the AWQ ops.")
            try:
                L = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise AwqHipError(f"cannot load {LIB_PATH}: {e}") from e
            missing = [s for s in EXPORTS + AUX_EXPORTS if not hasattr(L, s)]
            if missing:
                raise AwqHipError(f"{LIB_PATH} lacks symbols {missing}")
            _bind(L)
            if L.awq_hip_abi_version() != ABI_VERSION:
                raise AwqHipError(f"ABI version {L.awq_hip_abi_version()} != expected {ABI_VERSION}; rebuild")
            _lib = L
    return _lib


ERR_BAD_VARIANT = -7


def check(status: int, what: str):
    if status != 0:
        msg = load().awq_hip_status_string(status).decode()
        raise AwqHipError(f"{what} failed: {msg} (status {status})")
