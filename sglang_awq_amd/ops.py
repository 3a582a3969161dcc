"""torch custom ops `sgl_kernel::awq_dequantize` and `sgl_kernel::awq_gemm` backed by the gfx950 HIP
library (include/awq_hip.h).

Drop-in boundary:
  * same namespace and schema string as the reference registers for CUDA,
    "awq_dequantize(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor"
    (sgl-kernel/csrc/common_extension.cc:126-127); the implementation is registered on the CUDA
    dispatch key, which is the key PyTorch-ROCm uses for HIP tensors;
  * `awq_gemm(Tensor input, Tensor qweight, Tensor scales, Tensor qzeros, int split_k_iters) -> Tensor`
    does not exist natively in the reference; argument order and meaning are those of its Triton
    function awq_gemm_triton (awq_triton.py:284-339);
  * fake (meta) implementations with the correct arity — the reference's own fake for
    awq_dequantize is declared with six parameters against a three-parameter schema
    (awq.py:950-960) and cannot be used.

No CPU implementation is registered: CPU tensors raise NotImplementedError from the dispatcher,
a missing HIP library raises AwqHipError.  Outputs and the split-K workspace come from the torch
caching allocator; the C side only enqueues kernels on the current stream (graph-capture safe).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import AwqHipError  # noqa: F401  (re-exported)

_DTYPE_CODE = {torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16, torch.float32: _lib.DTYPE_F32}
_WORKSPACE_BYTES = 4096 + (32 << 20)       # counters + the slab budget awq_gemm_workspace_bytes() never exceeds
_workspaces = {}

_SCHEMA_DEQUANT = "awq_dequantize(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor"
_SCHEMA_GEMM = "awq_gemm(Tensor input, Tensor qweight, Tensor scales, Tensor qzeros, int split_k_iters) -> Tensor"
_SCHEMA_LINEAR = "awq_linear(Tensor input, Tensor qweight, Tensor scales, Tensor qzeros, Tensor? bias) -> Tensor"


def _vp(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_awq_tensors(qweight, scales, qzeros):
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        # the reference calls data_ptr<int>(), which throws for any other dtype (awq_kernel.cu:201-202)
        raise RuntimeError("awq: qweight and qzeros must be int32 tensors")
    if scales.dtype not in _DTYPE_CODE:
        raise RuntimeError(f"awq: scales must be float16, bfloat16 or float32, got {scales.dtype}")
    if qweight.dim() != 2 or scales.dim() != 2 or qzeros.dim() != 2:
        raise RuntimeError("awq: qweight, scales and qzeros must be 2-D")
    if not (qweight.is_contiguous() and scales.is_contiguous() and qzeros.is_contiguous()):
        raise RuntimeError("awq: qweight, scales and qzeros must be contiguous (row-major)")
    if not (qweight.device == scales.device == qzeros.device):
        raise RuntimeError("awq: qweight, scales and qzeros must be on the same device")
    K, C = qweight.shape
    G = scales.shape[0]
    if K == 0 or C == 0 or G == 0 or K % G != 0:
        raise RuntimeError(f"awq: K={K} must be a positive multiple of the number of groups {G}")
    N = C * 8
    if tuple(scales.shape) != (G, N):
        raise RuntimeError(f"awq: scales must be [{G}, {N}], got {tuple(scales.shape)}")
    if tuple(qzeros.shape) != (G, C):
        raise RuntimeError(f"awq: qzeros must be [{G}, {C}], got {tuple(qzeros.shape)}")
    return K, N, K // G


class _on_device:
    """Make `dev` current only when it is not already (torch.cuda.device costs microseconds)."""

    def __init__(self, dev: torch.device):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _workspace(dev: torch.device) -> torch.Tensor:
    """One scratch buffer per device (the reference runs one forward thread per GPU process,
    engine.py:931).  Only the arrival counters at its head need to start at zero; every call
    leaves them zero again, so the buffer is initialised once, outside any timed or captured
    region when the first call is a warm-up (as it is in the reference's graph-capture flow)."""
    ws = _workspaces.get(dev.index)
    if ws is None:
        ws = torch.empty(_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        ws[:4096].zero_()
        _workspaces[dev.index] = ws
    return ws


# --------------------------------------------------------------------------- implementations (HIP tensors)
def _awq_dequantize_hip(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    K, N, g = _check_awq_tensors(qweight, scales, qzeros)
    lib = _lib.load()
    dev = qweight.device
    with _on_device(dev):
        out = torch.empty((K, N), dtype=scales.dtype, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.awq_dequantize(_vp(qweight), _vp(scales), _vp(qzeros), _vp(out), K, N, g,
                                _DTYPE_CODE[scales.dtype], ctypes.c_void_p(stream))
    _lib.check(rc, "awq_dequantize")
    return out


def _gemm_impl(input, qweight, scales, qzeros, bias, split_k_iters, variant=_lib.GEMM_AUTO, tune=0):
    K, N, g = _check_awq_tensors(qweight, scales, qzeros)
    if input.dim() != 2 or input.shape[1] != K:
        raise RuntimeError(f"awq_gemm: input must be [M, {K}], got {tuple(input.shape)}")
    if input.dtype != scales.dtype:
        raise RuntimeError(f"awq_gemm: input dtype {input.dtype} must equal scales dtype {scales.dtype}")
    if input.device != qweight.device:
        raise RuntimeError("awq_gemm: input must be on the weights' device")
    if input.stride(1) != 1:
        input = input.contiguous()
    if split_k_iters <= 0 or split_k_iters > 32 or split_k_iters & (split_k_iters - 1):
        raise RuntimeError("awq_gemm: split_k_iters must be a power of two in [1, 32]")  # awq_triton.py:307-308
    if bias is not None:
        if bias.dtype != scales.dtype or bias.dim() != 1 or bias.shape[0] != N or bias.device != input.device:
            raise RuntimeError(f"awq_gemm: bias must be a [{N}] {scales.dtype} tensor on the same device")
        bias = bias.contiguous()
    M = input.shape[0]
    ldx = input.stride(0) if M > 1 else max(input.stride(0), K)
    lib = _lib.load()
    dev = input.device
    with _on_device(dev):
        y = torch.empty((M, N), dtype=scales.dtype, device=dev)
        if M == 0:
            return y
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _workspace(dev)
        need = lib.awq_gemm_workspace_bytes(M, K, N, g, _DTYPE_CODE[scales.dtype])
        if need > ws.numel():
            # prefill-sized call on a large matrix: room for the on-the-fly re-layout; a per-call buffer from the caching
            # allocator (nothing in it needs to start at zero — the arrival counters are not used on that path)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        rc = lib.awq_gemm_ex(_vp(input), ldx, _vp(qweight), _vp(scales), _vp(qzeros), _vp(bias), _vp(y), _vp(ws),
                             ws.numel(), M, K, N, g, _DTYPE_CODE[scales.dtype], split_k_iters, variant, tune,
                             ctypes.c_void_p(stream))
    _lib.check(rc, "awq_gemm")
    return y


def _awq_gemm_hip(input, qweight, scales, qzeros, split_k_iters: int) -> torch.Tensor:
    return _gemm_impl(input, qweight, scales, qzeros, None, split_k_iters)


def _awq_linear_hip(input, qweight, scales, qzeros, bias) -> torch.Tensor:
    return _gemm_impl(input, qweight, scales, qzeros, bias, 1)


# --------------------------------------------------------------------------- fake (meta) implementations
def _awq_dequantize_fake(qweight, scales, qzeros):
    return qweight.new_empty((qweight.shape[0], qweight.shape[1] * 8), dtype=scales.dtype)


def _awq_gemm_fake(input, qweight, scales, qzeros, split_k_iters):
    return input.new_empty((input.shape[0], qweight.shape[1] * 8), dtype=scales.dtype)


def _awq_linear_fake(input, qweight, scales, qzeros, bias):
    return input.new_empty((input.shape[0], qweight.shape[1] * 8), dtype=scales.dtype)


# --------------------------------------------------------------------------- registration
def _has_op(ns: str, name: str) -> bool:
    try:
        return hasattr(getattr(torch.ops, ns), name)
    except Exception:
        return False


_libs = []


def _register(ns: str, schema: str, impl, fake):
    name = schema.split("(")[0]
    lib = torch.library.Library(ns, "FRAGMENT")
    _libs.append(lib)          # keep alive: registrations die with the Library object
    if not _has_op(ns, name):
        lib.define(schema)
    lib.impl(name, impl, "CUDA")
    try:
        torch.library.register_fake(f"{ns}::{name}", fake, lib=lib)
    except Exception:          # a fake already registered by another owner of the schema
        pass


_register("sgl_kernel", _SCHEMA_DEQUANT, _awq_dequantize_hip, _awq_dequantize_fake)
_register("sgl_kernel", _SCHEMA_GEMM, _awq_gemm_hip, _awq_gemm_fake)
_register("sglang_awq_amd", _SCHEMA_LINEAR, _awq_linear_hip, _awq_linear_fake)


# --------------------------------------------------------------------------- python entry points
def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """Same call as sgl_kernel.awq_dequantize (sgl-kernel/python/sgl_kernel/gemm.py:8-11)."""
    return torch.ops.sgl_kernel.awq_dequantize.default(qweight, scales, qzeros)


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor,
             split_k_iters: int = 1) -> torch.Tensor:
    """Fused dequantise + GEMM; argument order of awq_gemm_triton (awq_triton.py:289-294)."""
    return torch.ops.sgl_kernel.awq_gemm.default(input, qweight, scales, qzeros, split_k_iters)


def awq_linear(input, qweight, scales, qzeros, bias=None) -> torch.Tensor:
    """awq_gemm with the bias add of AWQLinearMethod.apply fused into the epilogue (awq.py:449-450)."""
    return torch.ops.sglang_awq_amd.awq_linear.default(input, qweight, scales, qzeros, bias)


def awq_repack(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> Optional[torch.Tensor]:
    """One-time MFMA-fragment-major copy of an AWQ weight for the decode kernel (include/awq_hip.h awq_repack).
    Returns a uint8 tensor, or None when the shape / dtype is not supported (callers keep the plain ops)."""
    K, N, g = _check_awq_tensors(qweight, scales, qzeros)
    lib = _lib.load()
    code = _DTYPE_CODE[scales.dtype]
    nbytes = lib.awq_repacked_bytes(K, N, g, code)
    if nbytes == 0:
        return None
    dev = qweight.device
    with _on_device(dev):
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.awq_repack(_vp(qweight), _vp(scales), _vp(qzeros), _vp(packed), K, N, g, code,
                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "awq_repack")
    return packed


def awq_gemm_repacked(input: torch.Tensor, packed: torch.Tensor, K: int, N: int, group_size: int,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = input @ W (+ bias) from the repacked copy, any M (GEMV passes of 32 rows up to 160 rows, the MFMA-bound
    tiled kernel beyond), fp16.  Same numerics as awq_gemm / awq_linear."""
    if input.dim() != 2 or input.shape[1] != K or input.dtype != torch.float16:
        raise RuntimeError(f"awq_gemm_repacked: input must be fp16 [M, {K}], got {input.dtype} {tuple(input.shape)}")
    if input.stride(1) != 1:
        input = input.contiguous()
    if bias is not None and (bias.dtype != torch.float16 or bias.shape != (N,) or not bias.is_contiguous()):
        raise RuntimeError(f"awq_gemm_repacked: bias must be a contiguous fp16 [{N}] tensor")
    M = input.shape[0]
    ldx = input.stride(0) if M > 1 else max(input.stride(0), K)
    lib = _lib.load()
    dev = input.device
    with _on_device(dev):
        y = torch.empty((M, N), dtype=torch.float16, device=dev)
        rc = lib.awq_gemm_repacked(_vp(input), ldx, _vp(packed), _vp(bias), _vp(y), M, K, N, group_size, _lib.DTYPE_F16,
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "awq_gemm_repacked")
    return y


def awq_gemm_variant(input, qweight, scales, qzeros, variant: int, tune: int = 0, bias=None) -> torch.Tensor:
    """Force a kernel variant (tests / A-B benchmarks); see include/awq_hip.h awq_gemm_ex."""
    return _gemm_impl(input, qweight, scales, qzeros, bias, 1, variant, tune)
