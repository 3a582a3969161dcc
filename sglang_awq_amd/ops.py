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
import os
from typing import Optional

import torch
from torch.multiprocessing.reductions import StorageWeakRef

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


def _workspace(dev: torch.device, stream: int) -> torch.Tensor:
    """Scratch per (device, stream): the split-K arrival counters at its head, the fp32 slabs and (M >= 33) the target of
    the on-the-fly re-layout are live for the duration of one call, and calls on one stream are ordered — two streams of
    one device must never share them (alternate-stream overlap, capture on one stream while another runs).  Only the
    counters need to start at zero; every call leaves them zero again, so a buffer is initialised once.  Never during stream
    capture: the allocation would come from the graph's private pool and the zero-fill would become a graph node that has
    not run — a first call on a capturing stream raises; warm the op up on the capture stream first (as the reference's
    graph runner does, cuda_graph_runner.py:716-719) or call `prepare_stream_workspaces()`."""
    key = (dev.index, stream)
    ws = _workspaces.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sglang_awq_amd: the AWQ op needs its per-stream scratch, and the first call on this stream is inside "
                               "a graph capture. Run the op once eagerly on the capture stream (warm-up), or call "
                               "sglang_awq_amd.ops.prepare_stream_workspaces(stream) before capturing.")
        ws = torch.empty(_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        ws[:4096].zero_()
        _workspaces[key] = ws
    return ws


def prepare_stream_workspaces(stream: Optional["torch.cuda.Stream"] = None, device: Optional[torch.device] = None) -> None:
    """Create (eagerly, on the current stream) the scratch buffers that calls on `stream` will use, so that a capture on that
    stream never has to allocate: the op's split-K workspace and the decode harness' attention scratch."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    s = torch.cuda.current_stream(dev) if stream is None else stream
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("prepare_stream_workspaces must run outside graph capture")
    key = (dev.index, s.cuda_stream)
    if key not in _workspaces:
        ws = torch.empty(_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        ws[:4096].zero_()
        _workspaces[key] = ws
    from . import aux_ops

    aux_ops.prepare_attention_workspace(dev, s.cuda_stream)
    torch.cuda.current_stream(dev).synchronize()          # the zero-fills are done before any other stream may use the buffers


# --------------------------------------------------------------------------- repacked copies kept by the drop-in op
# `sgl_kernel.awq_gemm` receives the checkpoint tensors on every call; the kernels that reach the HBM roofline want the
# MFMA-fragment-major layout (awq_repack).  Serving weights are static, so the op can keep one repacked copy per weight it has
# seen, keyed on the three tensors' storage addresses and validated on every hit by (a) the tensors' version counters and
# (b) weak references to the storages (a freed weight whose address is reused can never alias a stale copy).
# (a) has a hole that no check at call time can close: the reference's weight loaders and its hot-update paths write through
# `param.data.copy_(w)` (layers/parameter.py:59,124), which bumps the version counter of the temporary `.data` alias, not of
# the tensor the op receives.  The cache is therefore only ON when something guarantees an invalidation after such writes:
#   * SGLANG_AWQ_AMD_OP_CACHE unset ("auto", the default): off until `sgl_kernel_compat.install()` has hooked the
#     reference's reload paths (ModelRunner.update_weights_from_* and load_weights_and_postprocess call
#     `weight_update.weights_updated`, which clears this cache and re-lays this package's layers out), or until the caller
#     takes responsibility with `awq_gemm_cache_enable(True)`;
#   * SGLANG_AWQ_AMD_OP_CACHE=1: on from the start (the caller calls awq_gemm_cache_clear() after in-place weight writes);
#   * SGLANG_AWQ_AMD_OP_CACHE=0: always off.
# The copy is made on the first eager call for a weight — never during stream capture (a capture-time miss runs the
# checkpoint-layout kernel, whose fp32 summation order differs: same tolerance, not bit-identical to the cached route) —
# costs awq_repacked_bytes() of HBM per weight (cap SGLANG_AWQ_AMD_OP_CACHE_GB, default 64; never more than half of the memory free
# on the device when a copy is about to be made), and is dropped by
# awq_gemm_cache_clear().  Results on a hit are bit-identical to awq_gemm_repacked (same kernels).  Tensors without a version
# counter (created under torch.inference_mode()) bypass the cache.
_op_cache = {}
_OP_CACHE_MODE = {"1": "on", "0": "off"}.get(os.environ.get("SGLANG_AWQ_AMD_OP_CACHE", ""), "auto")
_OP_CACHE_ENABLED = _OP_CACHE_MODE == "on"
_OP_CACHE_MAX_BYTES = int(float(os.environ.get("SGLANG_AWQ_AMD_OP_CACHE_GB", "64")) * (1 << 30))
_op_cache_bytes = 0


def _reload_hooks_installed() -> None:
    """Called by sgl_kernel_compat.install() once the reference's weight-reload paths invalidate the cache ("auto" mode)."""
    global _OP_CACHE_ENABLED
    if _OP_CACHE_MODE == "auto":
        _OP_CACHE_ENABLED = True


def awq_gemm_cache_clear() -> None:
    """Drop every repacked copy the awq_gemm op holds (frees the HBM once in-flight kernels have finished)."""
    global _op_cache_bytes
    _op_cache.clear()
    _op_cache_bytes = 0


def awq_gemm_cache_enable(flag: bool) -> None:
    """Turn the op's repacked-copy cache on or off at run time (off: every call runs the checkpoint-layout kernels)."""
    global _OP_CACHE_ENABLED
    _OP_CACHE_ENABLED = bool(flag)
    if not flag:
        awq_gemm_cache_clear()


def awq_gemm_cache_info() -> dict:
    return {"entries": len(_op_cache), "bytes": _op_cache_bytes, "enabled": _OP_CACHE_ENABLED, "mode": _OP_CACHE_MODE}


def _op_cached_repack(qweight, scales, qzeros, K, N, g):
    """The op's repacked copy of (qweight, scales, qzeros), made on a miss; None if unsupported / disabled / capturing."""
    global _op_cache_bytes
    if not _OP_CACHE_ENABLED or scales.dtype not in (torch.float16, torch.bfloat16):
        return None
    key = (qweight.device.index, qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(), K, N, g)
    ent = _op_cache.get(key)
    try:
        versions = (qweight._version, scales._version, qzeros._version)
    except RuntimeError:                                  # inference tensors track no version counter: nothing to validate against
        return None
    if ent is not None:
        packed, vers, refs = ent
        if vers == versions and not any(r.expired() for r in refs):
            return packed
        del _op_cache[key]
        _op_cache_bytes -= packed.numel()
    if torch.cuda.is_current_stream_capturing():
        return None                                   # fill on an eager (warm-up) call only
    nbytes = _lib.load().awq_repacked_bytes(K, N, g, _DTYPE_CODE[scales.dtype])
    if nbytes == 0 or _op_cache_bytes + nbytes > _OP_CACHE_MAX_BYTES:
        return None
    # the copies are made on the first forward, after a serving stack has sized its KV pool: never take more than half of what the
    # device has free at that moment (the call then runs the checkpoint-layout kernel, slower but allocation-free)
    free_bytes, _total = torch.cuda.mem_get_info(qweight.device)
    if nbytes > free_bytes // 2:
        return None
    packed = awq_repack(qweight, scales, qzeros)
    if packed is None:
        return None
    refs = tuple(StorageWeakRef(t.untyped_storage()) for t in (qweight, scales, qzeros))
    _op_cache[key] = (packed, versions, refs)
    _op_cache_bytes += nbytes
    return packed


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


def _gemm_impl(input, qweight, scales, qzeros, bias, split_k_iters, variant=_lib.GEMM_AUTO, tune=0, use_cache=True):
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
        if use_cache and variant == _lib.GEMM_AUTO and M > 0 and ldx % 8 == 0 and input.data_ptr() % 16 == 0:
            packed = _op_cached_repack(qweight, scales, qzeros, K, N, g)
            if packed is not None:
                try:
                    return awq_gemm_repacked(input, packed, K, N, g, bias)
                except AwqHipError:                   # e.g. 32 rows on a very wide strip: no repacked variant
                    pass
        y = torch.empty((M, N), dtype=scales.dtype, device=dev)
        if M == 0:
            return y
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _workspace(dev, stream)
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
    # this package's own op: its caller (AWQLinearMethod) either holds a repacked copy already or asked for none
    # (repack=False, or the fallback after a BAD_VARIANT on the repacked route) — never a second, hidden copy
    return _gemm_impl(input, qweight, scales, qzeros, bias, 1, use_cache=False)


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


def check_packed(packed: torch.Tensor, K: int, N: int, group_size: int, dev: torch.device, dtype: torch.dtype = torch.float16) -> None:
    """A repacked buffer is opaque bytes: the kernels index it from (K, N, group_size) alone, so a mismatching buffer would be
    read out of bounds.  Refuse anything but a uint8 tensor of exactly awq_repacked_bytes(K, N, group_size) on `dev`."""
    need = _lib.load().awq_repacked_bytes(K, N, group_size, _DTYPE_CODE.get(dtype, -1))
    if need == 0:
        raise AwqHipError(f"awq repacked layout does not support K={K} N={N} group_size={group_size} {dtype} "
                          "(fp16 / bf16, K % 128 == 0, group_size a multiple of 128 or 32 / 64)")
    if packed.dtype != torch.uint8 or packed.numel() != need or not packed.is_contiguous() or packed.device != dev:
        raise RuntimeError(f"awq: repacked buffer must be a contiguous uint8 tensor of {need} bytes on {dev} for K={K} N={N} "
                           f"group_size={group_size}, got {packed.dtype} x {packed.numel()} on {packed.device}")


def awq_gemm_repacked(input: torch.Tensor, packed: torch.Tensor, K: int, N: int, group_size: int,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = input @ W (+ bias) from the repacked copy, any M, in the dtype of `input` — fp16 or bf16, which must be the dtype of
    the scales the copy was made from (the copy stores their bits).  Same numerics as awq_gemm / awq_linear."""
    dt = input.dtype
    if input.dim() != 2 or input.shape[1] != K or dt not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"awq_gemm_repacked: input must be fp16 / bf16 [M, {K}], got {input.dtype} {tuple(input.shape)}")
    if input.stride(1) != 1:
        input = input.contiguous()
    if bias is not None and (bias.dtype != dt or bias.shape != (N,) or not bias.is_contiguous()):
        raise RuntimeError(f"awq_gemm_repacked: bias must be a contiguous {dt} [{N}] tensor")
    M = input.shape[0]
    ldx = input.stride(0) if M > 1 else max(input.stride(0), K)
    lib = _lib.load()
    dev = input.device
    check_packed(packed, K, N, group_size, dev, dt)
    with _on_device(dev):
        y = torch.empty((M, N), dtype=dt, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if M > 8 and lib.awq_gemm_repacked_workspace_bytes(M, K, N, group_size, _DTYPE_CODE[dt]):
            # 9 .. 32 rows on a narrow matrix (split-K GEMV) or 33 rows up with few MFMA tiles (split-K tiles): both want the
            # per-(device, stream) scratch (counters + fp32 partials)
            ws = _workspace(dev, stream)
            rc = lib.awq_gemm_repacked_ws(_vp(input), ldx, _vp(packed), _vp(bias), _vp(y), _vp(ws), ws.numel(), M, K, N, group_size,
                                          _DTYPE_CODE[dt], ctypes.c_void_p(stream))
        else:
            rc = lib.awq_gemm_repacked(_vp(input), ldx, _vp(packed), _vp(bias), _vp(y), M, K, N, group_size, _DTYPE_CODE[dt],
                                       ctypes.c_void_p(stream))
    _lib.check(rc, "awq_gemm_repacked")
    return y


def awq_gemm_variant(input, qweight, scales, qzeros, variant: int, tune: int = 0, bias=None) -> torch.Tensor:
    """Force a kernel variant (tests / A-B benchmarks); see include/awq_hip.h awq_gemm_ex."""
    return _gemm_impl(input, qweight, scales, qzeros, bias, 1, variant, tune)
