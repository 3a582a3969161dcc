"""CPU oracle for the AWQ int4 quantized-linear path (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package `sglang_awq_amd/` imports this
module; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may.  The product path is the HIP library behind `include/awq_hip.h` and fails loudly when
that library is missing.

This is a restatement (not a copy) of the reference's arithmetic, function by function:

  unpack_awq_int4      <- awq_triton.py:56-69 (shifts), :351-355; tests' reverse_awq_order
                          (sgl-kernel/tests/test_awq_dequant.py:9-22)
  awq_dequantize       <- op sgl_kernel::awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:126-221;
                          torch form awq_triton.py:342-368
  awq_gemm             <- awq_triton.py:289-339 (argument order / shapes) with the numerics of
                          AWQLinearMethod.apply, awq.py:446-447 (dequantise to the scale dtype,
                          then a matmul that accumulates wider than fp16)
  awq_linear_apply     <- awq.py:434-451
  shard_*              <- parameter.py:93-124, :242-286, :539-550; awq.py:372-385

Parity pin: checked bit-for-bit against outputs of the reference's own CPU-runnable functions
(`awq_dequantize_decomposition`, the two test files' `awq_dequantize_torch`, and the Triton
kernels under TRITON_INTERPRET=1) that `tests/golden/make_golden.py` captured in the build
container; see tests/test_oracle_golden.py.

Representation: fp16 tensors are np.float16, fp32 are np.float32, bf16 tensors are np.uint16
holding the raw bits (numpy has no bfloat16).  `dtype` arguments are "f16" | "bf16" | "f32".
"""
from __future__ import annotations

import numpy as np

# logical column j of a packed word lives in nibble AWQ_NIBBLE_OF_COL[j]
AWQ_NIBBLE_OF_COL = (0, 4, 1, 5, 2, 6, 3, 7)
PACK_FACTOR = 8


# --------------------------------------------------------------------------- dtype helpers
def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    nan = np.isnan(x)
    out = rounded.astype(np.uint16)
    if nan.any():
        out = np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x0040)).astype(np.uint16), out)
    return out


def to_f64(a: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "bf16":
        return bf16_bits_to_f32(a).astype(np.float64)
    return a.astype(np.float64)


def from_f64(a: np.ndarray, dtype: str) -> np.ndarray:
    """Round a float64 array once into the storage dtype."""
    if dtype == "f16":
        return a.astype(np.float16)
    if dtype == "f32":
        return a.astype(np.float32)
    if dtype == "bf16":
        # f64 -> f32 (RNE) -> bf16 (RNE) would round twice.  Make the intermediate a
        # round-to-odd fp32 (truncate toward zero, set the last bit if inexact); a final RNE
        # from a round-to-odd value with >= 2 extra bits equals a single rounding.
        a = np.asarray(a, dtype=np.float64)
        t = a.astype(np.float32)
        inexact = t.astype(np.float64) != a
        away = np.abs(t.astype(np.float64)) > np.abs(a)
        t = np.where(inexact & away, np.nextafter(t, np.float32(0)), t).astype(np.float32)
        u = t.view(np.uint32) | inexact.astype(np.uint32)
        return f32_to_bf16_bits(u.view(np.float32))
    raise ValueError(dtype)


def dtype_of(a: np.ndarray) -> str:
    if a.dtype == np.float16:
        return "f16"
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.uint16:
        return "bf16"
    raise TypeError(f"unsupported scale/activation array dtype {a.dtype}")


# --------------------------------------------------------------------------- integer unpack
def unpack_awq_int4(packed: np.ndarray) -> np.ndarray:
    """int32 [R, C] -> uint8 [R, 8C] in logical column order.

    Logical column 8c+j is nibble AWQ_NIBBLE_OF_COL[j] of word c (awq_triton.py:56-69).  The
    shift is done on the unsigned reinterpretation so words with bit 31 set unpack correctly
    (the reference relies on `& 0xF` after an arithmetic shift for the same effect).
    """
    if packed.dtype != np.int32 or packed.ndim != 2:
        raise TypeError("packed AWQ tensor must be a 2-D int32 array")
    u = np.ascontiguousarray(packed).view(np.uint32)
    out = np.empty((u.shape[0], u.shape[1], PACK_FACTOR), dtype=np.uint8)
    for j, nib in enumerate(AWQ_NIBBLE_OF_COL):
        out[:, :, j] = (u >> np.uint32(4 * nib)) & np.uint32(0xF)
    return out.reshape(u.shape[0], u.shape[1] * PACK_FACTOR)


def pack_awq_int4(vals: np.ndarray) -> np.ndarray:
    """Inverse of unpack_awq_int4: uint8 [R, N] (values 0..15) -> int32 [R, N/8]."""
    r, n = vals.shape
    if n % PACK_FACTOR:
        raise ValueError("N must be a multiple of 8")
    v = vals.astype(np.uint32).reshape(r, n // PACK_FACTOR, PACK_FACTOR)
    word = np.zeros((r, n // PACK_FACTOR), dtype=np.uint32)
    for j, nib in enumerate(AWQ_NIBBLE_OF_COL):
        word |= (v[:, :, j] & np.uint32(0xF)) << np.uint32(4 * nib)
    return word.view(np.int32)


def check_awq_shapes(qweight, scales, qzeros):
    """Shape contract of the op (awq_kernel.cu:187-199, awq_triton.py:246-250)."""
    if qweight.ndim != 2 or scales.ndim != 2 or qzeros.ndim != 2:
        raise ValueError("qweight, scales and qzeros must be 2-D")
    k, nc = qweight.shape
    groups = scales.shape[0]
    if groups == 0 or k % groups:
        raise ValueError("K must be a multiple of the number of groups")
    g = k // groups
    n = nc * PACK_FACTOR
    if scales.shape != (groups, n):
        raise ValueError(f"scales must be [{groups}, {n}], got {scales.shape}")
    if qzeros.shape != (groups, nc):
        raise ValueError(f"qzeros must be [{groups}, {nc}], got {qzeros.shape}")
    return k, n, g


# --------------------------------------------------------------------------- dequantize
def awq_dequantize(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> np.ndarray:
    """W[k, n] = (q[k, n] - z[k // g, n]) * s[k // g, n], one rounding in the scale dtype.

    (q - z) is an exact integer in [-15, 15]; the product with an 11-bit (fp16) or 8-bit
    (bf16) significand is exact in fp32, so "multiply in fp32, round once" reproduces both the
    CUDA kernel's sub.f16x2 + mul.rn.f16x2 (awq_kernel.cu:151-158) and torch's CPU half/bf16
    multiply used by awq_triton.py:363-367.
    """
    k, n, g = check_awq_shapes(qweight, scales, qzeros)
    dt = dtype_of(scales)
    q = unpack_awq_int4(qweight).astype(np.int16)
    z = unpack_awq_int4(qzeros).astype(np.int16)
    diff = (q.reshape(k // g, g, n) - z[:, None, :]).astype(np.float32)
    if dt == "f16":
        prod = diff * scales.astype(np.float32)[:, None, :]
        return prod.astype(np.float16).reshape(k, n)
    if dt == "bf16":
        prod = diff * bf16_bits_to_f32(scales)[:, None, :]
        return f32_to_bf16_bits(prod).reshape(k, n)
    return (diff * scales[:, None, :]).reshape(k, n).astype(np.float32)


# --------------------------------------------------------------------------- fused GEMM
def awq_gemm_exact(x: np.ndarray, qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> np.ndarray:
    """float64 value of x @ dequant(W): the dequantised weight is rounded to the scale dtype
    first (as awq.py:446 materialises it), the contraction is carried in float64.  This is the
    un-rounded quantity every "accumulate wider than fp16, round once" GEMM approximates."""
    k, n, _ = check_awq_shapes(qweight, scales, qzeros)
    if x.ndim != 2 or x.shape[1] != k:
        raise ValueError(f"input must be [M, {k}]")
    dt = dtype_of(scales)
    if dtype_of(x) != dt:
        raise TypeError("input and scales must share a dtype")
    w = to_f64(awq_dequantize(qweight, scales, qzeros), dt)
    return to_f64(x, dt) @ w


def awq_gemm(x, qweight, scales, qzeros, split_k_iters: int = 1) -> np.ndarray:
    """`awq_gemm(input, qweight, scales, qzeros, split_k_iters)` (awq_triton.py:289-339 order).

    Result dtype = scales dtype, shape [M, N].  split_k_iters must be a power of two <= 32
    (awq_triton.py:307-308); it only partitions the K loop and does not change the value the
    oracle returns (exact sum, one rounding)."""
    if split_k_iters <= 0 or split_k_iters & (split_k_iters - 1) or split_k_iters > 32:
        raise ValueError("split_k_iters must be a power of two in [1, 32]")
    return from_f64(awq_gemm_exact(x, qweight, scales, qzeros), dtype_of(scales))


# --------------------------------------------------------------------------- linear method
def awq_linear_apply(x, qweight, scales, qzeros, bias=None) -> np.ndarray:
    """AWQLinearMethod.apply (awq.py:434-451): flatten leading dims, y = x2d @ dequant(W) rounded
    to the activation dtype, in-place `add_(bias)` (a second rounding), reshape back."""
    dt = dtype_of(scales)
    n = qweight.shape[-1] * PACK_FACTOR
    lead = x.shape[:-1]
    x2d = x.reshape(-1, x.shape[-1])
    y = awq_gemm(x2d, qweight, scales, qzeros)
    if bias is not None:
        y = from_f64(to_f64(y, dt) + to_f64(bias, dt)[None, :], dt)
    return y.reshape(lead + (n,))


# --------------------------------------------------------------------------- TP sharding
def check_partition(input_size_per_partition: int, output_size_per_partition: int, group_size: int):
    """create_weights' two legality checks (awq.py:372-385)."""
    if input_size_per_partition % group_size != 0:
        raise ValueError("The input size is not aligned with the quantized weight shape. "
                         "This can be caused by too large tensor parallel size.")
    if output_size_per_partition % PACK_FACTOR != 0:
        raise ValueError("The output size is not aligned with the quantized weight shape. "
                         "This can be caused by too large tensor parallel size.")


def shard_column_parallel(qweight, scales, qzeros, rank: int, tp: int):
    """Column-parallel shard (parameter.py:93-124 with the packed adjustment :539-550): the
    output dim is split evenly; packed tensors are indexed in units of 8 logical columns."""
    n = scales.shape[1]
    if n % tp:
        raise ValueError("N must divide by tp")
    n_r = n // tp
    check_partition(qweight.shape[0], n_r, qweight.shape[0] // scales.shape[0])
    c0, c1 = rank * n_r, (rank + 1) * n_r
    return (qweight[:, c0 // PACK_FACTOR:c1 // PACK_FACTOR], scales[:, c0:c1],
            qzeros[:, c0 // PACK_FACTOR:c1 // PACK_FACTOR])


def shard_row_parallel(qweight, scales, qzeros, rank: int, tp: int):
    """Row-parallel shard (parameter.py:242-286): the input dim of all three tensors is split
    evenly, which needs whole quantisation groups per rank (awq.py:372-377)."""
    k = qweight.shape[0]
    groups = scales.shape[0]
    g = k // groups
    if k % tp:
        raise ValueError("K must divide by tp")
    k_r = k // tp
    check_partition(k_r, scales.shape[1], g)
    g_r = k_r // g
    return (qweight[rank * k_r:(rank + 1) * k_r], scales[rank * g_r:(rank + 1) * g_r],
            qzeros[rank * g_r:(rank + 1) * g_r])


def row_parallel_reference(x, qweight, scales, qzeros, tp: int, bias=None) -> np.ndarray:
    """RowParallelLinear.forward (linear.py:1388-1414): each rank multiplies its K-slice, bias is
    added on rank 0 only, partial outputs (already rounded to the activation dtype) are summed by
    the all-reduce.  Returns the float64 sum of the per-rank rounded partials."""
    dt = dtype_of(scales)
    k = qweight.shape[0]
    k_r = k // tp
    total = None
    for r in range(tp):
        qw, s, qz = shard_row_parallel(qweight, scales, qzeros, r, tp)
        part = awq_linear_apply(x[..., r * k_r:(r + 1) * k_r], qw, s, qz, bias if r == 0 else None)
        p64 = to_f64(part, dt)
        total = p64 if total is None else total + p64
    return total
