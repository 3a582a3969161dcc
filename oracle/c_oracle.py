"""ctypes loader for oracle/_build/libawq_oracle.so (the C restatement in awq_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of awq_oracle.c.  numpy in / numpy out; bf16 arrays are
np.uint16 bit patterns as in awq_ref.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libawq_oracle.so")
_DT = {"f16": 0, "bf16": 1, "f32": 2}
_NP = {"f16": np.float16, "bf16": np.uint16, "f32": np.float32}
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "awq_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.awq_oracle_unpack.argtypes = [vp, vp, i64, i64]
        L.awq_oracle_dequantize.argtypes = [vp, vp, vp, vp, i64, i64, i64, ci, ci]
        L.awq_oracle_gemm.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, ci]
        for f in (L.awq_oracle_unpack, L.awq_oracle_dequantize, L.awq_oracle_gemm, L.awq_oracle_num_threads):
            f.restype = ci
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dt(a: np.ndarray) -> str:
    return {np.dtype(np.float16): "f16", np.dtype(np.uint16): "bf16", np.dtype(np.float32): "f32"}[a.dtype]


def num_threads() -> int:
    return lib().awq_oracle_num_threads()


def unpack(packed: np.ndarray) -> np.ndarray:
    packed = np.ascontiguousarray(packed, dtype=np.int32)
    out = np.empty((packed.shape[0], packed.shape[1] * 8), dtype=np.uint8)
    rc = lib().awq_oracle_unpack(_p(packed), _p(out), packed.shape[0], packed.shape[1])
    if rc:
        raise ValueError(f"awq_oracle_unpack rc={rc}")
    return out


def dequantize(qweight, scales, qzeros, threads: int = 0) -> np.ndarray:
    qweight = np.ascontiguousarray(qweight, dtype=np.int32)
    qzeros = np.ascontiguousarray(qzeros, dtype=np.int32)
    scales = np.ascontiguousarray(scales)
    dt = _dt(scales)
    K, N = qweight.shape[0], qweight.shape[1] * 8
    g = K // scales.shape[0]
    out = np.empty((K, N), dtype=_NP[dt])
    rc = lib().awq_oracle_dequantize(_p(qweight), _p(scales), _p(qzeros), _p(out), K, N, g, _DT[dt], threads)
    if rc:
        raise ValueError(f"awq_oracle_dequantize rc={rc}")
    return out


def gemm(x, qweight, scales, qzeros, want_exact: bool = False, threads: int = 0):
    x = np.ascontiguousarray(x)
    qweight = np.ascontiguousarray(qweight, dtype=np.int32)
    qzeros = np.ascontiguousarray(qzeros, dtype=np.int32)
    scales = np.ascontiguousarray(scales)
    dt = _dt(scales)
    if _dt(x) != dt:
        raise TypeError("x and scales must share a dtype")
    M, K = x.shape
    N = qweight.shape[1] * 8
    g = K // scales.shape[0]
    y = np.empty((M, N), dtype=_NP[dt])
    ye = np.empty((M, N), dtype=np.float64) if want_exact else None
    rc = lib().awq_oracle_gemm(_p(x), _p(qweight), _p(scales), _p(qzeros), _p(y), _p(ye), M, K, N, g, _DT[dt], threads)
    if rc:
        raise ValueError(f"awq_oracle_gemm rc={rc}")
    return (y, ye) if want_exact else y
