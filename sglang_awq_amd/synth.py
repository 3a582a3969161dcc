"""Deterministic synthetic AWQ tensors (numpy only, counter-based, version-independent).

The reference's tests draw inputs with torch's RNG (test/srt/quant/test_awq_dequant.py:83-104);
digests committed under tests/golden/ must be reproducible on the GPU box whatever torch/numpy
version generated them, so inputs come from a splitmix64 counter hash instead.

Families:
  "R"  reference-test distribution: packed words in [0, int32.max) (bit 31 clear), scales and
       activations uniform [0, 1)
  "F"  full-range packed words (bit 31 varies), otherwise as R
  "A"  AWQ-realistic: nibbles uniform 0..15, scales 0.005 + 0.015*u, activations ~ N(0, 1) * x_std
"""
from __future__ import annotations

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_seed(seed: int, stream: int) -> int:
    return (seed * 0x100000001B3 + stream * 0xD6E8FEB86659FD93 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF


def rand_u32(shape, seed: int, stream: int) -> np.ndarray:
    n = int(np.prod(shape))
    h = splitmix64(np.arange(n, dtype=np.uint64), _stream_seed(seed, stream))
    return (h >> np.uint64(32)).astype(np.uint32).reshape(shape)


def rand_uniform(shape, seed: int, stream: int) -> np.ndarray:
    """float64 uniform in [0, 1) with 24 random bits."""
    return (rand_u32(shape, seed, stream) >> np.uint32(8)).astype(np.float64) * (1.0 / (1 << 24))


def rand_normal(shape, seed: int, stream: int) -> np.ndarray:
    u1 = 1.0 - rand_uniform(shape, seed, stream)
    u2 = rand_uniform(shape, seed, stream + 1000003)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def cast(a: np.ndarray, dtype: str) -> np.ndarray:
    """float64 -> storage array ("f16" np.float16 | "bf16" np.uint16 bits | "f32" np.float32)."""
    if dtype == "f16":
        return a.astype(np.float16)
    if dtype == "f32":
        return a.astype(np.float32)
    if dtype == "bf16":
        return _f32_to_bf16_bits(a.astype(np.float32))
    raise ValueError(dtype)


def make_awq_weights(K: int, N: int, g: int, dtype: str = "f16", family: str = "A", seed: int = 1234):
    """(qweight int32 [K, N/8], scales [K/g, N], qzeros int32 [K/g, N/8])."""
    if g in (-1, 0):
        g = K
    if N % 8 or K % g:
        raise ValueError("need N % 8 == 0 and K % g == 0")
    qw = rand_u32((K, N // 8), seed, 1)
    qz = rand_u32((K // g, N // 8), seed, 2)
    u = rand_uniform((K // g, N), seed, 3)
    if family == "R":
        qw, qz = qw >> np.uint32(1), qz >> np.uint32(1)
        s = u
    elif family == "F":
        s = u
    elif family == "A":
        s = 0.005 + 0.015 * u
    else:
        raise ValueError(family)
    return qw.view(np.int32), cast(s, dtype), qz.view(np.int32)


def make_activations(M: int, K: int, dtype: str = "f16", family: str = "A", seed: int = 1234, x_std: float = 1.0):
    if family in ("R", "F"):
        return cast(rand_uniform((M, K), seed, 4), dtype)
    return cast(rand_normal((M, K), seed, 5) * x_std, dtype)


def make_bias(N: int, dtype: str = "f16", seed: int = 1234) -> np.ndarray:
    return cast(rand_normal((N,), seed, 7) * 0.1, dtype)
