"""Shared helpers for the parity tests (numpy <-> torch, tolerances)."""
import numpy as np
import torch

from oracle import awq_ref

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def to_torch(a: np.ndarray, device="cpu") -> torch.Tensor:
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(device)
    return torch.from_numpy(np.ascontiguousarray(a).copy()).to(device)


def to_np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint8)


def ulp(v: np.ndarray, dt: str) -> np.ndarray:
    """Spacing of the storage dtype at |v| (normal range floor at the smallest normal)."""
    mant = {"f16": 10, "bf16": 7, "f32": 23}[dt]
    emin = {"f16": -14, "bf16": -126, "f32": -126}[dt]
    e = np.floor(np.log2(np.maximum(np.abs(v), 2.0 ** emin)))
    return 2.0 ** (e - mant)


def assert_gemm_close(y: np.ndarray, exact: np.ndarray, dt: str, atol: float = 1e-3, what: str = ""):
    """y (storage dtype) against the oracle's un-rounded float64 value: a correctly rounded result of
    an fp32-accumulated sum is within half an output ulp of the exact value plus the accumulation
    error; the north star allows 1e-3 absolute for the latter (it is ~1e-5 in practice)."""
    got = awq_ref.to_f64(y, dt)
    err = np.abs(got - exact)
    bound = 0.5 * ulp(exact, dt) * (1 + 1e-6) + atol
    bad = err > bound
    assert not bad.any(), f"{what}: {bad.sum()} elements off, worst {err.max():.3e} (bound there {bound.flat[err.argmax()]:.3e})"
    if dt == "f32":
        # an fp32 sum of fp32 products cannot be the correctly rounded exact sum; bound it instead
        assert np.all(err <= 1e-5 * np.maximum(np.abs(exact), 1.0) + 1e-5), f"{what}: fp32 result too far from the exact sum ({err.max():.3e})"
        return
    # and nearly all of them must be the correctly rounded value itself
    rounded = awq_ref.to_f64(awq_ref.from_f64(exact, dt), dt)
    frac = float((got != rounded).mean())
    assert frac < 0.02, f"{what}: {frac:.4f} of outputs differ from the correctly rounded exact sum"
