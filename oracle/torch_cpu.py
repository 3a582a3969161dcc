"""PyTorch-CPU leg of the oracle: the baseline that bench.py times beside the GPU path.

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/awq_ref.py header).  The reference's CPU-runnable
form of this path is eager PyTorch: `awq_dequantize_decomposition` (awq_triton.py:342-368) followed
by `torch.matmul` — the two steps of AWQLinearMethod.apply (awq.py:446-447).  This file restates
those two steps with torch CPU ops so `cpu_baseline` measures the same kind of work on the GPU
box's host cores, and provides torch<->numpy bridges (bf16 travels as uint16 bit patterns).
"""
from __future__ import annotations

import numpy as np
import torch

# 4 * nibble index of logical column j (order [0,4,1,5,2,6,3,7], awq_triton.py:351)
_SHIFT_OF_COL = (0, 16, 4, 20, 8, 24, 12, 28)


def to_np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def from_np(a: np.ndarray, bf16: bool = False) -> torch.Tensor:
    if a.dtype == np.uint16 or bf16:
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(np.ascontiguousarray(a).copy())


def dequantize_cpu(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """Eager-torch dequantise on CPU tensors: unpack both packed tensors nibble by nibble in the
    AWQ column order, subtract per group, multiply by the group scale in the scale dtype."""
    K, C = qweight.shape
    G = scales.shape[0]
    g = K // G
    shifts = torch.tensor(_SHIFT_OF_COL, dtype=torch.int32)
    q = torch.bitwise_and(torch.bitwise_right_shift(qweight.unsqueeze(-1), shifts), 0xF)
    z = torch.bitwise_and(torch.bitwise_right_shift(qzeros.unsqueeze(-1), shifts), 0xF)
    q = q.reshape(G, g, C * 8).to(scales.dtype)
    z = z.reshape(G, 1, C * 8).to(scales.dtype)
    return ((q - z) * scales.unsqueeze(1)).reshape(K, C * 8)


def linear_cpu(x: torch.Tensor, qweight, scales, qzeros, bias=None) -> torch.Tensor:
    """dequantise -> matmul -> add_(bias) -> reshape, on CPU (awq.py:434-451)."""
    out_shape = x.shape[:-1] + (qweight.shape[-1] * 8,)
    out = torch.matmul(x.reshape(-1, x.shape[-1]), dequantize_cpu(qweight, scales, qzeros))
    if bias is not None:
        out.add_(bias)
    return out.reshape(out_shape)
