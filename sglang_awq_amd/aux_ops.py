"""ctypes wrappers of include/awq_aux.h: fused RMSNorm(+residual), RoPE + KV-cache write, SiLU-and-mul for the
decode harness (sglang_awq_amd/llama.py).  fp16 HIP tensors, launch-only (graph-capturable)."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib


def _vp(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def add_rmsnorm(h: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, eps: float) -> torch.Tensor:
    """h += delta (in place, if delta is given); returns rmsnorm(h) * weight."""
    assert h.dtype == torch.float16 and h.is_contiguous() and h.dim() == 2
    out = torch.empty_like(h)
    rc = _lib.load().awq_aux_add_rmsnorm(_vp(h), _vp(delta), _vp(weight), _vp(out), h.shape[0], h.shape[1], float(eps), _stream(h))
    _lib.check(rc, "awq_aux_add_rmsnorm")
    return out


def rope_kv(qkv: torch.Tensor, pos: torch.Tensor, cos_table: torch.Tensor, sin_table: torch.Tensor, k_cache: torch.Tensor,
            v_cache: torch.Tensor, num_heads: int, num_kv_heads: int, head_dim: int) -> None:
    """Rotate q / k of qkv [B, (Hq + 2 Hkv) D] in place and write this token's k, v into the caches [Bmax, Hkv, S, D]."""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and pos.dtype == torch.int64
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and cos_table.dtype == torch.float32
    rc = _lib.load().awq_aux_rope_kv(_vp(qkv), _vp(pos), _vp(cos_table), _vp(sin_table), _vp(k_cache), _vp(v_cache), qkv.shape[0],
                                     num_heads, num_kv_heads, head_dim, k_cache.shape[2], _stream(qkv))
    _lib.check(rc, "awq_aux_rope_kv")


def decode_attention(qkv: torch.Tensor, pos: torch.Tensor, cos_table: torch.Tensor, sin_table: torch.Tensor, k_cache: torch.Tensor,
                     v_cache: torch.Tensor, num_heads: int, num_kv_heads: int, head_dim: int) -> torch.Tensor:
    """RoPE + KV-cache write + one-token attention in one launch; qkv [B, (Hq + 2 Hkv) D] is not modified.
    Returns [B, Hq * D]."""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and pos.dtype == torch.int64
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and cos_table.dtype == torch.float32
    out = torch.empty((qkv.shape[0], num_heads * head_dim), dtype=torch.float16, device=qkv.device)
    rc = _lib.load().awq_aux_decode_attention(_vp(qkv), _vp(pos), _vp(cos_table), _vp(sin_table), _vp(k_cache), _vp(v_cache), _vp(out),
                                              qkv.shape[0], num_heads, num_kv_heads, head_dim, k_cache.shape[2],
                                              float(head_dim) ** -0.5, _stream(qkv))
    _lib.check(rc, "awq_aux_decode_attention")
    return out


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    assert gate_up.dtype == torch.float16 and gate_up.is_contiguous() and gate_up.dim() == 2
    inter = gate_up.shape[1] // 2
    act = torch.empty((gate_up.shape[0], inter), dtype=torch.float16, device=gate_up.device)
    rc = _lib.load().awq_aux_silu_mul(_vp(gate_up), _vp(act), gate_up.shape[0], inter, _stream(gate_up))
    _lib.check(rc, "awq_aux_silu_mul")
    return act
