"""ctypes wrappers of include/awq_aux.h: fused RMSNorm(+residual), RoPE + KV-cache write + attention, SiLU-and-mul for the
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


_attn_ws = {}
_attn_ws_retired = []


_ATTN_WS_MIN = 1 << 20


def _attention_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Scratch of the split-S attention (tickets + partials) per (device, stream): the tickets are live during a call, so two
    streams of one device never share them.  Zero-filled once, every call leaves the tickets zero.  Never created during graph
    capture (the memory would come from the graph's private pool and the zero-fill would be an un-run graph node): a miss while
    capturing raises — warm up on the capture stream or call ops.prepare_stream_workspaces() first."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sglang_awq_amd: split-S attention needs its per-stream scratch and the first call on this stream is "
                               "inside a graph capture; warm up on the capture stream or call ops.prepare_stream_workspaces(stream)")
        if ws is not None:
            _attn_ws_retired.append(ws)          # a captured graph may still point at it
        ws = torch.zeros(max(nbytes, _ATTN_WS_MIN), dtype=torch.uint8, device=dev)
        _attn_ws[key] = ws
    return ws


def prepare_attention_workspace(dev: torch.device, stream: int, nbytes: int = _ATTN_WS_MIN) -> None:
    key = (dev.index, stream)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _attn_ws_retired.append(ws)
        _attn_ws[key] = torch.zeros(max(nbytes, _ATTN_WS_MIN), dtype=torch.uint8, device=dev)


def decode_attention(qkv: torch.Tensor, pos: torch.Tensor, cos_table: torch.Tensor, sin_table: torch.Tensor, k_cache: torch.Tensor,
                     v_cache: torch.Tensor, num_heads: int, num_kv_heads: int, head_dim: int, num_splits: int = 1) -> torch.Tensor:
    """RoPE + KV-cache write + one-token attention in one launch; qkv [B, (Hq + 2 Hkv) D] is not modified.
    num_splits > 1 spreads each (sequence, head) over that many workgroups (long contexts at small batch).
    Returns [B, Hq * D]."""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and pos.dtype == torch.int64
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and cos_table.dtype == torch.float32
    lib = _lib.load()
    B = qkv.shape[0]
    out = torch.empty((B, num_heads * head_dim), dtype=torch.float16, device=qkv.device)
    ws, ws_bytes = None, 0
    if num_splits > 1:
        ws_bytes = lib.awq_aux_decode_attention_workspace_bytes(B, num_heads, head_dim, num_splits)
        ws = _attention_workspace(qkv.device, ws_bytes)
    rc = lib.awq_aux_decode_attention(_vp(qkv), _vp(pos), _vp(cos_table), _vp(sin_table), _vp(k_cache), _vp(v_cache), _vp(out),
                                      B, num_heads, num_kv_heads, head_dim, k_cache.shape[2], float(head_dim) ** -0.5,
                                      int(num_splits), _vp(ws), ws_bytes, _stream(qkv))
    _lib.check(rc, "awq_aux_decode_attention")
    return out


def argmax_advance(logits: torch.Tensor, tokens: torch.Tensor, pos: torch.Tensor) -> None:
    """tokens[b] = argmax(logits[b]) (first maximum), pos[b] += 1, in one launch (the greedy tail of a decode step)."""
    assert logits.dtype == torch.float16 and logits.dim() == 2 and logits.is_contiguous()
    assert tokens.dtype == torch.int64 and pos.dtype == torch.int64 and tokens.is_contiguous() and pos.is_contiguous()
    rc = _lib.load().awq_aux_argmax_advance(_vp(logits), _vp(tokens), _vp(pos), logits.shape[0], logits.shape[1], _stream(logits))
    _lib.check(rc, "awq_aux_argmax_advance")


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    assert gate_up.dtype == torch.float16 and gate_up.is_contiguous() and gate_up.dim() == 2
    inter = gate_up.shape[1] // 2
    act = torch.empty((gate_up.shape[0], inter), dtype=torch.float16, device=gate_up.device)
    rc = _lib.load().awq_aux_silu_mul(_vp(gate_up), _vp(act), gate_up.shape[0], inter, _stream(gate_up))
    _lib.check(rc, "awq_aux_silu_mul")
    return act


def interleave_gate_up(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor):
    """Column-permuted copies of a merged gate_up AWQ weight ([K, 2I] logical columns, gate then up) in which the
    16-column groups alternate gate / up — the order awq_aux_gemv_repacked_fused(silu_mul=1) expects of the tensors
    its `packed` argument was repacked from.  I % 16 == 0."""
    N = scales.shape[1]
    inter = N // 2
    if N % 32 or qweight.shape[1] * 8 != N:
        raise ValueError(f"interleave_gate_up: need 2I columns with I % 16 == 0, got N = {N}")
    grp = torch.arange(N // 16, device=scales.device)
    src = torch.where(grp % 2 == 0, grp // 2, inter // 16 + grp // 2)                 # source 16-column group of each output group
    cols = (src.view(-1, 1) * 16 + torch.arange(16, device=scales.device)).reshape(-1)
    words = (src.view(-1, 1) * 2 + torch.arange(2, device=scales.device)).reshape(-1)   # 8 logical columns per int32
    return qweight[:, words].contiguous(), scales[:, cols].contiguous(), qzeros[:, words].contiguous()


def gemv_repacked_fused(packed: torch.Tensor, K: int, N: int, group_size: int, x: Optional[torch.Tensor] = None,
                        norm: Optional[tuple] = None, silu_mul: bool = False):
    """Repacked decode GEMV with RMSNorm(+residual) prologue and / or SiLU-mul epilogue (include/awq_aux.h).
    norm = (h, delta, weight, eps): returns (y, h + delta); else x is the input and (y, None) is returned.
    Returns None when there is no fused kernel for the shape (callers run the separate ops)."""
    src = norm[0] if norm is not None else x
    assert src is not None and src.dtype == torch.float16 and src.dim() == 2 and src.is_contiguous() and src.shape[1] == K
    M = src.shape[0]
    dev = src.device
    from .ops import check_packed

    check_packed(packed, K, N, group_size, dev)
    y = torch.empty((M, N // 2 if silu_mul else N), dtype=torch.float16, device=dev)
    h_out = None
    if norm is not None:
        h, delta, w, eps = norm
        assert delta.shape == h.shape and delta.is_contiguous() and w.is_contiguous() and w.shape == (K,)
        h_out = torch.empty_like(h)
        args = (_vp(h), _vp(delta), _vp(w), _vp(h_out), float(eps))
    else:
        args = (None, None, None, None, 0.0)
    rc = _lib.load().awq_aux_gemv_repacked_fused(_vp(x), K, _vp(packed), _vp(y), M, K, N, group_size, _lib.DTYPE_F16, *args,
                                                 1 if silu_mul else 0, _stream(src))
    if rc == _lib.ERR_BAD_VARIANT:
        return None
    _lib.check(rc, "awq_aux_gemv_repacked_fused")
    return y, h_out
