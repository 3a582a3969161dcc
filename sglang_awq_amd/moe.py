"""AWQ mixture-of-experts method for MI355X.

Counterpart of `AWQMoEMethod` in the reference (python/sglang/srt/layers/quantization/awq.py:661-852): same parameter
names and shapes (`w13_qweight [E, K, 2I/8]`, `w2_qweight [E, I, K/8]`, `w13_scales [E, K/g, 2I]`, `w2_scales [E, I/g, K]`,
`w13_qzeros`, `w2_qzeros`), created by `create_weights`, consumed after `process_weights_after_loading`.  The reference
re-lays the experts out for NVIDIA's Marlin MoE kernels (`awq_marlin_moe_repack`, awq.py:760-815) and runs
`fused_marlin_moe`; on ROCm it has no native path ("HIP does not support fused_marlin_moe currently", awq.py:70) and AWQ
experts go through the Triton kernels of `moe_wna16.py`.  Here every expert gets the MFMA-fragment-major copy of the dense
path (`awq_repack`; w13 from gate / up column-interleaved tensors so SiLU·mul is the GEMV's epilogue), and a forward is

    decode-sized batches (tokens x top_k <= MOE_GEMV_MAX_SLOTS): two launches — `awq_aux_moe_gemv` over the (token, expert)
        pairs with the SiLU·mul epilogue, then again for w2 with the routed weight applied to the fp32 sums — and one sum
        over the top_k partial rows (graph-capturable: no host synchronisation, expert ids stay on the device);
    larger batches: tokens grouped by expert on the host (one synchronisation), one fused dense call per active expert.

Arithmetic (what the reference's fused MoE computes, fused_moe.py fused_experts_impl): per (token, expert) pair
act = fp16(silu(fp16 gate)) * fp16 up of the fp16-rounded w13 output, y = fp16(routed weight * fp32 sums of act @ W2),
output = fp16(sum over the token's pairs).  No fixture of the reference pins MoE outputs (its only AWQ-MoE tests are e2e
accuracy runs, test/srt/quant/test_awq.py:15-44): the tests compare against the oracle's dense linear composed in numpy —
**parity unpinned** beyond the dense path's pin.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .awq import AWQConfig


def select_experts(router_logits: torch.Tensor, top_k: int, renormalize: bool = True):
    """softmax -> top-k (-> renormalise), the default routing of the reference (layers/moe/topk.py fused_topk)."""
    probs = torch.softmax(router_logits.float(), dim=-1)
    topk_weights, topk_ids = torch.topk(probs, top_k, dim=-1)
    if renormalize:
        topk_weights = topk_weights / topk_weights.sum(dim=-1, keepdim=True)
    return topk_weights, topk_ids.to(torch.int32)


class AWQMoEMethod:
    """create_weights / process_weights_after_loading / apply for a layer of AWQ-quantised experts."""

    MOE_GEMV_MAX_SLOTS = 64          # (token, expert) pairs served by the one-row-per-slot launch; beyond: grouped by expert

    def __init__(self, quant_config: AWQConfig):
        if quant_config.weight_bits != 4:
            raise ValueError("AWQMoEMethod only supports 4bit now.")        # awq.py:665-666
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, num_experts: int, hidden_size: int, intermediate_size_per_partition: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        pf, g = self.quant_config.pack_factor, self.quant_config.group_size
        E, K, I = num_experts, hidden_size, intermediate_size_per_partition
        if K % g or I % g or (2 * I) % pf or K % pf:
            raise ValueError("The expert sizes are not aligned with the quantized weight shape.")

        def reg(name, *shape, dtype=torch.int32):
            p = torch.nn.Parameter(torch.empty(*shape, dtype=dtype), requires_grad=False)
            for k, v in extra_weight_attrs.items():
                setattr(p, k, v)
            layer.register_parameter(name, p)

        reg("w13_qweight", E, K, 2 * I // pf)
        reg("w2_qweight", E, I, K // pf)
        reg("w13_scales", E, K // g, 2 * I, dtype=params_dtype)
        reg("w2_scales", E, I // g, K, dtype=params_dtype)
        reg("w13_qzeros", E, K // g, 2 * I // pf)
        reg("w2_qzeros", E, I // g, K // pf)
        layer.num_experts, layer.hidden_size, layer.intermediate_size_per_partition = E, K, I

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        from . import aux_ops, ops

        E, K, I = layer.num_experts, layer.hidden_size, layer.intermediate_size_per_partition
        g = self.quant_config.group_size
        if not layer.w13_qweight.is_cuda or layer.w13_scales.dtype != torch.float16:
            raise NotImplementedError("AWQMoEMethod: fp16 experts on a HIP device only")
        lib = _lib.load()
        b13, b2 = lib.awq_repacked_bytes(K, 2 * I, g, _lib.DTYPE_F16), lib.awq_repacked_bytes(I, K, g, _lib.DTYPE_F16)
        if b13 == 0 or b2 == 0 or g % 128:
            raise NotImplementedError(f"AWQMoEMethod: no fragment-major layout for K={K} I={I} group_size={g} (fp16, multiples of 128)")
        dev = layer.w13_qweight.device
        layer.w13_packed = torch.empty((E, b13), dtype=torch.uint8, device=dev)
        layer.w2_packed = torch.empty((E, b2), dtype=torch.uint8, device=dev)
        for e in range(E):
            il = aux_ops.interleave_gate_up(layer.w13_qweight[e], layer.w13_scales[e], layer.w13_qzeros[e])
            layer.w13_packed[e].copy_(ops.awq_repack(*il))
            layer.w2_packed[e].copy_(ops.awq_repack(layer.w2_qweight[e].contiguous(), layer.w2_scales[e].contiguous(), layer.w2_qzeros[e].contiguous()))

    def _moe_gemv(self, x, packed, expert_ids, slot_scale, slots, x_div, K, N, silu):
        g = self.quant_config.group_size
        y = torch.empty((slots, N // 2 if silu else N), dtype=torch.float16, device=x.device)
        rc = _lib.load().awq_aux_moe_gemv(ctypes.c_void_p(x.data_ptr()), x.stride(0), int(x_div), ctypes.c_void_p(packed.data_ptr()),
                                          packed.stride(0), ctypes.c_void_p(expert_ids.data_ptr()),
                                          ctypes.c_void_p(slot_scale.data_ptr()) if slot_scale is not None else None,
                                          ctypes.c_void_p(y.data_ptr()), slots, K, N, g, _lib.DTYPE_F16, 1 if silu else 0,
                                          ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc == _lib.ERR_BAD_VARIANT:
            return None
        _lib.check(rc, "awq_aux_moe_gemv")
        return y

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        """x [T, K] fp16, topk_weights [T, top_k] fp32, topk_ids [T, top_k] int32 -> [T, K]."""
        from . import aux_ops, ops

        E, K, I = layer.num_experts, layer.hidden_size, layer.intermediate_size_per_partition
        g = self.quant_config.group_size
        if x.dim() != 2 or x.shape[1] != K or x.dtype != torch.float16:
            raise RuntimeError(f"AWQMoEMethod.apply: x must be fp16 [tokens, {K}]")
        T, top_k = topk_ids.shape
        x = x.contiguous()
        ids = topk_ids.to(torch.int32).contiguous().view(-1)
        wts = topk_weights.to(torch.float32).contiguous().view(-1)
        slots = T * top_k
        if 0 < slots <= self.MOE_GEMV_MAX_SLOTS:
            act = self._moe_gemv(x, layer.w13_packed, ids, None, slots, top_k, K, 2 * I, True)
            if act is not None:
                y = self._moe_gemv(act, layer.w2_packed, ids, wts, slots, 1, I, K, False)
                if y is not None:
                    return y.view(T, top_k, K).sum(dim=1, dtype=torch.float32).to(torch.float16)
        # grouped by expert (host-side grouping: one synchronisation; prefill-sized batches)
        out = torch.zeros((T, K), dtype=torch.float32, device=x.device)
        order = torch.argsort(ids, stable=True)
        counts = torch.bincount(ids, minlength=E).tolist()
        tok = (order // top_k)
        start = 0
        for e, n in enumerate(counts):
            if n == 0:
                continue
            sel = order[start:start + n]
            rows = tok[start:start + n]
            xe = x.index_select(0, rows)
            r = aux_ops.gemv_repacked_fused(layer.w13_packed[e], K, 2 * I, g, x=xe, silu_mul=True) if n <= 32 else None
            if r is not None:
                act = r[0]
            else:
                gu = ops.awq_gemm_repacked(xe, self._plain_w13(layer, e), K, 2 * I, g)
                act = aux_ops.silu_mul(gu)
            ye = ops.awq_gemm_repacked(act, layer.w2_packed[e], I, K, g)
            # (the dense kernels round each pair's sum to fp16 before the routed weight is applied, the decode route after: the
            # two routes can differ by an fp16 ulp of a pair's contribution)
            out.index_add_(0, rows, ye.float() * wts.index_select(0, sel).unsqueeze(1))
            start += n
        return out.to(torch.float16)

    def _plain_w13(self, layer, e):
        """Repacked copy of expert e's w13 in natural (gate | up) column order, made on first use (large batches only)."""
        from . import ops

        cache = layer.__dict__.setdefault("_w13_plain_packed", {})
        if e not in cache:
            cache[e] = ops.awq_repack(layer.w13_qweight[e].contiguous(), layer.w13_scales[e].contiguous(), layer.w13_qzeros[e].contiguous())
        return cache[e]
