"""AWQ mixture-of-experts method for MI355X.

Counterpart of `AWQMoEMethod` in the reference (python/sglang/srt/layers/quantization/awq.py:661-852): same parameter
names and shapes (`w13_qweight [E, K, 2I/8]`, `w2_qweight [E, I, K/8]`, `w13_scales [E, K/g, 2I]`, `w2_scales [E, I/g, K]`,
`w13_qzeros`, `w2_qzeros`), created by `create_weights`, consumed after `process_weights_after_loading`, and the same call
interface: `create_moe_runner(layer, moe_runner_config)` then `apply(layer, dispatch_output) -> CombineInput`
(awq.py:815-845), with `dispatch_output.hidden_states` / `dispatch_output.topk_output.{topk_weights, topk_ids}` read by
attribute (StandardDispatchOutput / StandardTopKOutput, layers/moe/token_dispatcher/standard.py:53-61, layers/moe/topk.py:161-166)
and a `StandardCombineInput`-shaped result (`.hidden_states`, standard.py:68-76) — so the reference's `FusedMoE` layer can call
it.  The reference re-lays the experts out for NVIDIA's Marlin MoE kernels (`awq_marlin_moe_repack`, awq.py:760-815) and runs
`fused_marlin_moe`; on ROCm it has no native path ("HIP does not support fused_marlin_moe currently", awq.py:70) and AWQ
experts go through the Triton kernels of `moe_wna16.py` after converting the AWQ nibble order (moe_wna16.py:390-412).  Here
every expert gets the MFMA-fragment-major copy of the dense path (`awq_repack` reads the AWQ order directly; w13 from gate / up
column-interleaved tensors so SiLU·mul is the GEMV's epilogue), and a forward is, without any host synchronisation
(graph-capturable at every size):

    few pairs (tokens x top_k <= max(MOE_SLOT_MAX_PAIRS, E / 2)): two launches of `awq_aux_moe_gemv` over the (token, expert) pairs, one
        grid row per pair — w13 with the SiLU·mul epilogue, w2 with the routed weight applied to the fp32 sums;
    more pairs: the pairs are sorted by expert ON THE DEVICE and cut into 16-row blocks that never straddle two experts (the
        reference's moe_align_block_size step), then two launches of `awq_aux_moe_gemv_blocks` — one expert stream per 16 rows
        instead of one per pair, same epilogues, same rounding points;
    prefill-sized batches (more than 16 rows per expert on average): the same with 64- or 128-row blocks on the MFMA tile kernel of the
        dense prefill path (`awq_aux_moe_gemm_blocks`);
    then one sum over each token's top_k rows.  Ids outside [0, E) (the reference marks the padded tokens of a graph batch
    with -1, layers/moe/topk.py:705-712) contribute zero.

Arithmetic (what the reference's fused MoE computes, fused_moe.py fused_experts_impl): per (token, expert) pair
act = fp16(silu(fp16 gate)) * fp16 up of the fp16-rounded w13 output, y = fp16(routed weight * fp32 sums of act @ W2),
output = fp16(sum over the token's pairs) (* routed_scaling_factor if the runner config carries one).  No fixture of the
reference pins MoE outputs (its only AWQ-MoE tests are e2e accuracy runs, test/srt/quant/test_awq.py:15-44): the tests compare
against the oracle's dense linear composed in numpy — **parity unpinned** beyond the dense path's pin.
"""
from __future__ import annotations

import ctypes
from typing import NamedTuple, Optional

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


class StandardCombineInput(NamedTuple):
    """What `apply` returns: the reference's StandardCombineInput (layers/moe/token_dispatcher/standard.py:68-76)."""

    hidden_states: torch.Tensor

    @property
    def format(self) -> str:
        return "standard"


class AWQMoEMethod:
    """create_weights / process_weights_after_loading / create_moe_runner / apply for a layer of AWQ-quantised experts."""

    # (token, expert) pairs served one grid row each; beyond: expert-sorted 16-row blocks.  Measured on Mixtral-8x7B-like experts
    # (E = 8, K = 4096, I = 14336, top-2; profiles/r03_time_moe.txt): a pair costs ~16-17 us on the slot route, the block route ~215 us
    # for up to 16 rows per expert: they cross near 13 pairs
    MOE_SLOT_MAX_PAIRS = 12
    MOE_GEMV_MAX_SLOTS = MOE_SLOT_MAX_PAIRS   # (name of rounds 1-2)

    # Average rows per expert from which the MFMA tile route replaces the 16-row block route: a 16-row block streams its expert once at
    # GEMV speed, so up to 16 rows per expert the block route costs one stream per active expert and wins or ties (Mixtral-like: 215 us
    # against ~340; DeepSeek-V3-like: 1.7 ms either way); from 17 rows it pays a second stream while a 64-row tile does not
    # (profiles/r03_time_moe_tile_route.txt).
    TILE_ROUTE_MIN_ROWS_PER_EXPERT = 16
    # Below this average load the tiles are 64 rows tall (less padding in every expert's last tile, two workgroups per CU), above it 128
    # (Mixtral-like experts at 1024 rows each: 3662 vs 3755 us; at 512 rows a tie; DeepSeek-V3-like at 512 rows: 64-row tiles 10 % faster)
    TILE_ROUTE_WIDE_ROWS_PER_EXPERT = 768

    @classmethod
    def slot_route_max_pairs(cls, num_experts: int) -> int:
        """Most (token, expert) pairs the slot route serves for a layer of `num_experts` experts.  With many experts a batch's pairs mostly
        hit DIFFERENT experts (up to E / 2 pairs at least 79 % of them are distinct in expectation), so grouping them saves little
        traffic and the slot route's one-row stream is faster per byte.  Measured on DeepSeek-V3-like routed experts (E = 256,
        K = 7168, I = 2048, top-8; profiles/r03_time_moe_deepseek_v3_shapes.txt): 16 / 32 / 64 / 128 pairs 89 / 165 / 298 / 568 us
        against 138 / 231 / 384 / 620 on the block route; 256 / 512 pairs 1112 / 2203 against 969 / 1330."""
        return max(cls.MOE_SLOT_MAX_PAIRS, int(num_experts) // 2)

    def __init__(self, quant_config: AWQConfig):
        if quant_config.weight_bits != 4:
            raise ValueError("AWQMoEMethod only supports 4bit now.")        # awq.py:665-666
        self.quant_config = quant_config
        self.moe_runner_config = None

    def create_moe_runner(self, layer: torch.nn.Module, moe_runner_config) -> None:
        """awq.py:815-820.  The config is read by attribute (MoeRunnerConfig, layers/moe/moe_runner/base.py:26-46); what this
        method cannot honour raises here rather than computing something else."""
        act = getattr(moe_runner_config, "activation", "silu")
        if act != "silu" or not getattr(moe_runner_config, "is_gated", True):
            raise NotImplementedError(f"AWQMoEMethod: gated SiLU experts only (activation={act!r})")
        if getattr(moe_runner_config, "apply_router_weight_on_input", False):
            raise NotImplementedError("AWQMoEMethod: apply_router_weight_on_input is not supported")
        if getattr(moe_runner_config, "no_combine", False):
            raise NotImplementedError("AWQMoEMethod: no_combine is not supported")
        self.moe_runner_config = moe_runner_config

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

    def _moe_gemv(self, x, packed, expert_ids, slot_scale, slots, x_div, K, N, silu, num_experts):
        g = self.quant_config.group_size
        y = torch.empty((slots, N // 2 if silu else N), dtype=torch.float16, device=x.device)
        rc = _lib.load().awq_aux_moe_gemv(ctypes.c_void_p(x.data_ptr()), x.stride(0), int(x_div), ctypes.c_void_p(packed.data_ptr()),
                                          packed.stride(0), int(num_experts), ctypes.c_void_p(expert_ids.data_ptr()),
                                          ctypes.c_void_p(slot_scale.data_ptr()) if slot_scale is not None else None,
                                          ctypes.c_void_p(y.data_ptr()), slots, K, N, g, _lib.DTYPE_F16, 1 if silu else 0,
                                          ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc == _lib.ERR_BAD_VARIANT:
            return None
        _lib.check(rc, "awq_aux_moe_gemv")
        return y

    def _moe_blocks(self, x, packed, row_map, block_expert, slot_scale, pairs, x_div, K, N, silu, tiles=0):
        g = self.quant_config.group_size
        # rows of padded / dropped pairs are never written by the kernel: the combine launch skips them by expert id (no zero-fill launch);
        # only the tensor-op fallback of the combine (K % 8 != 0) needs zeros
        alloc = torch.empty if K % 8 == 0 and N % 8 == 0 else torch.zeros
        y = alloc((pairs, N // 2 if silu else N), dtype=torch.float16, device=x.device)
        # tiles = 64 / 128: rows per block on the MFMA tile kernel; 0: 16-row blocks on the GEMV
        rows_arg = (int(tiles),) if tiles else ()
        fn = _lib.load().awq_aux_moe_gemm_blocks if tiles else _lib.load().awq_aux_moe_gemv_blocks
        rc = fn(ctypes.c_void_p(x.data_ptr()), x.stride(0), int(x_div), ctypes.c_void_p(packed.data_ptr()),
                                                 packed.stride(0), ctypes.c_void_p(row_map.data_ptr()),
                                                 ctypes.c_void_p(block_expert.data_ptr()), block_expert.numel(), *rows_arg,
                                                 ctypes.c_void_p(slot_scale.data_ptr()) if slot_scale is not None else None,
                                                 ctypes.c_void_p(y.data_ptr()), K, N, g, _lib.DTYPE_F16, 1 if silu else 0,
                                                 ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc == _lib.ERR_BAD_VARIANT:
            return None
        _lib.check(rc, "awq_aux_moe_gemm_blocks" if tiles else "awq_aux_moe_gemv_blocks")
        return y

    @staticmethod
    def align_blocks(ids: torch.Tensor, num_experts: int, block: int = 16):
        """Device-side counterpart of the reference's moe_align_block_size (layers/moe/fused_moe_triton/moe_align_block_size.py):
        ids [P] int32 -> (row_map [B * block] int32: pair indices sorted by expert, each expert's run padded with -1 to a multiple
        of `block`; block_expert [B] int32: expert of each block, -1 for unused blocks).  B = ceil(P / block) + num_experts is a
        static bound, so nothing here depends on the data on the host: no synchronisation, capturable.  Ids outside
        [0, num_experts) (padded tokens carry -1) are dropped."""
        P = ids.numel()
        dev = ids.device
        E = num_experts
        B = (P + block - 1) // block + E
        valid = (ids >= 0) & (ids < E)
        key = torch.where(valid, ids, torch.full_like(ids, E)).to(torch.int64)          # dropped pairs sort behind every expert
        order = torch.argsort(key, stable=True)                                          # pair indices, grouped by expert
        sorted_key = key[order]
        counts = torch.zeros(E + 1, dtype=torch.int64, device=dev).scatter_add_(0, key, torch.ones_like(key))
        padded = (counts[:E] + (block - 1)) // block * block
        starts = torch.cumsum(padded, 0) - padded                                        # first row of each expert in row_map
        first = torch.cumsum(counts, 0) - counts                                         # first position of each key in `order`
        rank = torch.arange(P, device=dev, dtype=torch.int64) - first[sorted_key]        # rank of a pair inside its expert's run
        dest = torch.where(sorted_key < E, starts[sorted_key.clamp(max=E - 1)] + rank, torch.full_like(rank, B * block))
        row_map = torch.full((B * block + 1,), -1, dtype=torch.int32, device=dev)
        row_map.scatter_(0, dest, order.to(torch.int32))                                 # dropped pairs land in the spare last slot
        row_map = row_map[:B * block].contiguous()
        blk_start = torch.arange(B, device=dev, dtype=torch.int64) * block
        ends = starts + padded
        e_of = torch.searchsorted(ends, blk_start, right=True)                           # expert whose padded run holds the block
        block_expert = torch.where(blk_start < ends[-1], e_of, torch.full_like(e_of, -1)).to(torch.int32)
        return row_map, block_expert

    @classmethod
    def align_blocks_device(cls, ids: torch.Tensor, num_experts: int, block: int = 16):
        """align_blocks as one small launch (`awq_aux_moe_align_blocks`: counting sort through LDS, one workgroup) instead of a dozen
        tensor ops (~100 us of launch boundaries per MoE layer, profiles/r03_time_moe.txt); the same contract except that the order of
        the rows inside an expert's run is not specified.  More than 1024 experts: the tensor-op form."""
        if not ids.is_cuda or num_experts > 1024:
            return cls.align_blocks(ids, num_experts, block)
        P = ids.numel()
        B = (P + block - 1) // block + num_experts
        row_map = torch.empty(B * block, dtype=torch.int32, device=ids.device)
        block_expert = torch.empty(B, dtype=torch.int32, device=ids.device)
        rc = _lib.load().awq_aux_moe_align_blocks_n(ctypes.c_void_p(ids.data_ptr()), P, num_experts, int(block),
                                                    ctypes.c_void_p(row_map.data_ptr()), ctypes.c_void_p(block_expert.data_ptr()), B,
                                                    ctypes.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream))
        _lib.check(rc, "awq_aux_moe_align_blocks_n")
        return row_map, block_expert

    def apply(self, layer: torch.nn.Module, dispatch_output, topk_weights: Optional[torch.Tensor] = None,
              topk_ids: Optional[torch.Tensor] = None):
        """The reference's call: apply(layer, dispatch_output) -> CombineInput (awq.py:822-845).  For direct use the tensors may be
        given instead: apply(layer, x, topk_weights, topk_ids) -> Tensor."""
        if isinstance(dispatch_output, torch.Tensor):
            return self.apply_tensors(layer, dispatch_output, topk_weights, topk_ids)
        x = dispatch_output.hidden_states
        topk = dispatch_output.topk_output
        out = self.apply_tensors(layer, x, topk.topk_weights, topk.topk_ids)
        cfg = self.moe_runner_config
        rsf = getattr(cfg, "routed_scaling_factor", None) if cfg is not None else None
        if rsf is not None and rsf != 1.0:
            out = out * rsf                                   # fused_experts_impl multiplies the combined output (moe_sum_reduce)
        if cfg is not None and getattr(cfg, "inplace", False) and out.shape == x.shape and out.dtype == x.dtype:
            x.copy_(out)
            out = x
        return StandardCombineInput(hidden_states=out)

    def apply_tensors(self, layer: torch.nn.Module, x: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        """x [T, K] fp16, topk_weights [T, top_k] fp32, topk_ids [T, top_k] int (ids outside [0, E) = padded) -> [T, K]."""
        E, K, I = layer.num_experts, layer.hidden_size, layer.intermediate_size_per_partition
        if x.dim() != 2 or x.shape[1] != K or x.dtype != torch.float16:
            raise RuntimeError(f"AWQMoEMethod.apply: x must be fp16 [tokens, {K}]")
        T, top_k = topk_ids.shape
        if T == 0:
            return x.new_empty((0, K))
        x = x.contiguous()
        ids = topk_ids.to(torch.int32).contiguous().view(-1)
        wts = topk_weights.to(torch.float32).contiguous().view(-1)
        pairs = T * top_k
        y = None
        if pairs <= self.slot_route_max_pairs(E):
            act = self._moe_gemv(x, layer.w13_packed, ids, None, pairs, top_k, K, 2 * I, True, E)
            if act is not None:
                y = self._moe_gemv(act, layer.w2_packed, ids, wts, pairs, 1, I, K, False, E)
        if y is None and pairs >= self.TILE_ROUTE_MIN_ROWS_PER_EXPERT * E:
            # prefill-sized batch: 64- or 128-row blocks on the MFMA tile kernel (each active expert streamed once per tile of its rows)
            rows = 64 if pairs < self.TILE_ROUTE_WIDE_ROWS_PER_EXPERT * E else 128
            row_map, block_expert = self.align_blocks_device(ids, E, rows)
            act = self._moe_blocks(x, layer.w13_packed, row_map, block_expert, None, pairs, top_k, K, 2 * I, True, tiles=rows)
            if act is not None:
                y = self._moe_blocks(act, layer.w2_packed, row_map, block_expert, wts, pairs, 1, I, K, False, tiles=rows)
        if y is None:
            row_map, block_expert = self.align_blocks_device(ids, E)
            act = self._moe_blocks(x, layer.w13_packed, row_map, block_expert, None, pairs, top_k, K, 2 * I, True)
            if act is None:
                raise NotImplementedError(f"AWQMoEMethod: no kernel for experts K={K} I={I} group_size={self.quant_config.group_size}")
            y = self._moe_blocks(act, layer.w2_packed, row_map, block_expert, wts, pairs, 1, I, K, False)
            if y is None:
                raise NotImplementedError(f"AWQMoEMethod: no kernel for experts K={K} I={I} group_size={self.quant_config.group_size}")
        if K % 8 == 0:                                        # one launch: fp32 sum over each token's pairs, one rounding
            out = torch.empty((T, K), dtype=torch.float16, device=x.device)
            rc = _lib.load().awq_aux_moe_sum(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(out.data_ptr()), T, top_k, K,
                                             ctypes.c_void_p(ids.data_ptr()), E,
                                             ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            _lib.check(rc, "awq_aux_moe_sum")
            return out
        return y.view(T, top_k, K).sum(dim=1, dtype=torch.float32).to(torch.float16)
