"""Minimal Llama decode harness around the AWQ linears (BASELINE configs 4-5).

Only the wiring either side of the hot path: the four quantised projections of every layer are this
package's `QKVParallelLinear` / `RowParallelLinear` / `MergedColumnParallelLinear` with
`AWQLinearMethod` — exactly the layers `LlamaAttention` / `LlamaMLP` build in the reference
(python/sglang/srt/models/llama.py:61-200).  Everything else (RMSNorm, neox RoPE, attention over a
static KV cache, SiLU-and-mul, greedy sampling) is plain PyTorch-ROCm ops: no AITER, no Triton, no
serving runtime.  The whole decode step is launch-only, so it is captured once into a HIP graph and
replayed (the reference's decode path replays graphs: model_executor/model_runner.py:2765-2771).

Weights are synthetic (random packed int4, realistic scales): there is no checkpoint in this
environment; the reference measures dummy-weight decode the same way (`--load-format dummy`,
weight_utils.py:1108-1137).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F

from .awq import AWQConfig
from .distributed import get_tensor_model_parallel_world_size
from .linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear


@dataclass
class LlamaConfig:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    qkv_bias: bool = False          # Qwen2-style checkpoints carry q/k/v biases; HF Llama's `attention_bias` sets qkv + o
    o_bias: bool = False
    mlp_bias: bool = False

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @staticmethod
    def llama2_7b() -> "LlamaConfig":
        return LlamaConfig()

    @staticmethod
    def llama2_70b() -> "LlamaConfig":
        return LlamaConfig(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64,
                           num_key_value_heads=8)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype) * weight


def apply_rope_neox(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, H, D]; cos / sin [B, 1, D/2] (rotate-half convention, as get_rope(..., is_neox_style=True))."""
    d = x.shape[-1] // 2
    x1, x2 = x[..., :d].float(), x[..., d:].float()
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).to(x.dtype)


class LlamaDecoderLayer(torch.nn.Module):
    def __init__(self, cfg: LlamaConfig, quant: AWQConfig, layer_id: int, max_batch: int, max_seq: int, dtype: torch.dtype):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.cfg = cfg
        self.num_heads = cfg.num_attention_heads // tp
        self.num_kv_heads = max(1, cfg.num_key_value_heads // tp)
        self.head_dim = cfg.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        p = f"model.layers.{layer_id}"
        self.qkv_proj = QKVParallelLinear(cfg.hidden_size, self.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads,
                                          bias=cfg.qkv_bias, quant_config=quant, params_dtype=dtype, prefix=f"{p}.self_attn.qkv_proj")
        self.o_proj = RowParallelLinear(cfg.num_attention_heads * self.head_dim, cfg.hidden_size, bias=cfg.o_bias,
                                        quant_config=quant, params_dtype=dtype, prefix=f"{p}.self_attn.o_proj")
        self.gate_up_proj = MergedColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size] * 2, bias=cfg.mlp_bias,
                                                       quant_config=quant, params_dtype=dtype, prefix=f"{p}.mlp.gate_up_proj")
        self.down_proj = RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, bias=cfg.mlp_bias, quant_config=quant,
                                           params_dtype=dtype, prefix=f"{p}.mlp.down_proj")
        self.input_layernorm = torch.nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype), requires_grad=False)
        self.post_attention_layernorm = torch.nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype), requires_grad=False)
        # static KV cache [B, H_kv, S, D]: fixed shapes keep the step graph-capturable
        self.register_buffer("k_cache", torch.zeros(max_batch, self.num_kv_heads, max_seq, self.head_dim, dtype=dtype), persistent=False)
        self.register_buffer("v_cache", torch.zeros(max_batch, self.num_kv_heads, max_seq, self.head_dim, dtype=dtype), persistent=False)

    FUSE_NORM_MAX_BATCH = 16
    FUSE_NORM = os.environ.get("SGLANG_AWQ_AMD_FOLD_NORM", "1") != "0"      # 0: the reference's norm order at every batch size
    attn_splits = 1            # workgroups per (sequence, head) in the decode attention; GraphedDecoder sets it from batch / context

    def norm_order(self, batch: int) -> str:
        """Which arithmetic order the RMSNorm in front of qkv_proj / gate_up_proj runs in at this batch size (reported by the
        benchmarks next to their tok/s): "folded" = inv_rms * ((v * w) W), the norm carried through the GEMV's linearity
        (gemv_rp2_kernel<NORM>; differs from the reference's order by where x is rounded, at most an fp16 ulp of x per element);
        "reference" = fp16(fp16(v * inv_rms) * w) W as models/llama.py:277-290 -> layernorm.py computes it (a separate norm launch)."""
        folded = (self.FUSE_NORM and batch <= self.FUSE_NORM_MAX_BATCH and getattr(self.qkv_proj, "awq_packed", None) is not None
                  and self.qkv_proj.bias is None)
        return "folded: inv_rms * ((v * w) W)" if folded else "reference: fp16(fp16(v * inv_rms) * w) W"

    @staticmethod
    def _awq_dims(lin):
        K = lin.qweight.shape[0]
        return K, lin.qweight.shape[1] * 8, K // lin.scales.shape[0]

    def _gate_up_interleaved(self):
        """Second repacked copy of gate_up whose 16-column groups alternate gate / up (built once, lazily): lets the
        GEMV's epilogue apply SiLU-mul.  None when the shape has no repacked form."""
        if not hasattr(self, "_gu_il"):
            from . import aux_ops, ops

            self._gu_il = None
            lin = self.gate_up_proj
            if getattr(lin, "awq_packed", None) is not None and lin.bias is None and (lin.qweight.shape[1] * 8) % 32 == 0:
                self._gu_il = ops.awq_repack(*aux_ops.interleave_gate_up(lin.qweight.data, lin.scales.data, lin.qzeros.data))
        return self._gu_il

    def forward_fused(self, h, delta, pos, cos_table, sin_table):
        """Same layer with its elementwise neighbours fused: h is the residual stream, delta the previous layer's MLP
        output still to be added.  Returns (h + delta + attention output, this layer's MLP output = the next delta).
        Five launches at decode batch sizes: qkv GEMV (RMSNorm + residual in its prologue), attention (RoPE + KV write
        inside), o_proj GEMV, gate_up GEMV (RMSNorm + residual prologue, SiLU-mul epilogue), down_proj GEMV; shapes
        without a fused kernel fall back to the separate launches."""
        from . import aux_ops

        B = h.shape[0]
        eps = self.cfg.rms_norm_eps
        # the norm is folded through the GEMV (gemv_rp2_kernel<NORM>: x' = (h + delta) * w staged per wave, inv_rms applied to
        # the fp32 sums in the epilogue): 4096 x 12288 at 1 / 2 / 4 / 8 rows 7.4 / 8.3 / 8.7 / 10.1 us against 8.9 / 9.2 / 11.0 /
        # 14.5 for the earlier prologue form and ~4.7 us for a separate norm launch; the SiLU-mul epilogue is free at every batch
        # size.  The folded form exists up to 4 staging chunks per lane (8 rows at K = 4096, 4 at K = 8192); beyond that the call
    # returns None and the norm is its own launch (as it is past 16 rows, where the GEMV runs two row tiles per fragment).
        fuse_norm = self.FUSE_NORM and B <= self.FUSE_NORM_MAX_BATCH
        qkv = None
        packed = getattr(self.qkv_proj, "awq_packed", None)
        if fuse_norm and packed is not None and self.qkv_proj.bias is None:
            r = aux_ops.gemv_repacked_fused(packed, *self._awq_dims(self.qkv_proj), norm=(h, delta, self.input_layernorm, eps))
            if r is not None:
                qkv, h = r
        if qkv is None:
            x = aux_ops.add_rmsnorm(h, delta, self.input_layernorm, eps)          # h += delta in place
            qkv, _ = self.qkv_proj(x)
        attn = aux_ops.decode_attention(qkv, pos, cos_table, sin_table, self.k_cache, self.v_cache, self.num_heads, self.num_kv_heads,
                                        self.head_dim, num_splits=self.attn_splits)
        o, _ = self.o_proj(attn.reshape(B, self.q_size))
        act = None
        gu_il = self._gate_up_interleaved() if B <= 32 else None
        if gu_il is not None:
            dims = self._awq_dims(self.gate_up_proj)
            r = None
            if fuse_norm:
                r = aux_ops.gemv_repacked_fused(gu_il, *dims, norm=(h, o, self.post_attention_layernorm, eps), silu_mul=True)
                if r is not None:
                    act, h = r
            if r is None:
                x = aux_ops.add_rmsnorm(h, o, self.post_attention_layernorm, eps)
                r = aux_ops.gemv_repacked_fused(gu_il, *dims, x=x, silu_mul=True)
                if r is not None:
                    act = r[0]
                else:
                    gu, _ = self.gate_up_proj(x)
                    act = aux_ops.silu_mul(gu)
        if act is None:
            x = aux_ops.add_rmsnorm(h, o, self.post_attention_layernorm, eps)
            gu, _ = self.gate_up_proj(x)
            act = aux_ops.silu_mul(gu)
        d, _ = self.down_proj(act)
        return h, d

    def forward(self, h: torch.Tensor, pos: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        B = h.shape[0]
        x = rms_norm(h, self.input_layernorm, self.cfg.rms_norm_eps)
        qkv, _ = self.qkv_proj(x)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q = apply_rope_neox(q.view(B, self.num_heads, self.head_dim), cos, sin)
        k = apply_rope_neox(k.view(B, self.num_kv_heads, self.head_dim), cos, sin)
        v = v.view(B, self.num_kv_heads, self.head_dim)
        idx = pos.view(B, 1, 1, 1).expand(B, self.num_kv_heads, 1, self.head_dim)
        self.k_cache[:B].scatter_(2, idx, k.unsqueeze(2))
        self.v_cache[:B].scatter_(2, idx, v.unsqueeze(2))
        attn = F.scaled_dot_product_attention(q.unsqueeze(2), self.k_cache[:B], self.v_cache[:B], attn_mask=mask,
                                              enable_gqa=self.num_heads != self.num_kv_heads)
        o, _ = self.o_proj(attn.reshape(B, self.q_size))
        h = h + o
        x = rms_norm(h, self.post_attention_layernorm, self.cfg.rms_norm_eps)
        gu, _ = self.gate_up_proj(x)
        half = gu.shape[-1] // 2
        act = F.silu(gu[:, :half]) * gu[:, half:]
        d, _ = self.down_proj(act)
        return h + d


class LlamaForCausalLM(torch.nn.Module):
    """Decode-only Llama with AWQ projections, fp16 embedding / lm_head (as AutoAWQ checkpoints keep them)."""

    def __init__(self, cfg: LlamaConfig, quant: AWQConfig, max_batch: int = 32, max_seq: int = 512, dtype: torch.dtype = torch.float16):
        super().__init__()
        self.cfg, self.max_batch, self.max_seq, self.dtype = cfg, max_batch, max_seq, dtype
        self.fused_aux = True        # fused RMSNorm / RoPE+KV / SiLU-mul kernels (aux_ops); False = plain torch ops
        self.embed_tokens = torch.nn.Parameter(torch.zeros(cfg.vocab_size, cfg.hidden_size, dtype=dtype), requires_grad=False)
        self.layers = torch.nn.ModuleList(LlamaDecoderLayer(cfg, quant, i, max_batch, max_seq, dtype) for i in range(cfg.num_hidden_layers))
        self.norm = torch.nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype), requires_grad=False)
        self.lm_head = torch.nn.Parameter(torch.zeros(cfg.vocab_size, cfg.hidden_size, dtype=dtype), requires_grad=False)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
        t = torch.arange(max_seq, dtype=torch.float32)
        freqs = torch.outer(t, inv)
        self.register_buffer("cos_table", freqs.cos(), persistent=False)
        self.register_buffer("sin_table", freqs.sin(), persistent=False)
        self.register_buffer("arange_seq", torch.arange(max_seq), persistent=False)
        self.register_buffer("zero_delta", torch.zeros(max_batch, cfg.hidden_size, dtype=dtype), persistent=False)

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 0):
        """Random packed int4 weights, scales 0.005..0.02 / sqrt-ish fan-in scaling so activations stay O(1)."""
        g = torch.Generator(device=self.embed_tokens.device)
        g.manual_seed(seed)
        dev = self.embed_tokens.device
        self.embed_tokens.copy_((torch.randn(self.embed_tokens.shape, device=dev, generator=g) * 0.5).to(self.dtype))
        self.lm_head.copy_((torch.randn(self.lm_head.shape, device=dev, generator=g) * 0.02).to(self.dtype))
        for layer in self.layers:
            for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
                K = lin.qweight.shape[0]
                lin.qweight.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, lin.qweight.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
                lin.qzeros.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, lin.qzeros.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
                s = (0.5 + torch.rand(lin.scales.shape, device=dev, generator=g)) * (0.15 / (K ** 0.5))   # W std ~ 1/sqrt(K)
                lin.scales.copy_(s.to(self.dtype))
                lin.process_weights_after_loading()
        return self

    def step(self, tokens: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """One decode step.  tokens [B] int64, pos [B] int64 (position of these tokens) -> next tokens [B]."""
        return self.logits(tokens, pos).argmax(-1)

    def logits(self, tokens: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        if self.fused_aux and self.dtype == torch.float16:
            from . import aux_ops

            h = self.embed_tokens[tokens]                       # residual stream, updated in place by add_rmsnorm
            delta = self.zero_delta[:h.shape[0]]
            for layer in self.layers:
                h, delta = layer.forward_fused(h, delta, pos, self.cos_table, self.sin_table)
            h = aux_ops.add_rmsnorm(h, delta, self.norm, self.cfg.rms_norm_eps)
            return torch.matmul(h, self.lm_head.t())
        h = self.embed_tokens[tokens]
        cos = self.cos_table[pos].unsqueeze(1)
        sin = self.sin_table[pos].unsqueeze(1)
        mask = (self.arange_seq.view(1, 1, 1, -1) <= pos.view(-1, 1, 1, 1))          # attend to cache slots <= pos
        for layer in self.layers:
            h = layer(h, pos, cos, sin, mask)
        h = rms_norm(h, self.norm, self.cfg.rms_norm_eps)
        return torch.matmul(h, self.lm_head.t())


class GraphedDecoder:
    """Captures `model.step` for a fixed batch size into one HIP graph; `run(n)` replays n decode steps with the
    token / position buffers advanced on device between replays (no host round trip inside the timed loop)."""

    def __init__(self, model: LlamaForCausalLM, batch: int, start_pos: int = 0):
        self.model, self.batch = model, batch
        dev = model.embed_tokens.device
        self.start_pos = start_pos
        self.tokens = torch.zeros(batch, dtype=torch.int64, device=dev)
        self.pos = torch.full((batch,), start_pos, dtype=torch.int64, device=dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.capture_error: Optional[str] = None    # why run() steps eagerly, if it does
        self.steps_taken = 0                      # host-side mirror of how far `pos` has advanced on the device
        if start_pos >= model.max_seq:
            raise ValueError(f"start_pos {start_pos} is outside the KV cache (max_seq {model.max_seq})")

    def _set_attention_splits(self):
        """Long context at small batch: one workgroup per (sequence, head) leaves most of the chip idle (batch 1, context
        1024: 16 us per layer); spread each over up to 16 workgroups until the grid covers the 256 CUs."""
        layer = self.model.layers[0]
        splits = 1
        if self.start_pos >= 384:
            # measured at context 1024 (7B, 32 heads; step time at 2 / 4 / 8 / 16 splits): batch 1: 1.410 / 1.347 / 1.316 / 1.343 ms,
            # batch 2: 1.511 / 1.463 / 1.449 / 1.535, batch 4: 1.694 / 1.673 / 1.683 / 1.751 -> two workgroups per CU, at most 8 splits
            splits = max(1, min(8, 512 // max(1, self.batch * layer.num_heads)))
        for lyr in self.model.layers:
            lyr.attn_splits = splits

    def _step(self):
        model = self.model
        if model.fused_aux and model.dtype == torch.float16 and model.cfg.vocab_size % 8 == 0:
            from . import aux_ops

            aux_ops.argmax_advance(model.logits(self.tokens, self.pos), self.tokens, self.pos)   # greedy tail in one launch
            return
        nxt = model.step(self.tokens, self.pos)
        self.tokens.copy_(nxt)
        self.pos.add_(1)

    @torch.no_grad()
    def capture(self, warmup: int = 2):
        self._set_attention_splits()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        if self.start_pos + self.steps_taken + warmup > self.model.max_seq:
            raise ValueError("not enough room in the KV cache for the warm-up steps")
        # the per-(device, stream) scratch of the split-K GEMV and the split-S attention is created eagerly, before anything is
        # captured on `s` (a first use inside the capture would raise: ops._workspace) — also with warmup = 0
        from . import _lib, aux_ops, ops

        layer0 = self.model.layers[0]
        dev = self.tokens.device
        ops.prepare_stream_workspaces(s, dev)
        if layer0.attn_splits > 1:
            need = _lib.load().awq_aux_decode_attention_workspace_bytes(self.batch, layer0.num_heads, layer0.head_dim, layer0.attn_splits)
            aux_ops.prepare_attention_workspace(dev, s.cuda_stream, need)
            torch.cuda.current_stream(dev).synchronize()
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        self.steps_taken += warmup                # the captured step itself does not execute
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        try:
            # capture on the SAME stream the warm-up steps ran on: the per-(device, stream) scratch buffers of the split-K GEMV
            # and the split-S attention already exist there (a capture-time miss raises instead of allocating, ops._workspace)
            with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):     # (the RCCL watchdog thread may touch the runtime meanwhile)
                self._step()
            self.graph = graph
        except Exception as e:       # e.g. a collective that cannot be captured on this stack: run() then steps eagerly
            import sys

            print(f"GraphedDecoder: graph capture failed ({e!r}); stepping eagerly", file=sys.stderr)
            self.graph = None
            self.capture_error = repr(e)
            torch.cuda.synchronize()
        return self

    def reset(self, start_pos: Optional[int] = None) -> None:
        """Rewind the device-side positions (the KV cache keeps whatever the earlier steps wrote)."""
        if start_pos is not None:
            if start_pos >= self.model.max_seq:
                raise ValueError(f"start_pos {start_pos} is outside the KV cache (max_seq {self.model.max_seq})")
            self.start_pos = start_pos
        self.pos.fill_(self.start_pos)
        self.steps_taken = 0

    @torch.no_grad()
    def run(self, steps: int) -> List[int]:
        """steps decode steps (graph replays, or eager steps when capture was not possible); returns the current tokens.
        Raises before the device-side position would leave the KV cache / rotary tables (the kernels index them by pos)."""
        if self.start_pos + self.steps_taken + steps > self.model.max_seq:
            raise ValueError(f"{steps} more steps would write past the KV cache: position {self.start_pos + self.steps_taken} of "
                             f"{self.model.max_seq}")
        self.steps_taken += steps
        for _ in range(steps):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
        return self.tokens.tolist()
