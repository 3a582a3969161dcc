"""Round-3 parity additions (-m gpu): the long straight-line and loop forms of the decode GEMV (gemv_rp2_kernel up to 32 units per
wave, gemv_rp3_kernel) against the oracle, AWQ-MoE behind the reference's call
interface with padded ids, weight updates through `param.data.copy_` followed by the hook, scratch buffers under graph capture,
bias loading through apply, outlier activations through the norm-folded GEMV."""
import ctypes
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from oracle import awq_ref, c_oracle
from sglang_awq_amd import _lib, synth
from tests.util import assert_gemm_close, to_np, to_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    from sglang_awq_amd import ops as _ops   # raises if the HIP library is missing: no fallback

    _lib.load()
    return _ops


def _dev(*arrs):
    return [to_torch(a, DEV) for a in arrs]


# (K, N): deep K on one column group per workgroup (7B down_proj), deep K on wide strips, ragged k-blocks per wave (KB = 86 -> 6 per
# wave, 4 left for the last wave; KB = 40 -> 3 per wave, waves 14 / 15 empty), wide strips (G = 7: 8192 x 28672 in miniature is too
# big for the oracle — 2048 x 28672 has the same strip width and 1 k-block per wave is straight-line, so 4096 x 27648 with G = 7)
LOOP_SHAPES = [(11008, 4096), (5120, 2048), (8192, 1280), (4096, 27648), (2048, 512), (28672, 1024), (8192, 14336)]


@pytest.mark.parametrize("K,N", LOOP_SHAPES)
def test_loop_form_gemv_vs_oracle(ops, K, N):
    """awq_gemm_repacked at every row count of the decode range on shapes the straight-line kernel cannot hold: within half an
    output ulp + 1e-3 of the exact sum (north star: 1e-3 fp16 atol), bias epilogue = y + bias with a second rounding."""
    qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", seed=K + N)
    dq, ds, dz = _dev(qw, s, qz)
    packed = ops.awq_repack(dq, ds, dz)
    for M in (1, 2, 3, 6, 8, 9, 12, 16):
        x = synth.make_activations(M, K, "f16", "A", seed=M + K)
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        y = ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, 128)
        assert_gemm_close(to_np(y), exact, "f16", what=f"loop form M={M} K={K} N={N}")
        if M in (1, 9):
            b = to_torch(synth.make_bias(N, "f16", 5), DEV)
            assert torch.equal(ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, 128, b), y + b)


def _moe_layer(E, K, I, g, seed0=100):
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.moe import AWQMoEMethod

    method = AWQMoEMethod(AWQConfig(4, g, True))
    layer = torch.nn.Module()
    method.create_weights(layer, E, K, I, torch.float16)
    w13 = [synth.make_awq_weights(K, 2 * I, g, "f16", "A", seed=seed0 + e) for e in range(E)]
    w2 = [synth.make_awq_weights(I, K, g, "f16", "A", seed=seed0 + 100 + e) for e in range(E)]
    for e in range(E):
        layer.w13_qweight.data[e].copy_(to_torch(w13[e][0])); layer.w13_scales.data[e].copy_(to_torch(w13[e][1])); layer.w13_qzeros.data[e].copy_(to_torch(w13[e][2]))
        layer.w2_qweight.data[e].copy_(to_torch(w2[e][0])); layer.w2_scales.data[e].copy_(to_torch(w2[e][1])); layer.w2_qzeros.data[e].copy_(to_torch(w2[e][2]))
    layer.to(DEV)
    method.process_weights_after_loading(layer)
    return method, layer


@pytest.mark.parametrize("T", [2, 24])
def test_moe_reference_interface_and_padded_ids(ops, T):
    """create_moe_runner + apply(layer, dispatch_output) -> CombineInput (reference awq.py:815-845) with duck-typed
    StandardDispatchOutput / StandardTopKOutput / MoeRunnerConfig; ids of -1 (the reference's padded tokens of a graph batch,
    layers/moe/topk.py:705-712) and ids >= E contribute zero on both routes — an all-padded token gives an exactly zero row —
    and the tensor form gives the same bits."""
    from sglang_awq_amd.moe import StandardCombineInput, select_experts

    E, K, I, top_k, g = 4, 2048, 512, 2, 128
    method, layer = _moe_layer(E, K, I, g, seed0=300)
    method.create_moe_runner(layer, types.SimpleNamespace(activation="silu", is_gated=True, apply_router_weight_on_input=False,
                                                          no_combine=False, inplace=False, routed_scaling_factor=None, top_k=top_k,
                                                          num_experts=E))
    x = to_torch(synth.make_activations(T, K, "f16", "A", seed=T, x_std=0.5), DEV)
    tw, ti = select_experts(to_torch(synth.make_activations(T, E, "f16", "A", seed=T + 9).astype(np.float32), DEV), top_k)
    full = method.apply(layer, x, tw, ti)
    ti_pad = ti.clone()
    ti_pad[T - 1, :] = -1                       # a padded token
    ti_pad[0, 1] = E + 3                        # an out-of-range id
    disp = types.SimpleNamespace(hidden_states=x, hidden_states_scale=None,
                                 topk_output=types.SimpleNamespace(topk_weights=tw, topk_ids=ti_pad, router_logits=None))
    out = method.apply(layer, disp)
    assert isinstance(out, StandardCombineInput) and out.format == "standard"
    y = out.hidden_states
    assert y.shape == (T, K) and torch.isfinite(y).all()
    assert torch.count_nonzero(y[T - 1]) == 0, "a token whose ids are all -1 must come out as zeros"
    assert torch.equal(y, method.apply(layer, x, tw, ti_pad))
    if T > 2:
        assert torch.equal(y[1:T - 1], full[1:T - 1]), "rows without padded ids must not change"
    # token 0 keeps only its first expert: its row equals the one-expert computation
    one = method.apply(layer, x[:1].contiguous(), tw[:1, :1].contiguous(), ti[:1, :1].contiguous())
    assert torch.equal(y[0], one[0])
    with pytest.raises(NotImplementedError):
        method.create_moe_runner(layer, types.SimpleNamespace(activation="gelu", is_gated=True))
    # routed_scaling_factor is applied to the combined output
    method.create_moe_runner(layer, types.SimpleNamespace(activation="silu", is_gated=True, routed_scaling_factor=2.0, inplace=False))
    assert torch.equal(method.apply(layer, disp).hidden_states, y * 2.0)


def test_moe_block_route_matches_slot_route(ops):
    """The expert-sorted 16-row blocks (awq_aux_moe_gemv_blocks, gemv_rp3_kernel with a row map) and the one-row-per-pair launch
    (awq_aux_moe_gemv) round at the same points: the same pairs through both give the same outputs (bit-identical up to the
    compiler's choice of a fused multiply-convert, see below)."""
    from sglang_awq_amd.moe import AWQMoEMethod, select_experts

    E, K, I, top_k, g = 4, 2048, 512, 2, 128
    method, layer = _moe_layer(E, K, I, g, seed0=500)
    T = 4
    x = to_torch(synth.make_activations(T, K, "f16", "A", seed=3, x_std=0.5), DEV)
    tw, ti = select_experts(to_torch(synth.make_activations(T, E, "f16", "A", seed=4).astype(np.float32), DEV), top_k)
    assert T * top_k <= AWQMoEMethod.slot_route_max_pairs(E)
    y_slot = method.apply(layer, x, tw, ti)
    saved = AWQMoEMethod.__dict__["slot_route_max_pairs"]
    try:
        AWQMoEMethod.slot_route_max_pairs = classmethod(lambda cls, num_experts: 0)
        y_blk = method.apply(layer, x, tw, ti)
    finally:
        AWQMoEMethod.slot_route_max_pairs = saved
    # (same arithmetic; hipcc may fuse `fp16(sum * routed_weight)` into one v_fma_mixlo_f16 — a single rounding of the exact product —
    # in one kernel and keep the fp32 multiply + conversion in the other: a rare one-ulp double-rounding difference, nothing else)
    diff = y_slot != y_blk
    assert float(diff.float().mean()) < 1e-3
    # one ulp of a pair's contribution (|y_pair| < 32 here -> 2^-5), which the sum over top_k then carries
    assert (y_slot.float() - y_blk.float()).abs().max().item() <= 2.0 ** -5


def test_weights_updated_after_data_copy(ops):
    """The reference reloads weights through `param.data.copy_` (layers/parameter.py:59,124), which no version counter shows.
    Without the hook both the layer's repacked copy and the op's cached copy keep serving the old weights (asserted: that is the
    hazard); `weights_updated(model)` — what sgl_kernel_compat.install() makes ModelRunner.update_weights_from_* call —
    brings both to the new values."""
    from sglang_awq_amd import weight_update
    from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod

    K, N = 1024, 2048
    w1 = synth.make_awq_weights(K, N, 128, "f16", "A", 11)
    w2 = synth.make_awq_weights(K, N, 128, "f16", "A", 22)
    x = synth.make_activations(3, K, "f16", "A", 5)
    xt = to_torch(x, DEV)
    _, e1 = c_oracle.gemm(x, *w1, want_exact=True)
    _, e2 = c_oracle.gemm(x, *w2, want_exact=True)
    method = AWQLinearMethod(AWQConfig(4, 128, True))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.float16, weight_loader=None)
    layer.quant_method = method
    for name, t in zip(("qweight", "scales", "qzeros"), w1):
        getattr(layer, name).data.copy_(to_torch(t))
    model = torch.nn.Sequential(layer).to(DEV)
    method.process_weights_after_loading(layer)
    ops.awq_gemm_cache_enable(True)
    try:
        assert_gemm_close(to_np(method.apply(layer, xt)), e1, "f16", what="before the update")
        y_op1 = torch.ops.sgl_kernel.awq_gemm(xt, layer.qweight, layer.scales, layer.qzeros, 1)
        assert ops.awq_gemm_cache_info()["entries"] == 1
        for name, t in zip(("qweight", "scales", "qzeros"), w2):
            getattr(layer, name).data.copy_(to_torch(t, DEV))          # how the reference's weight loaders write
        # the hazard: nothing noticed
        assert torch.equal(torch.ops.sgl_kernel.awq_gemm(xt, layer.qweight, layer.scales, layer.qzeros, 1), y_op1)
        # the hook
        assert weight_update.weights_updated(model) == 1
        assert ops.awq_gemm_cache_info()["entries"] == 0
        assert_gemm_close(to_np(method.apply(layer, xt)), e2, "f16", what="apply after weights_updated")
        assert_gemm_close(to_np(torch.ops.sgl_kernel.awq_gemm(xt, layer.qweight, layer.scales, layer.qzeros, 1)), e2, "f16",
                          what="op after weights_updated")
    finally:
        ops.awq_gemm_cache_clear()


def test_cache_is_opt_in_and_inference_tensors_bypass_it():
    """A fresh process: the op's cache is off until someone vouches for invalidation (ops._OP_CACHE_MODE "auto");
    sgl_kernel_compat.install() switches it on together with the reload hooks.  Weights created under torch.inference_mode() have
    no version counter: the op must still work (cache bypassed), not raise."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from sglang_awq_amd import ops, synth, sgl_kernel_compat
from tests.util import to_torch
assert ops.awq_gemm_cache_info()["enabled"] is False and ops.awq_gemm_cache_info()["mode"] == "auto"
K, N = 1024, 1024
qw, s, qz = [to_torch(t, "cuda:0") for t in synth.make_awq_weights(K, N, 128, "f16", "A", 1)]
x = to_torch(synth.make_activations(2, K, "f16", "A", 2), "cuda:0")
y0 = ops.awq_gemm(x, qw, s, qz, 1)
assert ops.awq_gemm_cache_info()["entries"] == 0
sgl_kernel_compat.install(force_module=True)
assert ops.awq_gemm_cache_info()["enabled"] is True
y1 = ops.awq_gemm(x, qw, s, qz, 1)
assert ops.awq_gemm_cache_info()["entries"] == 1
assert (y0.float() - y1.float()).abs().max().item() < 2e-2
with torch.inference_mode():
    qi, si, zi = qw.clone(), s.clone(), qz.clone()
    yi = ops.awq_gemm(x, qi, si, zi, 1)
assert ops.awq_gemm_cache_info()["entries"] == 1
assert torch.equal(yi.cpu(), y0.cpu())
print("OPT_IN_OK")
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "SGLANG_AWQ_AMD_OP_CACHE"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OPT_IN_OK" in r.stdout, r.stderr[-2000:]


def test_scratch_is_never_created_during_capture(ops):
    """A first call on a capturing stream that needs the per-stream scratch raises (allocation would come from the graph's private
    pool, the zero-fill would be an un-run graph node); after prepare_stream_workspaces(stream) the same capture works and
    replays to the eager bits — the split-K route at 16 rows on 11008 x 4096."""
    K, N, M = 11008, 4096, 16
    qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", 77)
    packed = ops.awq_repack(*_dev(qw, s, qz))
    x = to_torch(synth.make_activations(M, K, "f16", "A", 78), DEV)
    want = ops.awq_gemm_repacked(x, packed, K, N, 128)          # eager, on the current stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="scratch"):
        with torch.cuda.graph(g, stream=side):
            ops.awq_gemm_repacked(x, packed, K, N, 128)
    torch.cuda.synchronize()
    ops.prepare_stream_workspaces(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        y = ops.awq_gemm_repacked(x, packed, K, N, 128)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_graphed_decoder_captures_batch16_without_prior_call_on_the_capture_stream(ops):
    """GraphedDecoder.capture() warms up and captures on one stream: batch 16 at 7B widths (split-K down_proj, split-S attention off)
    must capture (no eager fallback) and replay to the eager tokens."""
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                      vocab_size=1024)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg, AWQConfig(4, 128, True), max_batch=16, max_seq=64).to(DEV)
    model.init_synthetic_(seed=3)
    dec = GraphedDecoder(model, 16, start_pos=8)
    dec.tokens.copy_(torch.arange(16, device=DEV) % cfg.vocab_size)
    ref = GraphedDecoder(model, 16, start_pos=8)
    ref.tokens.copy_(dec.tokens)
    dec.capture(warmup=2)
    assert dec.graph is not None and dec.capture_error is None, dec.capture_error
    got = dec.run(3)
    for layer in model.layers:
        layer.k_cache.zero_(); layer.v_cache.zero_()
    for _ in range(5):                                     # 2 warm-up steps + 3 timed ones, eagerly
        ref._step()
    assert got == ref.tokens.tolist()


def test_bias_loaded_and_applied_through_apply(ops):
    """Bias parameters carry the reference's weight_loader / output_dim = 0 (linear.py:348-358, 1286-1293): a q/k/v bias loads as
    shards of qkv_proj.bias and is applied by the fused bias epilogue (out = fp16(sum), then out + bias with a second rounding:
    awq.py:449-450)."""
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.linear import QKVParallelLinear, RowParallelLinear

    H, hd, nh, nkv = 1024, 128, 8, 2
    quant = AWQConfig(4, 128, True)
    qkv = QKVParallelLinear(H, hd, nh, nkv, bias=True, quant_config=quant, params_dtype=torch.float16, tp_rank=0, tp_size=1)
    N = (nh + 2 * nkv) * hd
    qw, s, qz = synth.make_awq_weights(H, N, 128, "f16", "A", 41)
    b = synth.make_bias(N, "f16", 42)
    sizes = {"q": nh * hd, "k": nkv * hd, "v": nkv * hd}
    off = 0
    for sid in ("q", "k", "v"):
        n = sizes[sid]
        qkv.qweight.weight_loader(qkv.qweight, to_torch(qw[:, off // 8:(off + n) // 8]), sid)
        qkv.qzeros.weight_loader(qkv.qzeros, to_torch(qz[:, off // 8:(off + n) // 8]), sid)
        qkv.scales.weight_loader(qkv.scales, to_torch(s[:, off:off + n]), sid)
        qkv.bias.weight_loader(qkv.bias, to_torch(b[off:off + n]), sid)
        off += n
    qkv.to(DEV)
    qkv.process_weights_after_loading()
    assert torch.equal(qkv.bias.cpu(), to_torch(b))
    for M in (1, 5, 40):
        x = synth.make_activations(M, H, "f16", "A", M)
        y, ob = qkv(to_torch(x, DEV))
        assert ob is None
        want = awq_ref.awq_linear_apply(x, qw, s, qz, b).astype(np.float64)
        got = to_np(y).astype(np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -14))) - 10)
        assert np.all(np.abs(got - want) <= 1.01 * ulp + 1e-3), f"M={M}: {np.abs(got - want).max()}"
    # row-parallel: replicated bias, loaded whole
    row = RowParallelLinear(H, 512, bias=True, quant_config=quant, params_dtype=torch.float16, tp_rank=0, tp_size=1)
    rb = synth.make_bias(512, "f16", 43)
    row.bias.weight_loader(row.bias, to_torch(rb))
    assert torch.equal(row.bias, to_torch(rb))


def test_norm_folded_gemv_with_outlier_channels(ops):
    """Massive residual activations (|h| ~ 8000 in a few channels) with a norm weight above 1: the folded form stages
    x' = fp16((h + delta) * w) un-normalised, which would overflow fp16; the harness must either stay finite and close to the fp32
    RMSNorm -> linear result, or decline (None) so the caller runs the separate norm."""
    from sglang_awq_amd import aux_ops

    K, N = 4096, 4096
    qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", 91)
    packed = ops.awq_repack(*_dev(qw, s, qz))
    h = synth.make_activations(2, K, "f16", "A", 92).astype(np.float32)
    h[:, [7, 1033, 4000]] = [8000.0, -6000.0, 7000.0]
    w = np.full(K, 1.0, np.float32); w[[7, 1033]] = 12.0            # 8000 * 12 > 65504
    ht, dt_, wt = to_torch(h.astype(np.float16), DEV), torch.zeros(2, K, dtype=torch.float16, device=DEV), to_torch(w.astype(np.float16), DEV)
    r = aux_ops.gemv_repacked_fused(packed, K, N, 128, norm=(ht, dt_, wt, 1e-5))
    hf = torch.from_numpy(h.astype(np.float16).astype(np.float32))
    xn = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.float16) * torch.from_numpy(w.astype(np.float16))
    _, exact = c_oracle.gemm(xn.numpy(), qw, s, qz, want_exact=True)
    if r is None:
        return                                               # declined: the caller runs add_rmsnorm + the plain GEMV
    y = to_np(r[0]).astype(np.float64)
    assert np.isfinite(y).all(), "the folded norm overflowed fp16 on outlier channels"
    scale = np.abs(exact).max()
    assert np.abs(y - exact).max() <= 2e-2 * scale + 1e-2, f"folded norm with outliers off by {np.abs(y - exact).max()} (scale {scale})"


@pytest.mark.parametrize("B", [1, 8, 32])
def test_7b_dimension_layers_match_fp32_reference(B):
    """The wiring the driver line's `decode_7b_tp1` number comes from, at its real size: 2 layers at Llama-2-7B dimensions (H 4096,
    32 heads, I 11008, vocab 32000), 1024+ positions already in the KV cache (batch 1 runs the 8-way split-S attention), folded norm ->
    qkv -> attention -> o_proj -> folded norm + SiLU-mul -> down (batch 32: separate norm, two row tiles), against the fp32 reference
    of test_gpu_llama.py built from the SAME dequantised weights; fused and unfused; then the graph replay against eager stepping."""
    from sglang_awq_amd import ops
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM
    from tests.test_gpu_llama import _ref_step

    cfg = LlamaConfig(num_hidden_layers=2)                  # every other field = Llama-2-7B
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.vocab_size) == (4096, 11008, 32, 32000)
    S, P0 = 1040, 1024
    with torch.device(DEV):
        model = LlamaForCausalLM(cfg, AWQConfig(4, 128, True), max_batch=B, max_seq=S)
    model.init_synthetic_(seed=11)
    W = [{name: ops.awq_dequantize(lin.qweight, lin.scales, lin.qzeros).float()
          for name, lin in (("qkv", layer.qkv_proj), ("o", layer.o_proj), ("gate_up", layer.gate_up_proj), ("down", layer.down_proj))}
         for layer in model.layers]
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    past_k = [(torch.randn(B, l.num_kv_heads, P0, cfg.head_dim, device=DEV, generator=gen) * 0.5).half() for l in model.layers]
    past_v = [(torch.randn(B, l.num_kv_heads, P0, cfg.head_dim, device=DEV, generator=gen) * 0.5).half() for l in model.layers]
    splits = max(1, min(8, 512 // (B * model.layers[0].num_heads)))           # what GraphedDecoder picks at this context
    assert splits == {1: 8, 8: 2, 32: 1}[B]

    def reset_caches():
        for l, pk, pv in zip(model.layers, past_k, past_v):
            l.k_cache.zero_(); l.v_cache.zero_()
            l.k_cache[:, :, :P0] = pk; l.v_cache[:, :, :P0] = pv
            l.attn_splits = splits

    with torch.no_grad():
        for fused in (False, True):
            model.fused_aux = fused
            reset_caches()
            kc = [torch.zeros(B, l.num_kv_heads, S, cfg.head_dim, device=DEV) for l in model.layers]
            vc = [torch.zeros(B, l.num_kv_heads, S, cfg.head_dim, device=DEV) for l in model.layers]
            for c, pk in zip(kc, past_k):
                c[:, :, :P0] = pk.float()
            for c, pv in zip(vc, past_v):
                c[:, :, :P0] = pv.float()
            tokens = (torch.arange(B, device=DEV) * 977 + 13) % cfg.vocab_size
            pos = torch.full((B,), P0, dtype=torch.int64, device=DEV)
            for step in range(3):
                got = model.logits(tokens, pos).float()
                want = _ref_step(model, W, tokens, pos, kc, vc)
                scale = want.abs().max().item()
                err = (got - want).abs().max().item()
                assert err <= 2e-2 * scale + 2e-2, f"B={B} fused={fused} step {step}: {err} at scale {scale}"
                tokens = want.argmax(-1)
                pos = pos + 1
            del kc, vc
    del W
    torch.cuda.empty_cache()
    # graph replay == eager stepping at this size (same kernels, same order)
    model.fused_aux = True
    reset_caches()
    eager = GraphedDecoder(model, B, start_pos=P0)
    eager._set_attention_splits()
    eager.tokens.copy_((torch.arange(B, device=DEV) * 31 + 7) % cfg.vocab_size)
    seq_eager = [eager.run(1) for _ in range(4)]
    reset_caches()
    graphed = GraphedDecoder(model, B, start_pos=P0)
    graphed.tokens.copy_((torch.arange(B, device=DEV) * 31 + 7) % cfg.vocab_size)
    graphed.capture(warmup=0)
    assert graphed.graph is not None, graphed.capture_error
    assert [graphed.run(1) for _ in range(4)] == seq_eager
    assert "folded" in model.layers[0].norm_order(B) if B <= 16 else "reference" in model.layers[0].norm_order(B)


@pytest.mark.parametrize("BS", [16, 64, 128])
@pytest.mark.parametrize("P,E", [(1, 8), (40, 8), (600, 8), (33, 4), (5, 1), (4096, 60), (7, 1024)])
def test_moe_align_blocks_kernel_contract(ops, P, E, BS):
    """awq_aux_moe_align_blocks_n (one-workgroup counting sort; 16-row blocks for the GEMV route, 128-row blocks for the MFMA tile route)
    against the contract of the tensor-op form: every valid pair exactly once, in a block of its own expert; padding and unused blocks are
    -1; ids outside [0, E) are dropped; experts in ascending block order."""
    from sglang_awq_amd.moe import AWQMoEMethod

    g = torch.Generator().manual_seed(P * 7 + E)
    ids = torch.randint(-1, E, (P,), dtype=torch.int32, generator=g)
    if P > 3:
        ids[2] = E + 5
    row_map, block_expert = AWQMoEMethod.align_blocks_device(ids.to(DEV), E, BS)
    torch.cuda.synchronize()
    row_map, block_expert = row_map.cpu(), block_expert.cpu()
    B = block_expert.numel()
    assert B == (P + BS - 1) // BS + E and row_map.numel() == BS * B
    ref_map, ref_be = AWQMoEMethod.align_blocks(ids, E, BS)
    assert torch.equal(block_expert, ref_be)
    seen = []
    for b in range(B):
        e = int(block_expert[b])
        rows = row_map[b * BS:(b + 1) * BS].tolist()
        if e < 0:
            assert all(r == -1 for r in rows)
            continue
        real = [r for r in rows if r >= 0]
        assert real and all(int(ids[r]) == e for r in real)
        seen += real
    assert sorted(seen) == [i for i in range(P) if 0 <= int(ids[i]) < E]
    # the same SET of rows per expert as the stable reference (the order inside an expert's run is free)
    for e in range(E):
        got = sorted(r for b in range(B) if int(block_expert[b]) == e for r in row_map[b * BS:(b + 1) * BS].tolist() if r >= 0)
        want = sorted(r for b in range(B) if int(ref_be[b]) == e for r in ref_map[b * BS:(b + 1) * BS].tolist() if r >= 0)
        assert got == want


@pytest.mark.parametrize("wide_from", [0, 1e18], ids=["128-row tiles", "64-row tiles"])
def test_moe_tile_route_matches_block_route(ops, wide_from):
    """Prefill-sized MoE batches run expert-sorted 128-row (or, for thinly loaded experts, 64-row) blocks on the MFMA tile kernel
    (awq_aux_moe_gemm_blocks: gemm_repacked_pipelined_kernel<4, MOE, EPI, MI>) instead of 16-row blocks on the GEMV (awq_aux_moe_gemv_blocks).  Same rounding points
    (fp16 gate_up -> silu * up in the epilogue -> routed weight on the fp32 sums -> one rounding); the fp32 summation order differs, so a
    rounding flips here and there: at most one fp16 ulp of an output's slots.  Padded tokens (ids -1, layers/moe/topk.py:705-712) give zero
    rows on both routes; ragged expert loads (one expert with a single row, one with none) and a width that is not a multiple of the
    256-column tile are covered."""
    from sglang_awq_amd.moe import AWQMoEMethod, select_experts

    E, K, I, top_k, g = 5, 2048, 640, 2, 128             # 2 I = 1280 = 5 tiles of 256; K = 2048 -> w2 width 2048
    method, layer = _moe_layer(E, K, I, g, seed0=700)
    T = 200
    x = to_torch(synth.make_activations(T, K, "f16", "A", seed=31, x_std=0.5), DEV)
    logits = to_torch(synth.make_activations(T, E, "f16", "A", seed=32).astype(np.float32), DEV)
    logits[:, 4] = -1e4                                    # expert 4 is never chosen ...
    tw, ti = select_experts(logits, top_k)
    ti[7, 1] = 4                                           # ... except by one pair
    ti[-8:] = -1                                           # padded tokens of a graph batch
    assert T * top_k >= AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT * E
    saved = (AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT, AWQMoEMethod.TILE_ROUTE_WIDE_ROWS_PER_EXPERT)
    try:
        AWQMoEMethod.TILE_ROUTE_WIDE_ROWS_PER_EXPERT = wide_from
        y_tile = method.apply(layer, x, tw, ti)
        AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT = 1e18
        y_blk = method.apply(layer, x, tw, ti)
    finally:
        AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT, AWQMoEMethod.TILE_ROUTE_WIDE_ROWS_PER_EXPERT = saved
    assert torch.isfinite(y_tile).all() and torch.count_nonzero(y_tile[-8:]) == 0 and torch.count_nonzero(y_blk[-8:]) == 0
    d = (y_tile.float() - y_blk.float()).abs()
    ulp = 2.0 ** (torch.floor(torch.log2(y_blk.float().abs().clamp_min(2.0 ** -14))) - 10)
    # an act element landing on the neighbouring half moves a w2 sum slightly; then one rounding per slot and one of the top_k sum
    assert (d <= 4 * ulp + 4e-3).all(), f"worst {d.max().item()} at |y| {y_blk.float().abs().flatten()[d.argmax()].item()}"
    assert float((y_tile != y_blk).float().mean()) < 0.25


def test_split_s_attention_same_workspace_across_batch_sizes():
    """The split-S decode attention keeps arrival tickets in its per-stream workspace.  A call with a larger batch than an earlier one on the
    same workspace must still find them zero (round 3 bug: the ticket header was B * Hq words, so batch 4 after batch 1 read the earlier
    call's partial sums as tickets — wrong results from batch 4 at 2..8 splits, NaN at batch 8).  Growing and shrinking batch sizes, every
    split count against the single-workgroup result."""
    from sglang_awq_amd import aux_ops

    torch.manual_seed(0)
    Hq = Hkv = 32
    D, S = 128, 1040
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32, device=DEV) / D))
    fr = torch.outer(torch.arange(S, dtype=torch.float32, device=DEV), inv)
    cos_t, sin_t = fr.cos().contiguous(), fr.sin().contiguous()
    for B in (1, 4, 2, 8, 16, 3):
        pos = torch.full((B,), 1024, dtype=torch.int64, device=DEV)
        qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=DEV).half()
        kc0 = (torch.randn(B, Hkv, S, D, device=DEV) * 0.5).half()
        vc0 = (torch.randn(B, Hkv, S, D, device=DEV) * 0.5).half()
        ref = aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kc0.clone(), vc0.clone(), Hq, Hkv, D, num_splits=1)
        for splits in (2, 4, 8, 16):
            for _ in range(2):
                o = aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kc0.clone(), vc0.clone(), Hq, Hkv, D, num_splits=splits)
                assert torch.isfinite(o).all(), f"B={B} splits={splits}"
                assert (o.float() - ref.float()).abs().max().item() <= 2e-3, f"B={B} splits={splits}"


def test_tiny_llama_long_context_every_batch_regime():
    """The harness at a context where split-S attention is on (>= 384 positions in the cache), across the batch regimes of the GEMV dispatcher
    (1, 2-8 folded norm, 9-16, 17-32 two row tiles, 33 = passes) with whatever split count GraphedDecoder picks, against the fp32 reference."""
    from sglang_awq_amd import ops
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM
    from tests.test_gpu_llama import _ref_step

    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4,
                      vocab_size=512, max_position_embeddings=1024)
    S, P0 = 448, 400
    with torch.device(DEV):
        model = LlamaForCausalLM(cfg, AWQConfig(4, 128, True), max_batch=33, max_seq=S)
    model.init_synthetic_(seed=5)
    W = [{name: ops.awq_dequantize(lin.qweight, lin.scales, lin.qzeros).float()
          for name, lin in (("qkv", layer.qkv_proj), ("o", layer.o_proj), ("gate_up", layer.gate_up_proj), ("down", layer.down_proj))}
         for layer in model.layers]
    gen = torch.Generator(device=DEV); gen.manual_seed(9)
    seen_splits = set()
    with torch.no_grad():
        for B in (1, 2, 5, 8, 9, 16, 17, 32, 33):
            dec = GraphedDecoder(model, B, start_pos=P0)
            dec._set_attention_splits()
            seen_splits.add(model.layers[0].attn_splits)
            past_k = [(torch.randn(B, l.num_kv_heads, P0, cfg.head_dim, device=DEV, generator=gen) * 0.5).half() for l in model.layers]
            past_v = [(torch.randn(B, l.num_kv_heads, P0, cfg.head_dim, device=DEV, generator=gen) * 0.5).half() for l in model.layers]
            kc = [torch.zeros(B, l.num_kv_heads, S, cfg.head_dim, device=DEV) for l in model.layers]
            vc = [torch.zeros(B, l.num_kv_heads, S, cfg.head_dim, device=DEV) for l in model.layers]
            for l, pk, pv, c1, c2 in zip(model.layers, past_k, past_v, kc, vc):
                l.k_cache.zero_(); l.v_cache.zero_()
                l.k_cache[:B, :, :P0] = pk; l.v_cache[:B, :, :P0] = pv
                c1[:, :, :P0] = pk.float(); c2[:, :, :P0] = pv.float()
            tokens = (torch.arange(B, device=DEV) * 37 + 5) % cfg.vocab_size
            pos = torch.full((B,), P0, dtype=torch.int64, device=DEV) + (torch.arange(B, device=DEV) % 3)       # ragged positions
            for step in range(2):
                got = model.logits(tokens, pos).float()
                want = _ref_step(model, W, tokens, pos, kc, vc)
                scale = want.abs().max().item()
                err = (got - want).abs().max().item()
                assert err <= 2e-2 * scale + 2e-2, f"B={B} splits={model.layers[0].attn_splits} step {step}: {err} at scale {scale}"
                tokens = want.argmax(-1)
                pos = pos + 1
    assert len(seen_splits) >= 3, seen_splits


@pytest.mark.parametrize("M,K,N,g,bias", [(128, 11008, 4096, 128, False), (100, 4096, 4096, 128, True), (65, 8192, 8192, 128, False),
                                            (96, 4096, 11008, 128, False), (33, 2048, 1032, 128, True), (384, 4096, 4096, 4096, False)])
def test_split_k_tiles_vs_oracle(ops, M, K, N, g, bias):
    """33 rows and up with few MFMA tiles (narrow or deep matrices): `awq_gemm_repacked_ws` splits K over workgroups on the tile kernel
    (gemm_repacked_pipelined_kernel with PfSplit: fp32 partials in the workspace, pf_splitk_reduce_kernel adds them in slice order, adds the
    bias and rounds once).  Against the oracle; run-to-run bit-identical (fixed order, no atomics); the no-workspace entry point (tiles /
    passes without the split) gives the same values up to the order of the fp32 sums; ragged M and N, the bias epilogue and a single
    quantisation group are covered; the route survives graph capture."""
    lib = _lib.load()
    need = lib.awq_gemm_repacked_workspace_bytes(M, K, N, g, 0)
    assert need > 4096, "the shape was meant to take the split-K tile route"
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + K + N)
    x = synth.make_activations(M, K, "f16", "A", seed=M + 11)
    b = synth.make_bias(N, "f16", 5) if bias else None
    packed = ops.awq_repack(*[to_torch(t, DEV) for t in (qw, s, qz)])
    xt = to_torch(x, DEV)
    bt = to_torch(b, DEV) if bias else None
    y1 = ops.awq_gemm_repacked(xt, packed, K, N, g, bias=bt)
    y2 = ops.awq_gemm_repacked(xt, packed, K, N, g, bias=bt)
    assert torch.equal(y1, y2), "split-K tiles: not run-to-run identical"
    _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
    if bias:
        from tests.util import ulp
        pre = exact.astype(np.float16).astype(np.float64)                     # the reference rounds the product ...
        want = (pre + b.astype(np.float64)).astype(np.float16).astype(np.float64)        # ... adds the bias and rounds again (awq.py:447-449)
        got = to_np(y1).astype(np.float64)
        # one ulp of the PRE-bias sum (a rounding-boundary flip of the fp32 sum against the exact one; several ulps of a smaller post-bias
        # result where the bias cancels) and one of the result (tests/test_gpu_parity.py: bias epilogue)
        assert np.all(np.abs(got - want) <= 1.01 * (ulp(want, "f16") + ulp(pre, "f16"))), f"bias epilogue M={M} K={K} N={N}"
    else:
        assert_gemm_close(to_np(y1), exact, "f16", what=f"split-K tiles M={M} K={K} N={N} g={g}")
    y0 = torch.empty_like(y1)
    rc = lib.awq_gemm_repacked(ctypes.c_void_p(xt.data_ptr()), K, ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(bt.data_ptr()) if bias else None,
                               ctypes.c_void_p(y0.data_ptr()), M, K, N, g, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert (y0 != y1).float().mean().item() < 0.05          # different summation order: a rounding flips here and there at most
    assert (y0.float() - y1.float()).abs().max().item() <= 2.0 ** (np.floor(np.log2(max(1e-3, float(y1.float().abs().max())))) - 9)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.prepare_stream_workspaces(side, torch.device(DEV))
        warm = ops.awq_gemm_repacked(xt, packed, K, N, g, bias=bt)
    torch.cuda.current_stream().wait_stream(side)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=side):
        cap = ops.awq_gemm_repacked(xt, packed, K, N, g, bias=bt)
    gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap, y1) and torch.equal(warm, y1)


@pytest.mark.parametrize("M,K,N", [(128, 11008, 4096), (96, 4096, 4096), (200, 2048, 1024)])
def test_op_on_checkpoint_tensors_middle_rows_vs_oracle(ops, M, K, N):
    """`sgl_kernel.awq_gemm` on checkpoint tensors, cache off, 33 rows and up: the op re-lays the weight out into the FRONT of its workspace
    and then runs the fragment-major kernels — whose split-K tile route wants scratch of its own.  The two must not share bytes (the route
    gets what lies behind the copy, or none): results against the oracle, and equal to the persistent-copy path."""
    g = 128
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + K)
    x = synth.make_activations(M, K, "f16", "A", seed=M)
    tq, ts, tz, tx = (to_torch(t, DEV) for t in (qw, s, qz, x))
    ops.awq_gemm_cache_enable(False)
    y_op = ops.awq_gemm(tx, tq, ts, tz, 1)
    _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
    assert_gemm_close(to_np(y_op), exact, "f16", what=f"op, on-the-fly re-layout, M={M} K={K} N={N}")
    packed = ops.awq_repack(tq, ts, tz)
    y_rp = ops.awq_gemm_repacked(tx, packed, K, N, g)
    assert_gemm_close(to_np(y_rp), exact, "f16", what=f"persistent copy, M={M} K={K} N={N}")
    # (the two may take different slice counts — the op's scratch is partly taken by the copy — so: equal up to fp32 summation order)
    assert (y_op != y_rp).float().mean().item() < 0.05


def test_moe_sum_matches_torch_and_skips_padded_pairs():
    """awq_aux_moe_sum (the MoE combine: fp32 sum over each token's pairs, one rounding) against the two torch launches it replaces; pairs
    whose expert id is outside [0, E) are skipped even when their rows hold NaN (the block / tile routes never write those rows)."""
    lib = _lib.load()
    T, top_k, K, E = 37, 6, 2048, 8
    g = torch.Generator(device=DEV).manual_seed(3)
    y = torch.randn(T * top_k, K, device=DEV, generator=g).half()
    ids = torch.randint(0, E, (T * top_k,), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
    ids[5] = -1; ids[6 * top_k:7 * top_k] = -1; ids[11] = E + 3
    y[5] = float("nan"); y[6 * top_k:7 * top_k] = float("nan"); y[11] = float("inf")
    out = torch.empty(T, K, dtype=torch.float16, device=DEV)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.awq_aux_moe_sum(vp(y), vp(out), T, top_k, K, vp(ids), E, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    valid = ((ids >= 0) & (ids < E)).view(T, top_k, 1)
    want = torch.where(valid, y.view(T, top_k, K).float(), torch.zeros((), device=DEV)).sum(dim=1)
    got = out.float()
    assert torch.isfinite(got).all() and torch.count_nonzero(got[6]) == 0
    ulp = 2.0 ** (torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -14))) - 10)
    assert ((got - want).abs() <= 0.51 * ulp + 1e-6).all()               # one rounding of the fp32 sum (summation order may differ in the last fp32 bit)
    out2 = torch.empty_like(out)
    assert lib.awq_aux_moe_sum(vp(y[: T * top_k]), vp(out2), T, top_k, K, None, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert torch.isnan(out2[0]).all() and torch.isnan(out2[6]).all()           # (without ids nothing is skipped: token 0 holds pair 5's NaN)
