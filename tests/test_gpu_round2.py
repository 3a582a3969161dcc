"""Round-2 parity additions (-m gpu): the Llama-2-70B TP = 8 per-rank shapes and the 7B o_proj through both public routes,
the drop-in op's repacked-copy cache (bit-identity, invalidation, capture behaviour), two streams at once, the decode-harness
neighbour kernels on their own, awq_dequantize at the DeepSeek kv_b_proj post-load shape, the decoder's position guard."""
import numpy as np
import pytest
import torch

from oracle import awq_ref, c_oracle
from sglang_awq_amd import _lib, synth
from tests.util import assert_gemm_close, bits, to_np, to_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sglang_awq_amd import ops as _ops   # raises if the HIP library is missing: no fallback

    _lib.load()
    # the op's repacked-copy cache is opt-in (ops.py: off until the reload hooks of sgl_kernel_compat.install() are in place or
    # the caller takes responsibility for invalidation): these tests own their weights and clear the cache themselves
    _ops.awq_gemm_cache_enable(True)
    return _ops


def _dev(*arrs):
    return [to_torch(a, DEV) for a in arrs]


# BASELINE configs[4]: Llama-2-70B at TP = 8, per-rank (K, N): qkv, o, gate_up, down; plus the 7B o_proj (linear.py:832-837)
SHAPES = [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192), (4096, 4096)]


@pytest.mark.parametrize("K,N", SHAPES)
def test_tp8_per_rank_shapes_via_apply_and_op(ops, K, N):
    """AWQLinearMethod.apply (repacked copy made at load) and torch.ops.sgl_kernel.awq_gemm (checkpoint tensors; cache on
    and off) against the exact-sum oracle at decode batch sizes, and against each other bit for bit where they run the
    same kernel."""
    from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod

    qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", seed=K + N)
    method = AWQLinearMethod(AWQConfig(4, 128, True))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.float16, weight_loader=None)
    layer.qweight.data.copy_(to_torch(qw)); layer.qzeros.data.copy_(to_torch(qz)); layer.scales.data.copy_(to_torch(s))
    layer.to(DEV)
    method.process_weights_after_loading(layer)
    assert layer.awq_packed is not None
    for M in (1, 8, 32):
        x = synth.make_activations(M, K, "f16", "A", seed=M + K)
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        xt = to_torch(x, DEV)
        y_apply = method.apply(layer, xt)
        assert_gemm_close(to_np(y_apply), exact, "f16", what=f"apply M={M} K={K} N={N}")
        ops.awq_gemm_cache_enable(True)
        y_op = torch.ops.sgl_kernel.awq_gemm(xt, layer.qweight, layer.scales, layer.qzeros, 1)
        assert torch.equal(y_op, y_apply), f"op (cached repacked copy) != apply, M={M}"
        ops.awq_gemm_cache_enable(False)
        try:
            y_ck = torch.ops.sgl_kernel.awq_gemm(xt, layer.qweight, layer.scales, layer.qzeros, 1)
        finally:
            ops.awq_gemm_cache_enable(True)
        assert_gemm_close(to_np(y_ck), exact, "f16", what=f"op on the checkpoint layout M={M} K={K} N={N}")
    ops.awq_gemm_cache_clear()


def test_op_cache_invalidation_and_capture(ops):
    """The op's repacked copy follows the weight: an in-place update (version counter), a freed-and-reused address
    (storage weak reference) and awq_gemm_cache_clear() all lead to fresh results; a miss during stream capture does not
    fill the cache (and still computes the right thing through the checkpoint-layout kernel)."""
    K, N = 1024, 2048
    x = synth.make_activations(4, K, "f16", "A", 9)
    xt = to_torch(x, DEV)
    w1 = synth.make_awq_weights(K, N, 128, "f16", "A", 101)
    w2 = synth.make_awq_weights(K, N, 128, "f16", "A", 202)
    ops.awq_gemm_cache_enable(True)
    ops.awq_gemm_cache_clear()
    qw, s, qz = _dev(*w1)
    y1 = ops.awq_gemm(xt, qw, s, qz, 1)
    assert ops.awq_gemm_cache_info()["entries"] == 1
    assert torch.equal(y1, ops.awq_gemm(xt, qw, s, qz, 1)) and ops.awq_gemm_cache_info()["entries"] == 1
    _, e1 = c_oracle.gemm(x, *w1, want_exact=True)
    _, e2 = c_oracle.gemm(x, *w2, want_exact=True)
    assert_gemm_close(to_np(y1), e1, "f16", what="cached op, first weights")
    # in-place update of the same tensors (what an RL weight sync does): versions change -> the copy is rebuilt
    qw.copy_(to_torch(w2[0], DEV)); s.copy_(to_torch(w2[1], DEV)); qz.copy_(to_torch(w2[2], DEV))
    y2 = ops.awq_gemm(xt, qw, s, qz, 1)
    assert_gemm_close(to_np(y2), e2, "f16", what="cached op after in-place weight update")
    assert ops.awq_gemm_cache_info()["entries"] == 1
    # free the weights, allocate new ones of the same shape (the caching allocator hands the same addresses back)
    ptrs = (qw.data_ptr(), s.data_ptr(), qz.data_ptr())
    del qw, s, qz
    qw, s, qz = _dev(*w1)
    reused = (qw.data_ptr(), s.data_ptr(), qz.data_ptr()) == ptrs
    y3 = ops.awq_gemm(xt, qw, s, qz, 1)
    assert_gemm_close(to_np(y3), e1, "f16", what=f"cached op after free + realloc (addresses reused: {reused})")
    # explicit clear
    ops.awq_gemm_cache_clear()
    info = ops.awq_gemm_cache_info()
    assert (info["entries"], info["bytes"], info["enabled"]) == (0, 0, True)
    # capture-time miss: not cached, still right
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.awq_gemm_cache_enable(False)
        ops.awq_gemm(xt, qw, s, qz, 1)                 # warm-up of the checkpoint-layout route on the capture stream (workspace)
        ops.awq_gemm_cache_enable(True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        yc = ops.awq_gemm(xt, qw, s, qz, 1)
    assert ops.awq_gemm_cache_info()["entries"] == 0
    g.replay()
    torch.cuda.synchronize()
    assert_gemm_close(to_np(yc), e1, "f16", what="op captured with an empty cache")
    # eager call fills it; a graph captured afterwards replays the repacked kernel, bit-identical to awq_gemm_repacked
    y4 = ops.awq_gemm(xt, qw, s, qz, 1)
    assert ops.awq_gemm_cache_info()["entries"] == 1
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        yc2 = ops.awq_gemm(xt, qw, s, qz, 1)
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(yc2, y4) and torch.equal(y4, ops.awq_gemm_repacked(xt, ops.awq_repack(qw, s, qz), K, N, 128))
    ops.awq_gemm_cache_clear()


def test_two_streams_do_not_share_split_k_scratch(ops):
    """The checkpoint-layout split-K kernel keeps arrival counters and fp32 slabs in a workspace: one per (device, stream).
    Two streams run M <= 16 GEMMs at the same time, many times over; every result must equal the single-stream one."""
    K, N = 4096, 4096
    ops.awq_gemm_cache_enable(False)
    try:
        wa = _dev(*synth.make_awq_weights(K, N, 128, "f16", "A", 11))
        wb = _dev(*synth.make_awq_weights(K, N, 128, "f16", "A", 12))
        xa = to_torch(synth.make_activations(3, K, "f16", "A", 13), DEV)
        xb = to_torch(synth.make_activations(16, K, "f16", "A", 14), DEV)
        ya = ops.awq_gemm(xa, *wa, 1)
        yb = ops.awq_gemm(xb, *wb, 1)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs_a, outs_b = [], []
        for _ in range(50):
            with torch.cuda.stream(s1):
                outs_a.append(ops.awq_gemm(xa, *wa, 1))
            with torch.cuda.stream(s2):
                outs_b.append(ops.awq_gemm(xb, *wb, 1))
        torch.cuda.synchronize()
        assert all(torch.equal(o, ya) for o in outs_a) and all(torch.equal(o, yb) for o in outs_b)
        keys = {k for k in ops._workspaces if k[0] == 0}
        assert len({k[1] for k in keys}) >= 3            # default stream + the two side streams each got their own
    finally:
        ops.awq_gemm_cache_enable(True)


@pytest.mark.parametrize("H", [256, 4096, 8192, 8200, 11008])
@pytest.mark.parametrize("with_delta", [False, True])
def test_add_rmsnorm_kernel_vs_fp32(H, with_delta):
    """awq_aux_add_rmsnorm on its own: 1, 2, 4 register chunks per thread (H = 256, 4096, 8192), the strided two-pass route
    (H > 8192), a ragged width (8200 = 1025 chunks of 8), with and without the residual add; fp32 torch reference with the
    kernel's rounding points (fp16 add, fp32 sum of squares, fp16(v * inv) * w)."""
    from sglang_awq_amd import aux_ops

    rows, eps = 5, 1e-5
    g = torch.Generator(device="cpu"); g.manual_seed(H + with_delta)
    h = torch.randn(rows, H, generator=g).half()
    d = (0.5 * torch.randn(rows, H, generator=g)).half()
    w = (1.0 + 0.25 * torch.randn(H, generator=g)).half()
    hd = h.to(DEV).clone()
    out = aux_ops.add_rmsnorm(hd, d.to(DEV) if with_delta else None, w.to(DEV), eps)
    v = (h + d) if with_delta else h                                   # fp16 add
    assert torch.equal(hd.cpu(), v), "h must hold h + delta afterwards (or be untouched)"
    inv = torch.rsqrt((v.double() ** 2).mean(-1, keepdim=True) + eps)
    want = ((v.double() * inv).half() * w).double()                    # fp16(v * inv) * w, fp16 product
    got = out.cpu().double()
    # v_rsq_f32 is good to ~1 ulp of fp32: fp16(v * inv) may land on the neighbouring half for a few elements
    ulp = 2.0 ** (torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -14))) - 10)
    assert ((got - want).abs() <= 1.01 * ulp * (1 + w.double().abs())).all()
    assert float((got != want).double().mean()) < 0.02


@pytest.mark.parametrize("B,I", [(1, 11008), (3, 128), (32, 28672), (7, 1024)])
def test_silu_mul_kernel_vs_torch(B, I):
    from sglang_awq_amd import aux_ops

    g = torch.Generator(device="cpu"); g.manual_seed(B * 7 + I)
    gu = (2.0 * torch.randn(B, 2 * I, generator=g)).half()
    got = aux_ops.silu_mul(gu.to(DEV)).cpu()
    gate, up = gu[:, :I].float(), gu[:, I:]
    silu = (gate / (1.0 + torch.exp(-gate))).half()                    # silu in fp32 rounded to fp16 ...
    want = silu * up                                                   # ... then an fp16 product
    diff = (got.double() - want.double()).abs()

    def ulp16(t):
        return 2.0 ** (torch.floor(torch.log2(t.double().abs().clamp_min(2.0 ** -14))) - 10)
    # __expf vs exp: the silu may land on the neighbouring half (one ulp of the silu, scaled by |up|), then one product rounding
    assert (diff <= 1.01 * (ulp16(silu) * up.double().abs() + ulp16(want))).all()
    assert float((got != want).double().mean()) < 0.02


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_dequantize_deepseek_kv_b_proj_post_load_shapes(ops, dt):
    """deepseek_v2.py:3433-3450 / longcat_flash.py:645-655 call awq_dequantize(qweight, scales, qzeros).T on kv_b_proj after
    loading: K = kv_lora_rank = 512, N = heads x (qk_nope 128 + v 128) = 32768 at TP = 1 and 4096 at TP = 8; bit-exact
    against the oracle in the scales' dtype, transposed view included."""
    for N in (32768, 4096):
        qw, s, qz = synth.make_awq_weights(512, N, 128, dt, "F", seed=N)
        w = ops.awq_dequantize(*_dev(qw, s, qz))
        want = awq_ref.awq_dequantize(qw, s, qz)
        assert np.array_equal(bits(to_np(w)), bits(want))
        assert np.array_equal(bits(to_np(w.T.contiguous())), bits(np.ascontiguousarray(want.T)))


def test_released_checkpoint_tensors(ops, monkeypatch):
    """SGLANG_AWQ_AMD_KEEP_CHECKPOINT=0: after the re-layout the checkpoint tensors are released; apply() runs every batch size
    from the repacked copy (32 rows on a wide strip included) and reproduces the kept-checkpoint results bit for bit."""
    from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod

    K, N = 1024, 2048
    qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", 77)

    def build():
        method = AWQLinearMethod(AWQConfig(4, 128, True))
        layer = torch.nn.Module()
        method.create_weights(layer, K, [N], K, N, torch.float16, weight_loader=None)
        layer.qweight.data.copy_(to_torch(qw)); layer.qzeros.data.copy_(to_torch(qz)); layer.scales.data.copy_(to_torch(s))
        layer.to(DEV)
        method.process_weights_after_loading(layer)
        return method, layer

    m_keep, l_keep = build()
    monkeypatch.setenv("SGLANG_AWQ_AMD_KEEP_CHECKPOINT", "0")
    m_rel, l_rel = build()
    assert l_rel.qweight.numel() == 0 and l_rel.scales.numel() == 0 and l_rel.awq_shape == (K, N, 128)
    b = to_torch(synth.make_bias(N, "f16", 5), DEV)
    for M in (1, 16, 32, 200, 1030):
        x = to_torch(synth.make_activations(M, K, "f16", "A", M), DEV)
        assert torch.equal(m_rel.apply(l_rel, x, b), m_keep.apply(l_keep, x, b)), M
    assert m_rel.apply(l_rel, torch.empty(0, K, dtype=torch.float16, device=DEV)).shape == (0, N)


def test_graphed_decoder_refuses_to_leave_the_kv_cache():
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=512, max_position_embeddings=64)
    with torch.device(DEV):
        model = LlamaForCausalLM(cfg, AWQConfig(4, 128, True), max_batch=2, max_seq=16)
    model.init_synthetic_(seed=1)
    with pytest.raises(ValueError):
        GraphedDecoder(model, 2, start_pos=16)
    dec = GraphedDecoder(model, 2, start_pos=10).capture(warmup=1)      # the warm-up step consumed position 10 (capture runs nothing)
    dec.run(5)                                                          # 11 .. 15: the last slot of the cache
    with pytest.raises(ValueError):
        dec.run(1)
    assert int(dec.pos.max()) == 16                                     # the device-side counter never went further
    dec.reset(3)
    dec.run(2)
    assert int(dec.pos.max()) == 5


@pytest.mark.parametrize("dt,g", [("bf16", 128), ("f16", 64), ("f16", 32), ("bf16", 64), ("bf16", 32), ("bf16", 512)])
def test_repacked_layout_bf16_and_small_groups_vs_oracle(ops, dt, g):
    """bf16 and the Triton path's small groups (g in {32, 64}: awq_triton.py:250) on the MFMA-fragment-major
    layout — awq_repack + awq_gemm_repacked over the decode range (one launch, two 16-row passes), the GEMV-pass range
    and the tiled kernel, ragged N and K, with and without bias; plus AWQLinearMethod.apply and the drop-in op's cached copy."""
    from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod

    for (K, N) in [(512, 1056), (1152, 72), (4096, 4096)]:
        if K % g:
            continue
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, "A", seed=K + N + g)
        dq, ds, dz = _dev(qw, s, qz)
        packed = ops.awq_repack(dq, ds, dz)
        assert packed is not None and packed.numel() == ops._lib.load().awq_repacked_bytes(K, N, g, 0)
        Ms = [1, 5, 16, 17, 32, 100, 300] if K < 4096 else [1, 16, 300]
        for M in Ms:
            x = synth.make_activations(M, K, dt, "A", seed=M + K)
            _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
            y = ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g)
            assert_gemm_close(to_np(y), exact, dt, what=f"repacked {dt} g={g} M={M} K={K} N={N}")
            if M in (1, 100):
                b = synth.make_bias(N, dt, 3)
                yb = ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g, to_torch(b, DEV))
                assert torch.equal(yb, y + to_torch(b, DEV))
                assert torch.equal(ops.awq_gemm(to_torch(x, DEV), dq, ds, dz, 1), y), "op (cached copy) != awq_gemm_repacked"
    ops.awq_gemm_cache_clear()
    # the linear method repacks these too
    K, N = 1024, 2048
    qw, s, qz = synth.make_awq_weights(K, N, g, dt, "A", seed=g)
    method = AWQLinearMethod(AWQConfig(4, g, True))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.bfloat16 if dt == "bf16" else torch.float16, weight_loader=None)
    layer.qweight.data.copy_(to_torch(qw)); layer.qzeros.data.copy_(to_torch(qz)); layer.scales.data.copy_(to_torch(s))
    layer.to(DEV)
    method.process_weights_after_loading(layer)
    assert layer.awq_packed is not None
    x = synth.make_activations(7, K, dt, "A", seed=1)
    _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
    assert_gemm_close(to_np(method.apply(layer, to_torch(x, DEV))), exact, dt, what=f"apply {dt} g={g}")


def test_repacked_small_group_one_hot_rows_reproduce_dequantize(ops):
    """One-hot activations through the repacked kernels select single rows of W: bit-equality with awq_dequantize for g = 32
    (every k-step its own scale / zero) in fp16 and bf16 — a wrong group index cannot hide behind a tolerance."""
    K, N, g = 512, 256, 32
    for dt in ("f16", "bf16"):
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, "F", seed=17)
        dq, ds, dz = _dev(qw, s, qz)
        W = ops.awq_dequantize(dq, ds, dz)
        packed = ops.awq_repack(dq, ds, dz)
        rows = [0, 31, 32, 63, 64, 100, 127, 128, 300, 511]
        x = torch.zeros(len(rows), K, dtype=W.dtype, device=DEV)
        for i, k in enumerate(rows):
            x[i, k] = 1.0
        assert torch.equal(ops.awq_gemm_repacked(x, packed, K, N, g), W[rows]), dt
        xb = torch.zeros(200, K, dtype=W.dtype, device=DEV)
        idx = torch.arange(200, device=DEV) * 7 % K
        xb[torch.arange(200, device=DEV), idx] = 1.0
        assert torch.equal(ops.awq_gemm_repacked(xb, packed, K, N, g), W[idx]), dt + " tiled"


@pytest.mark.parametrize("T", [1, 3, 8, 40, 300])
def test_awq_moe_method_vs_oracle(ops, T):
    """AWQMoEMethod (reference awq.py:661-852): experts on the fragment-major layout, decode batches through the
    expert-indirect GEMV (awq_aux_moe_gemv: SiLU-mul epilogue, routed weight on the fp32 sums), larger ones grouped by expert.
    Oracle: the dense linear's exact sums per (token, expert), composed in numpy in the order of the reference's fused MoE
    (fp16 gate_up -> fp16 silu * up -> fp16(weight * sum) -> fp16 sum over the token's experts).  Parity unpinned by the
    reference beyond the dense path (no MoE fixtures)."""
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.moe import AWQMoEMethod, select_experts

    E, K, I, top_k, g = 6, 2048, 1024, 2, 128
    method = AWQMoEMethod(AWQConfig(4, g, True))
    layer = torch.nn.Module()
    method.create_weights(layer, E, K, I, torch.float16)
    assert layer.w13_qweight.shape == (E, K, 2 * I // 8) and layer.w2_qweight.shape == (E, I, K // 8)
    assert layer.w13_scales.shape == (E, K // g, 2 * I) and layer.w2_qzeros.shape == (E, I // g, K // 8)
    w13 = [synth.make_awq_weights(K, 2 * I, g, "f16", "A", seed=100 + e) for e in range(E)]
    w2 = [synth.make_awq_weights(I, K, g, "f16", "A", seed=200 + e) for e in range(E)]
    for e in range(E):
        layer.w13_qweight.data[e].copy_(to_torch(w13[e][0])); layer.w13_scales.data[e].copy_(to_torch(w13[e][1])); layer.w13_qzeros.data[e].copy_(to_torch(w13[e][2]))
        layer.w2_qweight.data[e].copy_(to_torch(w2[e][0])); layer.w2_scales.data[e].copy_(to_torch(w2[e][1])); layer.w2_qzeros.data[e].copy_(to_torch(w2[e][2]))
    layer.to(DEV)
    method.process_weights_after_loading(layer)

    x = synth.make_activations(T, K, "f16", "A", seed=T, x_std=0.5)
    logits = synth.make_activations(T, E, "f16", "A", seed=T + 50).astype(np.float32)
    tw, ti = select_experts(to_torch(logits, DEV), top_k)
    assert ti.shape == (T, top_k) and torch.allclose(tw.sum(-1), torch.ones(T, device=DEV), atol=1e-6)
    out = to_np(method.apply(layer, to_torch(x, DEV), tw, ti)).astype(np.float64)
    tw_np, ti_np = tw.cpu().numpy().astype(np.float64), ti.cpu().numpy()

    want = np.zeros((T, K))
    scale = np.zeros((T, K))
    terms = np.zeros((T, K))
    absw2 = [np.abs(c_oracle.dequantize(*w2[e]).astype(np.float32)) for e in range(E)]
    for t in range(T):
        for k in range(top_k):
            e = int(ti_np[t, k])
            _, gu = c_oracle.gemm(x[t:t + 1], *w13[e], want_exact=True)
            gu = gu.astype(np.float16)
            gate, up = gu[:, :I].astype(np.float32), gu[:, I:]
            act = (gate / (1.0 + np.exp(-gate))).astype(np.float16) * up
            _, y = c_oracle.gemm(act, *w2[e], want_exact=True)
            ys = (tw_np[t, k] * y[0]).astype(np.float16).astype(np.float64)
            want[t] += ys
            scale[t] += np.abs(ys)
            terms[t] += tw_np[t, k] * (np.abs(act.astype(np.float32)) @ absw2[e])[0]      # sum_i |act_i W2[i, n]|: what a flipped act ulp acts on
    want = want.astype(np.float16).astype(np.float64)
    # the kernel's silu (__expf) and its fp32 gate sums (summation order differs per route) may land some act elements on a
    # neighbouring half: one fp16 ulp (2^-11 relative) of a share of the w2 terms — bounded through sum_i |act_i W2[i, n]|, which a
    # cancelling w2 sum can be far below; each slot and the final sum round once more
    tol = 2.5 * (2.0 ** (np.floor(np.log2(np.maximum(scale, 2.0 ** -14))) - 10)) + 2e-3 * (1.0 + scale) + 2.0 ** -13 * terms
    err = np.abs(out - want)
    worst = int((err / tol).argmax())
    assert np.all(err <= tol), (f"T={T}: {int((err > tol).sum())} of {err.size} outside; worst err {err.flat[worst]:.3e} (tolerance {tol.flat[worst]:.3e}) "
                                f"at scale {scale.flat[worst]:.3f}, element {np.unravel_index(worst, err.shape)}, got {out.flat[worst]:.5f} want {want.flat[worst]:.5f}")
    # every route applies the routed weight to the fp32 sums before the one rounding (slot route up to MOE_SLOT_MAX_PAIRS pairs,
    # expert-sorted 16-row blocks beyond): the same share of one-ulp differences on both
    assert float((out != want).mean()) < 0.25
    # no route holds a host synchronisation: each must capture into a graph and replay to the same bits
    xt = to_torch(x, DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = method.apply(layer, xt, tw, ti)
    torch.cuda.current_stream().wait_stream(side)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=side):
        captured = method.apply(layer, xt, tw, ti)
    gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)


def test_split_k_gemv_on_deep_narrow_matrices_vs_oracle(ops):
    """9 .. 32 rows on a deep, narrow matrix (K >= 8192, N <= 4096): `awq_gemm_repacked_ws` runs wide strips with K split across
    workgroups (awq_repacked_splitk.hip; fp32 partials through the per-stream workspace, last workgroup to arrive adds them in slice
    order).  Against the oracle; repeated calls are bit-identical (fixed summation order, counters left at zero); the no-workspace
    entry point gives the same values to fp32 summation order; the route survives graph capture."""
    import ctypes

    lib = _lib.load()
    for (M, K, N, g) in [(9, 8192, 1280, 128), (16, 11008, 4096, 128), (17, 8192, 4096, 128), (32, 11008, 4096, 128),
                         (32, 8192, 1032, 128), (12, 8192, 2048, 8192)]:
        assert lib.awq_gemm_repacked_workspace_bytes(M, K, N, g, 0) > 0
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + K + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + 7)
        packed = ops.awq_repack(*[to_torch(t, DEV) for t in (qw, s, qz)])
        xt = to_torch(x, DEV)
        y1 = ops.awq_gemm_repacked(xt, packed, K, N, g)
        y2 = ops.awq_gemm_repacked(xt, packed, K, N, g)
        assert torch.equal(y1, y2), f"split-K M={M} K={K} N={N}: not run-to-run identical"
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        assert_gemm_close(to_np(y1), exact, "f16", what=f"split-K M={M} K={K} N={N} g={g}")
        # the entry point without scratch (one strip per workgroup): same values up to the order of the fp32 sums
        y0 = torch.empty_like(y1)
        rc = lib.awq_gemm_repacked(ctypes.c_void_p(xt.data_ptr()), K, ctypes.c_void_p(packed.data_ptr()), None, ctypes.c_void_p(y0.data_ptr()),
                                   M, K, N, g, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert_gemm_close(to_np(y0), exact, "f16", what=f"one-strip M={M} K={K} N={N}")
        assert (y0 != y1).float().mean().item() < 0.02          # different summation order: a rounding flips here and there at most
    # graph capture: the workspace of the capturing stream is allocated by the eager warm-up call
    M, K, N, g = 24, 11008, 4096, 128
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=5)
    packed = ops.awq_repack(*[to_torch(t, DEV) for t in (qw, s, qz)])
    xt = to_torch(synth.make_activations(M, K, "f16", "A", seed=6), DEV)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        want = ops.awq_gemm_repacked(xt, packed, K, N, g)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            got = ops.awq_gemm_repacked(xt, packed, K, N, g)
        for _ in range(3):
            graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_random_shapes_every_dispatch_route(ops):
    """A seeded subset of tools/fuzz_gpu.py: random (M, K, N, group size, dtype) through the op, the repacked entry point, strided
    rows and the bias epilogue, against the C oracle — GEMV, split-K, passes, K-split tiles, pipelined tiles, bf16 / small groups."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_gpu.py")
    spec = importlib.util.spec_from_file_location("fuzz_gpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(60, 11, verbose=False) == 60
