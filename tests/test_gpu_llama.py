"""Decode harness on a tiny Llama: the AWQ-linear model against an fp32 PyTorch
reference built from the SAME dequantised weights (awq_dequantize is bit-exact, see test_gpu_parity),
over several decode steps so the KV cache, RoPE positions and the graph replay are exercised."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_step(model, W, tokens, pos, kc, vc):
    """fp32 reference of LlamaForCausalLM.logits with dense weights W[layer][name] ([K, N] fp32)."""
    import torch.nn.functional as F

    cfg = model.cfg
    h = model.embed_tokens[tokens].float()
    B = tokens.shape[0]
    D = cfg.head_dim
    cos = model.cos_table[pos].unsqueeze(1)
    sin = model.sin_table[pos].unsqueeze(1)

    def norm(x, w):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps) * w.float()

    def rope(x):
        d = D // 2
        x1, x2 = x[..., :d], x[..., d:]
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)

    for li, layer in enumerate(model.layers):
        x = norm(h, layer.input_layernorm).half().float()
        qkv = (x @ W[li]["qkv"]).half().float()
        q, k, v = qkv.split([layer.q_size, layer.kv_size, layer.kv_size], -1)
        q = rope(q.view(B, layer.num_heads, D)).half().float()
        k = rope(k.view(B, layer.num_kv_heads, D)).half().float()
        v = v.view(B, layer.num_kv_heads, D)
        for b in range(B):
            kc[li][b, :, pos[b]] = k[b]
            vc[li][b, :, pos[b]] = v[b]
        rep = layer.num_heads // layer.num_kv_heads
        out = torch.empty(B, layer.num_heads, D, device=h.device)
        for b in range(B):
            L = int(pos[b]) + 1
            kk = kc[li][b, :, :L].repeat_interleave(rep, 0)
            vv = vc[li][b, :, :L].repeat_interleave(rep, 0)
            att = torch.softmax((q[b].unsqueeze(1) @ kk.transpose(1, 2)) * D ** -0.5, -1)
            out[b] = (att @ vv).squeeze(1)
        o = (out.reshape(B, -1).half().float() @ W[li]["o"]).half().float()
        h = (h.half() + o.half()).float()
        x = norm(h, layer.post_attention_layernorm).half().float()
        gu = (x @ W[li]["gate_up"]).half().float()
        half = gu.shape[-1] // 2
        act = (F.silu(gu[:, :half]).half() * gu[:, half:].half()).float()
        d = (act @ W[li]["down"]).half().float()
        h = (h.half() + d.half()).float()
    h = norm(h, model.norm).half().float()
    return h @ model.lm_head.float().t()


def test_tiny_llama_decode_matches_fp32_reference_and_graph_replay():
    from sglang_awq_amd import ops
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=64)
    quant = AWQConfig(4, 128, True)
    with torch.device(DEV):
        model = LlamaForCausalLM(cfg, quant, max_batch=20, max_seq=32)
    model.init_synthetic_(seed=3)
    W = []
    for layer in model.layers:
        W.append({name: ops.awq_dequantize(lin.qweight, lin.scales, lin.qzeros).float()
                  for name, lin in (("qkv", layer.qkv_proj), ("o", layer.o_proj), ("gate_up", layer.gate_up_proj), ("down", layer.down_proj))})
    # batch 3: norm prologue + SiLU-mul epilogue inside the GEMVs; 6: separate norm launch, fused epilogue; 20: all separate
    for B in (3, 6, 20):
        kc = [torch.zeros(B, l.num_kv_heads, 32, cfg.head_dim, device=DEV) for l in model.layers]
        vc = [torch.zeros(B, l.num_kv_heads, 32, cfg.head_dim, device=DEV) for l in model.layers]
        with torch.no_grad():
            for fused in (False, True):              # plain torch neighbours, then the fused aux kernels
                model.fused_aux = fused
                for layer in model.layers:
                    layer.k_cache.zero_(); layer.v_cache.zero_()
                for t_ in kc + vc:
                    t_.zero_()
                tokens = (torch.arange(B, device=DEV) * 37 + 5) % cfg.vocab_size
                pos = torch.arange(B, device=DEV) % 3          # ragged start positions (cache slots below are zeros)
                for step in range(4):
                    got = model.logits(tokens, pos).float()
                    want = _ref_step(model, W, tokens, pos, kc, vc)
                    scale = want.abs().max().item()
                    assert (got - want).abs().max().item() <= 2e-2 * scale + 2e-2, f"B={B} fused={fused} step {step}"
                    tokens = want.argmax(-1)
                    pos = pos + 1

    # graph replay produces the same token stream as eager stepping
    for layer in model.layers:
        layer.k_cache.zero_(); layer.v_cache.zero_()
    eager = GraphedDecoder(model, batch=2)
    eager.tokens.copy_(torch.tensor([7, 9], device=DEV))
    seq_eager = [eager.run(1) for _ in range(5)]
    for layer in model.layers:
        layer.k_cache.zero_(); layer.v_cache.zero_()
    graphed = GraphedDecoder(model, batch=2)
    graphed.tokens.copy_(torch.tensor([7, 9], device=DEV))
    graphed.capture(warmup=0)
    # capture itself ran no step (warmup=0); the graph holds exactly one
    seq_graph = [graphed.run(1) for _ in range(5)]
    assert seq_graph == seq_eager


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 8), (128, 8, 2), (64, 12, 4)])
def test_decode_attention_kernel_vs_fp32_reference(D, Hq, Hkv):
    """awq_aux_decode_attention (RoPE + KV write + one-token attention) against plain fp32 PyTorch: first token
    (pos 0), ragged positions across the batch, positions beyond one pass of the 4-deep position loop, GQA."""
    from sglang_awq_amd import aux_ops

    torch.manual_seed(D + Hq)
    B, S = 5, 300
    pos = torch.tensor([0, 1, 17, 130, 299], dtype=torch.int64, device=DEV)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=DEV).half()
    kc = (torch.randn(B + 1, Hkv, S, D, device=DEV)).half()        # one more batch slot than used: must stay untouched
    vc = (torch.randn(B + 1, Hkv, S, D, device=DEV)).half()
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32, device=DEV) / D))
    freqs = torch.outer(torch.arange(S, dtype=torch.float32, device=DEV), inv)
    cos_t, sin_t = freqs.cos().contiguous(), freqs.sin().contiguous()
    for b_ in range(B):                        # rows past pos are never read: poison them
        kc[b_, :, int(pos[b_]) + 1:] = float("nan")
        vc[b_, :, int(pos[b_]) + 1:] = float("nan")
    kc0, vc0, qkv0 = kc.clone(), vc.clone(), qkv.clone()

    out = aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kc, vc, Hq, Hkv, D)
    torch.cuda.synchronize()
    # split-S: same result up to fp32 summation order, identical cache writes, tickets left at zero (second call works)
    for splits in (2, 5, 16):
        kc2, vc2 = kc0.clone(), vc0.clone()
        for rep in range(2):
            out_s = aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kc2, vc2, Hq, Hkv, D, num_splits=splits)
            assert (out_s.float() - out.float()).abs().max().item() <= 2e-3, f"splits={splits} rep={rep}"
        assert torch.equal(kc2.view(torch.int16), kc.view(torch.int16)) and torch.equal(vc2.view(torch.int16), vc.view(torch.int16))
    assert torch.equal(qkv, qkv0)                                    # input not modified

    def rope(x, c, s):
        d = D // 2
        x1, x2 = x[..., :d].float(), x[..., d:].float()
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1).half()

    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], -1)
    rep = Hq // Hkv
    for b in range(B):
        p = int(pos[b])
        c, s = cos_t[p], sin_t[p]
        qb = rope(q[b].view(Hq, D), c, s)
        kb = rope(k[b].view(Hkv, D), c, s)
        vb = v[b].view(Hkv, D)
        assert torch.equal(kc[b, :, p], kb) and torch.equal(vc[b, :, p], vb)            # cache write, bit-exact
        keep = torch.ones(S, dtype=torch.bool, device=DEV)
        keep[p] = False
        assert torch.equal(kc[b][:, keep].view(torch.int16), kc0[b][:, keep].view(torch.int16))
        assert torch.equal(vc[b][:, keep].view(torch.int16), vc0[b][:, keep].view(torch.int16))
        kk = kc[b, :, :p + 1].float().repeat_interleave(rep, 0)
        vv = vc[b, :, :p + 1].float().repeat_interleave(rep, 0)
        att = torch.softmax((qb.float().unsqueeze(1) @ kk.transpose(1, 2)) * D ** -0.5, -1)
        want = (att @ vv).squeeze(1).reshape(-1)
        err = (out[b].float() - want).abs().max().item()
        assert err <= 2e-3 + 2e-3 * want.abs().max().item(), f"b={b} pos={p} err={err}"
    assert torch.equal(kc[B], kc0[B]) and torch.equal(vc[B], vc0[B])
    assert torch.isfinite(out).all()


def test_argmax_advance_kernel_matches_torch():
    """Greedy tail of a decode step: first-maximum argmax over fp16 logits (ties, -inf rows, ragged vocabulary) + pos += 1."""
    from sglang_awq_amd import aux_ops

    torch.manual_seed(5)
    for V in (32000, 512, 8200):
        B = 5
        logits = torch.randn(B, V, device=DEV).half()
        logits[1, 17] = logits[1].max() + 1
        logits[1, V - 3] = logits[1, 17]                         # tie: the first index wins
        logits[2] = float("-inf")
        logits[2, V - 1] = -1000.0
        logits[3, :] = 0.5                                        # all equal -> index 0
        tokens = torch.zeros(B, dtype=torch.int64, device=DEV)
        pos = torch.arange(B, dtype=torch.int64, device=DEV) * 3
        aux_ops.argmax_advance(logits, tokens, pos)
        assert torch.equal(tokens, logits.float().argmax(-1))
        assert torch.equal(pos, torch.arange(B, device=DEV) * 3 + 1)
