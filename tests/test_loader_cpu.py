"""AutoAWQ checkpoint ingestion on CPU: a synthetic model directory in the on-disk format
(config.json with quantization_config, safetensors with HF names) loaded through the stacked-parameter mapping
and the TP-sharding weight loaders."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from sglang_awq_amd import distributed as tpd
from sglang_awq_amd import synth
from sglang_awq_amd.loader import (iterate_safetensors, load_llama_awq, load_llama_config, load_quant_config, load_weights)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a).copy())


HF = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
          vocab_size=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=64)


def _write_checkpoint(path, quant_in_config=True, biases=None):
    """biases: None, "qwen2" (q / k / v biases, model_type qwen2) or "llama" (attention_bias + mlp_bias: every projection)."""
    os.makedirs(path, exist_ok=True)
    hf = dict(HF)
    if biases == "qwen2":
        hf["model_type"] = "qwen2"
    elif biases == "llama":
        hf["attention_bias"], hf["mlp_bias"] = True, True
    q = {"quant_method": "awq", "bits": 4, "group_size": 128, "zero_point": True, "version": "gemm"}
    if quant_in_config:
        hf["quantization_config"] = q
    else:
        with open(os.path.join(path, "quant_config.json"), "w") as f:
            json.dump({"w_bit": 4, "q_group_size": 128, "zero_point": True}, f)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf, f)
    head = HF["hidden_size"] // HF["num_attention_heads"]
    tensors, truth = {}, {}
    tensors["model.embed_tokens.weight"] = torch.randn(HF["vocab_size"], HF["hidden_size"]).half()
    tensors["model.norm.weight"] = torch.rand(HF["hidden_size"]).half()
    tensors["lm_head.weight"] = torch.randn(HF["vocab_size"], HF["hidden_size"]).half()
    for li in range(HF["num_hidden_layers"]):
        shapes = {"self_attn.q_proj": (HF["hidden_size"], HF["num_attention_heads"] * head),
                  "self_attn.k_proj": (HF["hidden_size"], HF["num_key_value_heads"] * head),
                  "self_attn.v_proj": (HF["hidden_size"], HF["num_key_value_heads"] * head),
                  "self_attn.o_proj": (HF["num_attention_heads"] * head, HF["hidden_size"]),
                  "mlp.gate_proj": (HF["hidden_size"], HF["intermediate_size"]),
                  "mlp.up_proj": (HF["hidden_size"], HF["intermediate_size"]),
                  "mlp.down_proj": (HF["intermediate_size"], HF["hidden_size"])}
        for i, (mod, (K, N)) in enumerate(shapes.items()):
            qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", 100 * li + i)
            base = f"model.layers.{li}.{mod}"
            tensors[base + ".qweight"], tensors[base + ".scales"], tensors[base + ".qzeros"] = _t(qw), _t(s), _t(qz)
            truth[base] = (qw, s, qz)
            if biases == "llama" or (biases == "qwen2" and mod in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")):
                tensors[base + ".bias"] = _t(synth.make_bias(N, "f16", 7 * li + i))
                truth[base + ".bias"] = tensors[base + ".bias"].numpy()
        tensors[f"model.layers.{li}.input_layernorm.weight"] = torch.rand(HF["hidden_size"]).half()
        tensors[f"model.layers.{li}.post_attention_layernorm.weight"] = torch.rand(HF["hidden_size"]).half()
        tensors[f"model.layers.{li}.self_attn.rotary_emb.inv_freq"] = torch.rand(head // 2)      # must be skipped
    # two shards, as real checkpoints are
    names = sorted(tensors)
    save_file({k: tensors[k] for k in names[::2]}, os.path.join(path, "model-00001-of-00002.safetensors"))
    save_file({k: tensors[k] for k in names[1::2]}, os.path.join(path, "model-00002-of-00002.safetensors"))
    return tensors, truth


@pytest.mark.parametrize("quant_in_config", [True, False])
def test_load_awq_llama_directory_tp1(tmp_path, quant_in_config):
    torch.manual_seed(0)
    tensors, truth = _write_checkpoint(str(tmp_path), quant_in_config)
    q = load_quant_config(str(tmp_path))
    assert (q.weight_bits, q.group_size, q.zero_point) == (4, 128, True)
    cfg = load_llama_config(str(tmp_path))
    assert cfg.num_key_value_heads == 2 and cfg.head_dim == 64
    assert len(dict(iterate_safetensors(str(tmp_path)))) == len(tensors)

    model = load_llama_awq(str(tmp_path), device=None, max_batch=2, max_seq=16)
    assert torch.equal(model.embed_tokens, tensors["model.embed_tokens.weight"])
    assert torch.equal(model.lm_head, tensors["lm_head.weight"]) and torch.equal(model.norm, tensors["model.norm.weight"])
    for li, layer in enumerate(model.layers):
        qs = [truth[f"model.layers.{li}.self_attn.{n}_proj"] for n in "qkv"]
        assert np.array_equal(layer.qkv_proj.qweight.numpy(), np.concatenate([t[0] for t in qs], 1))
        assert np.array_equal(layer.qkv_proj.scales.numpy(), np.concatenate([t[1] for t in qs], 1))
        assert np.array_equal(layer.qkv_proj.qzeros.numpy(), np.concatenate([t[2] for t in qs], 1))
        g, u = truth[f"model.layers.{li}.mlp.gate_proj"], truth[f"model.layers.{li}.mlp.up_proj"]
        assert np.array_equal(layer.gate_up_proj.qweight.numpy(), np.concatenate([g[0], u[0]], 1))
        for mod, lin in (("self_attn.o_proj", layer.o_proj), ("mlp.down_proj", layer.down_proj)):
            t = truth[f"model.layers.{li}.{mod}"]
            assert np.array_equal(lin.qweight.numpy(), t[0]) and np.array_equal(lin.scales.numpy(), t[1]) and np.array_equal(lin.qzeros.numpy(), t[2])
        assert torch.equal(layer.input_layernorm, tensors[f"model.layers.{li}.input_layernorm.weight"])
        assert getattr(layer.qkv_proj, "awq_packed", None) is None        # CPU tensors: no repacked copy is made


def test_load_awq_llama_directory_tp2_shards(tmp_path):
    """Rank 1 of 2: q / k / v by heads, gate / up by halves (column parallel), o / down by rows."""
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.llama import LlamaForCausalLM

    torch.manual_seed(1)
    _, truth = _write_checkpoint(str(tmp_path))
    tpd.set_tensor_parallel_group(tpd.TensorParallelGroup(None, 1, 2))
    try:
        model = LlamaForCausalLM(load_llama_config(str(tmp_path)), load_quant_config(str(tmp_path)), max_batch=1, max_seq=8)
        stats = load_weights(model, iterate_safetensors(str(tmp_path)))
    finally:
        tpd.set_tensor_parallel_group(tpd.TensorParallelGroup(None, 0, 1))
    assert stats["skipped"] == HF["num_hidden_layers"]                    # the rotary inv_freq tensors
    layer = model.layers[1]
    q, k, v = (truth[f"model.layers.1.self_attn.{n}_proj"] for n in "qkv")
    want_scales = np.concatenate([q[1][:, 128:256], k[1][:, 64:128], v[1][:, 64:128]], 1)     # 2 q heads + 1 kv head of rank 1
    assert np.array_equal(layer.qkv_proj.scales.numpy(), want_scales)
    g, u = truth["model.layers.1.mlp.gate_proj"], truth["model.layers.1.mlp.up_proj"]
    assert np.array_equal(layer.gate_up_proj.qweight.numpy(), np.concatenate([g[0][:, 32:64], u[0][:, 32:64]], 1))
    d = truth["model.layers.1.mlp.down_proj"]
    assert np.array_equal(layer.down_proj.qweight.numpy(), d[0][256:512]) and np.array_equal(layer.down_proj.qzeros.numpy(), d[2][2:4])
    o = truth["model.layers.1.self_attn.o_proj"]
    assert np.array_equal(layer.o_proj.scales.numpy(), o[1][1:2])


def test_loader_rejects_non_awq_and_missing_config(tmp_path):
    with open(tmp_path / "config.json", "w") as f:
        json.dump(dict(HF, quantization_config={"quant_method": "gptq", "bits": 4}), f)
    with pytest.raises(ValueError, match="not an AWQ checkpoint"):
        load_quant_config(str(tmp_path))
    with open(tmp_path / "config.json", "w") as f:
        json.dump(HF, f)
    with pytest.raises(ValueError, match="no AWQ quantisation config"):
        load_quant_config(str(tmp_path))
    with pytest.raises(ValueError, match="safetensors"):
        list(iterate_safetensors(str(tmp_path)))


@pytest.mark.parametrize("biases", ["qwen2", "llama"])
def test_bias_tensors_load_and_shard(tmp_path, biases):
    """`*.bias` names go through the stacked mapping and the parameters' weight loaders like the weights (reference:
    linear.py:348-358, 1286-1293 — `output_dim: 0` for column-parallel layers, replicated for row-parallel ones).  TP = 1: the
    fused qkv / gate_up biases are the concatenations; rank 1 of 2: q by heads, k / v by KV heads, gate / up by halves, the
    o_proj / down_proj biases whole (rank 0 alone adds them, linear.py:1401)."""
    from sglang_awq_amd.llama import LlamaForCausalLM

    torch.manual_seed(2)
    _, truth = _write_checkpoint(str(tmp_path), biases=biases)
    cfg = load_llama_config(str(tmp_path))
    assert cfg.qkv_bias and cfg.o_bias == (biases == "llama") and cfg.mlp_bias == (biases == "llama")
    model = load_llama_awq(str(tmp_path), device=None, max_batch=1, max_seq=8)
    b = lambda li, mod: truth[f"model.layers.{li}.{mod}.bias"]
    for li, layer in enumerate(model.layers):
        want = np.concatenate([b(li, f"self_attn.{n}_proj") for n in "qkv"])
        assert np.array_equal(layer.qkv_proj.bias.numpy(), want)
        if biases == "llama":
            assert np.array_equal(layer.o_proj.bias.numpy(), b(li, "self_attn.o_proj"))
            assert np.array_equal(layer.gate_up_proj.bias.numpy(), np.concatenate([b(li, "mlp.gate_proj"), b(li, "mlp.up_proj")]))
            assert np.array_equal(layer.down_proj.bias.numpy(), b(li, "mlp.down_proj"))
        else:
            assert layer.o_proj.bias is None and layer.gate_up_proj.bias is None and layer.down_proj.bias is None
    tpd.set_tensor_parallel_group(tpd.TensorParallelGroup(None, 1, 2))
    try:
        m2 = LlamaForCausalLM(cfg, load_quant_config(str(tmp_path)), max_batch=1, max_seq=8)
        stats = load_weights(m2, iterate_safetensors(str(tmp_path)))
    finally:
        tpd.set_tensor_parallel_group(tpd.TensorParallelGroup(None, 0, 1))
    assert stats["skipped"] == HF["num_hidden_layers"]                    # only the rotary inv_freq tensors: no bias was dropped
    layer = m2.layers[1]
    want = np.concatenate([b(1, "self_attn.q_proj")[128:256], b(1, "self_attn.k_proj")[64:128], b(1, "self_attn.v_proj")[64:128]])
    assert np.array_equal(layer.qkv_proj.bias.numpy(), want)
    if biases == "llama":
        assert np.array_equal(layer.gate_up_proj.bias.numpy(), np.concatenate([b(1, "mlp.gate_proj")[256:512], b(1, "mlp.up_proj")[256:512]]))
        assert np.array_equal(layer.o_proj.bias.numpy(), b(1, "self_attn.o_proj"))          # replicated
        assert np.array_equal(layer.down_proj.bias.numpy(), b(1, "mlp.down_proj"))
