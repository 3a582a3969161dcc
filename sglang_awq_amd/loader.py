"""AutoAWQ checkpoint ingestion for the decode harness.

What the reference does, restated for a local model directory (there is no network here):
  * quantisation config: `config.json["quantization_config"]` first, else the first of
    AWQConfig.get_config_filenames() found in the directory (model_loader/weight_utils.py:164-260,
    awq.py:142-158: keys `w_bit|bits`, `q_group_size|group_size`, `zero_point`, `modules_to_not_convert`);
  * tensors: every `*.safetensors` file, names as AutoAWQ writes them
    (`model.layers.N.self_attn.q_proj.qweight|qzeros|scales`, ...), opened with safetensors (no pickle);
  * stacked parameters: q/k/v -> `qkv_proj` shards "q"/"k"/"v", gate/up -> `gate_up_proj` shards 0/1
    (models/llama.py:560-632); each parameter's `weight_loader` does the tensor-parallel slicing
    (parameter.py, linear.py `weight_loader_v2`); `*.bias` tensors go the same way (a q/k/v bias is a shard of
    `qkv_proj.bias`, column-sharded; an o_proj / down_proj bias is replicated: linear.py:348-358, 1286-1293);
  * afterwards `process_weights_after_loading` on every linear (model_loader/loader.py:616-632).
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Iterable, Iterator, Optional, Tuple

import torch

from .awq import AWQConfig
from .llama import LlamaConfig, LlamaForCausalLM

# (parameter name part, checkpoint name part, shard id) — models/llama.py:561-568
STACKED_PARAMS_MAPPING = [
    (".qkv_proj", ".q_proj", "q"),
    (".qkv_proj", ".k_proj", "k"),
    (".qkv_proj", ".v_proj", "v"),
    (".gate_up_proj", ".gate_proj", 0),
    (".gate_up_proj", ".up_proj", 1),
]


def load_quant_config(model_dir: str) -> AWQConfig:
    cfg_path = os.path.join(model_dir, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as f:
            hf = json.load(f)
        q = hf.get("quantization_config")
        if q is not None:
            if q.get("quant_method", "awq") != "awq":
                raise ValueError(f"not an AWQ checkpoint: quant_method={q.get('quant_method')!r}")
            return AWQConfig.from_config(q)
    for name in AWQConfig.get_config_filenames():
        p = os.path.join(model_dir, name)
        if os.path.exists(p):
            with open(p) as f:
                return AWQConfig.from_config(json.load(f))
    raise ValueError(f"no AWQ quantisation config found in {model_dir}")


def load_llama_config(model_dir: str) -> LlamaConfig:
    with open(os.path.join(model_dir, "config.json")) as f:
        hf = json.load(f)
    return LlamaConfig(hidden_size=hf["hidden_size"], intermediate_size=hf["intermediate_size"],
                       num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=hf["num_attention_heads"],
                       num_key_value_heads=hf.get("num_key_value_heads", hf["num_attention_heads"]), vocab_size=hf["vocab_size"],
                       rms_norm_eps=hf.get("rms_norm_eps", 1e-5), rope_theta=hf.get("rope_theta", 10000.0),
                       max_position_embeddings=hf.get("max_position_embeddings", 4096),
                       # HF Llama: attention_bias covers q/k/v/o, mlp_bias gate/up/down; Qwen2 always has q/k/v biases
                       qkv_bias=bool(hf.get("attention_bias", False)) or hf.get("model_type") == "qwen2",
                       o_bias=bool(hf.get("attention_bias", False)) and hf.get("model_type") != "qwen2",
                       mlp_bias=bool(hf.get("mlp_bias", False)))


def iterate_safetensors(model_dir: str) -> Iterator[Tuple[str, torch.Tensor]]:
    from safetensors import safe_open

    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise ValueError(f"no *.safetensors files in {model_dir}")
    for path in files:
        with safe_open(path, framework="pt", device="cpu") as f:
            for name in f.keys():
                yield name, f.get_tensor(name)


def _module_param_name(ckpt_name: str) -> str:
    """Checkpoint (HF) name -> parameter name of this package's LlamaForCausalLM."""
    name = ckpt_name
    name = re.sub(r"^model\.layers\.(\d+)\.self_attn\.", r"layers.\1.", name)
    name = re.sub(r"^model\.layers\.(\d+)\.mlp\.", r"layers.\1.", name)
    name = re.sub(r"^model\.layers\.(\d+)\.(input_layernorm|post_attention_layernorm)\.weight$", r"layers.\1.\2", name)
    name = {"model.embed_tokens.weight": "embed_tokens", "model.norm.weight": "norm", "lm_head.weight": "lm_head"}.get(name, name)
    return name


def load_weights(model: LlamaForCausalLM, weights: Iterable[Tuple[str, torch.Tensor]]) -> Dict[str, int]:
    """Counterpart of LlamaForCausalLM.load_weights (models/llama.py:560-632).  Returns load statistics."""
    params = dict(model.named_parameters())
    stats = {"loaded": 0, "skipped": 0}
    for ckpt_name, tensor in weights:
        if "rotary_emb.inv_freq" in ckpt_name or "rotary_emb.cos_cached" in ckpt_name or "rotary_emb.sin_cached" in ckpt_name:
            stats["skipped"] += 1
            continue
        for param_part, weight_part, shard_id in STACKED_PARAMS_MAPPING:
            if weight_part not in ckpt_name:
                continue
            name = _module_param_name(ckpt_name.replace(weight_part, param_part))
            if name not in params:
                stats["skipped"] += 1          # the layer was built without that parameter (e.g. bias=False in the config)
                break
            p = params[name]
            p.weight_loader(p, tensor, shard_id)
            stats["loaded"] += 1
            break
        else:
            name = _module_param_name(ckpt_name)
            if name not in params:
                stats["skipped"] += 1
                continue
            p = params[name]
            loader = getattr(p, "weight_loader", None)
            if loader is not None:
                loader(p, tensor)
            else:
                if p.data.shape != tensor.shape:
                    raise ValueError(f"{ckpt_name}: shape {tuple(tensor.shape)} does not match parameter {tuple(p.data.shape)}")
                p.data.copy_(tensor)
            stats["loaded"] += 1
    return stats


def load_llama_awq(model_dir: str, device: Optional[torch.device] = None, max_batch: int = 32, max_seq: int = 512) -> LlamaForCausalLM:
    """Build the decode-harness Llama from a local AutoAWQ model directory and load its weights."""
    quant = load_quant_config(model_dir)
    cfg = load_llama_config(model_dir)
    model = LlamaForCausalLM(cfg, quant, max_batch=max_batch, max_seq=max_seq)
    load_weights(model, iterate_safetensors(model_dir))
    if device is not None:
        model.to(device)
    for layer in model.layers:
        for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
            lin.process_weights_after_loading()
    return model
