"""AWQ quantization config and linear method for MI355X.

Counterpart of `AWQConfig` / `AWQLinearMethod` in the reference
(python/sglang/srt/layers/quantization/awq.py:89-179, :352-451): same constructor arguments, same
parameter names, shapes and legality checks, same three-method plugin interface — so a layer built
by the reference's `LinearBase` machinery can use it unchanged.  What differs is `apply()`:

  reference   awq_dequantize(...) -> full [K, N] fp16 temp -> torch.matmul -> add_(bias)
              (every decode step re-materialises every weight matrix, ~8.7x the bytes at M = 1)
  here        one fused gfx950 launch (`awq_linear`: int4 tiles are dequantised in registers and
              fed to MFMA; bias added in the epilogue with the reference's two roundings)

`apply_mode="dequant_matmul"` keeps the reference's two-step structure on top of this package's
`awq_dequantize` op (useful as an A/B baseline and for M beyond the fused kernels' sweet spot).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import torch

from .base_config import LinearMethodBase, QuantizationConfig, QuantizeMethodBase
from .parameter import GroupQuantScaleParameter, PackedvLLMParameter


def is_layer_skipped_awq(prefix: str, modules_to_not_convert: List[str]) -> bool:
    return any(module_name in prefix for module_name in modules_to_not_convert)


class AWQConfig(QuantizationConfig):
    """Config class for AWQ (awq.py:89-179).  Only 4-bit weights are supported."""

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool,
                 modules_to_not_convert: Optional[List[str]] = None) -> None:
        super().__init__()
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.modules_to_not_convert = modules_to_not_convert or []
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for "
                             f"AWQ, but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return (f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"zero_point={self.zero_point}, modules_to_not_convert={self.modules_to_not_convert})")

    def get_scaled_act_names(self) -> List[str]:
        return []

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        # the reference lists fp16 only off-NPU (awq.py:129-130); the gfx950 kernels also take bf16
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        weight_bits = cls.get_from_keys(config, ["w_bit", "bits"])
        group_size = cls.get_from_keys(config, ["q_group_size", "group_size"])
        zero_point = cls.get_from_keys(config, ["zero_point"])
        modules_to_not_convert = cls.get_from_keys_or(config, ["modules_to_not_convert"], None)
        return cls(weight_bits, group_size, zero_point, modules_to_not_convert)

    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        from .linear import LinearBase, UnquantizedLinearMethod

        if isinstance(layer, LinearBase):
            if is_layer_skipped_awq(prefix, self.modules_to_not_convert):
                return UnquantizedLinearMethod()
            return AWQLinearMethod(self)
        return None


class AWQLinearMethod(LinearMethodBase):
    """Linear method for AWQ (awq.py:352-451)."""

    # with a repacked copy every batch size runs on it (GEMV passes up to 160 rows, MFMA-tiled kernel beyond)
    REPACKED_MAX_M = 1 << 30

    def __init__(self, quant_config: AWQConfig, apply_mode: Optional[str] = None, repack: Optional[bool] = None):
        self.quant_config = quant_config
        if repack is None:
            repack = os.environ.get("SGLANG_AWQ_AMD_REPACK", "1") != "0"
        self.repack = repack
        mode = apply_mode or os.environ.get("SGLANG_AWQ_AMD_APPLY", "fused")
        if mode not in ("fused", "dequant_matmul"):
            raise ValueError(f"apply_mode must be 'fused' or 'dequant_matmul', got {mode!r}")
        self.apply_mode = mode

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        group_size = self.quant_config.group_size
        if group_size == -1:
            group_size = input_size_per_partition
        if input_size_per_partition % group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % self.quant_config.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")

        weight_loader = extra_weight_attrs.get("weight_loader")
        pf = self.quant_config.pack_factor
        qweight = PackedvLLMParameter(
            data=torch.empty(input_size_per_partition, output_size_per_partition // pf, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=pf, weight_loader=weight_loader)
        qzeros = PackedvLLMParameter(
            data=torch.empty(input_size_per_partition // group_size, output_size_per_partition // pf, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=pf, weight_loader=weight_loader)
        scales = GroupQuantScaleParameter(
            data=torch.empty(input_size_per_partition // group_size, output_size_per_partition, dtype=params_dtype),
            input_dim=0, output_dim=1, weight_loader=weight_loader)
        layer.register_parameter("qweight", qweight)
        layer.register_parameter("qzeros", qzeros)
        layer.register_parameter("scales", scales)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        if getattr(layer, "awq_shape", None) is not None and layer.qweight.numel() == 0:
            # SGLANG_AWQ_AMD_KEEP_CHECKPOINT=0 released the checkpoint tensors of this layer after the first re-layout: there is
            # nothing left to re-derive the copy from, and nothing a weight loader could have written into
            raise RuntimeError("sglang_awq_amd: this layer's checkpoint tensors were released (SGLANG_AWQ_AMD_KEEP_CHECKPOINT=0); "
                               "weights cannot be reloaded in place. Run with SGLANG_AWQ_AMD_KEEP_CHECKPOINT=1 for deployments "
                               "that update weights, or rebuild the layer (create_weights) before loading.")
        # the kernels consume the on-disk AutoAWQ layout directly: no repack (awq.py:429-432)
        layer.qweight = torch.nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = torch.nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = torch.nn.Parameter(layer.scales.data, requires_grad=False)
        # Optional one-time MFMA-fragment-major copy for decode (the CDNA4 analogue of the reference's
        # awq_marlin_repack step, awq.py AWQMarlinLinearMethod.process_weights_after_loading).  The original
        # tensors stay: awq_dequantize and the prefill kernel consume the checkpoint layout.
        layer.awq_packed = None
        if layer.qweight.is_cuda:
            from . import ops

            ops.awq_gemm_cache_clear()           # (re)loaded weights: copies the drop-in op made of the old values are stale
            if self.repack and self.apply_mode == "fused" and layer.scales.dtype in (torch.float16, torch.bfloat16):
                layer.awq_packed = ops.awq_repack(layer.qweight.data, layer.scales.data, layer.qzeros.data)
                # SGLANG_AWQ_AMD_KEEP_CHECKPOINT=0: a fused-only deployment never reads the checkpoint tensors again (every batch
                # size runs on the repacked copy); releasing them halves the weight memory (70B at TP = 1: 35 GB).  The
                # parameters stay registered with their shapes' metadata (`awq_shape`) but empty storage, so
                # `sgl_kernel.awq_dequantize(layer.qweight, ...)` on such a layer raises instead of reading freed memory.
                if (layer.awq_packed is not None and os.environ.get("SGLANG_AWQ_AMD_KEEP_CHECKPOINT", "1") == "0"
                        and self._every_batch_has_repacked_variant(layer)):
                    K, C = layer.qweight.shape
                    layer.awq_shape = (K, C * self.quant_config.pack_factor, K // layer.scales.shape[0])
                    for name in ("qweight", "qzeros", "scales"):
                        t = getattr(layer, name)
                        setattr(layer, name, torch.nn.Parameter(torch.empty(0, dtype=t.dtype, device=t.device), requires_grad=False))

    def _every_batch_has_repacked_variant(self, layer) -> bool:
        """Before the checkpoint tensors are released: run the repacked route once at every decode batch-size class and at a
        prefill size, so a (K, N, g) for which one of them has no instantiation (AWQ_ERR_BAD_VARIANT — the route `apply` would
        cover with the checkpoint-layout kernel) keeps its checkpoint tensors instead of failing at serving time."""
        from . import ops

        K = layer.qweight.shape[0]
        N = layer.qweight.shape[1] * self.quant_config.pack_factor
        g = K // layer.scales.shape[0]
        try:
            for m in (1, 8, 16, 17, 32, 64, 256):
                ops.awq_gemm_repacked(torch.zeros((m, K), dtype=layer.scales.dtype, device=layer.qweight.device), layer.awq_packed, K, N, g)
        except ops.AwqHipError:
            return False
        return True

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        from . import ops

        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        packed = getattr(layer, "awq_packed", None)
        released = getattr(layer, "awq_shape", None)         # checkpoint tensors released (SGLANG_AWQ_AMD_KEEP_CHECKPOINT=0)
        if released is not None:
            K, N, g = released
        else:
            K, N, g = qweight.shape[0], qweight.shape[-1] * self.quant_config.pack_factor, qweight.shape[0] // max(scales.shape[0], 1)
        out_shape = x.shape[:-1] + (N,)
        reshaped_x = x.reshape(-1, x.shape[-1])
        if reshaped_x.shape[0] == 0:
            return x.new_empty(out_shape)
        if packed is not None and reshaped_x.shape[0] <= self.REPACKED_MAX_M and reshaped_x.shape[0] > 0:
            if released is not None:
                out = ops.awq_gemm_repacked(reshaped_x, packed, K, N, g, bias)     # no checkpoint-layout route left: errors propagate
            else:
                try:
                    out = ops.awq_gemm_repacked(reshaped_x, packed, K, N, g, bias)
                except ops.AwqHipError:     # e.g. M = 32 on a very wide strip: reduction scratch over the LDS guard
                    out = ops.awq_linear(reshaped_x, qweight, scales, qzeros, bias)
        elif self.apply_mode == "fused":
            out = ops.awq_linear(reshaped_x, qweight, scales, qzeros, bias)
        else:
            out = torch.matmul(reshaped_x, ops.awq_dequantize(qweight, scales, qzeros))
            if bias is not None:
                out.add_(bias)
        return out.reshape(out_shape)
