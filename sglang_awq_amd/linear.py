"""Tensor-parallel linear layers that host a `LinearMethodBase` (here: AWQ).

Counterparts of the reference's `LinearBase`, `ReplicatedLinear`, `ColumnParallelLinear`,
`MergedColumnParallelLinear`, `QKVParallelLinear` and `RowParallelLinear`
(python/sglang/srt/layers/linear.py:135-172, :255-450, :461-776, :779-1000, :1210-1422): same
constructor arguments, same `forward` contract `(output, output_bias)`, same sharding:

  column-parallel  the output dim is split; no communication unless gather_output
  row-parallel     the input dim is split (whole quantisation groups per rank, awq.py:372-377);
                   bias is applied on rank 0 only (linear.py:1401); partial outputs are summed
                   with one all-reduce (linear.py:1407-1408)

Only the v2 weight-loading protocol is implemented (AWQLinearMethod is in the reference's
WEIGHT_LOADER_V2_SUPPORTED list, linear.py:52-70): parameters shard themselves (parameter.py).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn import Parameter

from .base_config import LinearMethodBase, QuantizationConfig, QuantizeMethodBase
from .distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                          tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce)
from .parameter import ShardedParameter


def divide(a: int, b: int) -> int:
    if a % b != 0:
        raise ValueError(f"{a} is not divisible by {b}")
    return a // b


class UnquantizedLinearMethod(LinearMethodBase):
    """fp16/bf16 weight [out, in]; used for `modules_to_not_convert` and unquantised layers."""

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        weight = ShardedParameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
                                  input_dim=1, output_dim=0, weight_loader=extra_weight_attrs.get("weight_loader"))
        layer.register_parameter("weight", weight)

    def apply(self, layer, x, bias=None):
        return F.linear(x, layer.weight, bias)


class LinearBase(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = ""):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        self.params_dtype = params_dtype if params_dtype is not None else torch.get_default_dtype()
        self.quant_config = quant_config
        self.prefix = prefix
        if quant_config is None:
            self.quant_method: Optional[QuantizeMethodBase] = UnquantizedLinearMethod()
        else:
            self.quant_method = quant_config.get_quant_method(self, prefix=prefix)

    def _make_bias(self, size: int, bias: bool, weight_loader=None):
        """The reference gives `bias` a weight_loader and `output_dim: 0` (linear.py:348-358, 1286-1293): a column-parallel
        bias is sharded along the output dim like its weight (per logical matrix for merged / QKV layers), a row-parallel bias
        is replicated (and added on rank 0 only, linear.py:1401)."""
        if bias:
            self.bias = ShardedParameter(torch.zeros(size, dtype=self.params_dtype), input_dim=0, output_dim=0,
                                         weight_loader=weight_loader)
        else:
            self.register_parameter("bias", None)

    def process_weights_after_loading(self):
        self.quant_method.process_weights_after_loading(self)


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size, output_size, bias=True, skip_bias_add=False, params_dtype=None, quant_config=None, prefix=""):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        self.quant_method.create_weights(self, input_size, [output_size], input_size, output_size, self.params_dtype,
                                         weight_loader=self.weight_loader_v2)
        self._make_bias(output_size, bias, self.weight_loader_v2)

    def weight_loader_v2(self, param: ShardedParameter, loaded_weight: torch.Tensor):
        param.load_column_parallel_weight(loaded_weight, tp_rank=0, use_presharded_weights=True)

    def forward(self, x) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        bias = self.bias if not self.skip_bias_add else None
        return self.quant_method.apply(self, x, bias), (self.bias if self.skip_bias_add else None)


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None,
                 output_sizes: Optional[List[int]] = None, prefix: str = "", tp_rank: Optional[int] = None,
                 tp_size: Optional[int] = None, use_presharded_weights: bool = False):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        self.gather_output = gather_output
        self.use_presharded_weights = use_presharded_weights
        self.tp_rank = get_tensor_model_parallel_rank() if tp_rank is None else tp_rank
        self.tp_size = get_tensor_model_parallel_world_size() if tp_size is None else tp_size
        self.output_size_per_partition = divide(output_size, self.tp_size)
        self.output_partition_sizes = [self.output_size_per_partition]
        if hasattr(self, "output_sizes"):      # merged / qkv subclasses set this before super().__init__
            self.output_partition_sizes = [divide(s, self.tp_size) for s in self.output_sizes]
        self.quant_method.create_weights(self, self.input_size, self.output_partition_sizes, self.input_size,
                                         self.output_size, self.params_dtype, weight_loader=self.weight_loader_v2)
        self._make_bias(self.output_size_per_partition, bias, self.weight_loader_v2)

    def weight_loader_v2(self, param: ShardedParameter, loaded_weight: torch.Tensor):
        param.load_column_parallel_weight(loaded_weight, tp_rank=self.tp_rank,
                                          use_presharded_weights=self.use_presharded_weights)

    def forward(self, input_):
        bias = self.bias if not self.skip_bias_add else None
        output_parallel = self.quant_method.apply(self, input_, bias)
        output = tensor_model_parallel_all_gather(output_parallel) if self.gather_output else output_parallel
        return output, (self.bias if self.skip_bias_add else None)

    def extra_repr(self) -> str:
        return (f"in_features={self.input_size}, output_features={self.output_size_per_partition}, "
                f"bias={self.bias is not None}, tp_size={self.tp_size}, gather_output={self.gather_output}")


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices fused along the output dim (e.g. gate_proj + up_proj); each
    logical matrix is sharded separately when loaded (linear.py:461-776)."""

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None, prefix: str = "",
                 tp_rank: Optional[int] = None, tp_size: Optional[int] = None, use_presharded_weights: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias, gather_output, skip_bias_add, params_dtype, quant_config,
                         None, prefix, tp_rank, tp_size, use_presharded_weights)

    def weight_loader_v2(self, param: ShardedParameter, loaded_weight: torch.Tensor, loaded_shard_id: Optional[int] = None):
        if loaded_shard_id is None:
            # checkpoint already holds the fused tensor: split it and load shard by shard
            offset = 0
            for i, size in enumerate(self.output_sizes):
                sz, off = size, offset
                if param.packed_dim is not None and param.packed_dim == param.output_dim:
                    sz, off = param.adjust_shard_indexes_for_packing(size, offset)
                self.weight_loader_v2(param, loaded_weight.narrow(param.output_dim, off, sz), i)
                offset += size
            return
        if not 0 <= loaded_shard_id < len(self.output_sizes):
            raise ValueError(f"shard id {loaded_shard_id} out of range")
        shard_offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        shard_size = self.output_sizes[loaded_shard_id] // self.tp_size
        param.load_merged_column_weight(loaded_weight, shard_offset=shard_offset, shard_size=shard_size,
                                        tp_rank=self.tp_rank, use_presharded_weights=self.use_presharded_weights)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection, parallel over heads; KV heads are replicated when there are fewer of
    them than ranks (linear.py:779-1000)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: Optional[int] = None,
                 bias: bool = True, skip_bias_add: bool = False, params_dtype=None, quant_config=None, prefix: str = "",
                 tp_rank: Optional[int] = None, tp_size: Optional[int] = None, load_presharded_attn: bool = False):
        self.hidden_size = hidden_size
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        tp_rank = get_tensor_model_parallel_rank() if tp_rank is None else tp_rank
        tp_size = get_tensor_model_parallel_world_size() if tp_size is None else tp_size
        self.num_heads = divide(self.total_num_heads, tp_size)
        if tp_size >= self.total_num_kv_heads:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = divide(tp_size, self.total_num_kv_heads)
        else:
            self.num_kv_heads = divide(self.total_num_kv_heads, tp_size)
            self.num_kv_head_replicas = 1
        self.q_proj_shard_size = self.num_heads * head_size
        self.kv_proj_shard_size = self.num_kv_heads * head_size
        self.output_sizes = [self.num_heads * head_size * tp_size, self.num_kv_heads * head_size * tp_size,
                             self.num_kv_heads * head_size * tp_size]
        super().__init__(hidden_size, sum(self.output_sizes), bias, False, skip_bias_add, params_dtype, quant_config,
                         None, prefix, tp_rank, tp_size, load_presharded_attn)

    def _shard_offset(self, shard_id: str) -> int:
        return {"q": 0, "k": self.q_proj_shard_size, "v": self.q_proj_shard_size + self.kv_proj_shard_size}[shard_id]

    def _shard_size(self, shard_id: str) -> int:
        return self.q_proj_shard_size if shard_id == "q" else self.kv_proj_shard_size

    def weight_loader_v2(self, param: ShardedParameter, loaded_weight: torch.Tensor, loaded_shard_id: Optional[str] = None):
        if loaded_shard_id is None:
            offset = 0
            for sid, size in (("q", self.total_num_heads * self.head_size),
                              ("k", self.total_num_kv_heads * self.head_size),
                              ("v", self.total_num_kv_heads * self.head_size)):
                sz, off = size, offset
                if param.packed_dim is not None and param.packed_dim == param.output_dim:
                    sz, off = param.adjust_shard_indexes_for_packing(size, offset)
                self.weight_loader_v2(param, loaded_weight.narrow(param.output_dim, off, sz), sid)
                offset += size
            return
        if loaded_shard_id not in ("q", "k", "v"):
            raise ValueError(f"bad qkv shard id {loaded_shard_id!r}")
        param.load_qkv_weight(loaded_weight, shard_offset=self._shard_offset(loaded_shard_id),
                              shard_size=self._shard_size(loaded_shard_id), shard_id=loaded_shard_id,
                              num_heads=self.num_kv_head_replicas, tp_rank=self.tp_rank,
                              use_presharded_weights=self.use_presharded_weights)


class RowParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = True, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype=None, reduce_results: bool = True, quant_config=None,
                 prefix: str = "", tp_rank: Optional[int] = None, tp_size: Optional[int] = None,
                 use_presharded_weights: bool = False):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        self.use_presharded_weights = use_presharded_weights
        self.tp_rank = get_tensor_model_parallel_rank() if tp_rank is None else tp_rank
        self.tp_size = get_tensor_model_parallel_world_size() if tp_size is None else tp_size
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self.quant_method.create_weights(self, self.input_size_per_partition, [self.output_size], self.input_size,
                                         self.output_size, self.params_dtype, weight_loader=self.weight_loader_v2)
        if not reduce_results and bias and not skip_bias_add:
            raise ValueError("When not reduce the results, adding bias to the results can lead to incorrect results")
        self._make_bias(self.output_size, bias, self._load_replicated_bias)

    def weight_loader_v2(self, param: ShardedParameter, loaded_weight: torch.Tensor):
        param.load_row_parallel_weight(loaded_weight, tp_rank=self.tp_rank,
                                       use_presharded_weights=self.use_presharded_weights)

    @staticmethod
    def _load_replicated_bias(param: ShardedParameter, loaded_weight: torch.Tensor):
        """The bias of a row-parallel layer has no input dim to split: every rank holds all of it (the reference's v1
        weight_loader with input_dim = None, linear.py:1296-1340) and rank 0 alone adds it."""
        param.load_column_parallel_weight(loaded_weight, tp_rank=0, use_presharded_weights=True)

    def forward(self, input_, skip_all_reduce: bool = False):
        if self.input_is_parallel:
            input_parallel = input_
        else:
            k = self.input_size_per_partition
            # a strided view: the fused kernel takes the row stride, no copy (include/awq_hip.h `ldx`)
            input_parallel = input_[..., self.tp_rank * k:(self.tp_rank + 1) * k]
        # bias only on rank 0 so it is added once across the group (linear.py:1399-1401)
        bias_ = None if (self.tp_rank > 0 or self.skip_bias_add) else self.bias
        output_parallel = self.quant_method.apply(self, input_parallel, bias_)
        if self.reduce_results and self.tp_size > 1 and not skip_all_reduce:
            output = tensor_model_parallel_all_reduce(output_parallel)
        else:
            output = output_parallel
        return output, (self.bias if self.skip_bias_add else None)

    def extra_repr(self) -> str:
        return (f"input_features={self.input_size_per_partition}, output_features={self.output_size}, "
                f"bias={self.bias is not None}, tp_size={self.tp_size}, reduce_results={self.reduce_results}")
