"""Parameters that know how to take their tensor-parallel shard out of a full checkpoint tensor.

Counterpart of the reference's `PackedvLLMParameter` / `GroupQuantScaleParameter`
(python/sglang/srt/layers/parameter.py:93-286, :539-550), reduced to what AWQ needs:
  * `output_dim` sharding for column-parallel layers (plain, merged gate/up, fused q/k/v),
  * `input_dim` sharding for row-parallel layers,
  * a packed dimension: when 8 int4 share an int32 along the output dim, shard sizes and offsets
    expressed in logical columns are divided by `packed_factor` before indexing.
The loaders copy into `param.data` in place, so they work on any device.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch.nn import Parameter


class ShardedParameter(Parameter):
    def __new__(cls, data: torch.Tensor, **kwargs):
        return super().__new__(cls, data=data, requires_grad=False)

    def __init__(self, data: torch.Tensor, input_dim: int, output_dim: int,
                 weight_loader: Optional[Callable] = None, packed_dim: Optional[int] = None, packed_factor: int = 1):
        self._input_dim = input_dim
        self._output_dim = output_dim
        self._packed_dim = packed_dim
        self._packed_factor = packed_factor
        self._weight_loader = weight_loader

    input_dim = property(lambda self: self._input_dim)
    output_dim = property(lambda self: self._output_dim)
    packed_dim = property(lambda self: self._packed_dim)
    packed_factor = property(lambda self: self._packed_factor)
    weight_loader = property(lambda self: self._weight_loader)

    # ---- index math -----------------------------------------------------------------------
    def adjust_shard_indexes_for_packing(self, shard_size: int, shard_offset: int):
        """Logical-column (size, offset) -> packed-word (size, offset) (parameter.py:539-550)."""
        if self._packed_dim is None or self._packed_factor == 1:
            return shard_size, shard_offset
        if shard_size % self._packed_factor or shard_offset % self._packed_factor:
            raise ValueError(f"shard ({shard_offset}, {shard_size}) is not aligned to the pack factor {self._packed_factor}")
        return shard_size // self._packed_factor, shard_offset // self._packed_factor

    def _copy(self, dst: torch.Tensor, src: torch.Tensor):
        if dst.shape != src.shape:
            raise ValueError(f"shard shape mismatch: parameter slice {tuple(dst.shape)} vs checkpoint slice {tuple(src.shape)}")
        dst.copy_(src)

    # ---- loaders --------------------------------------------------------------------------
    def load_column_parallel_weight(self, loaded_weight: torch.Tensor, tp_rank: int = 0, use_presharded_weights: bool = False):
        if not use_presharded_weights:
            n = self.data.shape[self._output_dim]
            loaded_weight = loaded_weight.narrow(self._output_dim, tp_rank * n, n)
        self._copy(self.data, loaded_weight)

    def load_row_parallel_weight(self, loaded_weight: torch.Tensor, tp_rank: int = 0, use_presharded_weights: bool = False):
        if not use_presharded_weights:
            k = self.data.shape[self._input_dim]
            loaded_weight = loaded_weight.narrow(self._input_dim, tp_rank * k, k)
        self._copy(self.data, loaded_weight)

    def _packed_along_output(self) -> bool:
        return self._packed_dim is not None and self._packed_dim == self._output_dim

    def load_merged_column_weight(self, loaded_weight: torch.Tensor, shard_offset: int, shard_size: int, tp_rank: int = 0,
                                  use_presharded_weights: bool = False, **_):
        """One logical matrix (e.g. gate or up) of a merged column-parallel layer: it lands at
        [shard_offset, shard_offset + shard_size) of this rank's output dim (logical columns)."""
        if self._packed_along_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size, shard_offset)
        dst = self.data.narrow(self._output_dim, shard_offset, shard_size)
        if not use_presharded_weights:
            loaded_weight = loaded_weight.narrow(self._output_dim, tp_rank * shard_size, shard_size)
        self._copy(dst, loaded_weight)

    def load_qkv_weight(self, loaded_weight: torch.Tensor, shard_offset: int, shard_size: int, shard_id: str,
                        num_heads: int, tp_rank: int = 0, use_presharded_weights: bool = False, **_):
        """q, k or v of a fused QKV layer.  `num_heads` is the KV replication factor: with fewer KV
        heads than ranks, `num_heads` consecutive ranks share one KV shard (parameter.py:175-223)."""
        if self._packed_along_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size, shard_offset)
        dst = self.data.narrow(self._output_dim, shard_offset, shard_size)
        src_rank = tp_rank if shard_id == "q" else tp_rank // num_heads
        if not use_presharded_weights:
            loaded_weight = loaded_weight.narrow(self._output_dim, src_rank * shard_size, shard_size)
        self._copy(dst, loaded_weight)


class PackedvLLMParameter(ShardedParameter):
    """int4-in-int32 weights / zero points (qweight, qzeros)."""

    def __init__(self, data, input_dim, output_dim, packed_dim, packed_factor, weight_loader=None):
        super().__init__(data, input_dim, output_dim, weight_loader, packed_dim, packed_factor)


class GroupQuantScaleParameter(ShardedParameter):
    """per-group scales: unpacked, sharded along both dims like the weight."""

    def __init__(self, data, input_dim, output_dim, weight_loader=None):
        super().__init__(data, input_dim, output_dim, weight_loader)
