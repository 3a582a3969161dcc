"""The plugin seam the AWQ method plugs into, mirrored from the reference so code written against
`LinearMethodBase` / `QuantizationConfig` (python/sglang/srt/layers/quantization/base_config.py:16-82,
:112-228) reads the same here.  Only what the AWQ linear path uses is present."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch


class QuantizeMethodBase(ABC):
    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        """Create the layer's parameters and set them as attributes of `layer`."""
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return


class LinearMethodBase(QuantizeMethodBase):
    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class QuantizationConfig(ABC):
    def __init__(self):
        super().__init__()
        self.packed_modules_mapping: Dict[str, List[str]] = dict()

    @abstractmethod
    def get_name(self) -> str: ...

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]: ...

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int: ...

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]: ...

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig": ...

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]: ...

    @abstractmethod
    def get_scaled_act_names(self) -> List[str]: ...

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default
