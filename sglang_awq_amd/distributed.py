"""Tensor-parallel group for the AWQ linears: one process per GPU, torch.distributed over RCCL
(backend "nccl" IS RCCL on ROCm) on xGMI, gloo on CPU for tests.

Counterpart of the slice of the reference the row/column-parallel linears touch:
`tensor_model_parallel_all_reduce` / `_all_gather` (distributed/communication_op.py:11-13) and
`GroupCoordinator.all_reduce` (distributed/parallel_state.py:544-623, incl. the world_size == 1
bypass at :561-562).  The reference can route the reduce through custom / quick all-reduce kernels,
mscclpp or torch symmetric memory; here it is RCCL only (north star), in place, on the compute
stream, so it is capturable in a HIP graph with the GEMMs around it.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class TensorParallelGroup:
    def __init__(self, group: Optional[dist.ProcessGroup], rank: int, world_size: int):
        self.device_group = group
        self.rank = rank
        self.world_size = world_size

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the group (returns `t`); no-op for a single rank."""
        if self.world_size == 1:
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.device_group)
        return t

    def all_gather(self, t: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """Concatenate every rank's tensor along `dim` (column-parallel gather_output)."""
        if self.world_size == 1:
            return t
        if dim < 0:
            dim += t.dim()
        t = t.contiguous()
        flat = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(flat, t, group=self.device_group)      # rank-major along dim 0
        parts = flat.view((self.world_size,) + tuple(t.shape)).movedim(0, dim)   # [..., world, size_dim, ...]
        shape = list(t.shape)
        shape[dim] *= self.world_size
        return parts.reshape(shape)

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.device_group)


_TP: TensorParallelGroup = TensorParallelGroup(None, 0, 1)


def init_tensor_parallel(backend: Optional[str] = None, device: Optional[torch.device] = None) -> TensorParallelGroup:
    """Create the TP group over all ranks of the job.  Reads RANK / WORLD_SIZE / MASTER_* from the
    environment (torchrun); initialises the default process group if needed."""
    global _TP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not dist.is_initialized():
        _TP = TensorParallelGroup(None, 0, 1)
        return _TP
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, **kwargs)
    _TP = TensorParallelGroup(dist.group.WORLD, dist.get_rank(), dist.get_world_size())
    return _TP


def set_tensor_parallel_group(group: TensorParallelGroup):
    global _TP
    _TP = group


def destroy_tensor_parallel():
    global _TP
    _TP = TensorParallelGroup(None, 0, 1)
    if dist.is_initialized():
        dist.destroy_process_group()


def get_tp_group() -> TensorParallelGroup:
    return _TP


def get_tensor_model_parallel_rank() -> int:
    return _TP.rank


def get_tensor_model_parallel_world_size() -> int:
    return _TP.world_size


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    return _TP.all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    return _TP.all_gather(input_, dim)
