"""Keeping derived weight copies in step with in-place weight updates.

This package derives data from the checkpoint tensors at load time: the MFMA-fragment-major copy `layer.awq_packed` made by
`AWQLinearMethod.process_weights_after_loading`, the gate/up-interleaved copy of the decode harness, the experts' copies of
`AWQMoEMethod`, and the copies the drop-in op `sgl_kernel.awq_gemm` keeps per weight (ops._op_cache).  The reference updates
weights in place after loading — `ModelRunner.update_weights_from_disk / _from_tensor / _from_distributed / _from_ipc`
(python/sglang/srt/model_executor/model_runner.py:969, 1191, 1281, 2932) go through `model.load_weights`, i.e.
`param.data.copy_` (layers/parameter.py:59, 124) — and of those only `update_weights_from_disk` re-runs
`process_weights_after_loading` (via `DefaultModelLoader.load_weights_and_postprocess`, model_loader/loader.py:616-632).  A write
through `.data` bumps no version counter the op could see, so every derived copy would silently go stale.

`weights_updated(model)` is the one thing to call after such a write: it drops the op's cache and re-derives the copies of every
layer of `model` that uses this package's methods.  `sgl_kernel_compat.install()` wraps the reference's update paths so that
they call it; `tests/test_compat_cpu.py` checks that against a stub of those classes, `tests/test_gpu_round3.py` checks on the GPU
that `param.data.copy_(new)` followed by the hook gives the new weights' results.
"""
from __future__ import annotations

import functools
from typing import Iterable, Optional

import torch

_DERIVED_ATTRS = ("_gu_il", "_w13_plain_packed")      # lazily built copies hanging off modules (decode harness, MoE)


def _methods_of_this_package():
    from .awq import AWQLinearMethod
    from .moe import AWQMoEMethod

    return (AWQLinearMethod, AWQMoEMethod)


def weights_updated(model: Optional[torch.nn.Module] = None, modules: Optional[Iterable[torch.nn.Module]] = None) -> int:
    """Call after weights were written in place.  Clears the drop-in op's repacked-copy cache and re-runs
    `process_weights_after_loading` of every module of `model` (or of `modules`) whose `quant_method` belongs to this package,
    so `layer.awq_packed` and friends are rebuilt from the new values.  Returns the number of layers re-derived."""
    from . import ops

    ops.awq_gemm_cache_clear()
    mods = []
    if model is not None:
        mods.extend(m for _, m in model.named_modules())
    if modules is not None:
        mods.extend(modules)
    ours = _methods_of_this_package()
    n = 0
    for m in mods:
        for attr in _DERIVED_ATTRS:
            if attr in m.__dict__:
                del m.__dict__[attr]
        qm = getattr(m, "quant_method", None)
        if qm is not None and isinstance(qm, ours):
            qm.process_weights_after_loading(m)
            n += 1
    return n


def _model_of(obj):
    """The nn.Module a reference ModelRunner / loader call worked on: `self.model`, or the first Module among the arguments."""
    m = getattr(obj, "model", None)
    return m if isinstance(m, torch.nn.Module) else None


def wrap_update_method(fn):
    """Wrap a ModelRunner.update_weights_* method: after it returns, re-derive every copy made from the old weights."""
    if getattr(fn, "_sglang_awq_amd_wrapped", False):
        return fn

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        out = fn(self, *args, **kwargs)
        weights_updated(_model_of(self))
        return out

    wrapper._sglang_awq_amd_wrapped = True
    wrapper._sglang_awq_amd_original = fn
    return wrapper


def wrap_load_weights_and_postprocess(fn):
    """Wrap DefaultModelLoader.load_weights_and_postprocess(model, weights, target_device) (a staticmethod in the reference): it
    already re-runs process_weights_after_loading of every module, so only the op's cache and the lazily derived copies need dropping."""
    if getattr(fn, "_sglang_awq_amd_wrapped", False):
        return fn

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        out = fn(*args, **kwargs)
        from . import ops

        ops.awq_gemm_cache_clear()
        model = next((a for a in args if isinstance(a, torch.nn.Module)), kwargs.get("model"))
        if isinstance(model, torch.nn.Module):
            for _, m in model.named_modules():
                for attr in _DERIVED_ATTRS:
                    m.__dict__.pop(attr, None)
        return out

    wrapper._sglang_awq_amd_wrapped = True
    wrapper._sglang_awq_amd_original = fn
    return wrapper
