"""`sgl_kernel`-shaped module for the AWQ path, and the zero-edit binding on the reference's HIP branch.

The reference binds its AWQ op by name at import time (awq.py:62-77):
    CUDA / XPU:  from sgl_kernel import awq_dequantize
    HIP:         from ...awq_triton import awq_dequantize_triton as awq_dequantize
and `AWQLinearMethod.apply` reads that module global at CALL time (awq.py:446).  `install()` therefore

  1. makes `import sgl_kernel; sgl_kernel.awq_dequantize / sgl_kernel.awq_gemm` resolve to the gfx950 ops: it attaches
     the two names to an already-imported real `sgl_kernel` (whose ROCm build registers no AWQ op at all,
     common_extension_rocm.cc:21-198) or, when no `sgl_kernel` is importable, publishes this module under that name
     — what the `_is_cuda` / `_is_xpu` branches pick up;
  2. rebinds the global `awq_dequantize` of `sglang.srt.layers.quantization.awq` to `torch.ops.sgl_kernel.awq_dequantize`
     — what the `_is_hip` branch needs, since it bound the Triton kernel.  If that module is already imported it is
     patched in place; otherwise a post-import hook patches it when it is imported later.  No file of the reference is
     edited; `uninstall()` restores the original binding;
  3. optionally (`register_config=True`) puts this package's AWQConfig under "awq" in the reference's method registry
     (quantization/__init__.py:58), so new layers get this package's AWQLinearMethod (fused kernel, one launch) instead
     of dequantise + matmul;
  4. (`hook_weight_updates`, default on) wraps the reference's in-place weight-update paths —
     `ModelRunner.update_weights_from_disk / _from_tensor / _from_distributed / _from_ipc`
     (model_executor/model_runner.py:969, 1191, 1281, 2932) and `DefaultModelLoader.load_weights_and_postprocess`
     (model_loader/loader.py:616-632) — so that each of them ends with `weight_update.weights_updated(model)`: the op's
     repacked-copy cache is dropped and this package's layers re-derive their copies.  Those paths write through
     `param.data.copy_`, which no version counter shows; with the hooks in place the op's cache is safe and is switched on
     (ops._OP_CACHE_MODE "auto").  Already-imported modules are patched in place, the others when they are imported.
"""
import importlib
import importlib.abc
import sys

from .ops import awq_dequantize, awq_gemm  # noqa: F401  (registers torch.ops.sgl_kernel.* as a side effect)

__all__ = ["awq_dequantize", "awq_gemm", "install", "uninstall"]

REFERENCE_AWQ_MODULE = "sglang.srt.layers.quantization.awq"
REFERENCE_REGISTRY_MODULE = "sglang.srt.layers.quantization"
REFERENCE_RUNNER_MODULE = "sglang.srt.model_executor.model_runner"
REFERENCE_LOADER_MODULE = "sglang.srt.model_loader.loader"
RUNNER_UPDATE_METHODS = ("update_weights_from_disk", "update_weights_from_tensor", "update_weights_from_distributed",
                         "update_weights_from_ipc")
_SAVED = "_sglang_awq_amd_saved_awq_dequantize"


def _patch_reference_awq(mod) -> None:
    """Point the reference module's `awq_dequantize` global at the gfx950 op (the name apply() resolves per call)."""
    if getattr(mod, "awq_dequantize", None) is awq_dequantize:
        return
    if not hasattr(mod, _SAVED):
        setattr(mod, _SAVED, getattr(mod, "awq_dequantize", None))
    mod.awq_dequantize = awq_dequantize


def _patch_reference_runner(mod) -> None:
    """ModelRunner.update_weights_*: re-derive every copy made from the old weights once the update has returned."""
    from .weight_update import wrap_update_method

    cls = getattr(mod, "ModelRunner", None)
    if cls is None:
        return
    for name in RUNNER_UPDATE_METHODS:
        fn = cls.__dict__.get(name)
        if callable(fn):
            setattr(cls, name, wrap_update_method(fn))


def _patch_reference_loader(mod) -> None:
    """DefaultModelLoader.load_weights_and_postprocess (a staticmethod): drop the op's cache after a (re)load."""
    from .weight_update import wrap_load_weights_and_postprocess

    cls = getattr(mod, "DefaultModelLoader", None)
    if cls is None:
        return
    raw = cls.__dict__.get("load_weights_and_postprocess")
    if isinstance(raw, staticmethod):
        cls.load_weights_and_postprocess = staticmethod(wrap_load_weights_and_postprocess(raw.__func__))
    elif callable(raw):
        cls.load_weights_and_postprocess = wrap_load_weights_and_postprocess(raw)


def _unpatch_class(cls, names) -> None:
    for name in names:
        raw = cls.__dict__.get(name)
        fn = raw.__func__ if isinstance(raw, staticmethod) else raw
        orig = getattr(fn, "_sglang_awq_amd_original", None)
        if orig is not None:
            setattr(cls, name, staticmethod(orig) if isinstance(raw, staticmethod) else orig)


class _PatchOnImport(importlib.abc.MetaPathFinder):
    """Finds a module of the reference with the remaining finders and wraps its loader so the module is patched right after
    it has executed (its import-time ladder has run by then).  `patches`: module name -> patch function."""

    def __init__(self, patches):
        self.patches = dict(patches)

    def find_spec(self, fullname, path=None, target=None):
        patch = self.patches.get(fullname)
        if patch is None:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
                continue
            inner = spec.loader.exec_module

            def exec_module(module, _inner=inner, _patch=patch):
                _inner(module)
                _patch(module)

            spec.loader.exec_module = exec_module
            return spec
        return None


_hook = None


def install(force_module: bool = False, patch_reference: bool = True, register_config: bool = False,
            hook_weight_updates: bool = True):
    """Expose awq_dequantize / awq_gemm as attributes of `sgl_kernel`, (patch_reference) bind the reference's
    AWQLinearMethod to the gfx950 op on its HIP branch and (hook_weight_updates) make the reference's in-place weight-update
    paths invalidate every repacked copy; returns the `sgl_kernel` module."""
    global _hook
    this = sys.modules[__name__]
    target = sys.modules.get("sgl_kernel")
    if target is None and not force_module:
        try:
            target = importlib.import_module("sgl_kernel")
        except Exception:
            target = None
    if target is None or force_module:
        sys.modules["sgl_kernel"] = this
        target = this
    else:
        target.awq_dequantize = awq_dequantize
        target.awq_gemm = awq_gemm
    wanted = {}
    if patch_reference:
        wanted[REFERENCE_AWQ_MODULE] = _patch_reference_awq
    if hook_weight_updates:
        wanted[REFERENCE_RUNNER_MODULE] = _patch_reference_runner
        wanted[REFERENCE_LOADER_MODULE] = _patch_reference_loader
    pending = {}
    for name, patch in wanted.items():
        mod = sys.modules.get(name)
        if mod is not None:
            patch(mod)
        else:
            pending[name] = patch
    if pending:
        if _hook is None:
            _hook = _PatchOnImport(pending)
            sys.meta_path.insert(0, _hook)
        else:
            _hook.patches.update(pending)
    if hook_weight_updates:
        from . import ops

        ops._reload_hooks_installed()          # the op's repacked-copy cache is safe now ("auto" mode switches it on)
    if register_config:
        from .awq import AWQConfig

        reg = sys.modules.get(REFERENCE_REGISTRY_MODULE)
        if reg is None:
            try:
                reg = importlib.import_module(REFERENCE_REGISTRY_MODULE)
            except Exception:
                reg = None
        for name in ("BASE_QUANTIZATION_METHODS", "QUANTIZATION_METHODS"):
            table = getattr(reg, name, None) if reg is not None else None
            if isinstance(table, dict):
                table["awq"] = AWQConfig
    return target


def uninstall() -> None:
    """Undo install()'s patches of the reference modules and remove the post-import hook (the `sgl_kernel` names stay)."""
    global _hook
    if _hook is not None and _hook in sys.meta_path:
        sys.meta_path.remove(_hook)
    _hook = None
    mod = sys.modules.get(REFERENCE_AWQ_MODULE)
    if mod is not None and hasattr(mod, _SAVED):
        mod.awq_dequantize = getattr(mod, _SAVED)
        delattr(mod, _SAVED)
    runner = getattr(sys.modules.get(REFERENCE_RUNNER_MODULE), "ModelRunner", None)
    if runner is not None:
        _unpatch_class(runner, RUNNER_UPDATE_METHODS)
    loader = getattr(sys.modules.get(REFERENCE_LOADER_MODULE), "DefaultModelLoader", None)
    if loader is not None:
        _unpatch_class(loader, ("load_weights_and_postprocess",))
