"""sgl_kernel_compat.install(): the zero-edit binding on the reference's HIP branch, exercised on CPU with a stub package that
reproduces the reference's import-time ladder (python/sglang/srt/layers/quantization/awq.py:55-77: `_is_cuda` -> sgl_kernel,
`_is_hip` -> Triton) and its call-time lookup of the module global in AWQLinearMethod.apply (awq.py:446).  The real package
cannot be imported here (ModuleNotFoundError: pybase64)."""
import importlib
import sys
import textwrap

import pytest

STUB_AWQ = textwrap.dedent('''
    _is_cuda, _is_hip = False, True                      # what is_cuda() / is_hip() return on a ROCm box
    if _is_cuda:
        from sgl_kernel import awq_dequantize
    elif _is_hip:
        def awq_dequantize_triton(qweight, scales, qzeros):
            return "triton"
        awq_dequantize = awq_dequantize_triton           # awq.py:66-69


    class AWQLinearMethod:
        def apply(self, layer, x, bias=None):
            return awq_dequantize                        # awq.py:446 resolves the module global per call
''')
STUB_RUNNER = textwrap.dedent('''
    class ModelRunner:                                   # model_executor/model_runner.py: the in-place weight-update paths
        def __init__(self, model):
            self.model = model
            self.calls = []

        def update_weights_from_disk(self, model_path, load_format):
            self.calls.append("disk")
            return True, "Success"

        def update_weights_from_tensor(self, named_tensors, load_format=None):
            for name, t in named_tensors:
                dict(self.model.named_parameters())[name].data.copy_(t)      # how the reference's loaders write (parameter.py:59)
            self.calls.append("tensor")
            return True, "Success"

        def update_weights_from_distributed(self, names, dtypes, shapes, group_name):
            self.calls.append("distributed")
            return True, "Success"
''')
STUB_LOADER = textwrap.dedent('''
    class DefaultModelLoader:                            # model_loader/loader.py:616-632
        @staticmethod
        def load_weights_and_postprocess(model, weights, target_device):
            for _, module in model.named_modules():
                qm = getattr(module, "quant_method", None)
                if qm is not None:
                    qm.process_weights_after_loading(module)
            return "postprocessed"
''')
STUB_REGISTRY = "BASE_QUANTIZATION_METHODS = {'awq': object, 'gptq': object}\nQUANTIZATION_METHODS = {**BASE_QUANTIZATION_METHODS}\n"


@pytest.fixture
def stub_reference(tmp_path, monkeypatch):
    pkg = tmp_path / "sglang" / "srt" / "layers" / "quantization"
    pkg.mkdir(parents=True)
    for d in (tmp_path / "sglang", tmp_path / "sglang" / "srt", tmp_path / "sglang" / "srt" / "layers"):
        (d / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text(STUB_REGISTRY)
    (pkg / "awq.py").write_text(STUB_AWQ)
    for sub, name, text in (("model_executor", "model_runner.py", STUB_RUNNER), ("model_loader", "loader.py", STUB_LOADER)):
        d = tmp_path / "sglang" / "srt" / sub
        d.mkdir()
        (d / "__init__.py").write_text("")
        (d / name).write_text(text)
    monkeypatch.syspath_prepend(str(tmp_path))
    saved = {k: v for k, v in sys.modules.items() if k == "sglang" or k.startswith("sglang.") or k == "sgl_kernel"}
    for k in saved:
        del sys.modules[k]
    importlib.invalidate_caches()
    yield
    from sglang_awq_amd import sgl_kernel_compat

    sgl_kernel_compat.uninstall()
    for k in [k for k in sys.modules if k == "sglang" or k.startswith("sglang.") or k == "sgl_kernel"]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_install_then_import_patches_hip_branch(stub_reference):
    from sglang_awq_amd import sgl_kernel_compat as compat

    sk = compat.install()
    assert sk.awq_dequantize is compat.awq_dequantize and sk.awq_gemm is compat.awq_gemm
    import sgl_kernel

    assert sgl_kernel.awq_dequantize is compat.awq_dequantize
    mod = importlib.import_module(compat.REFERENCE_AWQ_MODULE)         # imported AFTER install(): patched by the post-import hook
    assert mod.awq_dequantize is compat.awq_dequantize
    assert mod.AWQLinearMethod().apply(None, None) is compat.awq_dequantize    # what apply() will call
    assert getattr(mod, "awq_dequantize_triton")(0, 0, 0) == "triton"           # the Triton function itself is untouched
    compat.uninstall()
    assert mod.awq_dequantize is mod.awq_dequantize_triton


def test_import_then_install_patches_in_place_and_registers_config(stub_reference):
    from sglang_awq_amd import sgl_kernel_compat as compat
    from sglang_awq_amd.awq import AWQConfig

    mod = importlib.import_module(compat.REFERENCE_AWQ_MODULE)         # the reference is already up: Triton is bound
    assert mod.AWQLinearMethod().apply(None, None)(0, 0, 0) == "triton"
    compat.install(register_config=True)
    assert mod.awq_dequantize is compat.awq_dequantize
    reg = importlib.import_module(compat.REFERENCE_REGISTRY_MODULE)
    assert reg.QUANTIZATION_METHODS["awq"] is AWQConfig and reg.BASE_QUANTIZATION_METHODS["awq"] is AWQConfig
    assert reg.QUANTIZATION_METHODS["gptq"] is object
    compat.install()                                                   # idempotent
    assert mod.awq_dequantize is compat.awq_dequantize
    compat.uninstall()
    assert mod.awq_dequantize is mod.awq_dequantize_triton


def test_patch_can_be_declined(stub_reference):
    from sglang_awq_amd import sgl_kernel_compat as compat

    compat.install(patch_reference=False)
    mod = importlib.import_module(compat.REFERENCE_AWQ_MODULE)
    assert mod.awq_dequantize is mod.awq_dequantize_triton


class _CountingMethod:
    """Stands in for this package's AWQLinearMethod in the hook tests (no GPU here): counts re-layouts."""

    def __init__(self):
        self.n = 0

    def process_weights_after_loading(self, layer):
        self.n += 1


def _model_with_our_method(monkeypatch):
    import torch

    from sglang_awq_amd import weight_update

    method = _CountingMethod()
    monkeypatch.setattr(weight_update, "_methods_of_this_package", lambda: (_CountingMethod,))
    lin = torch.nn.Linear(4, 4, bias=False)
    lin.quant_method = method
    lin.__dict__["_gu_il"] = "stale interleaved copy"
    return torch.nn.Sequential(lin), lin, method


@pytest.mark.parametrize("import_first", [False, True])
def test_install_hooks_the_weight_update_paths(stub_reference, monkeypatch, import_first):
    """install(hook_weight_updates=True): every ModelRunner.update_weights_from_* (model_runner.py:969, 1191, 1281) ends with
    weights_updated(self.model) — the op's cache is dropped, the layers of this package re-derive their copies, lazily derived
    copies are forgotten — and load_weights_and_postprocess (loader.py:616-632) drops the cache; the op's cache goes from
    "auto: off" to on because invalidation is now guaranteed.  Patched in place or on import; uninstall() restores the originals."""
    import torch

    from sglang_awq_amd import ops
    from sglang_awq_amd import sgl_kernel_compat as compat

    monkeypatch.setattr(ops, "_OP_CACHE_MODE", "auto")
    monkeypatch.setattr(ops, "_OP_CACHE_ENABLED", False)
    if import_first:
        runner_mod = importlib.import_module(compat.REFERENCE_RUNNER_MODULE)
        loader_mod = importlib.import_module(compat.REFERENCE_LOADER_MODULE)
    compat.install()
    assert ops.awq_gemm_cache_info()["enabled"] is True
    if not import_first:
        runner_mod = importlib.import_module(compat.REFERENCE_RUNNER_MODULE)
        loader_mod = importlib.import_module(compat.REFERENCE_LOADER_MODULE)
    model, lin, method = _model_with_our_method(monkeypatch)
    cleared = []
    monkeypatch.setattr(ops, "awq_gemm_cache_clear", lambda: cleared.append(1))
    runner = runner_mod.ModelRunner(model)
    new = torch.full((4, 4), 3.0)
    assert runner.update_weights_from_tensor([("0.weight", new)]) == (True, "Success")
    assert torch.equal(lin.weight.data, new) and method.n == 1 and len(cleared) == 1 and "_gu_il" not in lin.__dict__
    assert runner.update_weights_from_disk("/x", "auto") == (True, "Success") and method.n == 2
    assert runner.update_weights_from_distributed([], [], [], "g") == (True, "Success") and method.n == 3
    assert runner.calls == ["tensor", "disk", "distributed"]
    n_clear = len(cleared)
    assert loader_mod.DefaultModelLoader.load_weights_and_postprocess(model, [], "cpu") == "postprocessed"
    assert method.n == 4 and len(cleared) == n_clear + 1            # the loader re-laid out itself; the hook only drops the cache
    compat.install()                                                # idempotent: nothing is wrapped twice
    runner.update_weights_from_tensor([("0.weight", new)])
    assert method.n == 5
    compat.uninstall()
    runner.update_weights_from_tensor([("0.weight", new)])
    assert method.n == 5, "uninstall() must restore the reference's own methods"


def test_hooks_can_be_declined(stub_reference, monkeypatch):
    from sglang_awq_amd import ops
    from sglang_awq_amd import sgl_kernel_compat as compat

    monkeypatch.setattr(ops, "_OP_CACHE_MODE", "auto")
    monkeypatch.setattr(ops, "_OP_CACHE_ENABLED", False)
    compat.install(hook_weight_updates=False)
    assert ops.awq_gemm_cache_info()["enabled"] is False            # nobody vouches for invalidation: the cache stays off
    runner_mod = importlib.import_module(compat.REFERENCE_RUNNER_MODULE)
    assert not hasattr(runner_mod.ModelRunner.update_weights_from_tensor, "_sglang_awq_amd_wrapped")
