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
STUB_REGISTRY = "BASE_QUANTIZATION_METHODS = {'awq': object, 'gptq': object}\nQUANTIZATION_METHODS = {**BASE_QUANTIZATION_METHODS}\n"


@pytest.fixture
def stub_reference(tmp_path, monkeypatch):
    pkg = tmp_path / "sglang" / "srt" / "layers" / "quantization"
    pkg.mkdir(parents=True)
    for d in (tmp_path / "sglang", tmp_path / "sglang" / "srt", tmp_path / "sglang" / "srt" / "layers"):
        (d / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text(STUB_REGISTRY)
    (pkg / "awq.py").write_text(STUB_AWQ)
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
