"""CPU-side checks: the C-ABI library loads, exports exactly what include/awq_hip.h declares, rejects
bad arguments before touching the GPU, and the host-side mirror of the reference interface behaves
like the reference (shapes, errors).  No compute calls: there is no GPU here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from sglang_awq_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def _header_functions(name="awq_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awq_[a-z_]+)\s*\(", text)))


def test_header_and_export_list_agree(lib):
    assert _header_functions() == sorted(_lib.EXPORTS)
    assert _header_functions("awq_aux.h") == sorted(_lib.AUX_EXPORTS)
    for sym in _lib.EXPORTS + _lib.AUX_EXPORTS:
        assert hasattr(lib, sym), sym


def test_shared_object_exports_symbols_with_c_linkage():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for sym in _lib.EXPORTS + _lib.AUX_EXPORTS:
        assert sym in names, f"{sym} not exported unmangled"


def test_abi_version_and_strings(lib):
    assert lib.awq_hip_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.awq_hip_build_info()
    assert lib.awq_hip_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6, -7, -100):
        assert lib.awq_hip_status_string(code) != lib.awq_hip_status_string(12345)


def test_argument_validation_without_gpu(lib):
    """Every rejection below returns before any HIP call."""
    buf = (ctypes.c_char * 4096)()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    p = (p + 15) & ~15
    vp = ctypes.c_void_p
    assert lib.awq_dequantize(None, vp(p), vp(p), vp(p), 128, 64, 128, 0, None) == -1
    assert lib.awq_dequantize(vp(p), vp(p), vp(p), None, 128, 64, 128, 0, None) == -1
    assert lib.awq_dequantize(vp(p), vp(p), vp(p), vp(p), 128, 60, 128, 0, None) == -2   # N % 8
    assert lib.awq_dequantize(vp(p), vp(p), vp(p), vp(p), 100, 64, 64, 0, None) == -2    # K % g
    assert lib.awq_dequantize(vp(p), vp(p), vp(p), vp(p), 0, 64, 64, 0, None) == -2
    assert lib.awq_dequantize(vp(p), vp(p), vp(p), vp(p), 128, 64, 128, 7, None) == -3
    assert lib.awq_dequantize(vp(p), vp(p + 2), vp(p), vp(p), 128, 64, 128, 0, None) == -6
    g = lambda **kw: lib.awq_gemm(kw.get("x", vp(p)), kw.get("ldx", 128), vp(p), vp(p), vp(p), None, kw.get("y", vp(p)),
                                  vp(p), 4096, kw.get("M", 1), 128, 64, 128, 0, kw.get("sk", 1), None)
    assert g(sk=3) == -4
    assert g(sk=64) == -4
    assert g(ldx=64) == -2
    assert g(x=None) == -1
    assert g(M=0) == 0            # empty batch: nothing to launch
    assert lib.awq_gemm_ex(vp(p), 128, vp(p), vp(p), vp(p), None, vp(p), vp(p), 4096, 1, 128, 64, 128, 0, 1, 99, 0, None) == -7


def test_aux_argument_validation_without_gpu(lib):
    """include/awq_aux.h entry points: rejections that return before any HIP call, and shapes the fused GEMV has no
    instantiation for (AWQ_ERR_BAD_VARIANT = -7: the caller runs the separate ops)."""
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.cast(buf, ctypes.c_void_p).value + 15) & ~15
    vp = ctypes.c_void_p
    f = lambda **kw: lib.awq_aux_gemv_repacked_fused(kw.get("x", vp(p)), kw.get("ldx", 4096), kw.get("packed", vp(p)), kw.get("y", vp(p)),
                                                     kw.get("M", 1), kw.get("K", 4096), kw.get("N", 4096), kw.get("g", 128), 0,
                                                     kw.get("h", None), kw.get("delta", None), kw.get("w", None), kw.get("h_out", None),
                                                     1e-5, kw.get("silu", 1), None)
    assert f(packed=None) == -1 and f(y=None) == -1 and f(x=None) == -1
    assert f(N=4092) == -2 and f(K=4100) == -2 and f(M=0) == -2 and f(ldx=128) == -2
    assert f(packed=vp(p + 4)) == -6
    assert f(M=33) == -7                                              # more than two row tiles
    assert f(g=64) == -7                                              # no repacked form for g = 64
    assert f(N=4104) == -7                                            # SiLU-mul needs whole (gate, up) pairs of 16-column groups
    assert f(h=vp(p), delta=None, w=vp(p), h_out=vp(p + 64), silu=0) == -7        # norm prologue without delta
    assert f(h=vp(p), delta=vp(p), w=vp(p), h_out=vp(p), silu=0) == -7             # h_out aliases h
    assert f(M=16, h=vp(p), delta=vp(p), w=vp(p), h_out=vp(p + 64), silu=0) == -7  # too many rows for the prologue
    assert f(K=32768, ldx=32768, h=vp(p), delta=vp(p), w=vp(p), h_out=vp(p + 64), silu=0) == -7   # > 8 k-blocks per wave
    att = lambda **kw: lib.awq_aux_decode_attention(kw.get("qkv", vp(p)), vp(p), vp(p), vp(p), vp(p), vp(p), vp(p), 1, 8, kw.get("hkv", 8),
                                                    kw.get("D", 128), 64, 1.0, kw.get("splits", 1), kw.get("ws", None), kw.get("wsb", 0), None)
    assert att(qkv=None) == -1
    assert att(hkv=3) == -2                                            # Hq % Hkv
    assert att(D=96) == -2                                             # head dim
    assert att(splits=0) == -2 and att(splits=17) == -2
    assert att(splits=4) == -5                                         # split-S without workspace
    need = lib.awq_aux_decode_attention_workspace_bytes(1, 8, 128, 4)
    assert need >= 8 * 4 + 8 * 4 * 130 * 4 and lib.awq_aux_decode_attention_workspace_bytes(1, 8, 128, 1) == 0
    assert att(splits=4, ws=vp(p), wsb=need - 1) == -5
    assert lib.awq_aux_add_rmsnorm(vp(p), None, vp(p), vp(p), 1, 100, 1e-5, None) == -2      # H % 8
    assert lib.awq_aux_silu_mul(vp(p), None, 1, 64, None) == -1
    # AWQ-MoE block alignment and the tile route (rejections before any HIP call)
    al = lambda **kw: lib.awq_aux_moe_align_blocks_n(kw.get("ids", vp(p)), kw.get("pairs", 100), kw.get("E", 8), kw.get("rows", 128), vp(p), vp(p),
                                                     kw.get("B", 9), None)
    assert al(ids=None) == -1
    assert al(rows=32) == -2 and al(rows=0) == -2                      # block sizes: 16, 64, 128
    assert al(B=8) == -2                                               # fewer blocks than ceil(pairs / rows) + experts
    assert al(E=2000, B=2001) == -7                                    # more than 1024 experts: the tensor-op form (caller)
    mg = lambda **kw: lib.awq_aux_moe_gemm_blocks(kw.get("x", vp(p)), kw.get("ldx", 4096), kw.get("x_div", 2), vp(p), kw.get("stride", 1 << 20), vp(p),
                                                  vp(p), kw.get("B", 9), kw.get("rows", 128), None, kw.get("y", vp(p)), kw.get("K", 4096),
                                                  kw.get("N", 4096), kw.get("g", 128), 0, kw.get("silu", 0), None)
    assert mg(x=None) == -1 and mg(y=None) == -1
    assert mg(rows=16) == -2 and mg(B=0) == -2 and mg(x_div=0) == -2 and mg(ldx=128) == -2
    assert mg(stride=(1 << 20) + 4) == -6
    assert mg(g=64) == -7                                              # no repacked form for g = 64
    assert mg(N=4104, silu=1) == -7                                    # SiLU-mul needs whole (gate, up) pairs of 16-column groups


def test_workspace_query(lib):
    small = lib.awq_gemm_workspace_bytes(1, 4096, 11008, 128, 0)
    assert 4096 <= small <= 4096 + (32 << 20)
    # fp16 calls beyond the decode range: room for the on-the-fly re-layout (head + one repacked copy); bf16 / g = 64 keep the slab size
    assert lib.awq_gemm_workspace_bytes(2048, 4096, 11008, 128, 0) == 4096 + lib.awq_repacked_bytes(4096, 11008, 128, 0)
    assert lib.awq_gemm_workspace_bytes(2048, 4096, 11008, 128, 1) <= 4096 + (32 << 20)
    assert lib.awq_gemm_workspace_bytes(2048, 4096, 11008, 64, 0) <= 4096 + (32 << 20)
    assert lib.awq_gemm_workspace_bytes(64, 4096, 11008, 128, 0) == 4096 + lib.awq_repacked_bytes(4096, 11008, 128, 0)
    assert lib.awq_gemm_workspace_bytes(32, 4096, 11008, 128, 0) <= 4096 + (32 << 20)
    assert lib.awq_gemm_workspace_bytes(0, 0, 0, 128, 0) >= 0


def test_ops_registered_with_reference_schema():
    from sglang_awq_amd import ops  # noqa: F401

    s = torch.ops.sgl_kernel.awq_dequantize.default._schema
    assert [a.name for a in s.arguments] == ["qweight", "scales", "qzeros"]          # common_extension.cc:126
    s = torch.ops.sgl_kernel.awq_gemm.default._schema
    assert [a.name for a in s.arguments] == ["input", "qweight", "scales", "qzeros", "split_k_iters"]  # awq_triton.py:289-294


def test_ops_have_no_cpu_fallback():
    from sglang_awq_amd import ops

    qw = torch.zeros(128, 8, dtype=torch.int32)
    qz = torch.zeros(1, 8, dtype=torch.int32)
    sc = torch.ones(1, 64, dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        ops.awq_dequantize(qw, sc, qz)
    with pytest.raises(NotImplementedError):
        ops.awq_gemm(torch.ones(1, 128, dtype=torch.float16), qw, sc, qz, 1)


def test_fake_impls_have_correct_arity_and_shapes():
    from sglang_awq_amd import ops  # noqa: F401

    qw = torch.empty(256, 16, dtype=torch.int32, device="meta")
    qz = torch.empty(2, 16, dtype=torch.int32, device="meta")
    sc = torch.empty(2, 128, dtype=torch.bfloat16, device="meta")
    out = torch.ops.sgl_kernel.awq_dequantize(qw, sc, qz)
    assert out.shape == (256, 128) and out.dtype == torch.bfloat16
    x = torch.empty(5, 256, dtype=torch.bfloat16, device="meta")
    y = torch.ops.sgl_kernel.awq_gemm(x, qw, sc, qz, 8)
    assert y.shape == (5, 128) and y.dtype == torch.bfloat16


def test_sgl_kernel_compat_install():
    import sys

    from sglang_awq_amd import sgl_kernel_compat

    saved = sys.modules.pop("sgl_kernel", None)
    try:
        mod = sgl_kernel_compat.install()
        import sgl_kernel

        assert sgl_kernel is mod
        assert callable(sgl_kernel.awq_dequantize) and callable(sgl_kernel.awq_gemm)
    finally:
        sys.modules.pop("sgl_kernel", None)
        if saved is not None:
            sys.modules["sgl_kernel"] = saved


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AwqHipError, match="no CPU fallback"):
        _lib.load()


def test_repacked_entry_points_refuse_shapes_without_a_layout(lib):
    """awq_gemm_repacked indexes `packed` from (K, N, group_size) alone: shapes the fragment-major layout does not exist for
    (fp32, K % 128, group sizes other than 32 / 64 / multiples of 128) must come back as AWQ_ERR_BAD_VARIANT before anything
    is launched; fp16 and bf16 with g in {32, 64, 128 k} have one (same bytes for both dtypes)."""
    buf = (ctypes.c_char * 8192)()
    p = (ctypes.cast(buf, ctypes.c_void_p).value + 15) & ~15
    vp = ctypes.c_void_p
    assert lib.awq_repacked_bytes(4096, 11008, 128, 0) == 688 * 32 * 1024 + 688 * 32 * 64
    assert lib.awq_repacked_bytes(4096, 11008, 128, 1) == lib.awq_repacked_bytes(4096, 11008, 128, 0)
    assert lib.awq_repacked_bytes(4096, 11008, 64, 0) == 688 * 32 * 1024 + 688 * 64 * 64
    assert lib.awq_repacked_bytes(4096, 11008, 32, 1) == 688 * 32 * 1024 + 688 * 128 * 64
    assert lib.awq_repacked_bytes(4096, 11008, 128, 2) == 0 and lib.awq_repacked_bytes(4096, 11008, 16, 0) == 0
    assert lib.awq_repacked_bytes(192, 64, 64, 0) == 0 and lib.awq_repacked_bytes(4096, 64, 4096 // 3 + 1, 0) == 0
    # K = 192 (not a multiple of 128), group 64: legal AWQ shape, no repacked layout -> -7, for every M route
    for M in (1, 16, 32, 64, 100, 200, 2048):
        assert lib.awq_gemm_repacked(vp(p), 192, vp(p), None, vp(p), M, 192, 64, 64, 0, None) == -7, M
    assert lib.awq_gemm_repacked(vp(p), 4096, vp(p), None, vp(p), 128, 4096, 4096, 128, 2, None) == -7      # fp32
    assert lib.awq_gemm_repacked(vp(p), 4096, vp(p), None, vp(p), 128, 4096, 4096, 16, 0, None) == -7       # g = 16
    assert lib.awq_gemm_repacked(None, 4096, vp(p), None, vp(p), 1, 4096, 4096, 128, 0, None) == -1
    # the workspace variant validates the same way (nothing is launched: every call fails before the dispatch)
    assert lib.awq_gemm_repacked_ws(vp(p), 4096, vp(p), None, vp(p), None, 0, 128, 4096, 4096, 16, 0, None) == -7
    assert lib.awq_gemm_repacked_ws(None, 4096, vp(p), None, vp(p), None, 0, 1, 4096, 4096, 128, 0, None) == -1
    # scratch is wanted for 9 .. 32 rows of a supported fp16 layout on a narrow matrix (split-K GEMV) and from 33 rows where the MFMA
    # tiling leaves most CUs idle (split-K tiles: up to 8 slices of M x N fp32, at most 32 MiB); never for one row, bf16, or many tiles
    assert lib.awq_gemm_repacked_workspace_bytes(1, 11008, 4096, 128, 0) == 0
    assert lib.awq_gemm_repacked_workspace_bytes(64, 11008, 4096, 128, 0) == 4096 + 8 * 64 * 4096 * 4
    assert lib.awq_gemm_repacked_workspace_bytes(512, 11008, 4096, 128, 0) == 4096 + 4 * 512 * 4096 * 4      # 32 MiB budget: 4 slices
    assert lib.awq_gemm_repacked_workspace_bytes(2048, 4096, 11008, 128, 0) == 0                                # 688 tiles: no split
    assert lib.awq_gemm_repacked_workspace_bytes(64, 11008, 4096, 128, 1) == 0
    assert lib.awq_gemm_repacked_workspace_bytes(32, 11008, 4096, 128, 1) == 0
    assert lib.awq_gemm_repacked_workspace_bytes(32, 4096, 22016, 128, 0) == 0
    assert 0 < lib.awq_gemm_repacked_workspace_bytes(32, 11008, 4096, 128, 0) <= 4096 + (32 << 20)
    # the fused decode variants exist for fp16 with g % 128 == 0 only
    assert lib.awq_aux_gemv_repacked_fused(vp(p), 192, vp(p), vp(p), 1, 192, 64, 64, 0, None, None, None, None, 0.0, 1, None) == -7
    assert lib.awq_aux_gemv_repacked_fused(vp(p), 4096, vp(p), vp(p), 1, 4096, 4096, 64, 0, None, None, None, None, 0.0, 1, None) == -7
    assert lib.awq_aux_gemv_repacked_fused(vp(p), 4096, vp(p), vp(p), 1, 4096, 4096, 128, 1, None, None, None, None, 0.0, 1, None) == -7


def test_python_shim_checks_the_repacked_buffer():
    """ops.check_packed: a repacked buffer of the wrong size / dtype / device is refused before any kernel could read past it."""
    from sglang_awq_amd import ops

    cpu = torch.device("cpu")
    need = 688 * 32 * 1024 + 688 * 32 * 64
    ops.check_packed(torch.empty(need, dtype=torch.uint8), 4096, 11008, 128, cpu)
    with pytest.raises(RuntimeError):
        ops.check_packed(torch.empty(need - 64, dtype=torch.uint8), 4096, 11008, 128, cpu)
    with pytest.raises(RuntimeError):
        ops.check_packed(torch.empty(need // 4, dtype=torch.int32), 4096, 11008, 128, cpu)
    with pytest.raises(RuntimeError):
        ops.check_packed(torch.empty(need, dtype=torch.uint8), 4096, 11008, 256, cpu)     # same K, N, other group size
    ops.check_packed(torch.empty(688 * 32 * 1024 + 688 * 64 * 64, dtype=torch.uint8), 4096, 11008, 64, cpu, torch.bfloat16)
    with pytest.raises(ops.AwqHipError):
        ops.check_packed(torch.empty(16, dtype=torch.uint8), 4096, 11008, 16, cpu)        # no layout for g = 16


def test_awq_moe_method_weight_shapes_cpu():
    """AWQMoEMethod.create_weights registers the reference's six parameters with its shapes (awq.py:669-757) and rejects
    misaligned expert sizes; no GPU needed."""
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.moe import AWQMoEMethod, select_experts

    m = AWQMoEMethod(AWQConfig(4, 128, True))
    layer = torch.nn.Module()
    m.create_weights(layer, 8, 4096, 14336, torch.float16, weight_loader=None)
    shapes = {n: tuple(p.shape) for n, p in layer.named_parameters()}
    assert shapes == {"w13_qweight": (8, 4096, 3584), "w2_qweight": (8, 14336, 512), "w13_scales": (8, 32, 28672),
                      "w2_scales": (8, 112, 4096), "w13_qzeros": (8, 32, 3584), "w2_qzeros": (8, 112, 512)}
    assert layer.w13_qweight.dtype == torch.int32 and layer.w2_scales.dtype == torch.float16
    with pytest.raises(ValueError):
        m.create_weights(torch.nn.Module(), 2, 4096, 100, torch.float16)
    with pytest.raises(ValueError):
        AWQMoEMethod(AWQConfig.__new__(AWQConfig)) if False else AWQMoEMethod(type("C", (), {"weight_bits": 8, "group_size": 128, "pack_factor": 4})())
    w, i = select_experts(torch.tensor([[0.0, 2.0, 1.0, -1.0]]), 2)
    assert i.tolist() == [[1, 2]] and abs(float(w.sum()) - 1.0) < 1e-6 and w[0, 0] > w[0, 1]
