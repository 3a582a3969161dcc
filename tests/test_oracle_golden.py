"""The oracle (oracle/awq_ref.py numpy, oracle/awq_oracle.c C) against golden vectors captured from
the reference's own CPU-runnable functions by tests/golden/make_golden.py.  Bit-exact for the
dequantise (integer unpack + one rounding); the reference's own tolerance for its Triton GEMM."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import awq_ref, c_oracle
from sglang_awq_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(GOLD, "awq_dequant_small.npz"))


def test_dequant_small_cases_bit_exact(small):
    n = int(small["n_dequant"])
    assert n >= 10
    for i in range(n):
        K, N, g, seed = (int(v) for v in small[f"dq{i}_meta"])
        dt, fam = str(small[f"dq{i}_dtype"]), str(small[f"dq{i}_family"])
        qw, s, qz = small[f"dq{i}_qweight"], small[f"dq{i}_scales"], small[f"dq{i}_qzeros"]
        # the stored inputs are what synth regenerates (keeps digests.json meaningful)
        rq, rs, rz = synth.make_awq_weights(K, N, g, dt, fam, seed)
        assert np.array_equal(rq, qw) and np.array_equal(rs.view(np.uint8), s.view(np.uint8)) and np.array_equal(rz, qz)
        want = small[f"dq{i}_out"]
        got_np = awq_ref.awq_dequantize(qw, s, qz)
        got_c = c_oracle.dequantize(qw, s, qz)
        assert got_np.dtype == want.dtype and got_np.shape == (K, N)
        assert np.array_equal(got_np.view(np.uint8), want.view(np.uint8)), f"numpy oracle differs, case {i}"
        assert np.array_equal(got_c.view(np.uint8), want.view(np.uint8)), f"C oracle differs, case {i}"


def test_dequant_reference_test_inputs_verbatim(small):
    for i in range(int(small["n_verbatim"])):
        qw, s, qz, want = (small[f"verb{i}_{k}"] for k in ("qweight", "scales", "qzeros", "out"))
        assert np.array_equal(awq_ref.awq_dequantize(qw, s, qz).view(np.uint16), want.view(np.uint16))
        assert np.array_equal(c_oracle.dequantize(qw, s, qz).view(np.uint16), want.view(np.uint16))


def test_unpack_order_and_roundtrip():
    rng = np.random.default_rng(0)
    vals = rng.integers(0, 16, size=(7, 40), dtype=np.uint8)
    packed = awq_ref.pack_awq_int4(vals)
    assert packed.dtype == np.int32 and packed.shape == (7, 5)
    assert np.array_equal(awq_ref.unpack_awq_int4(packed), vals)
    assert np.array_equal(c_oracle.unpack(packed), vals)
    # one word by hand: nibble p of the word holds logical column Q[p] = [0,2,4,6,1,3,5,7][p]
    word = np.array([[0x76543210]], dtype=np.uint32).view(np.int32)
    assert awq_ref.unpack_awq_int4(word).tolist() == [[0, 4, 1, 5, 2, 6, 3, 7]]
    # bit 31 set (never exercised by the reference tests: randint(0, int32.max))
    word = np.array([[0xF0000000]], dtype=np.uint32).view(np.int32)
    assert awq_ref.unpack_awq_int4(word).tolist() == [[0, 0, 0, 0, 0, 0, 0, 15]]


def _digest_cases():
    with open(os.path.join(GOLD, "digests.json")) as f:
        return json.load(f)["cases"]


def test_dequant_digests_c_oracle():
    """Every digest case (reference test grids + the BASELINE shapes) through the C oracle."""
    cases = _digest_cases()
    assert len(cases) >= 60
    for c in cases:
        qw, s, qz = synth.make_awq_weights(c["K"], c["N"], c["g"], c["dtype"], c["family"], c["seed"])
        assert _sha(qw) + _sha(s)[:16] + _sha(qz)[:16] == c["inputs_sha256"], "synthetic inputs drifted"
        out = c_oracle.dequantize(qw, s, qz)
        assert _sha(out) == c["out_sha256"], f"C oracle digest mismatch for {c}"


def test_dequant_digests_numpy_oracle_subset():
    """numpy oracle on the headline shape and a few of the grid cases (it is the slower one)."""
    cases = [c for c in _digest_cases() if (c["K"], c["N"]) == (4096, 11008) or c["K"] * c["N"] <= 2 ** 21]
    assert any(c["K"] == 4096 and c["N"] == 11008 and c["g"] == 128 for c in cases)
    for c in cases:
        qw, s, qz = synth.make_awq_weights(c["K"], c["N"], c["g"], c["dtype"], c["family"], c["seed"])
        assert _sha(awq_ref.awq_dequantize(qw, s, qz)) == c["out_sha256"]


def test_gemm_against_reference_triton_outputs():
    """awq_gemm_triton's own outputs (fp32 inputs, K = 128, split-K 1 and 8) at the tolerance the
    reference test uses (atol = rtol = 1e-1, test_awq_dequant.py:171) — and far tighter against the
    reference's matmul(x, dequant), since the oracle and that matmul both carry fp32-or-wider sums."""
    z = np.load(os.path.join(GOLD, "awq_gemm_triton_f32.npz"))
    n = int(z["n_cases"])
    assert n == 8 * 3 * 4 * 2
    for i in range(n):
        M, K, N, g, sk = (int(v) for v in z[f"g{i}_meta"])
        x, qw, s, qz = z[f"g{i}_x"], z[f"g{i}_qweight"], z[f"g{i}_scales"], z[f"g{i}_qzeros"]
        got = awq_ref.awq_gemm(x, qw, s, qz, sk)
        assert got.dtype == np.float32 and got.shape == (M, N)
        np.testing.assert_allclose(got, z[f"g{i}_triton"], atol=1e-1, rtol=1e-1)
        np.testing.assert_allclose(got, z[f"g{i}_matmul"], atol=1e-3, rtol=1e-5)
        np.testing.assert_allclose(c_oracle.gemm(x, qw, s, qz), z[f"g{i}_matmul"], atol=1e-3, rtol=1e-5)


def test_apply_against_reference_cpu_path():
    """dequantise + torch.matmul (+ add_ bias) executed by the reference's functions on CPU in the
    build container.  torch's CPU half matmul rounds differently from an exact sum, so the check is
    one output-ulp, not bit equality (third-party arithmetic: SURVEY.md §8c)."""
    z = np.load(os.path.join(GOLD, "awq_apply_cpu.npz"))
    for i in range(int(z["n_cases"])):
        M, K, N, g, seed, has_bias = (int(v) for v in z[f"a{i}_meta"])
        dt = str(z[f"a{i}_dtype"])
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, "A", seed)
        x = synth.make_activations(M, K, dt, "A", seed)
        b = synth.make_bias(N, dt, seed) if has_bias else None
        got = awq_ref.to_f64(awq_ref.awq_linear_apply(x, qw, s, qz, b), dt)
        want = awq_ref.to_f64(z[f"a{i}_out"], dt)
        ulp = np.maximum(np.abs(want), 2.0 ** -14) * (2.0 ** -10 if dt == "f16" else 2.0 ** -7)
        assert np.all(np.abs(got - want) <= 2.0 * ulp + 1e-3), f"case {i}: max err {np.abs(got - want).max()}"


def test_c_and_numpy_gemm_agree_fp16_bf16():
    for dt, seed in (("f16", 31), ("bf16", 32)):
        qw, s, qz = synth.make_awq_weights(256, 96, 64, dt, "A", seed)
        x = synth.make_activations(7, 256, dt, "A", seed)
        y_c, exact_c = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        exact_np = awq_ref.awq_gemm_exact(x, qw, s, qz)
        np.testing.assert_allclose(exact_c, exact_np, rtol=1e-12, atol=1e-12)
        y_np = awq_ref.awq_gemm(x, qw, s, qz)
        # both round one (nearly identical) double once; allow a tie to fall either way
        d = np.abs(awq_ref.to_f64(y_c, dt) - awq_ref.to_f64(y_np, dt))
        assert (d > 0).mean() < 1e-3


def test_shape_contract_errors():
    qw, s, qz = synth.make_awq_weights(128, 64, 32, "f16", "A", 1)
    with pytest.raises(ValueError):
        awq_ref.awq_dequantize(qw, s[:, :-8], qz)
    with pytest.raises(ValueError):
        awq_ref.awq_dequantize(qw, s, qz[:, :-1])
    with pytest.raises(ValueError):
        awq_ref.awq_gemm(synth.make_activations(2, 128), qw, s, qz, split_k_iters=3)
    with pytest.raises(ValueError):
        awq_ref.awq_gemm(synth.make_activations(2, 128), qw, s, qz, split_k_iters=64)


def test_torch_cpu_baseline_dequantize_bit_exact(small):
    """oracle/torch_cpu.dequantize_cpu — the eager-torch restatement that bench.py's cpu_baseline leg times — against the
    reference-generated golden outputs, bit for bit (fp16 and bf16 cases, every group size in the file)."""
    from oracle import torch_cpu

    checked = 0
    for i in range(int(small["n_dequant"])):
        dt = str(small[f"dq{i}_dtype"])
        if dt not in ("f16", "bf16"):
            continue
        qw, s, qz, want = (small[f"dq{i}_{k}"] for k in ("qweight", "scales", "qzeros", "out"))
        got = torch_cpu.to_np(torch_cpu.dequantize_cpu(torch_cpu.from_np(qw), torch_cpu.from_np(s, bf16=dt == "bf16"), torch_cpu.from_np(qz)))
        assert np.array_equal(np.ascontiguousarray(got).view(np.uint8), np.ascontiguousarray(want).view(np.uint8)), f"case {i} ({dt})"
        checked += 1
    assert checked >= 8
    for i in range(int(small["n_verbatim"])):
        qw, s, qz, want = (small[f"verb{i}_{k}"] for k in ("qweight", "scales", "qzeros", "out"))
        got = torch_cpu.to_np(torch_cpu.dequantize_cpu(torch_cpu.from_np(qw), torch_cpu.from_np(s), torch_cpu.from_np(qz)))
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
