"""Parity of the HIP path (through the C ABI, via the torch ops) with the oracle, on a real MI355X.

Bars (BASELINE.json north_star): awq_dequantize bit-exact; fused GEMM within 1e-3 absolute of the
exact sum before the final rounding (plus the half-ulp the rounding itself adds) and, where outputs
stay below 1 in magnitude, within 1e-3 of the reference result outright.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import awq_ref, c_oracle
from sglang_awq_amd import _lib, synth
from tests.util import TORCH_DT, assert_gemm_close, bits, to_np, to_torch, ulp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sglang_awq_amd import ops as _ops   # raises if the HIP library is missing: no fallback

    _lib.load()
    # the op's repacked-copy cache is opt-in (ops.py: off until the reload hooks of sgl_kernel_compat.install() are in place or
    # the caller takes responsibility for invalidation): these tests own their weights and clear the cache themselves
    _ops.awq_gemm_cache_enable(True)
    return _ops


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _dev(*arrs):
    return [to_torch(a, DEV) for a in arrs]


# ------------------------------------------------------------------------------ awq_dequantize
def test_dequantize_golden_small_bit_exact(ops):
    z = np.load(os.path.join(GOLD, "awq_dequant_small.npz"))
    for i in range(int(z["n_dequant"])):
        qw, s, qz = z[f"dq{i}_qweight"], z[f"dq{i}_scales"], z[f"dq{i}_qzeros"]
        out = to_np(ops.awq_dequantize(*_dev(qw, s, qz)))
        assert out.shape == z[f"dq{i}_out"].shape
        assert np.array_equal(bits(out), bits(z[f"dq{i}_out"])), f"case {i} {z[f'dq{i}_meta']} {z[f'dq{i}_dtype']}"
    for i in range(int(z["n_verbatim"])):
        qw, s, qz = z[f"verb{i}_qweight"], z[f"verb{i}_scales"], z[f"verb{i}_qzeros"]
        out = to_np(ops.awq_dequantize(*_dev(qw, s, qz)))
        assert np.array_equal(bits(out), bits(z[f"verb{i}_out"]))


def test_dequantize_digests_reference_grids_and_baseline_shapes(ops):
    """Every digest case: the reference's two test grids (g = K, g in {32, 64, 128}; fp16 + bf16) and the
    BASELINE shapes incl. 4096 x 11008 g128, all three input families — sha256 of the output bytes."""
    with open(os.path.join(GOLD, "digests.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        qw, s, qz = synth.make_awq_weights(c["K"], c["N"], c["g"], c["dtype"], c["family"], c["seed"])
        out = ops.awq_dequantize(*_dev(qw, s, qz))
        assert out.dtype == TORCH_DT[c["dtype"]] and tuple(out.shape) == (c["K"], c["N"])
        assert _sha(to_np(out)) == c["out_sha256"], f"digest mismatch {c}"


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_dequantize_ragged_shapes_vs_oracle(ops, dt):
    # K not a multiple of the 32-row workgroup tile, packed columns not a multiple of 64, g = K,
    # one group spanning several lanes' row blocks, a single packed column
    for K, N, g in [(8, 8, 8), (40, 8, 8), (72, 520, 24), (100, 64, 100), (96, 1032, 32), (3, 16, 1), (130, 72, 65)]:
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, "F", seed=K * 7 + N)
        out = to_np(ops.awq_dequantize(*_dev(qw, s, qz)))
        want = awq_ref.awq_dequantize(qw, s, qz)
        assert np.array_equal(bits(out), bits(want)), (K, N, g, dt)


def test_dequantize_extreme_scales_bit_exact(ops):
    """fp16 subnormal / tiny / huge scales: the product must round (and underflow) like the oracle."""
    K, N, g = 64, 64, 32
    qw, _, qz = synth.make_awq_weights(K, N, g, "f16", "F", seed=5)
    raw = synth.rand_u32((K // g, N), 77, 9)
    s = (raw & 0x7BFF).astype(np.uint16).view(np.float16)          # every finite non-negative half pattern
    s[0, :8] = np.array([0, 6e-8, 1.2e-7, 6.1e-5, 6.104e-5, 65504, 4368, 1e-3], dtype=np.float16)
    out = to_np(ops.awq_dequantize(*_dev(qw, s, qz)))
    assert np.array_equal(bits(out), bits(awq_ref.awq_dequantize(qw, s, qz)))


def test_dequantize_errors(ops):
    qw, s, qz = _dev(*synth.make_awq_weights(128, 64, 32, "f16", "A", 1))
    with pytest.raises(RuntimeError):
        ops.awq_dequantize(qw.to(torch.int64), s, qz)
    with pytest.raises(RuntimeError):
        ops.awq_dequantize(qw, s[:, :-8], qz)
    with pytest.raises(RuntimeError):
        ops.awq_dequantize(qw, s, qz[:, :-1])
    with pytest.raises(RuntimeError):
        ops.awq_dequantize(qw.t().contiguous().t(), s, qz)            # non-contiguous
    with pytest.raises(RuntimeError):
        ops.awq_dequantize(qw, s.to(torch.float64), qz)


# ------------------------------------------------------------------------------ awq_gemm
def _gemm_case(ops, M, K, N, g, dt, family, seed, variant, tune=0, x_std=1.0, bias=False):
    qw, s, qz = synth.make_awq_weights(K, N, g, dt, family, seed)
    x = synth.make_activations(M, K, dt, family, seed, x_std=x_std)
    b = synth.make_bias(N, dt, seed) if bias else None
    dq, ds, dz, dx = _dev(qw, s, qz, x)
    db = to_torch(b, DEV) if bias else None
    y = to_np(ops.awq_gemm_variant(dx, dq, ds, dz, variant, tune, db))
    _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
    return y, exact, (qw, s, qz, x, b)


def test_gemm_reference_triton_grid_fp32(ops):
    """The reference's own GEMM test grid (fp32, K = 128, split-K 1 / 8) against its stored outputs."""
    z = np.load(os.path.join(GOLD, "awq_gemm_triton_f32.npz"))
    for i in range(int(z["n_cases"])):
        M, K, N, g, sk = (int(v) for v in z[f"g{i}_meta"])
        x, qw, s, qz = _dev(z[f"g{i}_x"], z[f"g{i}_qweight"], z[f"g{i}_scales"], z[f"g{i}_qzeros"])
        y = to_np(ops.awq_gemm(x, qw, s, qz, sk))
        assert y.dtype == np.float32 and y.shape == (M, N)
        np.testing.assert_allclose(y, z[f"g{i}_triton"], atol=1e-1, rtol=1e-1)     # the reference's bar
        np.testing.assert_allclose(y, z[f"g{i}_matmul"], atol=1e-3, rtol=1e-5)     # ours


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 7, 8, 13, 16])
def test_gemm_skinny_vs_oracle(ops, dt, M):
    # tune bits 8-15 force the number of K-slices S (0 = heuristic): 1 slice (no hand-off), odd counts,
    # more slices than k-steps / 4, ragged last column tile (N = 1056, 544, 96), g = 32 / 64 / K
    for (K, N, g, tune) in [(256, 512, 128, 0), (512, 1056, 128, 1 << 8), (1024, 96, 64, 3 << 8), (384, 544, 32, 0),
                            (512, 512, 128, 2 << 8), (2048, 1024, 128, 8 << 8), (4096, 512, 4096, 0),
                            (2048, 512, 128, 200 << 8), (1056 * 2, 1024, 32, 5 << 8)]:
        y, exact, _ = _gemm_case(ops, M, K, N, g, dt, "A", seed=M * 1000 + K + N, variant=_lib.GEMM_SKINNY, tune=tune)
        assert_gemm_close(y, exact, dt, what=f"skinny M={M} K={K} N={N} g={g} tune={tune} {dt}")


@pytest.mark.parametrize("M", [1, 2, 5, 8, 16, 17, 32])
def test_gemm_repacked_vs_oracle(ops, M):
    """MFMA-fragment-major re-layout + its decode kernel: column counts that are not a multiple
    of 16 (padded group), strips of 1..8 column groups, 1..4 k-blocks per wave (straight-line) and more (loop),
    g = 128 and g = K; N > 32768 (and M > 16 with more than 4 groups per CU) runs several rounds of narrow strips."""
    for (K, N, g) in [(128, 16, 128), (256, 72, 128), (512, 1056, 128), (1024, 4096, 128), (4096, 512, 4096),
                      (4096, 1024, 128), (11008 // 86 * 86, 256, 128), (2048, 11008, 128), (6144, 2048 * 11, 128),
                      (256, 32848, 128)]:
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M * 13 + K + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + K)
        b = synth.make_bias(N, "f16", 5)
        packed = ops.awq_repack(*_dev(qw, s, qz))
        assert packed is not None and packed.dtype == torch.uint8
        y = to_np(ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g))
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        assert_gemm_close(y, exact, "f16", what=f"repacked M={M} K={K} N={N} g={g}")
        yb = ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g, to_torch(b, DEV))
        assert torch.equal(yb, to_torch(y, DEV) + to_torch(b, DEV))
    assert ops.awq_repack(*_dev(*synth.make_awq_weights(256, 64, 16, "f16", "A", 1))) is None         # g = 16: no layout
    assert ops.awq_repack(*_dev(*synth.make_awq_weights(192, 64, 64, "f16", "A", 1))) is None         # K % 128 != 0: no layout
    assert ops.awq_repack(*_dev(*synth.make_awq_weights(256, 64, 128, "f32", "A", 1))) is None        # fp32: no layout
    # (g in {32, 64} and bf16 have one since round 2: tests/test_gpu_round2.py)


@pytest.mark.parametrize("M", [1, 3, 8])
def test_gemm_repacked_x_through_lds_vs_oracle(ops, M):
    """16-wave launches stage x through wave-private LDS (1, 2 or 4 chunks per lane): matrices >= 12 MB packed with
    2, 4 and 6 k-blocks per wave (the last only fits this way), strips of 1 and 3 column groups, ragged last k-block
    range (K = 11008: 86 k-blocks over 16 waves), and a batch too large for it (falls back to fragments in registers)."""
    for (K, N, g) in [(4096, 12288, 128), (8192, 4096, 128), (11008, 4096, 128)]:
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M * 17 + K + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + K + 9)
        b = synth.make_bias(N, "f16", 11)
        packed = ops.awq_repack(*_dev(qw, s, qz))
        y = to_np(ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g))
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        assert_gemm_close(y, exact, "f16", what=f"repacked (x via LDS) M={M} K={K} N={N}")
        yb = ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g, to_torch(b, DEV))
        assert torch.equal(yb, to_torch(y, DEV) + to_torch(b, DEV))
        # strided x (row stride > K), as a row-parallel slice of a wider activation
        wide = torch.zeros(M, K + 64, dtype=torch.float16, device=DEV)
        wide[:, :K] = to_torch(x, DEV)
        assert torch.equal(ops.awq_gemm_repacked(wide[:, :K], packed, K, N, g), to_torch(y, DEV))


def _interleave_np(qw, s, qz):
    """numpy restatement of aux_ops.interleave_gate_up: 16-column groups alternate gate / up."""
    N = s.shape[1]
    grp = np.arange(N // 16)
    src = np.where(grp % 2 == 0, grp // 2, N // 32 + grp // 2)
    cols = (src[:, None] * 16 + np.arange(16)).reshape(-1)
    words = (src[:, None] * 2 + np.arange(2)).reshape(-1)
    return qw[:, words], s[:, cols], qz[:, words]


@pytest.mark.parametrize("M", [1, 2, 4])
def test_gemv_repacked_fused_vs_oracle(ops, M):
    """Decode-harness fusions of the repacked GEMV (include/awq_aux.h): RMSNorm(+residual) prologue and SiLU-mul
    epilogue, separately and together, against numpy restatements around the oracle GEMM.  h + delta is bit-exact;
    y is held to the GEMM bound with the prologue's x rebuilt in numpy (rsq / summation-order differences can move
    an x element by one fp16 ulp, far below the 1e-3 allowance)."""
    from sglang_awq_amd import aux_ops

    eps = 1e-5
    for (K, N, g) in [(384, 96, 128), (512, 64, 128), (1024, 1056 * 2, 128), (4096, 12288, 128), (4096, 22016, 128), (2048, 4096, 2048),
                      (8192, 1024, 128), (8192, 10240, 128)]:       # the last: 70B qkv (3-group strips, 4 k-blocks per wave)
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M * 31 + K + N)
        h = synth.make_activations(M, K, "f16", "A", seed=M + K + 3)
        delta = synth.make_activations(M, K, "f16", "A", seed=M + K + 4)
        w = (1.0 + 0.25 * synth.make_activations(1, K, "f16", "A", seed=K + 5)[0].astype(np.float32)).astype(np.float16)
        v = (h + delta).astype(np.float16)                                         # fp16 add
        inv = 1.0 / np.sqrt((v.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)
        xn = (v.astype(np.float32) * inv.astype(np.float32)).astype(np.float16) * w   # fp16(v * inv) * w, fp16 multiply
        dev = [to_torch(t, DEV) for t in (h, delta, w)]
        packed = ops.awq_repack(*_dev(qw, s, qz))
        qwi, si, qzi = _interleave_np(qw, s, qz)
        ti = aux_ops.interleave_gate_up(*_dev(qw, s, qz))
        assert all(torch.equal(a, to_torch(np.ascontiguousarray(b), DEV)) for a, b in zip(ti, (qwi, si, qzi)))
        packed_il = ops.awq_repack(*ti)

        def silu_mul(gu_exact):           # gate_up rounded to fp16, silu in fp32 rounded to fp16, fp16 multiply
            gu = gu_exact.astype(np.float16)
            gate, up = gu[:, :N // 2].astype(np.float32), gu[:, N // 2:]
            return (gate / (1.0 + np.exp(-gate))).astype(np.float16) * up

        # 1. norm prologue only
        r = aux_ops.gemv_repacked_fused(packed, K, N, g, norm=(dev[0], dev[1], dev[2], eps))
        assert r is not None, f"no fused kernel for M={M} K={K} N={N}"
        y, h_out = r
        assert np.array_equal(to_np(h_out), v)
        # Two orders of the same arithmetic exist: the eager one, x = fp16(fp16(v * inv) * w) before the GEMM (what the separate
        # ops and the older prologue kernel compute), and the folded one of gemv_rp2_kernel<NORM>, y = inv * ((v * w) W) with
        # inv applied to the fp32 sums.  Whichever kernel ran must match ITS order to the GEMM bound, and both stay close.
        _, exact = c_oracle.gemm(xn, qw, s, qz, want_exact=True)
        xw = v * w                                                                  # fp16 product
        _, exact_f = c_oracle.gemm(xw, qw, s, qz, want_exact=True)
        exact_f = exact_f * inv
        got = to_np(y).astype(np.float64)
        err_e, err_f = np.abs(got - exact), np.abs(got - exact_f)
        ok_e = np.all(err_e <= 0.5 * ulp(exact, "f16") + 2e-3)
        ok_f = np.all(err_f <= 0.5 * ulp(exact_f, "f16") + 1e-3)
        assert ok_e or ok_f, f"norm-fused M={M} K={K} N={N}: worst vs eager order {err_e.max():.3e}, vs folded order {err_f.max():.3e}"
        ref = exact_f if ok_f else exact
        # v_rsq_f32 vs exact 1 / sqrt moves a few results across a rounding boundary: most, not 98 %, are the correctly rounded sum
        assert float((got != ref.astype(np.float16).astype(np.float64)).mean()) < 0.15
        # the two orders round x at different points: per element <= ~1.5 fp16 ulps of x apart, a random walk over K terms
        row_scale = np.sqrt((exact ** 2).mean(-1, keepdims=True))                   # (an output near 0 is a cancellation of terms this size)
        assert np.all(err_e <= 0.5 * ulp(exact, "f16") + 1e-4 * np.sqrt(K) * (1.0 + row_scale)), "folded and eager orders drifted apart"
        # 2. SiLU-mul epilogue only (x given)
        x = synth.make_activations(M, K, "f16", "A", seed=M + K + 6)
        r = aux_ops.gemv_repacked_fused(packed_il, K, N, g, x=to_torch(x, DEV), silu_mul=True)
        assert r is not None
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        want = silu_mul(exact).astype(np.float64)
        got = to_np(r[0]).astype(np.float64)
        assert got.shape == (M, N // 2)
        tol = 2.0 * ulp(want, "f16") + 2e-3 * (1.0 + np.abs(exact[:, N // 2:]))     # silu' <= 1.1: gate error 1e-3 scales by |up|
        assert np.all(np.abs(got - want) <= tol), f"silu-fused M={M} K={K} N={N}: worst {np.abs(got - want).max():.3e}"
        assert float((got != want).mean()) < 0.05
        # 3. both
        r = aux_ops.gemv_repacked_fused(packed_il, K, N, g, norm=(dev[0], dev[1], dev[2], eps), silu_mul=True)
        assert r is not None
        _, exact = c_oracle.gemm(xn, qw, s, qz, want_exact=True)                     # eager order again (step 2 used a given x)
        want_e = silu_mul(exact).astype(np.float64)
        want_f = silu_mul(exact_f).astype(np.float64)
        got = to_np(r[0]).astype(np.float64)
        tol = 2.0 * ulp(want_e, "f16") + 4e-3 * (1.0 + np.abs(exact[:, N // 2:]))
        d_e, d_f = np.abs(got - want_e), np.abs(got - want_f)
        assert np.all(d_e <= tol) or np.all(d_f <= tol), f"norm+silu M={M} K={K} N={N}: worst {d_e.max():.3e} / {d_f.max():.3e}"
        assert np.all(d_e <= tol + 2e-4 * np.sqrt(K) * (1.0 + row_scale) * (1.0 + np.abs(exact[:, N // 2:]) + np.abs(exact[:, :N // 2])))
        assert np.array_equal(to_np(r[1]), v)


def test_gemm_repacked_tiled_vs_oracle(ops):
    """M > 32 on the repacked copy.  Three routes (awq_capi.hip repacked_dispatch): 32-row GEMV passes (<= 96 rows, or
    <= 160 with many column tiles), 128 x 64 tiles with the K split inside the workgroup (> 96 rows, <= 64 wide tiles),
    hand-pipelined 128 x 256 tiles (the rest); ragged M / N against each tile, odd and even K-block counts."""
    for (M, K, N, g) in [(33, 128, 128, 128), (64, 256, 136, 128), (150, 128, 16648, 128),            # GEMV passes
                         (100, 512, 1056, 128), (256, 1024, 256, 1024), (257, 384, 264, 128), (300, 256, 2304, 128),
                         (1024, 128, 64, 128), (97, 640, 72, 128),                                      # K-split tiles
                         (600, 256, 4360, 128), (161, 384, 16640, 128), (520, 128, 3336, 128),          # pipelined tiles
                         (300, 512, 5640, 256), (200, 384, 16648, 384), (385, 768, 5896, 384),          # ... groups of 2 / 3 k-blocks (counted, not divided)
                         (200, 256, 8456, 128), (255, 128, 9000, 128)]:                                 # ... 128 x 128 tiles (two row tiles, wide matrix)
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M * 3 + K + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + K + 1)
        packed = ops.awq_repack(*_dev(qw, s, qz))
        y = to_np(ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, g))
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        assert_gemm_close(y, exact, "f16", what=f"repacked tiled M={M} K={K} N={N} g={g}")


def test_gemm_repacked_tiled_strided_rows(ops):
    """x as a column slice of a wider tensor (ldx > K) through the three M > 32 routes: the pipelined kernel addresses its x tile
    through a buffer descriptor with 32-bit offsets from the tile's first row."""
    for (M, K, N, g) in [(300, 256, 5640, 128), (257, 384, 264, 128), (150, 128, 2056, 128)]:
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + K + N)
        x = synth.make_activations(M, K + 72, "f16", "A", seed=M + 5)
        packed = ops.awq_repack(*_dev(qw, s, qz))
        xt = to_torch(x, DEV)[:, 40:40 + K]
        assert xt.stride(0) == K + 72 and not xt.is_contiguous()
        y = to_np(ops.awq_gemm_repacked(xt, packed, K, N, g))
        _, exact = c_oracle.gemm(np.ascontiguousarray(x[:, 40:40 + K]), qw, s, qz, want_exact=True)
        assert_gemm_close(y, exact, "f16", what=f"repacked tiled, strided x, M={M} K={K} N={N}")


def test_gemm_repacked_tiled_prefill_shape_one_hot(ops):
    K, N, g, M = 4096, 11008, 128, 2048
    dq, ds, dz = _dev(*synth.make_awq_weights(K, N, g, "f16", "A", 779))
    W = ops.awq_dequantize(dq, ds, dz)
    packed = ops.awq_repack(dq, ds, dz)
    x = torch.zeros(M, K, dtype=torch.float16, device=DEV)
    ks = (torch.arange(M, device=DEV) * 7 + 3) % K
    x[torch.arange(M, device=DEV), ks] = 1.0
    assert torch.equal(ops.awq_gemm_repacked(x, packed, K, N, g), W[ks])


def test_gemm_repacked_one_hot_rows_bit_exact_full_shape(ops):
    K, N, g = 4096, 11008, 128
    qw, s, qz = _dev(*synth.make_awq_weights(K, N, g, "f16", "F", 4243))
    W = ops.awq_dequantize(qw, s, qz)
    packed = ops.awq_repack(qw, s, qz)
    rows = [0, 1, 127, 128, 2047, 2048, 4095, 31, 32, 33, 1000, 3000, 4064, 555, 77, 4094]
    x = torch.zeros(len(rows), K, dtype=torch.float16, device=DEV)
    for m, k in enumerate(rows):
        x[m, k] = 1.0
    for M in (1, 7, 16):
        assert torch.equal(ops.awq_gemm_repacked(x[:M], packed, K, N, g), W[rows[:M]]), f"M={M}"


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_gemm_tiled_vs_oracle(ops, dt):
    """LDS-tiled MFMA kernel (prefill shapes): ragged M and N against the 128 x 128 tile, several K steps,
    g = 32 / 64 / 128 / K, more tiles than one XCD round."""
    for (M, K, N, g) in [(1, 128, 128, 128), (17, 256, 136, 64), (64, 1024, 1056, 128), (100, 512, 128, 32),
                         (128, 128, 256, 128), (129, 384, 264, 384), (300, 256, 2304, 128), (2048, 128, 128, 128)]:
        y, exact, _ = _gemm_case(ops, M, K, N, g, dt, "A", seed=M * 31 + K + N, variant=_lib.GEMM_TILED)
        assert_gemm_close(y, exact, dt, what=f"tiled M={M} K={K} N={N} g={g} {dt}")


def test_gemm_tiled_prefill_shape_properties(ops):
    """BASELINE configs[2] (M = 2048, K = 4096, N = 11008): one-hot rows must reproduce awq_dequantize rows
    bit for bit, and a random sample of outputs is checked against the exact-sum oracle."""
    K, N, g, M = 4096, 11008, 128, 2048
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", 777)
    dq, ds, dz = _dev(qw, s, qz)
    W = ops.awq_dequantize(dq, ds, dz)
    x = torch.zeros(M, K, dtype=torch.float16, device=DEV)
    ks = torch.arange(M, device=DEV) * 2 % K
    x[torch.arange(M, device=DEV), ks] = 1.0
    y = ops.awq_gemm(x, dq, ds, dz, 1)
    assert torch.equal(y, W[ks])
    xr = synth.make_activations(M, K, "f16", "A", 778)
    y = to_np(ops.awq_gemm(to_torch(xr, DEV), dq, ds, dz, 1))
    rows = [0, 1, 127, 128, 1000, 2047]
    _, exact = c_oracle.gemm(xr[rows], qw, s, qz, want_exact=True)
    assert_gemm_close(y[rows], exact, "f16", what="prefill sample rows")


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_gemm_generic_vs_oracle(ops, dt):
    for (M, K, N, g) in [(1, 128, 8, 128), (5, 96, 40, 32), (9, 200, 520, 100), (33, 256, 72, 64), (4, 64, 1024, 64)]:
        y, exact, _ = _gemm_case(ops, M, K, N, g, dt, "A", seed=M + K, variant=_lib.GEMM_GENERIC)
        assert_gemm_close(y, exact, dt, what=f"generic {M} {K} {N} {g} {dt}")


def test_gemm_auto_dispatch_all_m(ops):
    K, N, g = 512, 1024, 128
    for M in [1, 4, 16, 17, 32, 33, 64, 100, 256]:
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", M)
        x = synth.make_activations(M, K, "f16", "A", M)
        y = to_np(ops.awq_gemm(*_dev(x, qw, s, qz), 1))
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        assert_gemm_close(y, exact, "f16", what=f"auto M={M}")


def test_gemm_atol_1e3_on_unit_scale_outputs(ops):
    """With |y| < 1 the fp16 output ulp is < 1e-3, so the north star's bar applies to the rounded
    outputs directly: max |y_gpu - y_ref| <= 1e-3 against the correctly rounded oracle."""
    for M in (1, 8, 16):
        y, exact, (qw, s, qz, x, _) = _gemm_case(ops, M, 4096, 1024, 128, "f16", "A", seed=99 + M, variant=_lib.GEMM_AUTO, x_std=0.03)
        ref = awq_ref.to_f64(awq_ref.from_f64(exact, "f16"), "f16")
        assert np.abs(exact).max() < 1.0
        assert np.abs(awq_ref.to_f64(y, "f16") - ref).max() <= 1e-3


def test_gemm_one_hot_rows_reproduce_dequantize_bit_exact(ops):
    """Size-independent property at the full BASELINE shape: x = e_k selects row k of W, every other
    product is an exact zero, so awq_gemm must return awq_dequantize's row bit for bit."""
    K, N, g = 4096, 11008, 128
    qw, s, qz = _dev(*synth.make_awq_weights(K, N, g, "f16", "F", 4242))
    W = ops.awq_dequantize(qw, s, qz)
    rows = [0, 1, 127, 128, 2047, 2048, 4095, 31, 32, 33, 1000, 3000, 4064, 555, 77, 4094]
    x = torch.zeros(len(rows), K, dtype=torch.float16, device=DEV)
    for m, k in enumerate(rows):
        x[m, k] = 1.0
    for M in (1, 5, 16):
        y = ops.awq_gemm(x[:M], qw, s, qz, 1)
        assert torch.equal(y, W[rows[:M]]), f"M={M}"


def test_gemm_full_size_column_sums(ops):
    """x = ones at 4096 x 11008: y = column sums of W; checked against an fp64 sum of the GPU's own
    (bit-exact, see above) dequantised weight."""
    K, N, g = 4096, 11008, 128
    qw, s, qz = _dev(*synth.make_awq_weights(K, N, g, "f16", "A", 1234))
    W = ops.awq_dequantize(qw, s, qz)
    exact = W.double().sum(0, keepdim=True).cpu().numpy()
    x = torch.ones(1, K, dtype=torch.float16, device=DEV)
    assert_gemm_close(to_np(ops.awq_gemm(x, qw, s, qz, 1)), exact, "f16", what="column sums")


def test_gemm_deterministic_and_split_k_independent(ops):
    qw, s, qz = _dev(*synth.make_awq_weights(4096, 2048, 128, "f16", "A", 7))
    x = to_torch(synth.make_activations(3, 4096, "f16", "A", 7), DEV)
    y0 = ops.awq_gemm(x, qw, s, qz, 1)
    for sk in (1, 2, 8, 32):
        for _ in range(3):
            assert torch.equal(ops.awq_gemm(x, qw, s, qz, sk), y0)


def test_gemm_bias_epilogue_two_roundings(ops):
    for dt in ("f16", "bf16"):
        qw, s, qz = synth.make_awq_weights(512, 1024, 128, dt, "A", 3)
        x = synth.make_activations(4, 512, dt, "A", 3)
        b = synth.make_bias(1024, dt, 3)
        dq, ds, dz, dx, db = _dev(qw, s, qz, x, b)
        # awq_linear (this package's own op) never consults the drop-in op's repacked-copy cache: compare on the same kernels
        ops.awq_gemm_cache_enable(False)
        try:
            y = ops.awq_gemm(dx, dq, ds, dz, 1)
        finally:
            ops.awq_gemm_cache_enable(True)
        yb = ops.awq_linear(dx, dq, ds, dz, db)
        assert torch.equal(yb, y + db)          # torch's same-dtype add rounds once more, as add_ does
        want = awq_ref.awq_linear_apply(x, qw, s, qz, b)
        d = np.abs(awq_ref.to_f64(to_np(yb), dt) - awq_ref.to_f64(want, dt))
        assert (d > 0).mean() < 0.02


def test_gemm_strided_activation_slice(ops):
    """Row-parallel ranks pass a K-slice of a wider activation (linear.py:1395-1399): ldx > K."""
    qw, s, qz = _dev(*synth.make_awq_weights(512, 1024, 128, "f16", "A", 11))
    xfull = to_torch(synth.make_activations(6, 2048, "f16", "A", 11), DEV)
    xs = xfull[:, 512:1024]
    assert not xs.is_contiguous()
    assert torch.equal(ops.awq_gemm(xs, qw, s, qz, 1), ops.awq_gemm(xs.contiguous(), qw, s, qz, 1))


def test_gemm_errors(ops):
    qw, s, qz = _dev(*synth.make_awq_weights(128, 64, 32, "f16", "A", 1))
    x = torch.ones(2, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.awq_gemm(x, qw, s, qz, 3)
    with pytest.raises(RuntimeError):
        ops.awq_gemm(x, qw, s, qz, 64)
    with pytest.raises(RuntimeError):
        ops.awq_gemm(x[:, :64], qw, s, qz, 1)
    with pytest.raises(RuntimeError):
        ops.awq_gemm(x.float(), qw, s, qz, 1)
    assert ops.awq_gemm(x[:0], qw, s, qz, 1).shape == (0, 64)


def test_ops_are_graph_capturable(ops):
    """The decode path replays captured graphs (model_runner.py:2765-2771): launch-only, no sync."""
    qw, s, qz = _dev(*synth.make_awq_weights(1024, 1024, 128, "f16", "A", 21))
    x = to_torch(synth.make_activations(1, 1024, "f16", "A", 21), DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = ops.awq_gemm(x, qw, s, qz, 1)          # warm-up on the capture stream (workspace alloc)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y = ops.awq_gemm(x, qw, s, qz, 1)
        w = ops.awq_dequantize(qw, s, qz)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, eager)
    assert torch.equal(w, ops.awq_dequantize(qw, s, qz))


# ------------------------------------------------------------------------------ AWQLinearMethod
def test_awq_linear_method_apply_matches_oracle(ops):
    from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod

    cfg = AWQConfig(weight_bits=4, group_size=128, zero_point=True)
    for mode in ("fused", "dequant_matmul"):
        method = AWQLinearMethod(cfg, apply_mode=mode)
        layer = torch.nn.Module()
        method.create_weights(layer, 512, [768, 256], 512, 1024, torch.float16, weight_loader=None)
        assert layer.qweight.shape == (512, 128) and layer.qzeros.shape == (4, 128) and layer.scales.shape == (4, 1024)
        qw, s, qz = synth.make_awq_weights(512, 1024, 128, "f16", "A", 31)
        layer.qweight.data.copy_(to_torch(qw)); layer.qzeros.data.copy_(to_torch(qz)); layer.scales.data.copy_(to_torch(s))
        layer.to(DEV)
        method.process_weights_after_loading(layer)
        x = synth.make_activations(6, 512, "f16", "A", 31).reshape(2, 3, 512)
        b = synth.make_bias(1024, "f16", 31)
        y = method.apply(layer, to_torch(x, DEV), to_torch(b, DEV))
        assert y.shape == (2, 3, 1024)
        want = awq_ref.to_f64(awq_ref.awq_linear_apply(x, qw, s, qz, b), "f16")
        got = awq_ref.to_f64(to_np(y), "f16")
        from tests.util import ulp
        # One ulp of the PRE-bias sum (a rounding-boundary flip of the fp32-accumulated sum against the exact
        # one) — which can be several ulps of a smaller post-bias result when the bias cancels.  The vendor
        # BLAS of the dequant_matmul mode is third-party arithmetic: allow it a second flip.
        pre = awq_ref.to_f64(awq_ref.awq_linear_apply(x, qw, s, qz, None), "f16")
        tol = 1.01 if mode == "fused" else 2.02
        # ... and the bias add rounds once more, which can move the result by one further ulp of its own
        assert np.all(np.abs(got - want) <= tol * (ulp(want, "f16") + ulp(pre, "f16"))), mode


def test_gemv_repacked_fused_refuses_what_it_cannot_run(ops):
    """Shapes without a fused instantiation return None (AWQ_ERR_BAD_VARIANT at the C boundary) so callers run the
    separate ops: 16 rows of K = 4096 (prologue chunks per lane), K = 32768 (more than 8 k-blocks per wave)."""
    from sglang_awq_amd import aux_ops

    for (M, K, N) in [(16, 4096, 512), (1, 32768, 256)]:
        qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", seed=5)
        packed = ops.awq_repack(*_dev(qw, s, qz))
        h = torch.zeros(M, K, dtype=torch.float16, device=DEV)
        w = torch.ones(K, dtype=torch.float16, device=DEV)
        assert aux_ops.gemv_repacked_fused(packed, K, N, 128, norm=(h, h.clone(), w, 1e-5)) is None


@pytest.mark.parametrize("M", [17, 32])
def test_gemv_repacked_silu_epilogue_two_row_tiles(ops, M):
    """SiLU-mul epilogue on the two-row-tile variant (17..32 rows): straight-line (4 k-blocks per wave) and loop depth,
    strips of 2 and 6 groups (the latter with more than 64 KiB of reduction scratch)."""
    from sglang_awq_amd import aux_ops

    for (K, N, g) in [(4096, 2112, 128), (1024, 22016, 128), (4096, 22016, 128)]:
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M * 5 + K + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + K + 2)
        packed_il = ops.awq_repack(*aux_ops.interleave_gate_up(*_dev(qw, s, qz)))
        r = aux_ops.gemv_repacked_fused(packed_il, K, N, g, x=to_torch(x, DEV), silu_mul=True)
        assert r is not None, f"M={M} K={K} N={N}"
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        gu = exact.astype(np.float16)
        gate, up = gu[:, :N // 2].astype(np.float32), gu[:, N // 2:]
        want = ((gate / (1.0 + np.exp(-gate))).astype(np.float16) * up).astype(np.float64)
        got = to_np(r[0]).astype(np.float64)
        tol = 2.0 * ulp(want, "f16") + 2e-3 * (1.0 + np.abs(exact[:, N // 2:]))
        assert got.shape == (M, N // 2) and np.all(np.abs(got - want) <= tol), f"M={M} K={K} N={N}: worst {np.abs(got - want).max():.3e}"
        assert float((got != want).mean()) < 0.05


def test_awq_gemm_op_prefill_repacks_on_the_fly(ops):
    """The drop-in op at prefill-sized M (>= 1024) re-lays the weight out into workspace and runs the kernel of the
    repacked path: bit-identical to awq_gemm_repacked on a persistent copy, within the GEMM bound of the oracle (first
    rows), also on a matrix whose repacked copy exceeds the shim's persistent 32 MiB scratch and with a bias."""
    for (M, K, N) in [(1024, 512, 1056), (1100, 4096, 22016)]:
        qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", seed=M + N)
        x = synth.make_activations(M, K, "f16", "A", seed=M + 1)
        b = synth.make_bias(N, "f16", 3)
        dq, ds, dz = _dev(qw, s, qz)
        y = torch.ops.sgl_kernel.awq_gemm(to_torch(x, DEV), dq, ds, dz, 8)
        packed = ops.awq_repack(dq, ds, dz)
        assert torch.equal(y, ops.awq_gemm_repacked(to_torch(x, DEV), packed, K, N, 128))
        _, exact = c_oracle.gemm(x[:4], qw, s, qz, want_exact=True)
        assert_gemm_close(to_np(y[:4]), exact, "f16", what=f"awq_gemm op, prefill M={M} K={K} N={N}")
        yb = ops.awq_linear(to_torch(x, DEV), dq, ds, dz, to_torch(b, DEV))
        assert torch.equal(yb, y + to_torch(b, DEV))


def test_gemv_repacked_silu_epilogue_rounds_mode(ops):
    """SiLU-mul epilogue on a matrix too wide for one strip per CU (N > 32768: rounds of 4-group strips), as the 57344-wide
    gate_up of a 70B model; small K keeps the oracle cheap."""
    from sglang_awq_amd import aux_ops

    K, N, g = 256, 33280, 128
    for M in (1, 5):
        qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + 90)
        x = synth.make_activations(M, K, "f16", "A", seed=M + 91)
        packed_il = ops.awq_repack(*aux_ops.interleave_gate_up(*_dev(qw, s, qz)))
        r = aux_ops.gemv_repacked_fused(packed_il, K, N, g, x=to_torch(x, DEV), silu_mul=True)
        assert r is not None
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        gu = exact.astype(np.float16)
        gate, up = gu[:, :N // 2].astype(np.float32), gu[:, N // 2:]
        want = ((gate / (1.0 + np.exp(-gate))).astype(np.float16) * up).astype(np.float64)
        got = to_np(r[0]).astype(np.float64)
        tol = 2.0 * ulp(want, "f16") + 2e-3 * (1.0 + np.abs(exact[:, N // 2:]))
        assert np.all(np.abs(got - want) <= tol), f"M={M}: worst {np.abs(got - want).max():.3e}"
