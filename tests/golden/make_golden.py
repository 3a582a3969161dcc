#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own functions.

Runs only in the build container (needs /root/reference; that tree does not exist on the GPU
box and nothing under tests/ reads it at test time).  What is executed from the reference:

  * python/sglang/srt/layers/quantization/awq_triton.py, loaded standalone with importlib (its
    only imports are torch and triton): `awq_dequantize_decomposition` on CPU, and
    `awq_dequantize_triton` / `awq_gemm_triton` under TRITON_INTERPRET=1;
  * the function definitions `reverse_awq_order` and `awq_dequantize_torch` of
    sgl-kernel/tests/test_awq_dequant.py and test/srt/quant/test_awq_dequant.py, extracted by AST
    (the modules themselves import sgl_kernel / sglang, which are not installable here).

No reference source text is written anywhere: the outputs are arrays (npz) and sha256 digests
(json).  Inputs come from sglang_awq_amd.synth (deterministic) or, for the cases that mirror
the reference tests verbatim, from torch.manual_seed(0) exactly as those tests draw them; the
latter inputs are stored in the npz next to the outputs.

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz, digests.json)
"""
import ast
import hashlib
import importlib.util
import json
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sglang_awq_amd import synth  # noqa: E402

REF = "/root/reference"
REF_TRITON = f"{REF}/python/sglang/srt/layers/quantization/awq_triton.py"
REF_TESTS = [f"{REF}/sgl-kernel/tests/test_awq_dequant.py", f"{REF}/test/srt/quant/test_awq_dequant.py"]


def load_ref_triton():
    spec = importlib.util.spec_from_file_location("ref_awq_triton", REF_TRITON)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_test_oracle(path):
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("reverse_awq_order", "awq_dequantize_torch")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["awq_dequantize_torch"]


def t(a, dtype=None):
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def bits(x: torch.Tensor) -> np.ndarray:
    x = x.contiguous()
    if x.dtype == torch.bfloat16:
        return x.view(torch.int16).numpy().view(np.uint16)
    return x.numpy()


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = load_ref_triton()
    test_oracles = [load_test_oracle(p) for p in REF_TESTS]

    def ref_dequant_all(qw, s, qz, g, use_triton):
        """All importable reference dequantisers; asserts they agree bit for bit."""
        tq, ts, tz = t(qw), t(s), t(qz)
        out = bits(ref.awq_dequantize_decomposition(tq, ts, tz))
        for fn in test_oracles:
            o2 = fn(tq, ts, tz, g).to(ts.dtype)
            assert np.array_equal(bits(o2), out), "reference oracles disagree"
        if use_triton:
            o3 = ref.awq_dequantize_triton(tq, ts, tz)
            assert np.array_equal(bits(o3), out), "triton(interpret) disagrees with decomposition"
        return out

    # ------------------------------------------------------------------ small, stored in full
    small = {}
    small_cases = [
        # K, N, g, dtype, family, seed, triton?
        (128, 128, 128, "f16", "R", 1, True),
        (256, 64, 32, "f16", "R", 2, True),
        (256, 64, 64, "bf16", "R", 3, False),
        (512, 72, 512, "bf16", "R", 4, False),
        (128, 32, 128, "f16", "F", 5, True),
        (256, 128, 128, "f16", "A", 6, True),
        (64, 16, 32, "f32", "R", 7, True),
        (384, 8, 128, "f16", "F", 8, False),
        (128, 264, 64, "bf16", "F", 9, False),
        (96, 40, 32, "f32", "F", 10, False),
    ]
    for i, (K, N, g, dt, fam, seed, tri) in enumerate(small_cases):
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, fam, seed)
        out = ref_dequant_all(qw, s, qz, g, tri)
        small[f"dq{i}_meta"] = np.array([K, N, g, seed], dtype=np.int64)
        small[f"dq{i}_dtype"] = np.array(dt)
        small[f"dq{i}_family"] = np.array(fam)
        small[f"dq{i}_qweight"], small[f"dq{i}_scales"], small[f"dq{i}_qzeros"] = qw, s, qz
        small[f"dq{i}_out"] = out
    small["n_dequant"] = np.array(len(small_cases))

    # the reference Triton test draws its inputs with torch.manual_seed(0); mirror two of its
    # cases verbatim (test/srt/quant/test_awq_dequant.py:79-111) and store inputs + outputs.
    verb = []
    for rows, cols, g in [(128, 16, 32), (256, 32, -1)]:
        gg = rows if g == -1 else g
        torch.manual_seed(0)
        qweight = torch.randint(0, torch.iinfo(torch.int32).max, (rows, cols), dtype=torch.int32)
        scales = torch.rand(rows // gg, cols * 8, dtype=torch.float16)
        zeros = torch.randint(0, torch.iinfo(torch.int32).max, (rows // gg, cols), dtype=torch.int32)
        out = ref_dequant_all(qweight.numpy(), scales.numpy(), zeros.numpy(), gg, True)
        verb.append((qweight.numpy(), scales.numpy(), zeros.numpy(), out))
    for i, (a, b, c, d) in enumerate(verb):
        small[f"verb{i}_qweight"], small[f"verb{i}_scales"], small[f"verb{i}_qzeros"], small[f"verb{i}_out"] = a, b, c, d
    small["n_verbatim"] = np.array(len(verb))
    np.savez_compressed(os.path.join(HERE, "awq_dequant_small.npz"), **small)

    # ------------------------------------------------------------------ fused GEMM (Triton, interpret)
    # grid of test/srt/quant/test_awq_dequant.py:114-171: fp32 x / scales, K = 128, seed 0 per case
    gem = {}
    idx = 0
    for Mrows in [1, 2, 4, 8, 14, 17, 23, 32]:
        for Ncols in [16, 24, 32]:
            for g in [-1, 32, 64, 128]:
                for sk in [1, 8]:
                    K = 128
                    gg = K if g == -1 else g
                    torch.manual_seed(0)
                    x = torch.rand((Mrows, K), dtype=torch.float32)
                    qweight = torch.randint(0, torch.iinfo(torch.int32).max, (K, Ncols // 8), dtype=torch.int32)
                    qzeros = torch.randint(0, torch.iinfo(torch.int32).max, (K // gg, Ncols // 8), dtype=torch.int32)
                    scales = torch.rand((K // gg, Ncols), dtype=torch.float32)
                    tri = ref.awq_gemm_triton(x, qweight, scales, qzeros, sk)
                    w = ref.awq_dequantize_decomposition(qweight, scales, qzeros)
                    mm = torch.matmul(x, w)
                    assert torch.allclose(tri, mm, atol=1e-1, rtol=1e-1)
                    gem[f"g{idx}_meta"] = np.array([Mrows, K, Ncols, gg, sk], dtype=np.int64)
                    gem[f"g{idx}_x"], gem[f"g{idx}_qweight"] = x.numpy(), qweight.numpy()
                    gem[f"g{idx}_scales"], gem[f"g{idx}_qzeros"] = scales.numpy(), qzeros.numpy()
                    gem[f"g{idx}_triton"], gem[f"g{idx}_matmul"] = tri.numpy(), mm.numpy()
                    idx += 1
    gem["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "awq_gemm_triton_f32.npz"), **gem)

    # ------------------------------------------------------------------ apply() on CPU (fp16 / bf16)
    # the two steps of AWQLinearMethod.apply (awq.py:446-450) executed with the reference's own
    # dequantiser and torch CPU matmul; inputs from synth, outputs stored.
    app = {}
    apply_cases = [
        # M, K, N, g, dtype, family, seed, bias
        (1, 256, 64, 128, "f16", "A", 21, False),
        (5, 512, 128, 128, "f16", "A", 22, True),
        (16, 256, 72, 64, "f16", "A", 23, True),
        (3, 256, 64, 128, "bf16", "A", 24, True),
        (32, 1024, 256, 128, "f16", "A", 25, False),
    ]
    for i, (M, K, N, g, dt, fam, seed, has_bias) in enumerate(apply_cases):
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, fam, seed)
        x = synth.make_activations(M, K, dt, fam, seed)
        b = synth.make_bias(N, dt, seed) if has_bias else None
        w = ref.awq_dequantize_decomposition(t(qw), t(s), t(qz))
        out = torch.matmul(t(x), w)
        if b is not None:
            out.add_(t(b))
        app[f"a{i}_meta"] = np.array([M, K, N, g, seed, int(has_bias)], dtype=np.int64)
        app[f"a{i}_dtype"] = np.array(dt)
        app[f"a{i}_out"] = bits(out)
    app["n_cases"] = np.array(len(apply_cases))
    np.savez_compressed(os.path.join(HERE, "awq_apply_cpu.npz"), **app)

    # ------------------------------------------------------------------ large shapes, digests only
    digests = []
    big = [(4096, 11008, 128, "f16", "R", 100), (4096, 11008, 128, "f16", "F", 101),
           (4096, 11008, 128, "f16", "A", 1234), (4096, 11008, 128, "bf16", "F", 102),
           (11008, 4096, 128, "f16", "A", 103), (4096, 12288, 128, "f16", "A", 104),
           (4096, 4096, 128, "f16", "A", 105), (4096, 22016, 128, "f16", "A", 106)]
    # reference CUDA-op test grid (sgl-kernel/tests/test_awq_dequant.py:70-71), g = rows, both dtypes;
    # the 18944-row and 4736-column members are kept only where the product stays < 64M outputs.
    for rows in [3584, 18944, 128, 256, 512, 1024, 1536]:
        for cols in [448, 576, 4736, 16, 32, 64, 128, 72]:
            if rows * cols * 8 > 64 * 1024 * 1024:
                continue
            for dt in ("bf16", "f16"):
                big.append((rows, cols * 8, rows, dt, "R", 200 + len(big)))
    # Triton test grid adds g in {32, 64, 128} (test/srt/quant/test_awq_dequant.py:63-77)
    for rows, cols in [(3584, 448), (1024, 576), (512, 128), (256, 16)]:
        for g in (32, 64, 128):
            big.append((rows, cols * 8, g, "f16", "R", 400 + len(big)))
    for (K, N, g, dt, fam, seed) in big:
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, fam, seed)
        out = bits(ref.awq_dequantize_decomposition(t(qw), t(s), t(qz)))
        digests.append({"K": K, "N": N, "g": g, "dtype": dt, "family": fam, "seed": seed,
                        "inputs_sha256": sha(qw) + sha(s)[:16] + sha(qz)[:16], "out_sha256": sha(out)})
        print("digest", K, N, g, dt, fam, flush=True)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "awq_dequantize_decomposition "
                   "(awq_triton.py:342-368), cross-checked against both test oracles on the small cases",
                   "cases": digests}, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
