"""Randomised parity sweep on the GPU box: awq_gemm_repacked / the awq_gemm op / awq_dequantize against the C oracle over random
(M, K, N, group_size) — every dispatch route (GEMV, rp2, split-K, passes, K-split tiles, pipelined tiles of each width, generic
bf16 / small-group kernels).  Not a pytest (minutes of oracle time); run by hand after kernel changes:  python tools/fuzz_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from sglang_awq_amd import ops, synth  # noqa: E402
from tests.util import assert_gemm_close, to_np, to_torch  # noqa: E402

DEV = "cuda:0"


def run(cases: int, seed: int, verbose: bool = True) -> int:
    ops.awq_gemm_cache_enable(True)                        # the op's repacked-copy cache is opt-in; this sweep owns its weights
    rng = np.random.default_rng(seed)
    Ms = [1, 2, 3, 5, 8, 9, 13, 16, 17, 24, 31, 32, 33, 48, 64, 97, 130, 160, 161, 200, 256, 257, 300, 385, 520]
    done = 0
    for it in range(cases):
        M = int(rng.choice(Ms))
        kb = int(rng.integers(1, 13))                      # K = 128 .. 1536
        deep = rng.random() < 0.15
        if deep:
            kb = int(rng.choice([64, 72, 86]))             # deep matrices: the split-K GEMV's domain
        K = 128 * kb
        N = 8 * int(rng.integers(2, 40 if deep else 700)) # N % 8 == 0, ragged against 16 / 64 / 128 / 192 / 256
        long_ = (not deep) and rng.random() < 0.12         # round 3: the long straight-line GEMV (17 .. 32 units per wave) and the loop form
        if long_:
            M = int(rng.choice([1, 1, 13, 16]))
            if rng.random() < 0.5:
                K, N = 128 * int(rng.choice([144, 160, 224, 256])), 8 * int(rng.integers(2, 64))           # 9 .. 16 k-blocks per wave
            else:
                K, N = 128 * int(rng.choice([48, 64])), 8 * int(rng.integers(2560, 3584))                    # strips of 5 .. 7 column groups
        divs = [d for d in (32, 64, 128, 256, 384, 512, K) if K % d == 0 and d <= K]
        g = int(rng.choice(divs))
        dt = "bf16" if rng.random() < 0.2 else "f16"
        if M > 64 and K * N > 1536 * 4096:
            N = 8 * int(rng.integers(2, 300))
        qw, s, qz = synth.make_awq_weights(K, N, g, dt, "A", seed=it * 7 + 1)
        x = synth.make_activations(M, K, dt, "A", seed=it * 7 + 2)
        dq, ds, dz = (to_torch(t, DEV) for t in (qw, s, qz))
        xt = to_torch(x, DEV)
        _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
        what = f"case {it}: M={M} K={K} N={N} g={g} {dt}"
        # the op on checkpoint tensors (cache on: repacked route where the layout exists)
        y = torch.ops.sgl_kernel.awq_gemm(xt, dq, ds, dz, 1)
        assert_gemm_close(to_np(y), exact, dt, what=what + " (op)")
        packed = ops.awq_repack(dq, ds, dz)
        if packed is not None:
            y2 = ops.awq_gemm_repacked(xt, packed, K, N, g)
            assert_gemm_close(to_np(y2), exact, dt, what=what + " (repacked)")
            if it % 3 == 0:                                   # bias epilogue: the rounded sum plus bias, rounded again (awq.py:449-450)
                b = (torch.randn(N, device=DEV) * 0.5).to(xt.dtype)
                yb = ops.awq_gemm_repacked(xt, packed, K, N, g, b)
                assert torch.equal(yb, y2 + b), what + " (bias epilogue)"
            # strided rows
            xw = torch.zeros((M, K + 24), dtype=xt.dtype, device=DEV)
            xw[:, 8:8 + K] = xt
            y3 = ops.awq_gemm_repacked(xw[:, 8:8 + K], packed, K, N, g)
            assert torch.equal(y3, y2), what + " (strided x differs)"
        done += 1
        if it % 16 == 15:
            ops.awq_gemm_cache_clear()
        if verbose and it % 10 == 0:
            print(f"{it:4d} ok  {what}", flush=True)
    return done


def main():
    done = run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"FUZZ_OK {done} cases")


if __name__ == "__main__":
    main()
