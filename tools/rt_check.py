import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import c_oracle
from sglang_awq_amd import ops, synth
from tests.util import assert_gemm_close, to_np, to_torch
for (M, K, N, g) in [(17, 4096, 22016, 128), (32, 4096, 12288, 128), (24, 4096, 11008, 128), (32, 1152, 8200, 128), (19, 8192, 10240, 256), (32, 1280, 8192, 1280), (32, 4096, 28672, 128)]:
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", seed=M + K + N)
    x = synth.make_activations(M, K, "f16", "A", seed=M + 3)
    packed = ops.awq_repack(*[to_torch(t, "cuda:0") for t in (qw, s, qz)])
    for rep in range(2):
        y = to_np(ops.awq_gemm_repacked(to_torch(x, "cuda:0"), packed, K, N, g))
    _, exact = c_oracle.gemm(x, qw, s, qz, want_exact=True)
    assert_gemm_close(y, exact, "f16", what=f"rows M={M} K={K} N={N}")
    print("ok", M, K, N, flush=True)
