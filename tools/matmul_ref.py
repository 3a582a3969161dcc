"""Context measurement: vendor fp16 GEMM (hipBLASLt via torch.matmul) on prefill shapes, next to the fused AWQ path and
the reference's two-step structure (awq_dequantize + matmul).  Prints one line per (M, K, N)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sglang_awq_amd import ops, synth

dev = "cuda:0"
shapes = [(2048, 4096, 11008), (2048, 11008, 4096), (8192, 4096, 11008), (512, 4096, 11008)]
for (M, K, N) in shapes:
    qw, s, qz = (torch.from_numpy(t).to(dev) for t in synth.make_awq_weights(K, N, 128, "f16", "A", 7))
    x = torch.from_numpy(synth.make_activations(M, K, "f16", "A", 3)).to(dev)
    packed = ops.awq_repack(qw, s, qz)
    W = ops.awq_dequantize(qw, s, qz)

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    t_mm = timeit(lambda: torch.matmul(x, W))
    t_dq = timeit(lambda: ops.awq_dequantize(qw, s, qz))
    t_fused = timeit(lambda: ops.awq_gemm_repacked(x, packed, K, N, 128))
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: matmul {t_mm:8.1f} us ({fl / t_mm / 1e6:7.1f} TFLOP/s)  dequantize {t_dq:6.1f} us  "
          f"dequantize+matmul {t_mm + t_dq:8.1f} us  fused repacked {t_fused:8.1f} us ({fl / t_fused / 1e6:7.1f} TFLOP/s)", flush=True)
