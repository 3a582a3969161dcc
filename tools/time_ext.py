"""Quick timing of the bf16 / small-group repacked kernels against the fp16 g128 ones (graph replay, 16 rotating sets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sglang_awq_amd import ops

dev = torch.device("cuda:0")
K, N = 4096, 11008


def bench(dt, g, M, sets=16, reps=40):
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    packs = []
    for _ in range(sets):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // g, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        sc = (0.005 + 0.015 * torch.rand((K // g, N), device=dev, generator=gen)).to(dt)
        packs.append(ops.awq_repack(qw, sc, qz))
    x = torch.randn(M, K, device=dev, generator=gen).to(dt)
    f = lambda: [ops.awq_gemm_repacked(x, p, K, N, g) for p in packs]
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * sets)


for M in (1, 16, 2048):
    for dt, g in ((torch.float16, 128), (torch.bfloat16, 128), (torch.float16, 64), (torch.float16, 32), (torch.bfloat16, 32)):
        us = bench(dt, g, M, reps=40 if M < 100 else 4)
        print(f"M={M} {str(dt).split('.')[-1]} g={g}: {us:8.2f} us  {2 * M * K * N / us / 1e6:8.1f} TFLOP/s", flush=True)
