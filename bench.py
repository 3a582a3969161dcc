#!/usr/bin/env python3
"""Benchmark of the AWQ int4 quantized-linear hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the decode shape): awq_gemm at M = 1, K = 4096, N = 11008,
g = 128, fp16.  One STEP = one column-parallel AWQ linear followed by one row-parallel AWQ linear,
each rank holding a full 4096 x 11008 shard (weak scaling: the TP=N layer is N times wider), i.e.
two fused int4 GEMV launches per rank and, for N > 1, one RCCL all-reduce of the [M, 11008] fp16
partial sums.  The linears are this package's AWQLinearMethod as deployed: at load time
(process_weights_after_loading) the weights get a one-time MFMA-fragment-major copy and decode batches
run the kernel on that copy; the drop-in op on the checkpoint layout (`sgl_kernel.awq_gemm`) is timed
beside it and reported as `config.awq_gemm_op_checkpoint_layout`.  Weights rotate through `--sets` distinct copies (> 2x the 256 MiB Infinity Cache) so
the stream comes from HBM, not from cache; the cache-hot number is reported beside it.  Steps are
replayed from a captured HIP graph (the decode path of the reference replays graphs too), so the
timed region contains exactly K steps of device work and no Python.

Prints ONE JSON line (rank 0).  `value` is whole-job algorithmic GB/s: bytes every rank must move
(packed weight + activations + outputs = 23,455,232 B per GEMV) divided by the slowest rank's time.
`roofline` prices the dominant kernel (the fused GEMV) against 8 TB/s; `cpu_baseline` times the
reference's CPU form of the same linear (eager PyTorch dequantise + matmul) on this host.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this host driver
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_DIM, N_DIM, GROUP = 4096, 11008, 128
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# HBM bytes per M=1 launch from the PMC counters (separate rocprofv3 --pmc passes of tools/kbench, files
# profiles/r01_pmc_{fetch,write}_size_gemv_m1.csv): FETCH_SIZE 11,870.5 KiB x 2 (gfx950 reports half of a
# wide coalesced read stream; calibrated on a 22.5 MB linear read) + WRITE_SIZE 513.75 KiB
PMC_TRAFFIC_BYTES_M1_CHECKPOINT_LAYOUT = int((2 * 11870.5 + 513.75) * 1024)   # gemm_skinny_kernel (awq_gemm op)
PMC_TRAFFIC_BYTES_M1 = int((2 * 11812.0 + 21.5625) * 1024)                     # gemv_repacked_kernel (profiles/r01_pmc_*_repacked_m1.csv)
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16


def algorithmic_bytes(M: int) -> int:
    w = K_DIM * N_DIM // 2 + (K_DIM // GROUP) * (N_DIM // 2) + (K_DIM // GROUP) * N_DIM * 2   # 23,425,024
    return w + M * K_DIM * 2 + M * N_DIM * 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--m", type=int, default=1, help="rows of the activation (1 = decode shape; 2048 = prefill shape)")
    ap.add_argument("--sets", type=int, default=16, help="distinct weight sets rotated through (each 2 x 23.4 MB)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def cpu_baseline(budget_s: float, M: int):
    """The reference's CPU-runnable form of this linear — eager-PyTorch dequantise + matmul
    (awq_triton.py:342-368 + awq.py:447) — restated in oracle/torch_cpu.py, timed on this host."""
    import torch

    from oracle import torch_cpu
    from sglang_awq_amd import synth

    qw, s, qz = synth.make_awq_weights(K_DIM, N_DIM, GROUP, "f16", "A", 1234)
    x = synth.make_activations(M, K_DIM, "f16", "A", 1234)
    tq, ts, tz, tx = (torch.from_numpy(a.copy()) for a in (qw, s, qz, x))
    torch_cpu.linear_cpu(tx, tq, ts, tz)      # warm-up
    iters, t0 = 0, time.perf_counter()
    while True:
        torch_cpu.linear_cpu(tx, tq, ts, tz)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 200:
            break
    per = el / iters
    return {"value": round(algorithmic_bytes(M) / per / 1e9, 4), "unit": "GB/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_linear": round(per * 1e3, 2),
            "sample": f"{iters} x (eager torch CPU dequantise + matmul) of the same M={M} K={K_DIM} N={N_DIM} g={GROUP} fp16 linear, "
                      f"{el:.1f} s on {torch.get_num_threads()} threads ({os.cpu_count()} logical CPUs)"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        respawn_under_torchrun(args)          # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    from sglang_awq_amd import ops, synth
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.distributed import init_tensor_parallel
    from sglang_awq_amd.linear import ColumnParallelLinear, RowParallelLinear

    # BENCH_TEST_BACKEND=gloo + BENCH_TEST_ONE_DEVICE=1: rehearsal of the N > 1 code path on a one-GPU box (every
    # rank on device 0, collectives over gloo, no graph capture).  Never used for reported numbers.
    test_backend = os.environ.get("BENCH_TEST_BACKEND")
    dev = torch.device("cuda", 0 if os.environ.get("BENCH_TEST_ONE_DEVICE") else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        tp = init_tensor_parallel(backend=test_backend or "nccl", device=None if test_backend else dev)
    else:
        tp = init_tensor_parallel()
    M = args.m
    cfg = AWQConfig(weight_bits=4, group_size=GROUP, zero_point=True)

    # ---- synthetic layers: per-rank shard = the BASELINE shape (weak scaling) -------------------
    def make_layer(kind, seed, checked):
        if kind == "col":
            layer = ColumnParallelLinear(K_DIM, N_DIM * tp.world_size, bias=False, quant_config=cfg, params_dtype=torch.float16)
        else:
            layer = RowParallelLinear(K_DIM * tp.world_size, N_DIM, bias=False, quant_config=cfg, params_dtype=torch.float16)
        layer.to(dev)
        if checked:
            qw, s, qz = synth.make_awq_weights(K_DIM, N_DIM, GROUP, "f16", "A", seed)
            layer.qweight.data.copy_(torch.from_numpy(qw)); layer.scales.data.copy_(torch.from_numpy(s)); layer.qzeros.data.copy_(torch.from_numpy(qz))
        else:
            g = torch.Generator(device=dev); g.manual_seed(seed)
            layer.qweight.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
            layer.qzeros.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qzeros.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
            layer.scales.data.copy_((0.005 + 0.015 * torch.rand(layer.scales.shape, device=dev, generator=g)).half())
        layer.process_weights_after_loading()
        return layer

    sets = max(1, args.sets)
    cols = [make_layer("col", 1234 + 2 * i, i == 0) for i in range(sets)]
    rows = [make_layer("row", 1235 + 2 * i, i == 0) for i in range(sets)]
    x_np = synth.make_activations(M, K_DIM, "f16", "A", 1234 + rank)
    x_col = torch.from_numpy(x_np.copy()).to(dev)
    x_row = torch.from_numpy(synth.make_activations(M, K_DIM, "f16", "A", 4321 + rank).copy()).to(dev)

    # ---- parity spot-check of set 0 against the oracle (rank 0; the checker, not the measured path)
    if rank == 0:
        from oracle import c_oracle

        y = cols[0](x_col)[0]
        qw, s, qz = synth.make_awq_weights(K_DIM, N_DIM, GROUP, "f16", "A", 1234)
        Mc = min(M, 2)
        _, exact = c_oracle.gemm(x_np[:Mc], qw, s, qz, want_exact=True)
        got = y[:Mc].float().cpu().numpy().astype(np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(exact), 2.0 ** -14))) - 10)
        if not np.all(np.abs(got - exact) <= 0.5 * ulp + 1e-3):
            raise SystemExit("bench: GPU result does not match the oracle; refusing to time a wrong kernel")

    def step(i):
        c, r = cols[i % sets], rows[i % sets]
        c(x_col)
        r(x_row)                      # includes the all-reduce when tp > 1

    # ---- graphs of `sets` steps (one pass over every weight set) + a remainder graph ------------
    use_graph = not args.no_graph and not test_backend
    graphs = {}

    def capture(n):
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may touch the runtime while this thread captures (N > 1)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for i in range(n):
                step(i)
        return g

    def run_steps(n):
        if not use_graph:
            for i in range(n):
                step(i)
            return
        full, rem = divmod(n, sets)
        for _ in range(full):
            graphs[sets].replay()
        if rem:
            graphs[rem].replay()

    for i in range(min(sets, 4)):
        step(i)                        # eager warm-up: workspace allocation, RCCL communicator setup
    torch.cuda.synchronize()
    if use_graph:
        try:
            graphs[sets] = capture(sets)
            for n in {args.steps % sets, args.warmup % sets} - {0}:
                graphs[n] = capture(n)
        except Exception as e:         # e.g. a collective that cannot be captured on this stack
            if rank == 0:
                print(f"bench: graph capture failed ({e!r}); falling back to eager launches", file=sys.stderr)
            use_graph = False
            graphs.clear()
            torch.cuda.synchronize()

    def timed(n):
        """barrier + sync, n steps, sync + barrier; returns (wall seconds, HIP-event seconds)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tp.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        run_steps(n)
        e1.record()
        torch.cuda.synchronize(); tp.barrier()
        t1 = time.perf_counter()
        return t1 - t0, e0.elapsed_time(e1) * 1e-3

    run_steps(args.warmup)
    wall, ev = timed(args.steps)
    if world > 1:
        tmax = torch.tensor([wall, ev], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, ev = float(tmax[0]), float(tmax[1])

    # cache-hot variant: one weight set only (fits the 256 MiB Infinity Cache)
    hot = None
    if rank == 0 and world == 1:
        saved = (cols, rows, sets, dict(graphs))
        cols, rows, sets = cols[:1], rows[:1], 1
        graphs.clear()
        if use_graph:
            graphs[1] = capture(1)
        run_steps(50)
        _, ev_hot = timed(500)
        hot = ev_hot / 500
        cols, rows, sets, graphs = saved[0], saved[1], saved[2], saved[3]

    # the drop-in op on the checkpoint layout, same rotation of weight sets (N = 1 only)
    op_us = None
    if rank == 0 and world == 1 and (M <= 16 or M >= 1024):
        def op_pass():
            for i in range(sets):
                ops.awq_gemm(x_col, cols[i].qweight, cols[i].scales, cols[i].qzeros, 1)
        op_pass()
        torch.cuda.synchronize()
        g_op = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_op):
            op_pass()
        for _ in range(3):
            g_op.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = max(1, (1000 if M <= 16 else 100) // sets)
        for _ in range(reps):
            g_op.replay()
        e1.record()
        torch.cuda.synchronize()
        op_us = e0.elapsed_time(e1) * 1e3 / (reps * sets)

    # context only (not the metric): the same decode kernel on a 70B-class matrix, where the fixed per-launch costs
    # (kernel boundary, time to first load, reduction) amortise — what fraction of peak the kernel itself reaches
    big = None
    if rank == 0 and world == 1 and M == 1:
        BK, BN, nb = 8192, 28672, 5                      # 5 x 122 MB of packed weight: more than 2 x the Infinity Cache
        gen = torch.Generator(device=dev); gen.manual_seed(99)
        packs = []
        for _ in range(nb):
            bqw = torch.randint(-2 ** 31, 2 ** 31 - 1, (BK, BN // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
            bqz = torch.randint(-2 ** 31, 2 ** 31 - 1, (BK // GROUP, BN // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
            bsc = (0.005 + 0.015 * torch.rand((BK // GROUP, BN), device=dev, generator=gen)).half()
            packs.append(ops.awq_repack(bqw, bsc, bqz))
            del bqw, bqz, bsc
        xb = torch.randn(1, BK, device=dev, generator=gen).half()

        def big_pass():
            for pk in packs:
                ops.awq_gemm_repacked(xb, pk, BK, BN, GROUP)
        big_pass()
        torch.cuda.synchronize()
        g_big = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_big):
            big_pass()
        for _ in range(3):
            g_big.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            g_big.replay()
        e1.record()
        torch.cuda.synchronize()
        big_us = e0.elapsed_time(e1) * 1e3 / (40 * nb)
        big_bytes = BK * BN // 2 + (BK // GROUP) * BN // 2 + (BK // GROUP) * BN * 2 + BK * 2 + BN * 2
        big = {"shape": f"M=1 K={BK} N={BN} g={GROUP}", "us_per_launch": round(big_us, 2), "GBps": round(big_bytes / big_us / 1e3, 1),
               "frac_of_8TBps": round(big_bytes / big_us / 1e3 / HBM_PEAK_GBPS, 4)}
        del packs

    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    launches_per_step = 2
    per_launch = ev / (args.steps * launches_per_step) if world == 1 else None
    bytes_step = launches_per_step * algorithmic_bytes(M)
    value = world * bytes_step * args.steps / wall / 1e9
    flops_step = launches_per_step * 2 * M * K_DIM * N_DIM
    out = {
        "metric": "AWQ int4 GEMM GB/s (algorithmic bytes: packed weight + x + y) at the decode shape" if M <= 64 else
                  "AWQ int4 GEMM at the prefill shape (see roofline for TFLOP/s)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"awq_gemm M={M} K={K_DIM} N={N_DIM} g={GROUP} fp16 (BASELINE configs[{1 if M <= 64 else 2}]); step = column-parallel + "
                               f"row-parallel AWQ linear per rank" + (" + RCCL all-reduce [M,11008] fp16" if world > 1 else ""),
                   "weight_sets": sets, "graph_replay": use_graph, "parallelism": f"tp{world}",
                   "tflops": round(world * flops_step * args.steps / wall / 1e12, 3),
                   "weight_layout": "MFMA-fragment-major copy made once at load (awq_repack); checkpoint tensors kept",
                   "same_kernel_large_matrix": big,
                   "awq_gemm_op_checkpoint_layout": None if op_us is None else {
                       "us_per_launch": round(op_us, 3), "GBps": round(algorithmic_bytes(M) / op_us / 1e3, 1),
                       "frac_of_8TBps": round(algorithmic_bytes(M) / op_us / 1e3 / HBM_PEAK_GBPS, 4),
                       "pmc_traffic_bytes": PMC_TRAFFIC_BYTES_M1_CHECKPOINT_LAYOUT if M == 1 else None,
                       "tflops": round(2 * M * K_DIM * N_DIM / op_us / 1e6, 1),
                       "note": "split-K kernel on the AutoAWQ layout" if M <= 16 else
                               "prefill-sized call: the op repacks into workspace on the fly, then the fragment-major kernel"}},
    }
    if world == 1:
        ach = algorithmic_bytes(M) / per_launch / 1e9
        bound = "hbm" if M <= 64 else "mfma"
        if bound == "hbm":
            out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBPS, 4),
                               "traffic": PMC_TRAFFIC_BYTES_M1 if M == 1 else None,
                               "traffic_note": "bytes per launch, rocprofv3 PMC passes committed under profiles/ (not collected live)",
                               "kernel": "gemv_repacked_kernel", "us_per_launch": round(per_launch * 1e6, 3),
                               "cache_hot_GBps": round(algorithmic_bytes(M) / (hot / launches_per_step) / 1e9, 1) if hot else None}
        else:
            tf = 2 * M * K_DIM * N_DIM / per_launch / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None, "us_per_launch": round(per_launch * 1e6, 3)}
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, M)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
