#!/usr/bin/env python3
"""Benchmark of the AWQ int4 quantized-linear hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1], the decode shape): awq_gemm at M = 1, K = 4096, N = 11008,
g = 128, fp16.  One STEP = one column-parallel AWQ linear followed by one row-parallel AWQ linear, i.e.
two fused int4 GEMV launches per rank and, for N > 1, one RCCL all-reduce of the [M, N_out] fp16 partial sums.
  --scaling weak   (default) every rank holds a full 4096 x 11008 shard (the TP = N layer is N times wider);
  --scaling strong the ONE 4096 x 11008 layer is sharded as SURVEY §8(e) lists: column-parallel N / tp
                   (11008, 5504, 2752, 1376 columns), row-parallel K / tp (4096 ... 512 rows, whole groups),
                   all-reduce payload [M, 11008];
  --shapes 70b-tp8 the four per-rank linears of Llama-2-70B at TP = 8 (BASELINE configs[4]): qkv 8192 -> 1280,
                   o 1024 -> 8192 (+ AR), gate_up 8192 -> 7168, down 3584 -> 8192 (+ AR); with fewer than 8 ranks the
                   per-rank shapes are kept and the all-reduce runs over the ranks present.
The linears are this package's AWQLinearMethod as deployed: at load time (process_weights_after_loading) the
weights get a one-time MFMA-fragment-major copy and decode batches run the kernel on that copy.  Weights rotate
through `--sets` distinct copies (> 2x the 256 MiB Infinity Cache) so the stream comes from HBM, not from cache.
Steps are replayed from a captured HIP graph (the decode path of the reference replays graphs too), so the timed
region contains exactly K steps of device work and no Python.

Prints ONE JSON line (rank 0).  `value` is whole-job algorithmic GB/s: bytes every rank must move (packed weight +
activations + outputs = 23,455,232 B per GEMV at the headline shape) divided by the slowest rank's wall time over
the K timed steps (barrier + synchronize on both sides).  `roofline` prices the dominant kernel (the fused GEMV)
against 8 TB/s from HIP events around the same K steps (all K steps are one graph replay, kernel boundaries included);
`cpu_baseline` times the reference's CPU form of the same linear on this host.

At N = 1 the same line also carries the rest of BASELINE.json's metric, each measured in this run:
  `prefill`        awq_gemm at M = 2048 (configs[2]): us, TFLOP/s, roofline against 2.5 PFLOP/s dense fp16;
  `dequantize`     awq_dequantize 4096 x 11008 (configs[0]'s op on the GPU): us, GB/s against 113,602,560 B;
  `awq_gemm_op`    the drop-in `sgl_kernel.awq_gemm` op on the checkpoint tensors (M = 1 and M = 2048);
  `decode_7b_tp1`  Llama-2-7B-AWQ decode tok/s at batch 1 and 32 (configs[3]), median-latency method of the
                   reference's bench_one_batch.py:497-623.
"""
import argparse
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys
import tempfile
import time

# dmabuf IPC between processes: the platform this runs on documents it as required for RCCL / cross-process GPU
# memory sharing on its host driver (legacy IPC fails with hipIpcGetMemHandle: invalid argument); harmless at N = 1
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_DIM, N_DIM, GROUP = 4096, 11008, 128
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0          # ... 6.29 TB/s measured (float4 copy), the practical ceiling of a streaming kernel
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16 / bf16
# per-rank linears of Llama-2-70B at TP = 8 (hidden 8192, 64 q / 8 kv heads of 128, intermediate 28672): (kind, K, N)
SHAPES_70B_TP8 = [("col", 8192, 1280), ("row", 1024, 8192), ("col", 8192, 7168), ("row", 3584, 8192)]


def linear_bytes(M: int, K: int, N: int, g: int = GROUP) -> int:
    """Algorithmic bytes of one AWQ linear launch: packed weight + zeros + scales + x + y (fp16)."""
    return K * N // 2 + (K // g) * (N // 2) + (K // g) * N * 2 + M * K * 2 + M * N * 2


def algorithmic_bytes(M: int) -> int:
    return linear_bytes(M, K_DIM, N_DIM)          # 23,455,232 at M = 1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--m", type=int, default=1, help="rows of the activation of the headline workload (1 = decode shape)")
    ap.add_argument("--sets", type=int, default=16, help="distinct weight sets rotated through (each 2 x 23.4 MB)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--shapes", choices=["baseline", "70b-tp8"], default="baseline")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--sections", default="prefill,dequantize,op,decode,large",
                    help="extra records measured at N = 1 beside the headline (comma list; '' = none)")
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


def pmc_traffic(kernel_substr: str, tag: str = "*m1*"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r*_pmc_fetch_size_<tag>.csv and the
    matching write_size file; newest round first): 2 x FETCH_SIZE (gfx950 reports half of a wide coalesced read stream,
    MI355X_MICROARCH.md §HBM) + WRITE_SIZE, both in KiB per dispatch, averaged over the dispatches of the kernel (launches of a
    multi-launch operator are summed per call by the caller).  (None, []) if no committed pass has the kernel."""
    def mean_counter(path, counter):
        vals = []
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
        return sum(vals) / len(vals) if vals else None

    for fetch_path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_fetch_size_{tag}.csv")), reverse=True):
        write_path = fetch_path.replace("fetch_size", "write_size")
        if not os.path.exists(write_path):
            continue
        fk, wk = mean_counter(fetch_path, "FETCH_SIZE"), mean_counter(write_path, "WRITE_SIZE")
        if fk is not None and wk is not None:
            return int((2 * fk + wk) * 1024), [os.path.relpath(fetch_path, ROOT), os.path.relpath(write_path, ROOT)]
    return None, []


def pmc_traffic_sum(kernel_substr: str, tag: str, launches_per_call: int):
    """Traffic of an operator that is several launches of one kernel family per call (the prefill's wide + narrow tile regions):
    the per-dispatch mean x launches per call."""
    t, files = pmc_traffic(kernel_substr, tag)
    return (t * launches_per_call if t is not None else None), files


def rccl_summary(log_glob: str):
    """What RCCL said it chose (NCCL_DEBUG=INFO into per-process files): version, transports, channels, and any
    algorithm / protocol lines.  Best effort: the exact wording differs between RCCL releases."""
    info = {"version": None, "transports": [], "channels": None, "algo_proto": [], "lines_seen": 0}
    transports, algos = set(), []
    for path in sorted(glob.glob(log_glob)):
        try:
            with open(path, errors="replace") as f:
                for line in f:
                    info["lines_seen"] += 1
                    m = re.search(r"(RCCL|NCCL) version ([^\s]+)", line)
                    if m and not info["version"]:
                        info["version"] = f"{m.group(1)} {m.group(2)}"
                    m = re.search(r"via (P2P|SHM|NET)/([A-Za-z0-9_]+)", line)
                    if m:
                        transports.add(f"{m.group(1)}/{m.group(2)}")
                    m = re.search(r"(\d+) coll channels", line)
                    if m:
                        info["channels"] = int(m.group(1))
                    m = re.search(r"[Aa]lgo(?:rithm)?\s*[:=]?\s*(\w+).*?[Pp]roto(?:col)?\s*[:=]?\s*(\w+)", line)
                    if m:
                        ap = f"{m.group(1)}/{m.group(2)}"
                        if ap not in algos:
                            algos.append(ap)
        except OSError:
            pass
    info["transports"] = sorted(transports)
    info["algo_proto"] = algos[:8]
    return info


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        respawn_under_torchrun(args)          # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    test_backend = os.environ.get("BENCH_TEST_BACKEND")
    rccl_log = None
    if world > 1 and not test_backend:
        # ask RCCL what it picks; into files, not onto the JSON line's stdout (INIT + tuning + per-collective choice; a
        # graph replay logs nothing, so the timed region is unaffected unless capture falls back to eager launches)
        rccl_log = os.path.join(tempfile.gettempdir(), f"awq_bench_rccl_{os.environ.get('MASTER_PORT', '0')}")
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING,GRAPH")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log + ".%h.%p.log")

    import numpy as np
    import torch
    import torch.distributed as dist

    from sglang_awq_amd import ops, synth
    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.distributed import init_tensor_parallel
    from sglang_awq_amd.linear import ColumnParallelLinear, RowParallelLinear

    # BENCH_TEST_BACKEND=gloo + BENCH_TEST_ONE_DEVICE=1: rehearsal of the N > 1 code path on a one-GPU box (every
    # rank on device 0, collectives over gloo, no graph capture).  BENCH_TEST_CPU=1: the same on CPU tensors with the
    # kernels stubbed out (shapes, sharding and collectives only; tests/test_tp_cpu.py).  Never used for reported numbers.
    cpu_rehearsal = bool(os.environ.get("BENCH_TEST_CPU"))
    if cpu_rehearsal:
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", 0 if os.environ.get("BENCH_TEST_ONE_DEVICE") else local_rank)
        torch.cuda.set_device(dev)
    if world > 1:
        tp = init_tensor_parallel(backend=test_backend or "nccl", device=None if test_backend else dev)
    else:
        tp = init_tensor_parallel()
    M = args.m
    cfg = AWQConfig(weight_bits=4, group_size=GROUP, zero_point=True)

    def sync():
        if not cpu_rehearsal:
            torch.cuda.synchronize()

    # ---- which linears one step runs on this rank: (kind, K_total, N_total) as the TP layer sees them -----------
    if args.shapes == "70b-tp8":
        # per-rank shapes are those of TP = 8; the layer objects are built tp.world_size wide so the sharding code runs
        plan = [(kind, K * (tp.world_size if kind == "row" else 1), N * (tp.world_size if kind == "col" else 1))
                for kind, K, N in SHAPES_70B_TP8]
        scaling = "weak"
    elif args.scaling == "strong":
        if N_DIM % (8 * tp.world_size) or K_DIM % (GROUP * tp.world_size):
            raise SystemExit(f"strong scaling of {K_DIM} x {N_DIM} g{GROUP} is not legal at tp={tp.world_size} (awq.py:372-385)")
        plan = [("col", K_DIM, N_DIM), ("row", K_DIM, N_DIM)]
        scaling = "strong"
    else:
        plan = [("col", K_DIM, N_DIM * tp.world_size), ("row", K_DIM * tp.world_size, N_DIM)]
        scaling = "weak"
    rank_shapes = [(kind, K // (tp.world_size if kind == "row" else 1), N // (tp.world_size if kind == "col" else 1)) for kind, K, N in plan]
    bytes_step_rank = sum(linear_bytes(M, K, N) for _, K, N in rank_shapes)
    flops_step_rank = sum(2 * M * K * N for _, K, N in rank_shapes)
    ar_payloads = [M * N * 2 for kind, _, N in plan if kind == "row"] if tp.world_size > 1 else []

    def make_layer(kind, K, N, seed, checked):
        if kind == "col":
            layer = ColumnParallelLinear(K, N, bias=False, quant_config=cfg, params_dtype=torch.float16)
        else:
            layer = RowParallelLinear(K, N, bias=False, quant_config=cfg, params_dtype=torch.float16)
        layer.to(dev)
        if checked:
            qw, s, qz = synth.make_awq_weights(layer.qweight.shape[0], layer.qweight.shape[1] * 8, GROUP, "f16", "A", seed)
            layer.qweight.data.copy_(torch.from_numpy(qw)); layer.scales.data.copy_(torch.from_numpy(s)); layer.qzeros.data.copy_(torch.from_numpy(qz))
        else:
            g = torch.Generator(device=dev); g.manual_seed(seed)
            layer.qweight.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
            layer.qzeros.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qzeros.shape, dtype=torch.int64, device=dev, generator=g).to(torch.int32))
            layer.scales.data.copy_((0.005 + 0.015 * torch.rand(layer.scales.shape, device=dev, generator=g)).half())
        if cpu_rehearsal:
            layer.quant_method.apply = lambda lyr, x, bias=None: torch.zeros(x.shape[:-1] + (lyr.qweight.shape[1] * 8,), dtype=x.dtype)
        else:
            layer.process_weights_after_loading()
        return layer

    sets = max(1, args.sets)
    layer_sets = [[make_layer(kind, K, N, 1234 + 16 * i + j, i == 0 and j == 0) for j, (kind, K, N) in enumerate(plan)] for i in range(sets)]

    def make_x(rows):
        xs = []
        for j, (kind, K, N) in enumerate(rank_shapes):
            xs.append(torch.from_numpy(synth.make_activations(rows, K, "f16", "A", 1234 + 3087 * j + rank).copy()).to(dev))
        return xs

    xs = make_x(M)

    # ---- parity spot-check of set 0's first linear against the oracle (rank 0; the checker, not the measured path)
    if rank == 0 and not cpu_rehearsal:
        from oracle import c_oracle

        kind0, K0, N0 = rank_shapes[0]
        y = layer_sets[0][0](xs[0])[0]
        if tp.world_size == 1:
            qw, s, qz = synth.make_awq_weights(K0, N0, GROUP, "f16", "A", 1234)
            Mc = min(M, 2)
            _, exact = c_oracle.gemm(xs[0][:Mc].cpu().numpy(), qw, s, qz, want_exact=True)
            got = y[:Mc].float().cpu().numpy().astype(np.float64)
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(exact), 2.0 ** -14))) - 10)
            if not np.all(np.abs(got - exact) <= 0.5 * ulp + 1e-3):
                raise SystemExit("bench: GPU result does not match the oracle; refusing to time a wrong kernel")

    def step(i, layers=None, inputs=None):
        layers = layers if layers is not None else layer_sets[i % sets]
        inputs = inputs if inputs is not None else xs
        for layer, x in zip(layers, inputs):
            layer(x)                      # a row-parallel linear includes the all-reduce when tp > 1

    # ---- graphs: one holding all K timed steps (a single replay in the timed region), one for the warm-up -------
    use_graph = not args.no_graph and not test_backend and not cpu_rehearsal
    capture_error = None
    graphs = {}

    cap_stream = None

    def capture(n, fn=step):
        nonlocal cap_stream
        if cap_stream is None:
            # one capture stream for every graph of this run, with the ops' per-stream scratch created eagerly on it (a first use
            # inside a capture raises: sglang_awq_amd/ops.py)
            cap_stream = torch.cuda.Stream()
            cap_stream.wait_stream(torch.cuda.current_stream())
            ops.prepare_stream_workspaces(cap_stream, dev)
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may touch the runtime while this thread captures (N > 1)
        with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
            for i in range(n):
                fn(i)
        return g

    MAX_GRAPH_STEPS = 256

    def run_steps(n):
        if not use_graph:
            for i in range(n):
                step(i)
            return
        while n > 0:
            k = n if n in graphs else (MAX_GRAPH_STEPS if n >= MAX_GRAPH_STEPS and MAX_GRAPH_STEPS in graphs else sets if n >= sets and sets in graphs else 1)
            graphs[k].replay()
            n -= k

    for i in range(int(os.environ.get("BENCH_EAGER_PASSES", "1")) * sets):
        step(i)                        # eager set-up pass over every weight set: workspace allocation, RCCL communicator, page tables
    sync()
    if use_graph:
        try:
            for n in sorted({min(args.steps, MAX_GRAPH_STEPS), min(max(args.warmup, 1), MAX_GRAPH_STEPS), sets, 1}):
                graphs[n] = capture(n)
        except Exception as e:         # e.g. a collective that cannot be captured on this stack
            capture_error = repr(e)
            if rank == 0:
                print(f"bench: graph capture failed ({e!r}); falling back to eager launches", file=sys.stderr)
            use_graph = False
            graphs.clear()
            sync()
    if world > 1 and not args.no_graph and not test_backend and not cpu_rehearsal:
        # every rank replays or every rank launches eagerly: a rank whose capture failed must not leave the others replaying
        ok = torch.tensor([1 if use_graph else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and use_graph:
            use_graph = False
            graphs.clear()
            capture_error = capture_error or "capture failed on another rank"
        sync()


    def timed_wall(n):
        """The contract's timed region: barrier + synchronize, n steps, synchronize + barrier; wall seconds."""
        tp.barrier(); sync()
        t0 = time.perf_counter()
        run_steps(n)
        sync(); tp.barrier()
        return time.perf_counter() - t0

    def timed_events(n, runner=None):
        """The same n steps between HIP events on the launch stream (torch's current stream): device time of the kernels
        and their boundaries.  (Queuing the events behind a spin kernel, to keep the host's graph-launch latency out of
        the interval, measured the same: 6.963 vs 6.956 us per launch at K = 20 — one graph replay holds all K steps.)"""
        if cpu_rehearsal:
            return float("nan")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tp.barrier(); sync()
        e0.record()
        (runner or run_steps)(n)
        e1.record()
        sync()
        return e0.elapsed_time(e1) * 1e-3

    run_steps(args.warmup)
    # beyond the W warm-up steps: bring the chip to its steady clocks before the timed region (the first launches after an idle
    # period run up to 10 % slower; with K = 20 the whole timed region is ~0.3 ms).  Untimed; a fixed ~0.25 s of the same steps.
    clock_warm = 0
    if not cpu_rehearsal and world == 1:
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.25:
            run_steps(max(args.warmup, sets))
            clock_warm += max(args.warmup, sets)
            sync()
    wall = timed_wall(args.steps)
    ev = timed_events(args.steps)
    if world > 1:
        tmax = torch.tensor([wall, ev if ev == ev else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, ev = float(tmax[0]), float(tmax[1])

    sections = set(s for s in args.sections.split(",") if s) if (world == 1 and rank == 0 and not cpu_rehearsal and args.shapes == "baseline"
                                                                 and args.scaling == "weak") else set()

    def graph_time(fn_pass, launches_per_pass, min_launches, max_seconds=2.0):
        """us per launch of `fn_pass` (a pass over the rotating weight sets) replayed from a graph between HIP events."""
        fn_pass()
        sync()
        g = capture(1, lambda _i: fn_pass())
        for _ in range(2):
            g.replay()
        sync()
        reps = max(2, -(-min_launches // launches_per_pass))
        t = timed_events(reps, lambda n: [g.replay() for _ in range(n)])
        return t * 1e6 / (reps * launches_per_pass)

    # cache-hot variant: one weight set only (fits the 256 MiB Infinity Cache)
    hot = None
    if sections and M <= 64:
        HOT = 32                       # one graph of 32 steps on weight set 0 (a 2-launch graph would time the replay overhead)
        hot = graph_time(lambda: [step(0) for _ in range(HOT)], HOT * len(plan), 2000)

    # ---- prefill (BASELINE configs[2]): the same layers at M = 2048, MFMA-bound --------------------------------
    prefill = None
    if "prefill" in sections and M != 2048:
        MP = 2048
        xp = make_x(MP)
        us = graph_time(lambda: [step(i, inputs=xp) for i in range(sets)], sets * len(plan), 300)
        tf = 2 * MP * K_DIM * N_DIM / us / 1e6
        prefill = {"workload": f"awq_gemm M={MP} K={K_DIM} N={N_DIM} g={GROUP} fp16 (BASELINE configs[2]) via AWQLinearMethod.apply",
                   "kernel": "gemm_repacked_pipelined_kernel", "us_per_launch": round(us, 2), "tflops": round(tf, 1),
                   "GBps": round(linear_bytes(MP, K_DIM, N_DIM) / us / 1e3, 1),
                   "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None}}
        # two launches per call at this shape (wide + narrow tile regions): traffic = sum over both, from the committed PMC passes
        pt, pfiles = pmc_traffic_sum("gemm_repacked_pipelined_kernel", "prefill*", 2)
        prefill["roofline"]["traffic"] = pt
        prefill["roofline"]["traffic_note"] = (f"2 launches x mean (2 x FETCH_SIZE + WRITE_SIZE) per dispatch from {pfiles}; algorithmic bytes "
                                               f"{linear_bytes(MP, K_DIM, N_DIM)}: every 128-row tile re-reads its weight columns (served by L2 / Infinity Cache)")
        del xp

    # ---- awq_dequantize (the op of BASELINE configs[0], on the GPU): HBM-bound, 113,602,560 B per call ---------
    dequant = None
    if "dequantize" in sections:
        cols = [ls[0] for ls in layer_sets]
        nrot = min(sets, 8)                                    # 8 x (23.4 MB in + 90.2 MB out) of distinct buffers
        outs = [torch.empty((K_DIM, N_DIM), dtype=torch.float16, device=dev) for _ in range(nrot)]
        lib = ops._lib.load()
        import ctypes

        def dq_pass():
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for i in range(nrot):
                c = cols[i]
                rc = lib.awq_dequantize(ops._vp(c.qweight), ops._vp(c.scales), ops._vp(c.qzeros), ops._vp(outs[i]), K_DIM, N_DIM, GROUP,
                                        ops._lib.DTYPE_F16, st)
                if rc:
                    raise SystemExit(f"awq_dequantize failed: {rc}")
        us = graph_time(dq_pass, nrot, 400)
        nbytes = K_DIM * N_DIM // 2 + (K_DIM // GROUP) * (N_DIM // 2) + (K_DIM // GROUP) * N_DIM * 2 + K_DIM * N_DIM * 2   # 113,602,560
        # spot check against the op's own output path (bit-equality with the oracle is the -m gpu tests' job)
        w_op = ops.awq_dequantize(cols[0].qweight, cols[0].scales, cols[0].qzeros)
        if not torch.equal(w_op, outs[0]):
            raise SystemExit("bench: awq_dequantize C-ABI call and torch op disagree")
        dequant = {"workload": f"awq_dequantize K={K_DIM} N={N_DIM} g={GROUP} fp16 (sgl_kernel.awq_dequantize; through the C ABI into preallocated outputs)",
                   "kernel": "dequant_kernel", "us_per_launch": round(us, 2), "bytes": nbytes, "GBps": round(nbytes / us / 1e3, 1),
                   "roofline": {"bound": "hbm", "achieved": round(nbytes / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(nbytes / us / 1e3 / HBM_COPY_GBPS, 4),
                                "traffic": None}}
        dtr, dfiles = pmc_traffic("dequant_kernel", "dequant*")
        dequant["roofline"]["traffic"] = dtr
        dequant["roofline"]["traffic_note"] = f"2 x FETCH_SIZE + WRITE_SIZE per dispatch from {dfiles} (separate --pmc runs, not collected live)"
        del outs, w_op

    # ---- the drop-in op `sgl_kernel.awq_gemm` on the checkpoint tensors, same rotation of weight sets -----------
    op_rec = None
    if "op" in sections:
        cols = [ls[0] for ls in layer_sets]
        # the op's repacked-copy cache is opt-in (off until sgl_kernel_compat.install() has hooked the reference's weight-update
        # paths, or the caller vouches for invalidation): this benchmark owns its weights and never rewrites them
        cache_default = ops.awq_gemm_cache_info()
        ops.awq_gemm_cache_enable(True)
        op_rec = {"cache": True, "cache_default": f"{cache_default['mode']} (enabled={cache_default['enabled']} before this section switched it on)"}
        for rows in (1, 2048):
            xo = make_x(rows)[0]

            passes = 2 if rows == 1 else 1          # 32 launches per graph, as the other sections (a short graph times the replay boundary)

            def op_pass():
                for _ in range(passes):
                    for i in range(sets):
                        ops.awq_gemm(xo, cols[i].qweight, cols[i].scales, cols[i].qzeros, 1)
            us = graph_time(op_pass, sets * passes, 1000 if rows == 1 else 100)
            rec = {"us_per_launch": round(us, 3), "GBps": round(linear_bytes(rows, K_DIM, N_DIM) / us / 1e3, 1),
                   "tflops": round(2 * rows * K_DIM * N_DIM / us / 1e6, 1)}
            if rows == 1:
                rec["frac_of_8TBps"] = round(linear_bytes(rows, K_DIM, N_DIM) / us / 1e3 / HBM_PEAK_GBPS, 4)
                y_op = ops.awq_gemm(xo, cols[0].qweight, cols[0].scales, cols[0].qzeros, 1)
                rec["bit_identical_to_awq_gemm_repacked"] = bool(torch.equal(
                    y_op, ops.awq_gemm_repacked(xo, cols[0].awq_packed, K_DIM, N_DIM, GROUP)))
            else:
                rec["frac_of_2p5PF"] = round(2 * rows * K_DIM * N_DIM / us / 1e6 / MFMA_PEAK_TFLOPS, 4)
            op_rec[f"m{rows}"] = rec
        op_rec["note"] = ("torch.ops.sgl_kernel.awq_gemm(x, qweight, scales, qzeros, 1) on the AutoAWQ tensors; with the cache on, the op keeps "
                          "one MFMA-fragment-major copy per weight (made on the first eager call, validated by tensor versions and storage "
                          "weak references, dropped by the weight-update hooks of sgl_kernel_compat.install()) and runs the same kernels as "
                          "awq_gemm_repacked; with the cache off the checkpoint-layout split-K kernel runs (13 us at M = 1)")
        ops.awq_gemm_cache_enable(False)           # also drops the copies

    # context only (not the metric): the same decode kernel on a 70B-class matrix, where the fixed per-launch costs
    # (kernel boundary, time to first load, reduction) amortise — what fraction of peak the kernel itself reaches
    big = None
    if "large" in sections and M == 1:
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
        big_us = graph_time(lambda: [ops.awq_gemm_repacked(xb, pk, BK, BN, GROUP) for pk in packs], nb, 200)
        big_bytes = linear_bytes(1, BK, BN)
        big = {"shape": f"M=1 K={BK} N={BN} g={GROUP}", "us_per_launch": round(big_us, 2), "GBps": round(big_bytes / big_us / 1e3, 1),
               "frac_of_8TBps": round(big_bytes / big_us / 1e3 / HBM_PEAK_GBPS, 4)}
        del packs

    # ---- Llama-2-7B-AWQ decode, TP = 1 (BASELINE configs[3]) ----------------------------------------------------
    decode = None
    if "decode" in sections:
        del layer_sets, graphs
        torch.cuda.empty_cache()
        import bench_decode

        try:
            recs = bench_decode.measure("7b", [1, 32], steps=48, context=1024, dev=dev)
            decode = {"method": "median over 48 single-step latencies, each bracketed by torch.cuda.synchronize (bench_one_batch.py:497-623); "
                                "1024 positions already in the KV cache; greedy; synthetic weights; whole step replayed from one HIP graph",
                      "weight_GB_per_step": recs[0]["weight_GB_per_step"]}
            for r in recs:
                decode[f"bs{r['batch']}"] = {k: r[k] for k in ("value", "unit", "median_step_ms", "device_step_ms", "tok_per_s_device", "hbm_GBps_device",
                                                               "norm_order", "graph_replay")}
            if args.cpu_seconds > 0:
                from sglang_awq_amd.llama import LlamaConfig

                decode["cpu_baseline"] = bench_decode.decode_cpu_baseline(LlamaConfig.llama2_7b(), min(args.cpu_seconds, 8.0))
        except Exception as e:      # the headline must not be lost to the harness
            decode = {"error": repr(e)}

    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    launches_per_step = len(plan)
    per_launch = ev / (args.steps * launches_per_step) if (world == 1 and ev == ev) else None
    value = world * bytes_step_rank * args.steps / wall / 1e9
    headline_shape = args.shapes == "baseline"
    out = {
        "metric": "AWQ int4 GEMM GB/s (algorithmic bytes: packed weight + x + y) at the decode shape" if M <= 64 else
                  "AWQ int4 GEMM at the prefill shape (see roofline for TFLOP/s)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 6), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": (f"awq_gemm M={M} K={K_DIM} N={N_DIM} g={GROUP} fp16 (BASELINE configs[{1 if M <= 64 else 2}]); step = column-parallel + "
                                f"row-parallel AWQ linear per rank" if headline_shape else
                                f"Llama-2-70B TP=8 per-rank AWQ linears at M={M} (BASELINE configs[4]): qkv 8192->1280, o 1024->8192, gate_up 8192->7168, down 3584->8192")
                               + (f" + RCCL all-reduce {ar_payloads} B" if world > 1 else ""),
                   "per_rank_linears": [f"{kind} {K}x{N}" for kind, K, N in rank_shapes],
                   "weight_sets": sets, "extra_clock_warmup_steps": clock_warm, "graph_replay": use_graph, "graph_capture_error": capture_error, "parallelism": f"tp{world}",
                   "tflops": round(world * flops_step_rank * args.steps / wall / 1e12, 3),
                   "weight_layout": "MFMA-fragment-major copy made once at load (awq_repack); checkpoint tensors kept",
                   "same_kernel_large_matrix": big},
    }
    if world > 1:
        out["config"]["collective"] = {"backend": test_backend or "nccl (RCCL)", "all_reduce_bytes": ar_payloads, "in_graph": use_graph,
                                       "rccl": rccl_summary(rccl_log + ".*.log") if rccl_log else None}
    if per_launch is not None:
        if M <= 64:
            ach = bytes_step_rank / launches_per_step / per_launch / 1e9
            traffic, traffic_files = pmc_traffic("gemv_rp2_kernel") if (headline_shape and M == 1) else (None, [])
            out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(ach / HBM_COPY_GBPS, 4),
                               "traffic": traffic,
                               "traffic_note": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE read from the committed rocprofv3 PMC passes "
                                               f"{traffic_files} (separate --pmc runs, not collected live)",
                               "kernel": "gemv_rp2_kernel" if headline_shape else "gemv_rp2_kernel / gemv_repacked_kernel (per shape)",
                               "us_per_launch": round(per_launch * 1e6, 3),
                               "timing": f"HIP events around the {args.steps} timed steps ({args.steps * launches_per_step} launches, "
                                         f"{'one graph replay' if use_graph and args.steps <= MAX_GRAPH_STEPS else 'graph replays' if use_graph else 'eager'}); kernel boundaries are inside the interval",
                               "cache_hot_GBps": round(bytes_step_rank / launches_per_step / hot / 1e3, 1) if hot else None}
        else:
            tf = flops_step_rank / launches_per_step / per_launch / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None, "us_per_launch": round(per_launch * 1e6, 3)}
    for name, rec in (("prefill", prefill), ("dequantize", dequant), ("awq_gemm_op", op_rec), ("decode_7b_tp1", decode)):
        if rec is not None:
            out[name] = rec
    if world == 1 and args.cpu_seconds > 0 and not cpu_rehearsal:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, M)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
