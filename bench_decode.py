#!/usr/bin/env python3
"""Llama-2-7B-AWQ decode tok/s, TP=1, batch 1-32 (BASELINE configs[3]) on one MI355X; `--model 70b` is Llama-2-70B-AWQ
(configs[4]) — at TP=1 it fits one MI355X (35 GB packed + the repacked copies), and under
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench_decode.py --model 70b` it runs TP=N over RCCL.

Method of the reference's bench_one_batch.py (:497-623): per decode step `synchronize; tic; decode;
synchronize`, report the MEDIAN step latency and batch / latency as tok/s.  The decode step is this
package's minimal Llama (sglang_awq_amd/llama.py) with every projection an AWQ linear on the gfx950
kernels, replayed from one captured HIP graph; weights are synthetic (no checkpoint here; the reference
benchmarks `--load-format dummy` the same way).  Prints one JSON line per batch size and a final
summary line; `--cpu-seconds` bounds the CPU baseline (the reference's eager-torch CPU dequantise +
matmul, timed on ONE decoder layer's four linears and scaled by the layer count).

    python bench_decode.py [--model 7b] [--batches 1,2,4,8,16,32] [--steps 64] [--context 1024]
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this host driver
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def measure(model_name: str, batches, steps: int = 64, context: int = 1024, dev=None, verbose: bool = False, layers=None):
    """Decode tok/s of a synthetic Llama-AWQ model per batch size: median over `steps` single-step latencies, each bracketed by
    torch.cuda.synchronize (the method of the reference's bench_one_batch.py:497-623), plus the device-only time of the same
    step (16 graph replays back to back between HIP events).  Returns one dict per batch size."""
    import torch

    from sglang_awq_amd.awq import AWQConfig
    from sglang_awq_amd.distributed import get_tensor_model_parallel_world_size
    from sglang_awq_amd.llama import GraphedDecoder, LlamaConfig, LlamaForCausalLM

    world = get_tensor_model_parallel_world_size()
    if dev is None:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    if model_name == "7b":
        cfg = LlamaConfig.llama2_7b()
    elif model_name == "70b":
        cfg = LlamaConfig.llama2_70b()
    else:
        cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=8,
                          vocab_size=2048)
    if layers:
        cfg.num_hidden_layers = int(layers)           # a shortened stack (rehearsals, tests): per-layer work and collectives unchanged
    max_seq = context + steps + 32
    with torch.device(dev):
        model = LlamaForCausalLM(cfg, AWQConfig(4, 128, True), max_batch=max(batches), max_seq=max_seq)
    model.init_synthetic_(0)
    torch.cuda.synchronize()

    # weight bytes one decode step must stream (packed AWQ linears + fp16 lm_head): the HBM floor
    lin_bytes = sum(l.qweight.numel() * 4 + l.qzeros.numel() * 4 + l.scales.numel() * 2
                    for layer in model.layers for l in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj))
    head_bytes = model.lm_head.numel() * 2
    name = {"7b": "Llama-2-7B-AWQ", "70b": "Llama-2-70B-AWQ", "tiny": "tiny-llama"}[model_name]
    results = []
    for B in batches:
        dec = GraphedDecoder(model, B, start_pos=context)
        if os.environ.get("BENCH_TEST_BACKEND"):
            # rehearsal over gloo: a gloo collective cannot be captured (and a failed capture poisons the stream): step eagerly, say so
            dec._set_attention_splits()
            dec.capture_error = f"not captured: rehearsal backend {os.environ['BENCH_TEST_BACKEND']}"
        else:
            dec.capture(warmup=2)
        dec.run(3)
        lat = []
        for _ in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.run(1)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        med = statistics.median(lat)
        # device-only time of the same step: K replays back to back between HIP events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dec.reset()
        e0.record()
        for _ in range(16):
            dec.run(1)
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / 16
        r = {"metric": f"{name} decode tok/s TP={world}", "batch": B,
             "value": round(B / med, 1), "unit": "tok/s", "median_step_ms": round(med * 1e3, 4), "device_step_ms": round(dev_ms, 4),
             "tok_per_s_device": round(B / (dev_ms * 1e-3), 1), "context": context, "steps": steps,
             "weight_GB_per_step": round((lin_bytes + head_bytes) / 1e9, 3),
             "hbm_GBps_device": round((lin_bytes + head_bytes) / (dev_ms * 1e-3) / 1e9, 1), "data": "synthetic", "dtype": "f16",
             "graph_replay": dec.graph is not None, "graph_capture_error": dec.capture_error,
             "norm_order": model.layers[0].norm_order(B)}
        results.append(r)
        if verbose:
            print(json.dumps(r), flush=True)
        del dec
    del model
    torch.cuda.empty_cache()
    return results


def decode_cpu_baseline(cfg, budget_s: float):
    """The reference's CPU form of a decode step's AWQ linears — eager-torch dequantise + matmul (awq_triton.py:342-368 +
    awq.py:447, restated in oracle/torch_cpu.py) — timed on ONE decoder layer's four linears at batch 1 and scaled by the layer
    count (attention, norms and lm_head not counted: an upper bound on the CPU's tok/s).  Bounded by `budget_s` seconds."""
    import torch

    from oracle import torch_cpu
    from sglang_awq_amd import synth

    shapes = [(cfg.hidden_size, (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim), (cfg.hidden_size, cfg.hidden_size),
              (cfg.hidden_size, 2 * cfg.intermediate_size), (cfg.intermediate_size, cfg.hidden_size)]
    tens = []
    for i, (K, N) in enumerate(shapes):
        qw, s, qz = synth.make_awq_weights(K, N, 128, "f16", "A", 50 + i)
        x = synth.make_activations(1, K, "f16", "A", 60 + i)
        tens.append(tuple(torch.from_numpy(a.copy()) for a in (x, qw, s, qz)))
    for x, qw, s, qz in tens:
        torch_cpu.linear_cpu(x, qw, s, qz)
    n, t0 = 0, time.perf_counter()
    while n < 1 or time.perf_counter() - t0 < budget_s:
        for x, qw, s, qz in tens:
            torch_cpu.linear_cpu(x, qw, s, qz)
        n += 1
    per_layer = (time.perf_counter() - t0) / n
    return {"value": round(1.0 / (per_layer * cfg.num_hidden_layers), 4), "unit": "tok/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x the four AWQ linears of ONE decoder layer at batch 1 (eager torch CPU dequantise + matmul), "
                      f"{per_layer * 1e3:.0f} ms per layer, scaled by {cfg.num_hidden_layers} layers; attention / lm_head not counted"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=["7b", "70b", "tiny"])
    ap.add_argument("--batches", default="1,2,4,8,16,32")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--context", type=int, default=1024,
                    help="KV positions already in the cache when timing starts (bench_one_batch.py defaults to 1024 input tokens)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--layers", type=int, default=0, help="run only this many decoder layers (0 = the model's own count)")
    args = ap.parse_args()

    import torch

    from sglang_awq_amd.distributed import init_tensor_parallel
    from sglang_awq_amd.llama import LlamaConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # BENCH_TEST_BACKEND=gloo + BENCH_TEST_ONE_DEVICE=1: rehearsal of the N > 1 path on a one-GPU box (every rank on device 0,
    # collectives over gloo; graph capture of a gloo collective fails, so the decoder steps eagerly and says so).  Never for reported numbers.
    test_backend = os.environ.get("BENCH_TEST_BACKEND")
    dev = torch.device("cuda", 0 if os.environ.get("BENCH_TEST_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        init_tensor_parallel(backend=test_backend or "nccl", device=None if test_backend else dev)
    else:
        init_tensor_parallel()
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank):
        raise SystemExit(f"rank {rank}: injected failure (BENCH_TEST_FAIL_RANK)")       # a rank's failure must end the job non-zero
    batches = [int(b) for b in args.batches.split(",")]
    results = measure(args.model, batches, args.steps, args.context, dev, verbose=(rank == 0), layers=args.layers)
    cfg = {"7b": LlamaConfig.llama2_7b, "70b": LlamaConfig.llama2_70b}.get(args.model, lambda: LlamaConfig(
        hidden_size=512, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=8, vocab_size=2048))()

    cpu = decode_cpu_baseline(cfg, args.cpu_seconds) if (args.cpu_seconds > 0 and rank == 0) else None
    if rank == 0:
        print(json.dumps({"summary": {f"b{r['batch']}": r["value"] for r in results}, "unit": "tok/s", "cpu_baseline": cpu}), flush=True)


if __name__ == "__main__":
    main()
