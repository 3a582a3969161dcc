"""Decode harness under tensor parallelism on ONE GPU: two ranks (both on cuda:0, gloo collectives — the N > 1 code path
with a CPU-side transport, as the N = 8 run itself belongs to the driver) load the same synthetic AutoAWQ checkpoint
through the TP-sharding weight loaders and must reproduce the TP = 1 logits: column-parallel qkv / gate_up with the
fused norm prologue and SiLU-mul epilogue on per-rank shards (gate and up interleaved per rank), row-parallel
o_proj / down_proj + all-reduce, KV heads split across ranks."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS, B = 3, 3


def _run_decode(model, first_tokens):
    tokens = first_tokens.clone()
    pos = torch.zeros(B, dtype=torch.int64, device=DEV)
    outs = []
    with torch.no_grad():
        for _ in range(STEPS):
            lg = model.logits(tokens, pos).float()
            outs.append(lg.cpu())
            tokens = lg.argmax(-1)
            pos = pos + 1
    return torch.stack(outs)


def _worker(rank, world, port, ckpt, ref_path, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch as th

    from sglang_awq_amd import distributed as tpd
    from sglang_awq_amd.loader import load_llama_awq

    th.cuda.set_device(0)
    tpd.init_tensor_parallel(backend="gloo")
    try:
        model = load_llama_awq(ckpt, device=th.device(DEV), max_batch=4, max_seq=32)
        ref = th.load(ref_path, weights_only=True)
        got = _run_decode(model, ref["first"].to(DEV))
        scale = ref["logits"].abs().max().item()
        err = (got - ref["logits"]).abs().max().item()
        layer = model.layers[0]
        fused = getattr(layer.qkv_proj, "awq_packed", None) is not None and layer._gate_up_interleaved() is not None
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(f"{err} {scale} {int(fused)} {layer.num_heads} {layer.num_kv_heads}")
    finally:
        tpd.destroy_tensor_parallel()


def test_tiny_llama_tp2_matches_tp1(tmp_path):
    import torch.multiprocessing as mp

    from sglang_awq_amd.loader import load_llama_awq
    from tests.test_loader_cpu import _write_checkpoint

    torch.manual_seed(11)
    ckpt = str(tmp_path / "ckpt")
    _write_checkpoint(ckpt)
    model = load_llama_awq(ckpt, device=torch.device(DEV), max_batch=4, max_seq=32)
    first = torch.tensor([3, 77, 101], device=DEV)
    ref = _run_decode(model, first)
    ref_path = str(tmp_path / "ref.pt")
    torch.save({"first": first.cpu(), "logits": ref}, ref_path)
    del model

    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, ckpt, ref_path, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        err, scale, fused, heads, kv = open(tmp_path / f"rank{rank}.txt").read().split()
        assert int(fused) == 1, "the TP ranks did not take the fused (repacked) path"
        assert int(heads) == 2 and int(kv) == 1
        # fp16 partial sums are rounded before the all-reduce: allow a few output ulps of the logit scale
        assert float(err) <= 2e-2 * float(scale) + 2e-2, f"rank {rank}: TP=2 logits off by {err} (scale {scale})"


def _torchrun_bench_decode(extra_env, port):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_TEST_BACKEND="gloo", BENCH_TEST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench_decode.py"), "--model", "70b", "--layers", "2", "--batches", "1,8",
           "--steps", "4", "--context", "64", "--cpu-seconds", "0"]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)


def test_bench_decode_70b_two_ranks_rehearsal_and_failure_exit_code():
    """BASELINE configs[4] kept from rotting without an 8-GPU node: `bench_decode.py --model 70b` under torchrun at world 2 (both
    ranks on this GPU, gloo collectives, 2 of the 80 layers): per-rank shards of the 70B linears (KV heads split 8 -> 4 per rank),
    row-parallel all-reduces, the JSON contract; and a rank that dies must end the job with a non-zero exit code
    (reference: RowParallelLinear.forward + all-reduce, python/sglang/srt/layers/linear.py:1388-1414)."""
    import json

    port = 29700 + os.getpid() % 200
    r = _torchrun_bench_decode({}, port)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    recs = [l for l in lines if "batch" in l]
    assert [x["batch"] for x in recs] == [1, 8] and all(x["value"] > 0 and x["metric"].endswith("TP=2") for x in recs)
    assert all(x["graph_replay"] is False and x["graph_capture_error"] for x in recs)      # gloo cannot be captured: said, not hidden
    assert any("summary" in l for l in lines)
    r = _torchrun_bench_decode({"BENCH_TEST_FAIL_RANK": "1"}, port + 1)
    assert r.returncode != 0, "a failed rank must end the job with a non-zero exit code"
