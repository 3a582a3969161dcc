"""Tensor-parallel path on CPU (no GPU): packed-weight sharding against the oracle's restatement of
parameter.py, the legality checks of create_weights, and a world_size-2 gloo run of the
column-parallel -> row-parallel (+ all-reduce) wiring with the oracle standing in for the HIP ops."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import awq_ref, torch_cpu
from sglang_awq_amd import synth
from sglang_awq_amd.awq import AWQConfig, AWQLinearMethod
from sglang_awq_amd.linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                                   ReplicatedLinear, RowParallelLinear)


class OracleAWQLinearMethod(AWQLinearMethod):
    """Same parameters and checks as the product method; apply() runs the CPU oracle so the TP wiring
    can be exercised without a GPU (test-only: the product method has no CPU path)."""

    def apply(self, layer, x, bias=None):
        return torch_cpu.linear_cpu(x, layer.qweight.data, layer.scales.data, layer.qzeros.data, bias)


class OracleAWQConfig(AWQConfig):
    def get_quant_method(self, layer, prefix):
        return OracleAWQLinearMethod(self)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a).copy())


def _load(layer, qw, s, qz, shard_id=None):
    for name, full in (("qweight", qw), ("scales", s), ("qzeros", qz)):
        p = getattr(layer, name)
        if shard_id is None:
            p.weight_loader(p, _t(full))
        else:
            p.weight_loader(p, _t(full), shard_id)


CFG = AWQConfig(weight_bits=4, group_size=128, zero_point=True)


@pytest.mark.parametrize("tp", [1, 2, 4, 8])
def test_column_and_row_shards_match_oracle(tp):
    K, N, g = 1024, 2048, 128
    qw, s, qz = synth.make_awq_weights(K, N, g, "f16", "A", 5)
    for r in range(tp):
        col = ColumnParallelLinear(K, N, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=r, tp_size=tp)
        _load(col, qw, s, qz)
        wq, ws, wz = awq_ref.shard_column_parallel(qw, s, qz, r, tp)
        assert np.array_equal(col.qweight.numpy(), wq) and np.array_equal(col.scales.numpy(), ws) and np.array_equal(col.qzeros.numpy(), wz)
        row = RowParallelLinear(K, N, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=r, tp_size=tp)
        _load(row, qw, s, qz)
        wq, ws, wz = awq_ref.shard_row_parallel(qw, s, qz, r, tp)
        assert np.array_equal(row.qweight.numpy(), wq) and np.array_equal(row.scales.numpy(), ws) and np.array_equal(row.qzeros.numpy(), wz)


def test_merged_column_gate_up_shards():
    """gate_proj and up_proj are loaded separately into one fused parameter; each rank gets its slice
    of BOTH (linear.py:687-776); packed tensors index in units of 8 columns (parameter.py:539-550)."""
    K, inter, tp = 256, 1024, 4
    gq, gs, gz = synth.make_awq_weights(K, inter, 128, "f16", "A", 1)
    uq, us, uz = synth.make_awq_weights(K, inter, 128, "f16", "A", 2)
    for r in range(tp):
        layer = MergedColumnParallelLinear(K, [inter, inter], bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=r, tp_size=tp)
        _load(layer, gq, gs, gz, 0)
        _load(layer, uq, us, uz, 1)
        n_r = inter // tp
        want_q = np.concatenate([gq[:, r * n_r // 8:(r + 1) * n_r // 8], uq[:, r * n_r // 8:(r + 1) * n_r // 8]], axis=1)
        want_s = np.concatenate([gs[:, r * n_r:(r + 1) * n_r], us[:, r * n_r:(r + 1) * n_r]], axis=1)
        assert np.array_equal(layer.qweight.numpy(), want_q) and np.array_equal(layer.scales.numpy(), want_s)
        # a checkpoint that stores the fused tensor loads to the same thing
        fused = MergedColumnParallelLinear(K, [inter, inter], bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=r, tp_size=tp)
        _load(fused, np.concatenate([gq, uq], 1), np.concatenate([gs, us], 1), np.concatenate([gz, uz], 1))
        assert torch.equal(fused.qweight, layer.qweight) and torch.equal(fused.scales, layer.scales) and torch.equal(fused.qzeros, layer.qzeros)


def test_qkv_shards_with_kv_head_replication():
    """Llama-2-70B-like head layout at TP=8: 64 q heads, 8 kv heads, head 128 -> 8 q heads + 1 kv head per
    rank (qkv 8192 -> 1280 per rank); with 4 kv heads, two ranks share each kv head."""
    hidden, head = 256, 16
    for (nq, nkv, tp) in [(64, 8, 8), (32, 4, 8), (16, 16, 4)]:
        q = synth.make_awq_weights(hidden, nq * head, 128, "f16", "A", 11)
        k = synth.make_awq_weights(hidden, nkv * head, 128, "f16", "A", 12)
        v = synth.make_awq_weights(hidden, nkv * head, 128, "f16", "A", 13)
        for r in range(tp):
            layer = QKVParallelLinear(hidden, head, nq, nkv, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=r, tp_size=tp)
            _load(layer, *q, "q"); _load(layer, *k, "k"); _load(layer, *v, "v")
            qh = nq // tp
            kvh = max(nkv // tp, 1)
            kv_rank = r // layer.num_kv_head_replicas
            want_s = np.concatenate([q[1][:, r * qh * head:(r + 1) * qh * head],
                                     k[1][:, kv_rank * kvh * head:(kv_rank + 1) * kvh * head],
                                     v[1][:, kv_rank * kvh * head:(kv_rank + 1) * kvh * head]], axis=1)
            assert layer.scales.shape[1] == (qh + 2 * kvh) * head
            assert np.array_equal(layer.scales.numpy(), want_s)
            want_q = np.concatenate([q[0][:, r * qh * head // 8:(r + 1) * qh * head // 8],
                                     k[0][:, kv_rank * kvh * head // 8:(kv_rank + 1) * kvh * head // 8],
                                     v[0][:, kv_rank * kvh * head // 8:(kv_rank + 1) * kvh * head // 8]], axis=1)
            assert np.array_equal(layer.qweight.numpy(), want_q)


def test_row_parallel_legality_llama7b_down_proj():
    """F7: K = 11008 with g = 128 only shards 1 or 2 ways (awq.py:372-377)."""
    for tp in (1, 2):
        RowParallelLinear(11008, 4096, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=0, tp_size=tp)
    for tp in (4, 8):
        with pytest.raises(ValueError, match="too large tensor parallel size"):
            RowParallelLinear(11008, 4096, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=0, tp_size=tp)
    with pytest.raises(ValueError, match="too large tensor parallel size"):
        ColumnParallelLinear(4096, 8 * 4 + 4, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=0, tp_size=1)
    # Llama-2-70B per-rank shapes at TP = 8
    down = RowParallelLinear(28672, 8192, bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=7, tp_size=8)
    assert down.qweight.shape == (3584, 1024) and down.qzeros.shape == (28, 1024)
    gate_up = MergedColumnParallelLinear(8192, [28672, 28672], bias=False, quant_config=CFG, params_dtype=torch.float16, tp_rank=0, tp_size=8)
    assert gate_up.qweight.shape == (8192, 7168 // 8)


def test_config_parsing_and_skipped_modules():
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": 64, "zero_point": True, "modules_to_not_convert": ["lm_head"]})
    assert (cfg.weight_bits, cfg.group_size, cfg.pack_factor, cfg.get_name()) == (4, 64, 8, "awq")
    cfg2 = AWQConfig.from_config({"bits": 4, "group_size": 128, "zero_point": False})
    assert cfg2.group_size == 128 and cfg2.modules_to_not_convert == []
    with pytest.raises(ValueError):
        AWQConfig(weight_bits=8, group_size=128, zero_point=True)
    with pytest.raises(ValueError):
        AWQConfig.from_config({"bits": 4})
    head = ReplicatedLinear(64, 32, bias=False, quant_config=cfg, params_dtype=torch.float16, prefix="lm_head")
    assert hasattr(head, "weight") and not hasattr(head, "qweight")
    body = ReplicatedLinear(64, 32, bias=False, quant_config=cfg, params_dtype=torch.float16, prefix="model.layers.0.mlp.down_proj")
    assert body.qweight.shape == (64, 4) and body.qzeros.shape == (1, 4) and body.scales.shape == (1, 32)


def test_oracle_row_parallel_reference_matches_unsharded():
    qw, s, qz = synth.make_awq_weights(512, 256, 128, "f16", "A", 3)
    x = synth.make_activations(3, 512, "f16", "A", 3, x_std=0.05)
    b = synth.make_bias(256, "f16", 3)
    full = awq_ref.to_f64(awq_ref.awq_linear_apply(x, qw, s, qz, b), "f16")
    for tp in (2, 4):
        got = awq_ref.row_parallel_reference(x, qw, s, qz, tp, b)
        assert np.abs(got - full).max() < 4e-3     # per-rank fp16 partials: a few output ulps


# ------------------------------------------------------------------------------------------ gloo, world_size 2
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from sglang_awq_amd import distributed as tpd

    tp = tpd.init_tensor_parallel(backend="gloo")
    assert tp.world_size == world and tp.rank == rank
    cfg = OracleAWQConfig(weight_bits=4, group_size=128, zero_point=True)
    hidden, inter, M = 256, 512, 5
    gq, gs, gz = synth.make_awq_weights(hidden, inter, 128, "f16", "A", 1)
    uq, us, uz = synth.make_awq_weights(hidden, inter, 128, "f16", "A", 2)
    dq, ds, dz = synth.make_awq_weights(inter, hidden, 128, "f16", "A", 3)
    bias = _t(synth.make_bias(hidden, "f16", 4))
    x = _t(synth.make_activations(M, hidden, "f16", "A", 5, x_std=0.05))

    gate_up = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, quant_config=cfg, params_dtype=torch.float16)
    down = RowParallelLinear(inter, hidden, bias=True, quant_config=cfg, params_dtype=torch.float16)
    assert gate_up.tp_size == world and down.tp_rank == rank          # picked up from the TP group
    _load(gate_up, gq, gs, gz, 0); _load(gate_up, uq, us, uz, 1); _load(down, dq, ds, dz)
    down.bias.data.copy_(bias)
    for layer in (gate_up, down):
        layer.process_weights_after_loading()

    h, _ = gate_up(x)                                   # column-parallel: no communication
    n_r = inter // world
    act = torch.nn.functional.silu(h[:, :n_r].float()).half() * h[:, n_r:]
    y, _ = down(act)                                    # row-parallel: bias on rank 0, then all-reduce
    # gather_output column-parallel layer
    colg = ColumnParallelLinear(hidden, inter, bias=False, gather_output=True, quant_config=cfg, params_dtype=torch.float16)
    _load(colg, gq, gs, gz)
    yg, _ = colg(x)
    # row-parallel that splits a replicated input itself
    row2 = RowParallelLinear(inter, hidden, bias=False, input_is_parallel=False, quant_config=cfg, params_dtype=torch.float16)
    _load(row2, dq, ds, dz)
    full_act = _t(synth.make_activations(M, inter, "f16", "A", 6, x_std=0.05))
    y2, _ = row2(full_act)
    torch.save({"y": y, "yg": yg, "y2": y2}, os.path.join(out_dir, f"rank{rank}.pt"))
    tp.barrier()
    tpd.destroy_tensor_parallel()


def test_tp2_gloo_column_row_allreduce(tmp_path):
    world = 2
    mp.spawn(_tp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]
    # every rank holds the same all-reduced / all-gathered result
    for key in ("y", "yg", "y2"):
        assert torch.equal(outs[0][key], outs[1][key]), key

    hidden, inter, M = 256, 512, 5
    gq, gs, gz = synth.make_awq_weights(hidden, inter, 128, "f16", "A", 1)
    uq, us, uz = synth.make_awq_weights(hidden, inter, 128, "f16", "A", 2)
    dq, ds, dz = synth.make_awq_weights(inter, hidden, 128, "f16", "A", 3)
    bias = synth.make_bias(hidden, "f16", 4)
    x = synth.make_activations(M, hidden, "f16", "A", 5, x_std=0.05)
    g = awq_ref.awq_linear_apply(x, gq, gs, gz)
    u = awq_ref.awq_linear_apply(x, uq, us, uz)
    act = (torch.nn.functional.silu(_t(g).float()).half() * _t(u)).numpy()
    want = awq_ref.to_f64(awq_ref.awq_linear_apply(act, dq, ds, dz, bias), "f16")
    got = outs[0]["y"].double().numpy()
    assert got.shape == (M, hidden)
    assert np.abs(got - want).max() < 5e-3          # two fp16 partial sums instead of one: a few output ulps
    # gather = concatenation of the ranks' column shards, in rank order (torch's CPU half matmul, the stand-in
    # here, may differ from the exact-sum oracle by an output ulp)
    yg = outs[0]["yg"].double().numpy()
    assert yg.shape == (M, inter)
    assert np.abs(yg - awq_ref.to_f64(g, "f16")).max() <= 2.0 ** -10 * np.abs(yg).max()
    full_act = synth.make_activations(M, inter, "f16", "A", 6, x_std=0.05)
    want2 = awq_ref.row_parallel_reference(full_act, dq, ds, dz, 2)
    assert np.abs(outs[0]["y2"].double().numpy() - want2).max() < 2e-3


def _bench_rehearsal(extra, world=2):
    """bench.py's N > 1 path on CPU: gloo collectives, kernels stubbed out (BENCH_TEST_CPU=1) — shapes, sharding, the
    all-reduce and the JSON contract, not numbers."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_TEST_BACKEND="gloo", BENCH_TEST_CPU="1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--sets", "2", "--cpu-seconds", "0"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_strong_scaling_rehearsal_gloo():
    out = _bench_rehearsal(["--scaling", "strong"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    # column N / tp = 5504, row K / tp = 2048, all-reduce payload [M, 11008] fp16
    assert out["config"]["per_rank_linears"] == ["col 4096x5504", "row 2048x11008"]
    assert out["config"]["collective"]["all_reduce_bytes"] == [11008 * 2]
    assert out["config"]["graph_replay"] is False and out["value"] > 0


def test_bench_70b_tp8_shapes_rehearsal_gloo():
    out = _bench_rehearsal(["--shapes", "70b-tp8"])
    assert out["config"]["per_rank_linears"] == ["col 8192x1280", "row 1024x8192", "col 8192x7168", "row 3584x8192"]
    assert out["config"]["collective"]["all_reduce_bytes"] == [8192 * 2, 8192 * 2]
    assert out["scaling"] == "weak"
