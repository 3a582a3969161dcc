"""AWQ-MoE host logic on CPU: the device-side block alignment (no host synchronisation by construction: pure tensor ops with
static shapes) against a plain-Python restatement of the reference's moe_align_block_size contract, and the runner-config checks of
create_moe_runner (reference awq.py:815-820)."""
import types

import pytest
import torch

from sglang_awq_amd.awq import AWQConfig
from sglang_awq_amd.moe import AWQMoEMethod, StandardCombineInput, select_experts


@pytest.mark.parametrize("P,E", [(1, 8), (6, 8), (40, 8), (600, 8), (33, 4), (16, 3), (5, 1), (64, 60)])
def test_align_blocks_contract(P, E):
    """Every valid pair appears exactly once, in a block of its own expert, in ascending pair order inside an expert (stable);
    padding rows and unused blocks are -1; ids outside [0, E) (the reference pads graph batches with -1, topk.py:705-712) vanish;
    the number of blocks is the static bound ceil(P / 16) + E."""
    g = torch.Generator().manual_seed(P * 131 + E)
    ids = torch.randint(-1, E, (P,), dtype=torch.int32, generator=g)
    if P > 3:
        ids[2] = E + 5
    row_map, block_expert = AWQMoEMethod.align_blocks(ids, E)
    B = block_expert.numel()
    assert B == (P + 15) // 16 + E and row_map.numel() == B * 16 and row_map.dtype == torch.int32 and block_expert.dtype == torch.int32
    seen = []
    for b in range(B):
        e = int(block_expert[b])
        rows = row_map[b * 16:(b + 1) * 16].tolist()
        if e < 0:
            assert all(r == -1 for r in rows)
            continue
        real = [r for r in rows if r >= 0]
        assert real and all(int(ids[r]) == e for r in real)
        assert rows[:len(real)] == real                      # padding only at the end of a block
        seen += real
    assert sorted(seen) == [i for i in range(P) if 0 <= int(ids[i]) < E] and len(set(seen)) == len(seen)
    for e in range(E):
        rows = [r for b in range(B) if int(block_expert[b]) == e for r in row_map[b * 16:(b + 1) * 16].tolist() if r >= 0]
        assert rows == sorted(rows)
    es = [int(v) for v in block_expert.tolist() if v >= 0]
    assert es == sorted(es)                                   # experts in ascending order: one contiguous run of blocks each


def test_create_moe_runner_checks_and_combine_input():
    m = AWQMoEMethod(AWQConfig(4, 128, True))
    layer = torch.nn.Module()
    m.create_moe_runner(layer, types.SimpleNamespace(activation="silu", is_gated=True, top_k=2))
    assert m.moe_runner_config.top_k == 2
    for bad in (dict(activation="gelu"), dict(activation="silu", is_gated=False), dict(activation="silu", apply_router_weight_on_input=True),
                dict(activation="silu", no_combine=True)):
        with pytest.raises(NotImplementedError):
            m.create_moe_runner(layer, types.SimpleNamespace(**bad))
    out = StandardCombineInput(hidden_states=torch.zeros(2, 3))
    assert out.format == "standard" and out.hidden_states.shape == (2, 3) and out[0] is out.hidden_states
    w, i = select_experts(torch.randn(5, 8), 2)
    assert i.dtype == torch.int32 and torch.allclose(w.sum(-1), torch.ones(5))


def test_create_weights_shapes_match_the_reference():
    """awq.py:669-757: w13_qweight [E, K, 2I/8], w2_qweight [E, I, K/8], scales [E, K/g, 2I] / [E, I/g, K], qzeros packed likewise."""
    m = AWQMoEMethod(AWQConfig(4, 128, True))
    layer = torch.nn.Module()
    m.create_weights(layer, 4, 256, 512, torch.float16, weight_loader=lambda *a: None)
    assert layer.w13_qweight.shape == (4, 256, 128) and layer.w2_qweight.shape == (4, 512, 32)
    assert layer.w13_scales.shape == (4, 2, 1024) and layer.w2_scales.shape == (4, 4, 256)
    assert layer.w13_qzeros.shape == (4, 2, 128) and layer.w2_qzeros.shape == (4, 4, 32)
    assert callable(layer.w13_qweight.weight_loader)
    with pytest.raises(ValueError):
        AWQMoEMethod(AWQConfig(4, 128, True)).create_weights(torch.nn.Module(), 2, 200, 512, torch.float16)


def test_slot_route_threshold_grows_with_expert_count():
    """Few experts: pairs beyond 12 share experts, grouping them pays (Mixtral-like); many experts: up to E / 2 pairs mostly hit
    different experts and stay on the one-row-per-pair route (DeepSeek-V3-like, measured in profiles/r03_time_moe_deepseek_v3_shapes.txt)."""
    from sglang_awq_amd.moe import AWQMoEMethod

    assert AWQMoEMethod.slot_route_max_pairs(8) == 12
    assert AWQMoEMethod.slot_route_max_pairs(64) == 32
    assert AWQMoEMethod.slot_route_max_pairs(256) == 128
