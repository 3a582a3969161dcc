"""Timing of an AWQ-MoE decode step on the expert-indirect GEMV (default: Mixtral-8x7B-like expert shapes, E = 8, K = 4096, I = 14336, top-2).

  python tools/time_moe.py [E K I top_k [tokens,...]]        e.g. 256 7168 2048 8  (DeepSeek-V3-like routed experts)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sglang_awq_amd import _lib
if os.environ.get("AWQ_LAB_LIB"):                  # tools only: time another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["AWQ_LAB_LIB"])
from sglang_awq_amd.awq import AWQConfig
from sglang_awq_amd.moe import AWQMoEMethod, select_experts

dev = torch.device("cuda:0")
E, K, I, top_k = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 4096, 14336, 2)
g = 128
TOKENS = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1, 2, 4, 8, 16, 64, 256]
print(f"E = {E}, K = {K}, I = {I}, top_k = {top_k}, g = {g}", flush=True)
if os.environ.get("MOE_SLOT_MAX_PAIRS"):            # A/B of the route threshold
    AWQMoEMethod.slot_route_max_pairs = classmethod(lambda cls, num_experts: int(os.environ["MOE_SLOT_MAX_PAIRS"]))
if os.environ.get("MOE_TILE_MIN_ROWS"):
    AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT = float(os.environ["MOE_TILE_MIN_ROWS"])
if os.environ.get("MOE_TILE_WIDE_ROWS"):
    AWQMoEMethod.TILE_ROUTE_WIDE_ROWS_PER_EXPERT = float(os.environ["MOE_TILE_WIDE_ROWS"])
m = AWQMoEMethod(AWQConfig(4, g, True))
layer = torch.nn.Module()
m.create_weights(layer, E, K, I, torch.float16)
layer.to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(0)
for n, p in layer.named_parameters():
    if p.dtype == torch.int32:
        p.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, p.shape, dtype=torch.int64, device=dev, generator=gen).to(torch.int32))
    else:
        p.data.copy_((0.005 + 0.015 * torch.rand(p.shape, device=dev, generator=gen)).half())
m.process_weights_after_loading(layer)
wbytes = (layer.w13_packed.shape[1] + layer.w2_packed.shape[1])
for T in TOKENS:
    x = torch.randn(T, K, device=dev, generator=gen).half() * 0.5
    tw, ti = select_experts(torch.randn(T, E, device=dev, generator=gen), top_k)
    m.apply(layer, x, tw, ti); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = m.apply(layer, x, tw, ti)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    slots = T * top_k
    route = ("one grid row per pair" if slots <= AWQMoEMethod.slot_route_max_pairs(E) else
             ("expert-sorted 64-row tiles" if slots < AWQMoEMethod.TILE_ROUTE_WIDE_ROWS_PER_EXPERT * E else "expert-sorted 128-row tiles")
             if slots >= AWQMoEMethod.TILE_ROUTE_MIN_ROWS_PER_EXPERT * E else "expert-sorted 16-row blocks")
    active = len(set(ti.view(-1).tolist()))
    print(f"T={T:3d} pairs={slots:3d} ({route}, {active} active experts): {us:8.1f} us per MoE layer  ({active * wbytes / us / 1e3:7.1f} GB/s if every "
          f"active expert were streamed once)", flush=True)
