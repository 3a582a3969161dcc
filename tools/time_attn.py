"""Timing of the decode attention launch (RoPE + KV write + one-token attention) at 7B dimensions (32 heads, D = 128), HBM-cold:
a hipGraph of LAYERS calls, each on its own KV cache (LAYERS x batch x 16.8 MB at context 1024 does not fit the 256 MB MALL at any batch).

  python tools/time_attn.py [context=1024] [batches=1,2,4,8,32] [splits=1,2,4,8,16]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sglang_awq_amd import _lib, aux_ops
if os.environ.get("AWQ_LAB_LIB"):                  # tools only: time the laboratory build (make -C sglang_awq_amd/csrc lab)
    _lib.LIB_PATH = os.path.abspath(os.environ["AWQ_LAB_LIB"])

dev = torch.device("cuda:0")
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batches = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,32").split(",")]
splits_l = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,16").split(",")]
Hq = Hkv = 32
D = 128
S = ctx + 64
gen = torch.Generator(device=dev); gen.manual_seed(0)
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D))
fr = torch.outer(torch.arange(S, device=dev).float(), inv)
cos_t, sin_t = fr.cos().contiguous(), fr.sin().contiguous()
for B in batches:
    layers = max(4, min(32, int(600e6 // (B * Hkv * S * D * 4))))
    kcs = [torch.randn(B, Hkv, S, D, device=dev, generator=gen).half() for _ in range(layers)]
    vcs = [torch.randn(B, Hkv, S, D, device=dev, generator=gen).half() for _ in range(layers)]
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=dev, generator=gen).half()
    pos = torch.full((B,), ctx, dtype=torch.int64, device=dev)
    kv_bytes = B * Hkv * (ctx + 1) * D * 2 * 2
    row = []
    for ns in splits_l:
        if ns > 1 and B * Hq > 16384:
            continue
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            if ns > 1:
                aux_ops.prepare_attention_workspace(dev, st.cuda_stream, aux_ops._lib.load().awq_aux_decode_attention_workspace_bytes(B, Hq, D, ns))
            outs = [aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kcs[i], vcs[i], Hq, Hkv, D, num_splits=ns) for i in range(layers)]
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                outs = [aux_ops.decode_attention(qkv, pos, cos_t, sin_t, kcs[i], vcs[i], Hq, Hkv, D, num_splits=ns) for i in range(layers)]
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            e0.record(st)
            for _ in range(reps):
                gr.replay()
            e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * layers)
        row.append(f"splits {ns:2d}: {us:7.2f} us ({kv_bytes / us / 1e6:5.2f} TB/s)")
        if ns == splits_l[0]:
            ref = outs[0].float().clone()
        else:
            d = (outs[0].float() - ref).abs().max().item()
            row[-1] += f" d={d:.1e}"
    print(f"context {ctx} batch {B:3d} ({layers} caches, {kv_bytes / 1e6:.1f} MB of K+V per call): " + " | ".join(row), flush=True)
    del kcs, vcs
