"""RCCL sanity on a one-GPU box: a world-size-1 "nccl" (= RCCL) process group, an in-place all-reduce on the compute stream eagerly and
inside a HIP-graph capture (what bench.py / GraphedDecoder do at N > 1).  Not a scaling measurement — it shows that librccl loads, the
communicator initialises with HSA_ENABLE_IPC_MODE_LEGACY=0, and a captured all-reduce replays."""
import os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 100), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(8192, dtype=torch.float16, device="cuda") / 64
want = x.clone()
dist.all_reduce(x)
torch.cuda.synchronize()
assert torch.equal(x, want)
print("eager all_reduce ok; backend", dist.get_backend(), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        y = x * 2
        dist.all_reduce(y)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want * 2)
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print(f"captured all_reduce ok; {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per replay (mul + all_reduce, world 1)", flush=True)
except Exception as e:
    print("capture of the all_reduce failed:", repr(e), flush=True)
    sys.exit(1)
# the package's own path: TensorParallelGroup over the default group
from sglang_awq_amd import distributed as tpd
tp = tpd.TensorParallelGroup(dist.group.WORLD, 0, 1)
z = tp.all_reduce(x.clone())
assert torch.equal(z, x)
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
