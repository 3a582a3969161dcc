"""RCCL on the one-GPU box: the "nccl" backend of torch.distributed IS RCCL on ROCm.  A world-size-1 group cannot measure scaling, but it does
execute librccl: communicator initialisation under HSA_ENABLE_IPC_MODE_LEGACY=0, an in-place all-reduce on the compute stream, and the same
all-reduce captured into a HIP graph and replayed — the three things bench.py / GraphedDecoder rely on at N > 1 (reference:
GroupCoordinator.all_reduce, python/sglang/srt/distributed/parallel_state.py:544-623; pynccl.py:144-165)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_eager_and_captured_all_reduce():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_check.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, NCCL_DEBUG="WARN"))
    if r.returncode != 0 and "Could not read node" in (r.stdout + r.stderr):
        # seen on some hosts of the pool (two of ~ten runs in round 3): librccl aborts inside its own topology discovery ("NCCL WARN Could not
        # read node # 9", SIGABRT) before any communicator exists — a property of that host's sysfs as seen from the one-GPU container, not of
        # this package; every other failure of the script still fails the test
        pytest.skip("librccl could not read this host's topology (NCCL WARN Could not read node ...): RCCL cannot initialise here")
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "backend nccl" in r.stdout and "captured all_reduce ok" in r.stdout
