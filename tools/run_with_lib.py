"""Run a script of this repository against another build of the HIP library (A/B timing only):
  python3 tools/run_with_lib.py sglang_awq_amd/lib_p0/libawq_hip.so bench_decode.py --batches 1,8 ..."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from sglang_awq_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(os.path.join(root, script), run_name="__main__")
