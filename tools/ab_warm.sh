#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# A/B: does the 20-step figure depend on how long the GPU was busy right before the timed region?
set -e
mkdir -p gpurun_out
for w in 5 200 4000; do
  for rep in 1 2; do
    python3 bench.py --gpus 1 --steps 20 --warmup $w --sections "" --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('warmup', $w, 'ms_per_step', d['ms_per_step'], 'us/launch', d['roofline']['us_per_launch'], 'frac', d['roofline']['frac'])"
  done
done
python3 bench.py --gpus 1 --steps 2000 --warmup 5 --sections "" --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('steps 2000', 'ms_per_step', d['ms_per_step'], 'us/launch', d['roofline']['us_per_launch'], 'frac', d['roofline']['frac'])"
