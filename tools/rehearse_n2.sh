#!/bin/bash
# N = 2 rehearsal of bench.py on a one-GPU box: both ranks on device 0, collectives over gloo, real kernels (no graph capture).
export BENCH_TEST_BACKEND=gloo BENCH_TEST_ONE_DEVICE=1
for extra in "" "--scaling strong" "--shapes 70b-tp8"; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 10 --warmup 2 --sets 4 --cpu-seconds 0 $extra 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$extra', '| n_gpus', j['n_gpus'], 'scaling', j['scaling'], 'value', j['value'], 'GB/s ms/step', j['ms_per_step'], j['config']['per_rank_linears'], j['config']['collective']['all_reduce_bytes'], 'graph', j['config']['graph_replay'])"
done
