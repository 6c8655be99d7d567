#!/bin/bash
# Two (or $1) ranks sharing ONE GPU, rendezvous over gloo: the shared-palette step of bench.py with the partial sums exchanged
# (a) by torch.distributed per iteration, (b) through the library's loop over a host transport, (c) by the one-shot exchange over
# IPC-mapped mailboxes.  One GPU: the ranks' kernels share the CUs, so only the DIFFERENCE between the lines means anything.
N=${1:-2}
R=$(cd "$(dirname "$0")/.." && pwd)
export CNIIC_BENCH_BACKEND=gloo CNIIC_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for how in torch host mailbox; do
  CNIIC_COLLECTIVES=$how timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus $N --steps 5 --warmup 2 --cpu-sample 0 --no-extras 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('$how', 'ranks', d['n_gpus'], 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['config'].get('parallelism'), 'iterations', d['config'].get('iterations'))
" || exit 1
done
