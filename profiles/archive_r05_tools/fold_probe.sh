#!/bin/bash
# Two ranks sharing ONE GPU with the assign kernel's grid capped so that both ranks' blocks are resident side by side (CNIIC_KM_MAX_BLOCKS):
# the one-shot exchange as a kernel of its own against the exchange folded into the launches (CNIIC_MB_FOLD=1, testing build).  At full-size
# grids the folded exchange cannot run on a shared GPU (NOTES D); capped, only the DIFFERENCE between the two lines means anything.
N=${1:-2}; CAP=${2:-384}
R=$(cd "$(dirname "$0")/.." && pwd)
export CNIIC_BENCH_BACKEND=gloo CNIIC_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 CNIIC_USE_TESTING_LIB=1 CNIIC_COLLECTIVES=mailbox CNIIC_COLLECTIVE_TIMEOUT_MS=5000 CNIIC_KM_MAX_BLOCKS=$CAP
for fold in 0 1; do
  CNIIC_MB_FOLD=$fold timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus $N --steps 5 --warmup 2 --cpu-sample 0 --no-extras 2>/tmp/fold_err_$fold.txt | python3 -c "
import json, sys
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('fold $fold', 'ranks', d['n_gpus'], 'cap $CAP', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'iterations', d['config'].get('kmeans_iterations'), d['config'].get('parallelism')[-60:])
"
  tail -n 2 /tmp/fold_err_$fold.txt | cut -c1-200
done
