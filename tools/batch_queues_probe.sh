#!/bin/bash
# the reference's batch semantics (64 frames 1080p, one palette each) over the number of hardware queues the HIP runtime may use
# (GPU_MAX_HW_QUEUES, read at runtime start: default 4) and the worker streams
R=$(cd "$(dirname "$0")/.." && pwd)
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python3 $R/tools/batch_probe.py 64 2>&1 | grep -E "^launches|one frame"
done
