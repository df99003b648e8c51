#!/bin/bash
# Runs ON THE GPU BOX: the paired / lone launch time of two solvers' raw sweeps in N separate PROCESSES (tools/gpu_deal_probe.py with
# DEAL_ONCE=1), then the same with more hardware queues per process (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up).
for q in "" 8; do
  echo "== GPU_MAX_HW_QUEUES=${q:-default}"
  for i in 1 2 3 4 5 6 7 8; do if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi; DEAL_ONCE=1 python tools/gpu_deal_probe.py 4096 0.4 2>&1 | grep process; done
done
