#!/bin/bash
# Runs ON THE GPU BOX: slab groups (S = 2, 4 on one device) created in many process orders -- the creation-time dry run of the hand-off
# (SlabGroup::probeHandoff: timed; a slow or timed-out one is answered by other streams) must leave no group in the slow mode.
# Prints ms per run for every group, the dry run's us per sweep and the re-deals.   tools/gpu_slab_orders.sh [processes=10]
n=${1:-10}
orders=("4096 2048" "2048 4096" "2048 2048 4096" "4096 4096 2048" "2048 4096 2048 4096")
for i in $(seq 1 $n); do
  o=${orders[$(( (i - 1) % ${#orders[@]} ))]}
  echo "== process $i: grids $o, S = 1, 2, 4"
  SLABS=1,2,4 PLANEVERB_AMD_QUEUE_PROBE=2 python tools/gpu_slabs.py $o 2>&1 | grep -v "stream of" | sed -e 's/; halo.*//' -e 's/\[planeverb_amd\] slab group 0x[0-9a-f]*: /   /'
done
