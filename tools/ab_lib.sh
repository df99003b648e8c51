#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of two builds of the library (PLANEVERB_AMD_LIB) on the bench workloads.
#   tools/ab_lib.sh <other.so> <label>
for i in 1 2; do
  for v in $2 default; do
    if [ $v = $2 ]; then export PLANEVERB_AMD_LIB=$PWD/$1; else unset PLANEVERB_AMD_LIB; fi
    for args in "--steps 10" "--steps 10 --inflight 1" "--grid 2048 --scene BigRoom.pv --steps 20" "--grid 512 --scene Shoebox.pv --inflight 4 --steps 40" "--grid 512 --scene Shoebox.pv --inflight 2 --batch 0 --steps 10" "--grid 8192 --steps 3"; do
      python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-8s %-60s %.3e' % ('$v', '$args', d['value']))"
    done
  done
done
