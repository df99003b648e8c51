#!/bin/bash
# Runs ON THE GPU BOX: lone / paired launch time of two solvers' raw sweeps, both solvers created anew six times per process
# (tools/gpu_deal_probe.py DEAL_BOTH), for several placements of a solver's two buffer sets (PLANEVERB_AMD_SET_SKEW)
for sk in "" 0 4096 69632 1052672 2097152 3145728; do
  echo "== PLANEVERB_AMD_SET_SKEW=${sk:-unset (two allocations)}"
  if [ -n "$sk" ]; then export PLANEVERB_AMD_SET_SKEW=$sk; else unset PLANEVERB_AMD_SET_SKEW; fi
  DEAL_BOTH=6 python tools/gpu_deal_probe.py 4096 0.25 2>&1 | grep pair
done
