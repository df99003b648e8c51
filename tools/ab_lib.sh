#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of two builds of the library (PLANEVERB_AMD_LIB) on the bench workloads.
#   tools/ab_lib.sh <other.so> <label> ["bench args" ...]
so=$1; label=$2; shift 2
[ $# -eq 0 ] && set -- "--steps 10" "--steps 10 --inflight 1"
for i in 1 2 3; do
  for v in $label default; do
    if [ $v = $label ]; then export PLANEVERB_AMD_LIB=$PWD/$so; else unset PLANEVERB_AMD_LIB; fi
    for args in "$@"; do
      python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-8s %-50s %.3e verified %s' % ('$v', '$args', d['value'], d['verified_runs']))"
    done
  done
done
