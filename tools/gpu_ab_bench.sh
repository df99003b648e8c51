#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of bench.py argument sets (one line per invocation: value, ms per step, spread).
#   tools/gpu_ab_bench.sh <rounds> "<args A>" "<args B>" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for a in "$@"; do
    python bench.py --no-cpu-baseline --steps 8 --warmup 2 $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-40s value %.4e  ms/step %.3f  verified %s' % ('$a', d['value'], d['ms_per_step'], d.get('verified_runs')))
"
  done
done
