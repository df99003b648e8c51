#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of library builds (PLANEVERB_AMD_LIB) on bench.py, two runs in flight and one.
#   tools/gpu_ab_libs_bench.sh <rounds> <a.so> <b.so> ...      (names under planeverb_amd/)
rounds=$1; shift
for i in $(seq 1 $rounds); do
 for so in "$@"; do
  for fl in "" "--inflight 1"; do
   echo "$so $fl: $(PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so python bench.py --no-cpu-baseline --steps 8 --warmup 2 $fl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.4e  ms/step %.3f  verified %s' % (d['value'], d['ms_per_step'], d.get('verified_runs')))
")"
  done
 done
done
