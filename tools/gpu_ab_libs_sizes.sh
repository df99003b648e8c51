#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of library builds (PLANEVERB_AMD_LIB, names under planeverb_amd/) on the other bench
# workloads of tools/collect_profiles.sh (2048^2 BigRoom, 512^2 / 1024^2 Shoebox batched, 8192^2).
#   tools/gpu_ab_libs_sizes.sh <rounds> <a.so> <b.so> ...
rounds=$1; shift
for i in $(seq 1 $rounds); do
 for so in "$@"; do
  for fl in "--grid 2048 --scene BigRoom.pv" "--grid 1024 --scene Shoebox.pv --inflight 2 --batch 8" "--grid 512 --scene Shoebox.pv --inflight 2 --batch 0" "--grid 8192 --steps 4" "--grid 3072"; do
   echo "$so $fl: $(PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so python bench.py --no-cpu-baseline $fl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.4e  ms/step %.3f  verified %s' % (d['value'], d['ms_per_step'], d.get('verified_runs')))
")"
  done
 done
done
