#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of library builds (PLANEVERB_AMD_LIB, names under planeverb_amd/) on bench.py's headline AND its
# roofline.dense leg (the raw stencil on seeded random fields / on zero fields: rate, launch ms, shader clock, socket power).
#   tools/gpu_ab_libs_dense.sh <rounds> <a.so> <b.so> ...        [BENCH_ARGS="--grid 4096"]
rounds=$1; shift
for i in $(seq 1 $rounds); do
 for so in "$@"; do
   echo "$so: $(PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so python bench.py --no-cpu-baseline --steps 8 --warmup 2 --repeats 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; dn = r.get('dense', {})
        f = lambda t: '%.4e (%.4f ms, %.0f MHz, %.0f W)' % (dn[t]['value'], dn[t]['launch_ms_p50'], dn[t]['clock_mhz_median'] or 0, dn[t]['power_w_median'] or 0) if t in dn else '-'
        print('headline %.4e launch %.4f ms single %.4f ms | dense random %s zero %s ratio %.3f | verified %s' % (
            d['value'], r['launch_ms'], (r.get('single_run') or {}).get('launch_ms', 0), f('random'), f('zero'), dn.get('random_over_zero', 0), d.get('verified_runs')))
")"
 done
done
