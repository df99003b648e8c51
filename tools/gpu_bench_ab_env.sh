#!/bin/bash
# Runs ON THE GPU BOX: bench.py workloads with an environment switch off / on, alternating (value, ms per step, fdtd ms, analysis ms).
#   tools/gpu_bench_ab_env.sh <rounds> <ENV=VALUE> "<bench args>" ["<bench args>" ...]
rounds=$1; kv=$2; shift 2
for args in "$@"; do
  for i in $(seq 1 $rounds); do
    for on in 0 1; do
      if [ $on = 1 ]; then export "$kv"; else unset "${kv%%=*}"; fi
      echo "[$args] ${kv} $( [ $on = 1 ] && echo set || echo unset ): $(python bench.py --no-cpu-baseline --no-dense-leg $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.4e  ms/step %.3f  fdtd %.3f  analysis %.3f  reached %.0f' % (d['value'], d['ms_per_step'], d['fdtd_ms'], d['analysis_ms'], d['roofline']['analysis']['reached_cells_per_run']))
")"
    done
  done
done
