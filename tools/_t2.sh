#!/bin/bash
# tile-order sweep under 1 and 2 runs in flight
for g in 4096 8192; do
for inf in 2 1; do
for o in 1 2 4 6 8 12; do
  python bench.py --no-cpu-baseline --grid $g --inflight $inf --tile-order $o --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('grid $g inflight $inf order $o value %.4g fdtd_ms %.3f launch_ms %.4f' % (d['value'], d['fdtd_ms'], d['roofline']['launch_ms']))"
done; done; done
