#!/bin/bash
# Runs ON THE GPU BOX: shader clock (tools/clock_probe.hip) idle and beside the bench workloads.
hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe 2>/dev/null
echo idle:; /tmp/clock_probe 3
python bench.py --no-cpu-baseline --steps 200 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 14
echo "beside bench.py (2 runs in flight, 4096^2):"; /tmp/clock_probe 8
wait $BP
tail -1 /tmp/b.json | cut -c1-100
python bench.py --no-cpu-baseline --steps 100 --grid 8192 --open-field > /tmp/b.json 2>/dev/null &
BP=$!
sleep 16
echo "beside bench.py --grid 8192 --open-field:"; /tmp/clock_probe 5
wait $BP
