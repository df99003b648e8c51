#!/usr/bin/env python3
"""The raw stencil on seeded RANDOM fields against ZERO fields (bench.py's roofline.dense leg) for several tile configurations
and numbers of runs in flight: rate, launch p50 / p90, shader clock, socket power (profiles/r05_dense.txt).

    python tools/gpu_dense.py [grid=4096] [inflight=2] [K,rows ...]      e.g.  tools/gpu_dense.py 4096 2 12,36 10,36 8,40"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(0, 0)]
hooks = bench.GpuHooks(0)
size = bench.mode_a_size(grid)
res = 275
if os.environ.get("MODEB"):  # the 25 m scene at the resolution that gives this grid (Mode B's geometry: thick walls, 4 % general tiles)
    size, res = 25.0, {512: 2009, 1024: 4017, 2048: 8034, 4096: 16067, 8192: 32134}[grid]
print("# %d^2, HugeRoom.pv, %d run(s) in flight; raw stencil (PvAmdRunSteps), ~1.2 s per leg" % (grid, inflight))
print("# K rows | random: upd/s  launch p50/p90 ms  clock MHz (median/min)  W | zero: upd/s  p50/p90  clock  W | random/zero")
for K, rows in cfgs:
    opts = dict(steps_per_launch=K, tile_rows=rows) if K else {}
    if os.environ.get("MODEB"):
        opts["streaming_analysis"] = 1  # (no T x cells history at these T; the leg itself is raw stepping)
    solvers = [hooks.make_solver(size, res, **opts) for _ in range(inflight)]
    for s in solvers:
        s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    s = solvers[0]
    d = hooks.dense_leg(solvers, s.info.stepsPerLaunch, s.T, (s.gx + 1) * (s.gy + 1))
    if "skipped" in d:
        print("%2d %3d | skipped: %s" % (s.info.stepsPerLaunch, s.info.tileRows, d["skipped"]))
    else:
        r, z = d["random"], d["zero"]
        f = lambda x: "%.3e  %.4f/%.4f  %5.0f/%5.0f  %s" % (x["value"], x["launch_ms_p50"], x["launch_ms_p90"], x["clock_mhz_median"] or 0,
                                                            x["clock_mhz_min"] or 0, "%.0f" % x["power_w_median"] if x["power_w_median"] else "-")
        print("%2d %3d | %s | %s | %.3f" % (s.info.stepsPerLaunch, s.info.tileRows, f(r), f(z), d["random_over_zero"]), flush=True)
    for s in solvers:
        s.close()
