#!/usr/bin/env python3
"""Workload for the SQ / GRBM counter passes of tools/pmc_r02.sh: the raw K-step stencil (PvAmdRunSteps: the dominant
kernel pv_step_merged_kernel and nothing else) on the bench grid, started from all-zero or from all-non-zero random
fields, one run or two runs in flight (two solvers, two host threads; under --pmc the profiler serialises dispatches,
so the counters of both modes describe single launches -- the kernel-trace durations are what shows the overlap).

    python tools/sq_workload.py --fields zero|random [--grid 4096] [--launches 6] [--inflight 1|2] [--scene HugeRoom.pv]
"""
import argparse
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fields", default="zero")
ap.add_argument("--grid", type=int, default=4096)
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--inflight", type=int, default=1)
ap.add_argument("--scene", default="HugeRoom.pv")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--tile-order", type=int, default=-1)
ap.add_argument("--k", type=int, default=0)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--alt", type=int, default=-1)  # PVA_OPT_ALTERNATE_SWEEPS
ap.add_argument("--regions", type=int, default=-1)  # PVA_OPT_XCD_REGIONS
ap.add_argument("--segments", type=int, default=0)  # PVA_OPT_STREAM_ROWS: row-streaming air segments (-1 = tile kernels)
a = ap.parse_args()

dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
size = float((a.grid + 0.5) * dx)
kw = dict(steps_per_launch=a.k, tile_rows=a.rows) if a.k else {}
if a.tile_order >= 0:
    kw["tile_order"] = a.tile_order
if a.alt >= 0:
    kw["alternate_sweeps"] = a.alt
if a.regions >= 0:
    kw["xcd_regions"] = a.regions
solvers = [pv.Solver(size, size, 275, no_free_grid=1, stream_rows=a.segments, **kw) for _ in range(a.inflight)]
rng = np.random.default_rng(1)
for s in solvers:
    if a.scene != "none":
        s.load_scene(os.path.join(ROOT, "tests", "scenes", a.scene))
    if a.fields == "random":
        shp = (s.gx + 1, s.gy + 1)
        s.set_fields(*[(rng.random(shp, np.float32) - np.float32(0.5)) * np.float32(1e-3) for _ in range(3)])
K = solvers[0].info.stepsPerLaunch


def work(s):
    for _ in range(a.reps):
        s.run_steps(K * a.launches)


import time  # noqa: E402
work(solvers[0])  # warm
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(s,)) for s in solvers]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
cells = (solvers[0].gx + 1) * (solvers[0].gy + 1)
print("fields=%s inflight=%d grid=%d K=%d: %.3e cell-updates/s (wall, %d launches x %d reps per solver)" % (
    a.fields, a.inflight, a.grid, K, a.inflight * a.reps * a.launches * K * cells / dt, a.launches, a.reps))
for s in solvers:
    s.close()
