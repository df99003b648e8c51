#!/usr/bin/env python3
"""Timing-only proxy for split tiling of the dominant kernel (docs/experiments/split_tiling.md): phase 1 of a 4096^2 sweep
= today's 60-row tiles laid side by side without their 24 halo rows = today's kernel on a grid of 4096 x 36/60 rows; the
phase-2 boundary strips are priced from the same kernel on 12-row tiles.  One run at a time and two in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv

dx = float(np.float32(343.21) / np.float32(275) / np.float32(3.5))
def run(nx, ny, inflight, **opts):
    ss = [pv.Solver((nx + 0.5) * dx, (ny + 0.5) * dx, 275, no_free_grid=1, skip_analysis=1, **opts) for _ in range(inflight)]
    L = (nx * dx / 2, 0.0, ny * dx / 2)
    for s in ss:
        s.run(L)
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        for s in ss:
            s.run_async(L)
        for s in ss:
            s.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    k = ss[0].info.stepsPerLaunch
    T = ss[0].T
    for s in ss:
        s.close()
    return ms, ms / inflight / ((T + k - 1) // k) * 1e3
for inflight in (1, 2):
    a = run(4096, 4096, inflight)
    b = run(2458, 4096, inflight, steps_per_launch=12, tile_rows=36)
    print("inflight %d: 4096 x 4096 %.2f ms per %d run(s) = %.1f us per sweep and run | 2458 x 4096 (phase-1 proxy) %.2f ms = %.1f us per sweep (%.2f x)" % (inflight, a[0], inflight, a[1], b[0], b[1], b[0] / a[0]), flush=True)
