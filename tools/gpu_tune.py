#!/usr/bin/env python3
"""A/B of kernel configurations on the bench workload (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv

def run(n, reps=4, **opts):
    dx = 343.21 / 275 / 3.5
    size = (n + 0.5) * dx
    s = pv.Solver(size, size, 275, **opts)
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    L = (5, 0, 4)
    s.run(L)
    f, a = [], []
    for _ in range(reps):
        s.run(L)
        t = s.timings()
        f.append(t.fdtdMs); a.append(t.analysisMs)
    cells = (s.gx + 1) * (s.gy + 1)
    t = s.timings()
    print("n=%d %s: fdtd min %.2f med %.2f ms (%.3e upd/s) analysis %.2f  air %.1f us gen %.1f us" % (
        n, opts, min(f), float(np.median(f)), cells * s.T / (min(f) * 1e-3), min(a), t.airKernelMs * 1e3, t.generalKernelMs * 1e3))
    s.close()

if __name__ == "__main__":
    # usage: gpu_tune.py N K,rows[,tile_order] ...
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfgs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[2:]] or [(4, 32), (8, 24), (6, 28)]
    for rep in range(2):
        for c in cfgs:
            opts = dict(steps_per_launch=c[0], tile_rows=c[1])
            if len(c) > 2:
                opts["tile_order"] = c[2]
            try:
                run(n, **opts)
            except pv.PlaneverbError as e:
                print("%s: %s" % (opts, e))
