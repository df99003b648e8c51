#!/usr/bin/env python3
"""Wall time of a whole run (enqueue + device work + sync + queried outputs) on the presets and two launch-bound grids: one
line (for tools/gpu_ab_libs.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv
out = []
scene = os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv")
for res in (275, 375, 500, 750, 1000, 2009):
    with pv.Solver(25.0, 25.0, res, no_free_grid=1) as s:
        s.load_scene(scene)
        s.set_output_queries([(5.0, 0.0, 6.0)])
        for _ in range(3):
            s.run((5.0, 0.0, 4.0))
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                s.run((5.0, 0.0, 4.0))
                s.queried_outputs()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        out.append("%d^2 %.3f" % (s.gx, best))
print("  ".join(out))
