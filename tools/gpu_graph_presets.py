#!/usr/bin/env python3
"""Stencil time of the replayed tile-kernel graph (resident kernel off) on the presets and a few launch-bound grids: one line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
out = []
scene = os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv")
for res in (275, 750, 1500, 2009):
    with pv.Solver(25.0, 25.0, res, resident_kernel=2, no_free_grid=1) as s:
        s.load_scene(scene)
        ts = []
        for _ in range(6):
            s.run((5.0, 0.0, 4.0))
            ts.append(s.timings().fdtdMs)
        out.append("%d^2 %.3f" % (s.gx, min(ts[1:])))
print("  ".join(out))
