#!/usr/bin/env python3
"""Analysis time of a run (HIP events around the chain: far frame + encode + wet gain / decay time + direction) on the presets
and on BASELINE config 4's grid: one line (for tools/gpu_ab_libs.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv
out = []
scene = os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv")
for res in (275, 500, 750, 1000, 1500):
    with pv.Solver(25.0, 25.0, res, no_free_grid=1) as s:
        s.load_scene(scene)
        ts = []
        for _ in range(6):
            s.run((5.0, 0.0, 4.0))
            ts.append(s.timings().analysisMs)
        out.append("%d^2 %.3f" % (s.gx, min(ts[1:])))
dx = float(np.float32(343.21) / np.float32(275) / np.float32(3.5))
with pv.Solver((4096 + 0.5) * dx, (4096 + 0.5) * dx, 275, no_free_grid=1) as s:
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    ts = []
    for _ in range(4):
        s.run((5.0, 0.0, 4.0))
        ts.append(s.timings().analysisMs)
    out.append("4096^2 %.3f" % min(ts[1:]))
print("  ".join(out))
