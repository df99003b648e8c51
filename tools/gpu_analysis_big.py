#!/usr/bin/env python3
"""Analysis ms of lone runs on the bench's secondary workloads (many reached cells in a large window): BigRoom.pv at 2048^2 (70 015
reached cells), the open field at 4096^2 / 8192^2 (275 669), per decay-time form (PVA_OPT_RT60_LANES: 0 = chosen on the device).
    python tools/gpu_analysis_big.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

dx = float(np.float32(343.21) / np.float32(275) / np.float32(3.5))
for n, scene, L in ((2048, "BigRoom.pv", (5.0, 0.0, 4.0)), (4096, None, None), (8192, None, None)):
    out = []
    for lanes in (0, 4, 1):
        with pv.Solver((n + 0.5) * dx, (n + 0.5) * dx, 275, no_free_grid=1, rt60_lanes=lanes) as s:
            if scene:
                s.load_scene(os.path.join(ROOT, "tests", "scenes", scene))
            l = L or ((n // 2 + 0.5) * dx, 0.0, (n // 2 + 0.5) * dx)
            ts = []
            for _ in range(4):
                s.run(l)
                ts.append(s.timings().analysisMs)
            out.append("lanes %d: %.3f ms (%d reached)" % (lanes, min(ts[1:]), s.timings().reachedCells))
    print("%d^2 %s: %s" % (n, scene or "open field", "   ".join(out)), flush=True)
