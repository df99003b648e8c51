#!/usr/bin/env python3
"""Development aid: A/B of the persistent patch kernel (PVA_OPT_PATCH_KERNEL, csrc/pv_patch.h) against the one-wave-per-tile
kernel -- stencil-loop time of whole runs (HIP events around the launch loop), one run at a time and two in flight.

    python tools/gpu_patch.py [grid] [strip ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
strips = [int(v) for v in sys.argv[2:]] or [3]
dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
size = float((N + 0.5) * dx)
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")
L = [(5.0, 0.0, 4.0), (8.0, 0.0, 8.0)]


def one(tag, **opts):
    sv = [pv.Solver(size, size, 275, **opts) for _ in range(2)]
    for s in sv:
        s.load_scene(scene)
    cells = (sv[0].gx + 1) * (sv[0].gy + 1)
    T = sv[0].T
    for s in sv:
        s.run(L[0])
    loop, wall = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        sv[0].run(L[0])
        wall.append(time.perf_counter() - t0)
        loop.append(sv[0].timings().stepLoopMs)
    single = float(np.median(loop))
    sw = float(np.median(wall))
    # two in flight
    t2 = []
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(4):
            for i, s in enumerate(sv):
                s.run_async(L[i])
            for s in sv:
                s.sync()
        t2.append((time.perf_counter() - t0) / 8)
    dual = float(np.median(t2))
    out = sv[0].get_output((5.0, 0.0, 6.0)).as_array()
    print("%-28s loop %.3f ms  (%.0f us per 12-step sweep)  run wall %.3f ms = %.3e upd/s ; two in flight %.3f ms per run "
          "= %.3e upd/s ; out %s" % (tag, single, single * 12 / T * 1e3, sw * 1e3, cells * T / sw, dual * 1e3,
                                     cells * T / dual, out[:3]))
    for s in sv:
        s.close()


one("tile kernel (merged)", patch_kernel=0)
for st in strips:
    one("patch kernel, strip %d" % st, patch_kernel=1, patch_strip=st)
one("tile kernel again", patch_kernel=0)
