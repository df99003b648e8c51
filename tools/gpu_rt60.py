#!/usr/bin/env python3
"""Decay-time pass (csrc/pv_rt60.hip): the 16- / 4- / 1-lane forms against each other, bit for bit, and their analysis
times on the reference's presets and beyond (development aid; profiles/r04_rt60.txt).

    python tools/gpu_rt60.py [res ...] [size=25] [scene=SmallRoomScene.pv]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

size, scene, presets = 25.0, "SmallRoomScene.pv", []
for a in sys.argv[1:]:
    if a.startswith("size="):
        size = float(a[5:])
    elif a.startswith("scene="):
        scene = a[6:]
    else:
        presets.append(int(a))
scene = os.path.join(ROOT, "tests", "scenes", scene) if scene != "none" else None
L = (5.0, 0.0, 4.0)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    nn = ~(np.isnan(a) & np.isnan(b))
    z = (a == 0) & (b == 0)
    return np.array_equal(a.view(np.uint32)[nn & ~z], b.view(np.uint32)[nn & ~z])


print("# %g m, %s; analysis ms of a run (far frame + encode + wet / decay time + direction), run ms" % (size, scene))
LANES = [int(x) for x in os.environ.get("LANES", "16,4,0").split(",")]
print("# res  grid     T   reached |  lanes %s (0 = auto) | bit-identical" % LANES)
bad = 0
for res in presets or [275, 500, 750, 1000, 1500, 2009]:
    row, ref, ok = [], None, True
    for lanes in LANES:
        with pv.Solver(size, size, res, rt60_lanes=lanes) as s:
            if scene:
                s.load_scene(scene)
            for _ in range(2):
                s.run(L)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                s.run(L)
                ts.append((s.timings().analysisMs, (time.perf_counter() - t0) * 1e3))
            row.append(min(ts))
            r, d = s.results()
            if ref is None:
                ref = (r, d)
                reached = int((d != np.finfo(np.float32).max).sum())
                gx, T = s.gx, s.T
            else:
                ok = ok and same(ref[0], r) and same(ref[1], d)
    bad += 0 if ok else 1
    print("%5d %4d^2 %5d %8d | %s | %s" % (res, gx, T, reached, " ".join("%5.3f/%5.2f" % t for t in row), ok), flush=True)
sys.exit(1 if bad else 0)
