#!/usr/bin/env python3
"""Run time of the reference's resolution presets (PvTypes.h:22-30: 275 / 375 / 500 / 750 Hz) on the Sandbox's 25 m scene,
batch API and live module -- the grids a drop-in user actually runs (development aid; profiles/r03_presets.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

scene = os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv")
print("# 25 m x 25 m, SmallRoomScene.pv, listener (5,0,4), emitter (5,0,6)")
print("# res   grid   T     batch run ms   us per step   live ms/iteration")
opts = {}
SIZE = 25.0
args = []
for a in sys.argv[1:]:
    if a.startswith("size="):
        SIZE = float(a[5:])
    elif "=" in a:
        k, v = a.split("=")
        opts[k] = int(v)
    else:
        args.append(int(a))
live = not opts
for res in args or [275, 375, 500, 750, 1000, 1500]:
    with pv.Solver(SIZE, SIZE, res, **opts) as s:
        s.load_scene(scene)
        s.set_output_queries([(5.0, 0.0, 6.0)])
        for _ in range(3):
            s.run((5.0, 0.0, 4.0))
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            s.run((5.0, 0.0, 4.0))
            s.queried_outputs()
        batch_ms = (time.perf_counter() - t0) / reps * 1e3
        gx, T = s.gx, s.T
        k, rows = s.info.stepsPerLaunch, s.info.tileRows
    if not live:
        print("%5d  %4d^2  %5d   %8.3f   %8.3f   (K %d, tile rows %d)" % (res, gx, T, batch_ms, batch_ms * 1e3 / T, k, rows), flush=True)
        continue
    pv.Init(pv.Config((25.0, 25.0), res, 0, ".", 0, pv.pv_GPU))
    pv.SetListenerPosition((5.0, 0.0, 4.0))
    pv.LoadScene(scene)
    pv.Emit((5.0, 0.0, 6.0))
    pv.WaitIterations(pv.IterationCount() + 5, 60000)
    n0, t0 = pv.IterationCount(), time.perf_counter()
    time.sleep(1.0)
    n1, t1 = pv.IterationCount(), time.perf_counter()
    pv.Exit()
    print("%5d  %4d^2  %5d   %8.3f   %8.3f   %8.3f   (K %d, tile rows %d)" % (res, gx, T, batch_ms, batch_ms * 1e3 / T,
                                                                               1e3 * (t1 - t0) / max(1, n1 - n0), k, rows), flush=True)
