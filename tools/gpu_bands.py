#!/usr/bin/env python3
"""Row bands (PVA_OPT_ROW_BANDS): whole-run rate of ONE run at a time and of two runs in flight, per band count.
    python tools/gpu_bands.py [grid ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")
LS = [(5, 0, 4), (8, 0, 8), (12, 0, 6), (15, 0, 15)]
grids = [int(a) for a in sys.argv[1:]] or [4096]
for n in grids:
    size = float((n + 0.5) * dx)
    for bands in (1, 2, 3, 4, 6, 8, 12):
        for inflight in (1, 2):
            sv = [pv.Solver(size, size, 275, row_bands=bands) for _ in range(inflight)]
            for s in sv:
                s.load_scene(scene)
                s.set_output_queries([(5.0, 0.0, 6.0)])
                s.run(LS[0])
            reps = 12
            t0 = time.perf_counter()
            for r in range(reps):
                for i, s in enumerate(sv):
                    s.sync()
                    s.run_async(LS[(r + i) % 4])
            for s in sv:
                s.sync()
            dt = time.perf_counter() - t0
            cells = (sv[0].gx + 1) * (sv[0].gy + 1)
            t = sv[0].timings()
            print("grid %d bands %2d inflight %d: %.3e cell-updates/s whole run (%.3f ms per run; last run: step loop %.3f ms, "
                  "analysis %.3f ms)" % (n, bands, inflight, reps * inflight * cells * sv[0].T / dt,
                                         dt / (reps * inflight) * 1e3, t.stepLoopMs, t.analysisMs), flush=True)
            for s in sv:
                s.close()
