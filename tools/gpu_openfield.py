#!/usr/bin/env python3
"""Open-field runs (empty scene, listener at the grid centre: SURVEY.md 8d config 5's shape) -- stencil and analysis
time per run.  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
for n in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096, 8192]:
    dx = 343.21 / 275 / 3.5
    size = (n + 0.5) * dx
    s = pv.Solver(size, size, 275)
    L = (size / 2, 0, size / 2)
    s.run(L)
    s.run(L)
    t = s.timings()
    print("open field n=%d: fdtd %.2f ms analysis %.2f ms" % (n, t.fdtdMs, t.analysisMs))
    s.close()
