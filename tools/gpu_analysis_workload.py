#!/usr/bin/env python3
"""The all-cells-reached analysis workload: Shoebox.pv as the real 25 m room at 512^2 (BASELINE config 2,
Mode B: res 2009, fs 10547, T = 3179) -- every cell of the room has an impulse response.  N runs; prints reached cells, analysis
ms and IR/s.  Profiled by tools/collect_profiles.sh (kernel stats + the two HBM PMC passes) -> profiles/<round>_analysis_pmc.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = 2009
for a in sys.argv[2:]:
    if a.startswith("res="):
        res = int(a[4:])
with pv.Solver(25.0, 25.0, res, rt60_lanes=int(os.environ.get("LANES", "0")), analysis_fork=int(os.environ.get("FORK", "1"))) as s:
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "Shoebox.pv"))
    ana, fd = [], []
    for _ in range(n):
        s.run((5.0, 0.0, 4.0))
        t = s.timings()
        ana.append(t.analysisMs)
        fd.append(t.fdtdMs)
    reached = t.reachedCells
    print("Shoebox.pv 25 m at %d^2, T = %d: %d reached cells of %d; analysis %.3f ms (min of %d runs) = %.3e analysed IR/s; "
          "history read at least once = %.1f MB -> %.0f GB/s; stencil %.3f ms" % (
              s.gx, s.T, reached, s.gx * s.gy, min(ana), n, reached / (min(ana) * 1e-3), reached * 4.0 * s.T / 1e6,
              reached * 4.0 * s.T / (min(ana) * 1e-3) / 1e9, min(fd)))
