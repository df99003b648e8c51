#!/usr/bin/env python3
"""row-streaming segments vs tile kernels: stencil loop of one run, by grid size, tile configuration and segment count
(profiles/r02_segments.txt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
dx = 343.21 / 275 / 3.5
for n in (4096, 8192):
    size = (n + 0.5) * dx
    for K, R in ((12, 36), (8, 40)):
        for seg in (0, 512, 1024, 1536, 2048, 3072, 4096):
            if K == 12 and seg > 2048: continue
            s = pv.Solver(size, size, 275, stream_rows=seg, steps_per_launch=K, tile_rows=R)
            s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
            s.run((5, 0, 4))
            t = []
            for _ in range(4):
                s.run((5, 0, 4)); t.append(s.timings().fdtdMs)
            cells = (s.gx + 1) * (s.gy + 1)
            print("n=%d K=%d rows=%d segments=%-4d fdtd min %6.2f ms  %.3e cell-updates/s" % (
                n, K, R, seg, min(t), cells * s.T / (min(t) * 1e-3)), flush=True)
            s.close()
