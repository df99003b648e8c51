#!/usr/bin/env python3
"""Mode B (fixed 25 m scene, resolution scaled so that the grid is N x N) in streaming-analysis mode."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv

SCENE = os.environ.get("SCENE", "HugeRoom.pv")
FUSE = int(os.environ.get("FUSE", "-1"))  # PVA_OPT_STREAM_FUSE (-1: by grid size)
ALT = int(os.environ.get("ALT", "-1"))  # PVA_OPT_ALTERNATE_SWEEPS (-1: default)
REG = int(os.environ.get("REG", "-1"))  # PVA_OPT_XCD_REGIONS (-1: default)
RUNS = int(os.environ.get("RUNS", "1"))
for res in [int(a) for a in sys.argv[1:]] or [2009, 4017, 8034, 16067]:
    t0 = time.time()
    s = pv.Solver(25.0, 25.0, res, streaming_analysis=1, stream_fuse=FUSE, alternate_sweeps=ALT, xcd_regions=REG)
    t_init = time.time() - t0
    s.load_scene(os.path.join(ROOT, "tests", "scenes", SCENE))
    E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
    s.set_emitters(E)
    L = (5.0, 0.0, 4.0)
    for _ in range(RUNS):
        t0 = time.time()
        s.run(L)
        wall = time.time() - t0
    t = s.timings()
    cells = (s.gx + 1) * (s.gy + 1)
    print("fuse %d alt %d reg %d " % (FUSE, ALT, REG), end="")
    print("res %d grid %d^2 T=%d efree %.6g init %.1fs: run wall %.3f s, stencil+accum %.1f ms (%.3e upd/s), finalize %.2f ms, HBM %d MB" % (
        res, s.gx, s.T, s.efree, t_init, wall, t.fdtdMs, cells * s.T / (t.fdtdMs * 1e-3), t.analysisMs, s.info.deviceBytes >> 20))
    for e in E:
        print("   ", e, np.array2string(s.get_output(e).as_array(), precision=6))
    s.close()
