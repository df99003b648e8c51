#!/usr/bin/env python3
"""Development aid: a few runs with the patch kernel (or the tile kernel: argv[1] = 0) for rocprofv3 --kernel-trace --stats"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
strip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dbg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
size = float((N + 0.5) * dx)
with pv.Solver(size, size, 275, patch_kernel=m, patch_strip=strip | (dbg << 8)) as s:
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    for _ in range(4):
        if dbg:
            s.run_steps(12 * 10, True, (5.0, 0.0, 4.0))
        else:
            s.run((5.0, 0.0, 4.0))
    print("loop ms", s.timings().stepLoopMs)
