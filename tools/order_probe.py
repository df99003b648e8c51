#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
order = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dx = 343.21 / 275 / 3.5
s = pv.Solver((n + 0.5) * dx, (n + 0.5) * dx, 275, tile_order=order, skip_analysis=1)
s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
for _ in range(3):
    s.run((5, 0, 4))
print(order, s.timings().fdtdMs)
