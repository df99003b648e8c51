#!/usr/bin/env python3
"""Where a 4096^2 run's time goes (development aid): stencil from zero fields, from all-non-zero fields, and the
full run with the pulse and the history record."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, rows = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 0)
dx = 343.21 / 275 / 3.5
size = (n + 0.5) * dx
opts = dict(steps_per_launch=K, tile_rows=rows) if K else {}
s = pv.Solver(size, size, 275, **opts)
s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
cells = (s.gx + 1) * (s.gy + 1)
steps = 36 * s.info.stepsPerLaunch


def t_steps(label):
    best = 1e9
    for _ in range(3):
        s.run_steps(steps)
        best = min(best, s.timings().fdtdMs)
    print("%-44s %.3f ms / %d steps = %.1f us per launch  (%.3e upd/s)" % (
        label, best, steps, best * 1e3 / 36, cells * steps / (best * 1e-3)))


z = np.zeros((s.gx + 1, s.gy + 1), np.float32)
s.set_fields(z, z, z)
t_steps("zero fields, no record, no pulse")
rng = np.random.default_rng(0)
f = [rng.standard_normal(z.shape).astype(np.float32) * 1e-3 for _ in range(3)]
s.set_fields(*f)
t_steps("random fields, no record")
L = (5, 0, 4)
s.run(L)
best = 1e9
for _ in range(3):
    s.run(L)
    best = min(best, s.timings().fdtdMs)
nl = -(-s.T // s.info.stepsPerLaunch)
print("%-44s %.3f ms / %d steps = %.1f us per launch" % ("full run (pulse + record)", best, s.T, best * 1e3 / nl))
