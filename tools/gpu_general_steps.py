#!/usr/bin/env python3
"""Where a general block's time goes: one launch of the merged kernel with nsteps = 1 ... K (PvAmdRunSteps(n), n <= K: same loads and stores,
fewer steps), whole launch and -- with PV_PROBE_GENERAL_ONLY=1 -- cut off behind its general blocks.  time(n) = a + b n: a = dispatch + load phase
+ store phase, b = one step (profiles/r05_modeB_general.txt).
    [MODEB=1] [PV_PROBE_GENERAL_ONLY=1] python tools/gpu_general_steps.py [grid=4096]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size, res, opts = bench.mode_a_size(grid), 275, {}
if os.environ.get("MODEB"):
    size, res = 25.0, {512: 2009, 1024: 4017, 2048: 8034, 4096: 16067, 8192: 32134}[grid]
    opts["streaming_analysis"] = 1
s = pv.Solver(size, size, res, **opts)
s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
K = s.info.stepsPerLaunch
rows = []
for n in [1, 2, 3, 4, 6, 8, 10, 12]:
    if n > K:
        continue
    s.run_steps(n)
    t = []
    for _ in range(60):
        s.run_steps(n)
        t.append(s.timings().fdtdMs * 1e3)
    t = np.sort(np.asarray(t))
    rows.append((n, float(t[len(t) // 2]), float(t[3])))
ns = np.array([r[0] for r in rows], float)
ts = np.array([r[1] for r in rows], float)
b, a = np.polyfit(ns, ts, 1)
print("%s geometry, %d^2, K = %d%s: launch us (median / low) by steps per launch: %s" % (
    "Mode B" if os.environ.get("MODEB") else "Mode A", grid, K, ", general blocks only" if os.environ.get("PV_PROBE_GENERAL_ONLY") else "",
    "  ".join("%d: %.1f/%.1f" % r for r in rows)))
print("   fit: %.1f us + %.2f us per step" % (a, b))
s.close()
