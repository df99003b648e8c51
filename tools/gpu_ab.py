#!/usr/bin/env python3
"""Interleaved A/B of solver options in ONE process (development aid): python tools/gpu_ab.py N 'k=v,...' 'k=v,...'"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
n = int(sys.argv[1])
variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]]
dx = 343.21 / 275 / 3.5
size = (n + 0.5) * dx
solvers = []
for v in variants:
    s = pv.Solver(size, size, 275, **v)
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    s.run((5, 0, 4))
    solvers.append(s)
times = [[] for _ in variants]
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    for i, s in enumerate(solvers):
        s.run((5, 0, 4))
        times[i].append(s.timings().fdtdMs)
cells = (solvers[0].gx + 1) * (solvers[0].gy + 1)
for v, t in zip(variants, times):
    print("n=%d %s: fdtd min %.2f med %.2f ms  %.3e upd/s" % (n, v, min(t), float(np.median(t)), cells * solvers[0].T / (min(t) * 1e-3)))
