#!/usr/bin/env python3
"""Is the residual two-runs-in-flight mode (1.69 vs 1.78e12 at 4096^2, profiles/r05_placement.txt) a property of the process or of the second
solver's resources?  Solver A lives for the whole process; solver B (its streams AND its buffers) is created, timed beside A, and closed,
several times over; then both are re-created.
    python tools/gpu_mode_probe.py [grid=4096] [rounds=6]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dx = 343.21 / 275 / 3.5
size = (n + 0.5) * dx
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")
Ls = [(5, 0, 4), (8, 0, 8)]


def make():
    s = pv.Solver(size, size, 275)
    s.load_scene(scene)
    s.run(Ls[0])
    return s


def pair_rate(S, reps=10):
    def work(i):
        for _ in range(reps):
            S[i].run(Ls[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    return (S[0].gx + 1) * (S[0].gy + 1) * S[0].T * reps * 2 / dt


A = make()
out = []
for r in range(rounds):
    B = make()
    rates = [pair_rate([A, B]) for _ in range(3)]
    out.append("%.3f %.3f %.3f" % tuple(x / 1e12 for x in rates))
    B.close()
print("A kept, B re-created %d times (three timings each, e12 cell-updates/s):  %s" % (rounds, " | ".join(out)))
A.close()
out = []
for r in range(3):
    A, B = make(), make()
    out.append("%.3f %.3f" % tuple(pair_rate([A, B]) / 1e12 for _ in range(2)))
    A.close()
    B.close()
print("both re-created 3 times:  %s" % " | ".join(out))
