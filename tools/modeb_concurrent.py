#!/usr/bin/env python3
"""Mode B (25 m scene at N x N cells, streaming analysis) with B independent runs in flight on one GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv

res = int(sys.argv[1]) if len(sys.argv) > 1 else 16067
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
Ls = [(5.0, 0.0, 4.0), (8.0, 0.0, 8.0), (12.0, 0.0, 6.0), (15.0, 0.0, 15.0)]
S = []
for b in range(B):
    s = pv.Solver(25.0, 25.0, res, streaming_analysis=1)
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    s.set_emitters(E)
    S.append(s)
cells = (S[0].gx + 1) * (S[0].gy + 1)
T = S[0].T
for b, s in enumerate(S):
    s.run(Ls[b])
t0 = time.time()
for b, s in enumerate(S):
    s.run(Ls[b])
seq = time.time() - t0
t0 = time.time()
for b, s in enumerate(S):
    s.run_async(Ls[b])
for s in S:
    s.sync()
con = time.time() - t0
print("res %d grid %d^2 T=%d, %d runs: one at a time %.3f s (%.3e upd/s), in flight together %.3f s (%.3e upd/s)" % (
    res, S[0].gx, T, B, seq, B * cells * T / seq, con, B * cells * T / con))
