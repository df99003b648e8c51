#!/usr/bin/env python3
"""Mode B (25 m scene at N x N cells, sparse-emitter mode) with B independent runs in flight on one GPU, one HOST THREAD per run
(enqueueing a sparse-emitter run blocks its caller for the front phase: tools/modeb_concurrent.py's single thread serialises them).
    python tools/modeb_threads.py [res=16067] [B=2]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

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
outs = [None] * B
for b, s in enumerate(S):
    s.run(Ls[b])
    outs[b] = [s.get_output(e).as_array().copy() for e in E]
t0 = time.time()
for b, s in enumerate(S):
    s.run(Ls[b])
seq = time.time() - t0


def work(b):
    S[b].run(Ls[b])


th = [threading.Thread(target=work, args=(b,)) for b in range(B)]
t0 = time.time()
for t in th:
    t.start()
for t in th:
    t.join()
par = time.time() - t0
same = all((S[b].get_output(e).as_array().view("u4") == outs[b][i].view("u4")).all() for b in range(B) for i, e in enumerate(E))
print("res %d grid %d^2 T=%d, %d runs: one at a time %.3f s (%.3e upd/s), %d host threads %.3f s (%.3e upd/s), records identical: %s" % (
    res, S[0].gx, T, B, seq, B * cells * T / seq, B, par, B * cells * T / par, same))
