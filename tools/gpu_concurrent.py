import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
import planeverb_amd.api as pv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nsolv = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dx = 343.21 / 275 / 3.5
size = (n + 0.5) * dx
S = [pv.Solver(size, size, 275) for _ in range(nsolv)]
for s in S:
    s.load_scene("/root/repo/tests/scenes/HugeRoom.pv")
Ls = [(5, 0, 4), (8, 0, 8), (12, 0, 6), (15, 0, 15)]
for i, s in enumerate(S):
    s.run(Ls[i])
cells = (S[0].gx + 1) * (S[0].gy + 1)
T = S[0].T
reps = 6
# sequential
t0 = time.time()
for r in range(reps):
    for i, s in enumerate(S):
        s.run(Ls[i])
seq = time.time() - t0
def work(i):
    for r in range(reps):
        S[i].run(Ls[i])
t0 = time.time()
th = [threading.Thread(target=work, args=(i,)) for i in range(nsolv)]
[t.start() for t in th]; [t.join() for t in th]
con = time.time() - t0
print("n=%d solvers=%d: sequential %.3e upd/s, concurrent %.3e upd/s (x%.3f)" % (n, nsolv, cells*T*reps*nsolv/seq, cells*T*reps*nsolv/con, seq/con))
