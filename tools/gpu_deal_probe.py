#!/usr/bin/env python3
"""Does the paired launch time of two solvers depend on WHICH streams they were dealt?  (VERDICT r05 "next" item 6.)
Solver A stays; solver B is created again and again -- with 0 ... 7 idle aux streams in front of its own, which shifts the hardware
queues its streams are dealt (PVA_OPT_AUX_STREAMS), and several times with the same number -- and every pair is timed on REAL
sweeps: both solvers step the raw stencil from two host threads for a while, launch time = HIP events around each call's
back-to-back launches.  Prints p50 of the paired launch time per deal, and the lone launch time of A between the pairs.

    python tools/gpu_deal_probe.py [grid=4096] [seconds per pair=0.4]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
size = float((n + 0.5) * dx)
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")


def make(**kw):
    s = pv.Solver(size, size, 275, **kw)
    s.load_scene(scene)
    return s


def timed(solvers, seconds):
    K = solvers[0].info.stepsPerLaunch
    launches = 36
    per = [[] for _ in solvers]
    stop = [False]

    def work(i, sv):
        while not stop[0]:
            sv.run_steps(launches * K)
            per[i].append(sv.timings().fdtdMs / launches)
    for sv in solvers:
        sv.run_steps(K)
    th = [threading.Thread(target=work, args=(i, sv)) for i, sv in enumerate(solvers)]
    for t in th:
        t.start()
    time.sleep(seconds)
    stop[0] = True
    for t in th:
        t.join()
    return [float(np.median(p[1:] or p)) for p in per]


if os.environ.get("DEAL_ONCE"):  # one pair per PROCESS (tools/gpu_deal_procs.sh): does the launch time differ between processes?
    A, B = make(), make()
    lone = timed([A], 0.2)[0]
    a, b = timed([A, B], secs)
    print("process %d: lone %.4f  paired %.4f %.4f" % (os.getpid(), lone, a, b), flush=True)
    B.close()
    A.close()
    sys.exit(0)
if os.environ.get("DEAL_BOTH"):  # both solvers created anew for every pair, in ONE process: does the mode flip inside a process?
    keep = []
    for rep in range(int(os.environ["DEAL_BOTH"])):
        A, B = make(), make()
        a, b = timed([A, B], secs)
        print("pair %d: paired %.4f %.4f   lone A %.4f" % (rep, a, b, timed([A], 0.15)[0]), flush=True)
        B.close()
        A.close()
    sys.exit(0)
A = make()
print("# %d^2: paired launch ms (A, B) per deal of B; lone = A alone" % n)
print("lone A: %.4f" % timed([A], secs)[0])
for rep in range(2):
    for aux in (0, 1, 2, 3, 4, 5, 6, 7, 0, 0):
        B = make(aux_streams=aux)
        a, b = timed([A, B], secs)
        print("aux %d: paired A %.4f  B %.4f   (lone A after: %.4f)" % (aux, a, b, timed([A], 0.15)[0]), flush=True)
        B.close()
A.close()
