#!/usr/bin/env python3
"""Two host threads, each creating, running (replayed run graph: capture on the first run) and destroying solvers on the same
device at the same time -- what the live module's worker and a caller's own batch solver do.  A graph capture that another
thread's allocation / free / synchronous copy invalidates shows up as a failed run (development aid)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fails = []
def work(tid):
    for i in range(N):
        try:
            s = pv.Solver(25.0 + tid, 25.0 + tid, 275 + 50 * (i % 3))
            s.load_scene(os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv"))
            for k in range(3):
                s.run((5.0, 0.0, 4.0 + 0.1 * k))
            s.close()
        except Exception as e:  # noqa: BLE001
            fails.append((tid, i, str(e)))
t0 = time.time()
ts = [threading.Thread(target=work, args=(t,)) for t in range(int(os.environ.get("THREADS", "2")))]
for t in ts: t.start()
for t in ts: t.join()
print("%d threads x %d solvers x 3 runs in %.1f s: %d failures" % (len(ts), N, time.time() - t0, len(fails)))
for f in fails[:5]: print("  ", f)
sys.exit(1 if fails else 0)
