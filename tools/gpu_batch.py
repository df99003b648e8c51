"""Runs ON THE GPU BOX: whole-run throughput (stencil + record + analysis + per-emitter fetch) of G groups in flight x
B runs per batched launch, per grid size.  usage: gpu_batch.py N[,N..] "GxB GxB ..." [scene.pv]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import planeverb_amd.api as pv

sizes = [int(x) for x in sys.argv[1].split(",")]
combos = [tuple(int(v) for v in c.split("x")) for c in sys.argv[2].split()]
scene = sys.argv[3] if len(sys.argv) > 3 else "HugeRoom.pv"
dx = 343.21 / 275 / 3.5
LISTENERS = [(5, 4), (8, 8), (12, 6), (15, 15), (20, 5), (5, 20), (20, 20), (12.5, 18)]
for n in sizes:
    size = (n + 0.5) * dx
    for G, B in combos:
        groups = []
        for g in range(G):
            grp = []
            for b in range(B):
                s = pv.Solver(size, size, 275, **(pv.batch_solver_options(n) if B > 1 and n <= 2048 else {}))
                if scene != "none":
                    s.load_scene(os.path.join(ROOT, "tests", "scenes", scene))
                grp.append(s)
            groups.append(grp)
        cells = (groups[0][0].gx + 1) * (groups[0][0].gy + 1)
        T = groups[0][0].T
        nruns = 0

        def Ls(i, B):
            return [(LISTENERS[(i * B + b) % 8][0], 0.0, LISTENERS[(i * B + b) % 8][1]) for b in range(B)]

        def go(rounds):
            global nruns
            pending = [None] * G
            for i in range(rounds * G):
                g = i % G
                if pending[g] is not None:
                    for sv, L in zip(groups[g], pending[g]):
                        sv.sync()
                        sv.get_output((L[0], 0.0, L[2] + 2.0))
                ll = Ls(i, B)
                if B == 1:
                    groups[g][0].run_async(ll[0])
                else:
                    pv.run_batch(groups[g], ll, wait=False)
                pending[g] = ll
            for g in range(G):
                if pending[g] is not None:
                    for sv, L in zip(groups[g], pending[g]):
                        sv.sync()
                        sv.get_output((L[0], 0.0, L[2] + 2.0))
        go(2)
        rounds = max(3, min(40, int(3e10 / (cells * T * G * B))))
        t0 = time.perf_counter()
        go(rounds)
        dt = time.perf_counter() - t0
        runs = rounds * G * B
        print("n=%d groups=%d batch=%d: %.3f ms/run  %.4g cell-updates/s  (fdtd %.3f ms per batch, K=%d)" % (
            n, G, B, dt / runs * 1e3, runs * cells * T / dt, groups[0][0].timings().fdtdMs,
            groups[0][0].info.stepsPerLaunch), flush=True)
        for grp in groups:
            for s in grp:
                s.close()
