#!/usr/bin/env python3
"""Resident kernel (csrc/pv_resident.hip) against the replayed graph of tile-kernel launches on the reference's presets:
every map, field, history plane and emitter record compared bit for bit, and the run time of both (development aid;
profiles/r04_presets.txt).

    python tools/gpu_resident.py [res ...] [size=25] [scene=SmallRoomScene.pv] [reps=20] [check=1] [stress=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

size, reps, check, stress = 25.0, 20, 1, 0
scene = "SmallRoomScene.pv"
presets = []
for a in sys.argv[1:]:
    if a.startswith("size="):
        size = float(a[5:])
    elif a.startswith("scene="):
        scene = a[6:]
    elif a.startswith("reps="):
        reps = int(a[5:])
    elif a.startswith("check="):
        check = int(a[6:])
    elif a.startswith("stress="):
        stress = int(a[7:])
    else:
        presets.append(int(a))
scene = os.path.join(ROOT, "tests", "scenes", scene)
L, E = (5.0, 0.0, 4.0), (5.0, 0.0, 6.0)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    z = (a == 0) & (b == 0)  # sign of zero aside
    return a.shape == b.shape and np.array_equal(a.view(np.uint32)[~z], b.view(np.uint32)[~z])


def timed(s):
    for _ in range(3):
        s.run(L)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.run(L)
        s.queried_outputs()
    ms = (time.perf_counter() - t0) / reps * 1e3
    tm = s.timings()
    return ms, tm.fdtdMs, tm.analysisMs


print("# %g m x %g m, %s, listener %s, emitter %s" % (size, size, os.path.basename(scene), L, E))
print("# res  grid    T    tiles | graph: run ms (fdtd + analysis) | resident: run ms (fdtd + analysis) | x     | bit-identical")
bad = 0
for res in presets or [275, 375, 500, 750, 1000, 1500]:
    with pv.Solver(size, size, res, resident_kernel=2) as g, pv.Solver(size, size, res) as r:
        for s in (g, r):
            s.load_scene(scene)
            s.set_output_queries([E])
        if not r.info.residentKernel:
            print("%5d  %4d^2: the resident kernel does not take this grid" % (res, r.gx))
            continue
        gm = timed(g)
        rm = timed(r)
        ok = "-"
        if check:
            ok = same(g.queried_outputs()[0], r.queried_outputs()[0])
            rg, dg = g.results()
            rr, dr = r.results()
            ok = ok and same(rg, rr) and same(dg, dr)
            for fa, fb in zip(g.fields(), r.fields()):
                ok = ok and same(fa, fb)
            for t in (0, 1, 11, 12, 13, g.T // 2, g.T - 2, g.T - 1):
                ok = ok and same(g.history_plane(t), r.history_plane(t))
            # other listeners: corners, edges, inside a wall
            for Lx in ((0.1, 0.0, 0.1), (size - 0.2, 0.0, size - 0.2), (12.5, 0.0, 0.3), (7.3, 0.0, 19.1)):
                g.run(Lx)
                r.run(Lx)
                rg, dg = g.results()
                rr, dr = r.results()
                ok = ok and same(rg, rr) and same(dg, dr)
                for fa, fb in zip(g.fields(), r.fields()):
                    ok = ok and same(fa, fb)
            bad += 0 if ok else 1
        tiles = (-(-(r.gx + 1) // r.info.tileRows)) * (-(-(r.gy + 1) // r.info.tileCols))
        print("%5d  %4d^2 %5d  %4d | %7.3f (%6.3f + %5.3f) | %7.3f (%6.3f + %5.3f) | %4.2f | %s" % (
            res, r.gx, r.T, tiles, gm[0], gm[1], gm[2], rm[0], rm[1], rm[2], gm[0] / rm[0], ok), flush=True)
        if stress:
            # repeated runs with moving listeners, every run compared (hand-off staleness shows up as a mismatch)
            rng = np.random.default_rng(res)
            nbad = 0
            for i in range(stress):
                Lr = (float(rng.uniform(0.2, size - 0.2)), 0.0, float(rng.uniform(0.2, size - 0.2)))
                g.run(Lr)
                r.run(Lr)
                rg, dg = g.results()
                rr, dr = r.results()
                if not (same(rg, rr) and same(dg, dr) and all(same(fa, fb) for fa, fb in zip(g.fields(), r.fields()))):
                    nbad += 1
            print("#        stress: %d runs with random listeners, %d mismatches" % (stress, nbad), flush=True)
            bad += nbad
sys.exit(1 if bad else 0)
