#!/usr/bin/env python3
"""Fuzz of the NEAR BOX (round 6: the analysis' passes over the bounding box of the reached cells instead of the history window) where
the window is smaller than the grid: seeded random scenes -- rooms with a door, loose walls, several absorptions -- in grids of
1000^2 ... 1700^2 cells (Mode A, T = 435: an 873-cell window), a sequence of random listeners (inside the room, in the open grid, at
the grid's edge, the same position twice), two solvers on the same scene: the near-box passes against the window-wide ones
(PLANEVERB_AMD_NEAR_BOX=0).  After every run: the registered queries, single outputs, a random block, and (every other run) the whole
delay and result maps, bit for bit.  The window-wide passes themselves are what the oracle / reference campaigns check.

    python tools/gpu_fuzz_nearbox.py [first_seed] [scenes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dx = float(np.float32(343.21) / np.float32(275) / np.float32(3.5))
bad = 0
t0 = time.time()
for seed in range(seed0, seed0 + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1000, 1700))
    size = (n + 0.5) * dx
    # a room (four walls, one with a door) somewhere in the grid + loose walls
    rx, ry = rng.uniform(0.1, 0.6) * size, rng.uniform(0.1, 0.6) * size
    rw, rh = rng.uniform(15, 0.3 * size), rng.uniform(15, 0.3 * size)
    t = rng.uniform(0.6, 3.0)
    R = float(rng.choice([0.97, 0.9, 0.5, 0.999]))
    door = rng.uniform(0.2, 0.8) * rw
    boxes = [[rx + rw / 2, ry, rw + t, t, R], [rx + rw / 2, ry + rh, rw + t, t, R], [rx, ry + rh / 2, t, rh + t, R],
             [rx + rw, ry + rh / 2 - door / 2 - 2, t, rh - door, float(rng.choice([R, 0.8]))]]
    if rng.random() < 0.3:
        boxes = boxes[:3] + [[rx + rw, ry + rh / 2, t, rh + t, R]]  # closed
    for _ in range(int(rng.integers(0, 6))):
        boxes.append([rng.uniform(0, size), rng.uniform(0, size), rng.uniform(1, 40), rng.uniform(0.5, 3), float(rng.uniform(0.3, 0.99))])
    inside = lambda: (rx + rng.uniform(0.1, 0.9) * rw, 0.0, ry + rng.uniform(0.1, 0.9) * rh)
    anywhere = lambda: (rng.uniform(0, size), 0.0, rng.uniform(0, size))
    Ls = [inside(), inside(), anywhere(), inside(), (rng.uniform(0, 3), 0.0, rng.uniform(0, size)), anywhere()]
    Ls.insert(int(rng.integers(1, 5)), Ls[0])
    probes = [inside(), inside(), anywhere(), anywhere(), (rx - 1.0, 0.0, ry + rh / 2), (size - 0.5, 0.0, size - 0.5)]
    os.environ.pop("PLANEVERB_AMD_NEAR_BOX", None)
    a = pv.Solver(size, size, 275)
    os.environ["PLANEVERB_AMD_NEAR_BOX"] = "0"
    b = pv.Solver(size, size, 275)
    os.environ.pop("PLANEVERB_AMD_NEAR_BOX", None)
    ok = True
    for s in (a, b):
        for bx in boxes:
            s.add_geometry(bx)
    for k, L in enumerate(Ls):
        for s in (a, b):
            s.set_output_queries(probes)
            s.run(L)
        eq = lambda x, y: np.array_equal(np.asarray(x, np.float32).view(np.uint32), np.asarray(y, np.float32).view(np.uint32))
        ok = ok and eq(a.queried_outputs(), b.queried_outputs())
        ok = ok and all(eq(a.get_output(p).as_array(), b.get_output(p).as_array()) for p in probes[:3])
        r0, c0 = int(rng.integers(0, n - 300)), int(rng.integers(0, n - 400))
        ba, bb = a.results_block(r0, c0, 300, 400), b.results_block(r0, c0, 300, 400)
        ok = ok and eq(ba[0], bb[0]) and eq(ba[1], bb[1])
        ok = ok and a.timings().reachedCells == b.timings().reachedCells
        if k % 2 == 1:
            ra, da = a.results()
            rb, db = b.results()
            ok = ok and eq(da, db) and eq(ra, rb)
        if not ok:
            print("seed %d: MISMATCH at run %d (grid %d^2, listener %r)" % (seed, k, n, L), flush=True)
            break
    reached = a.timings().reachedCells
    a.close()
    b.close()
    bad += 0 if ok else 1
    print("seed %d: grid %d^2, %d boxes, %d runs, last run reached %d cells: %s" % (seed, n, len(boxes), len(Ls), reached, "identical" if ok else "MISMATCH"), flush=True)
print("%d scenes, %d mismatches, %.0f s" % (count, bad, time.time() - t0))
