#!/usr/bin/env python3
"""Sparse-emitter mode: forward sums inside the stencil (PVA_OPT_STREAM_FUSE = 1) against ring + accumulate pass for
every tile (= 0) on the SAME scene: whole result / delay maps and the registered emitters' records, bit for bit.
usage: modeb_fuse_check.py [res ...]   (25 m scene; env SCENE, TILE="K,rows" to force a tile configuration)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv

SCENE = os.environ.get("SCENE", "HugeRoom.pv")
TILE = os.environ.get("TILE", "")
NAMES = ["occlusion", "wetGain", "rt60", "lowpass", "dirX", "dirY", "srcDirX", "srcDirY"]
E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
L = (5.0, 0.0, 4.0)
bad = 0
for res in [int(a) for a in sys.argv[1:]] or [2009, 4017]:
    out = {}
    for fuse in (1, 0):
        opts = dict(streaming_analysis=1, stream_fuse=fuse)
        if TILE:
            k, r = TILE.split(",")
            opts.update(steps_per_launch=int(k), tile_rows=int(r))
        s = pv.Solver(25.0, 25.0, res, **opts)
        s.load_scene(os.path.join(ROOT, "tests", "scenes", SCENE))
        s.set_emitters(E)
        t0 = time.time()
        s.run(L)
        wall = time.time() - t0
        t = s.timings()
        r, d = s.results()
        em = np.stack([s.get_output(e).as_array() for e in E])
        out[fuse] = (r, d, em, s.fields())
        cells = (s.gx + 1) * (s.gy + 1)
        print("res %d grid %d^2 T=%d K=%d rows=%d fuse %d: wall %.3f s, stencil+accum %.1f ms (%.3e upd/s)" % (
            res, s.gx, s.T, s.info.stepsPerLaunch, s.info.tileRows, fuse, wall, t.fdtdMs,
            cells * s.T / (t.fdtdMs * 1e-3)), flush=True)
        s.close()
    (r1, d1, e1, f1), (r0, d0, e0, f0) = out[1], out[0]
    for name, a, b in [("delay", d1, d0), ("emitter records", e1, e0), ("pr", f1[0], f0[0]), ("vx", f1[1], f0[1]),
                       ("vy", f1[2], f0[2])] + [(NAMES[k], r1[..., k], r0[..., k]) for k in range(8)]:
        ne = a.view(np.uint32) != b.view(np.uint32)
        if ne.any():
            bad += 1
            idx = np.argwhere(ne)
            print("  MISMATCH %s: %d cells, first %s: %r vs %r" % (name, ne.sum(), idx[0], a[tuple(idx[0])], b[tuple(idx[0])]))
    print("  res %d: %s" % (res, "identical" if not bad else "DIFFERENT"), flush=True)
sys.exit(1 if bad else 0)
