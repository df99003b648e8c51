#!/usr/bin/env python3
"""row-streaming kernel: bit-equivalence with the tile kernels and speed (development aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
dx = 343.21 / 275 / 3.5

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
def same(a, b): return bool(((bits(a) == bits(b)) | ((a == 0) & (b == 0))).all())

n = 600
size = (n + 0.5) * dx
rng = np.random.default_rng(0)
init = [rng.standard_normal((n + 1, n + 1)).astype(np.float32) for _ in range(3)]
ref = None
for M in (0, 1, 2, 4, 7):
    s = pv.Solver(size, size, 275, no_free_grid=1, stream_rows=M)
    s.add_geometry([60, 70, 20, 1, 0.9]); s.add_geometry([120, 40, 1, 30, 0.7])
    s.set_fields(*init)
    s.run_steps(37)
    f = s.fields()
    if ref is None: ref = f
    print("M=%d raw 37 steps same as tiles:" % M, all(same(a, b) for a, b in zip(f, ref)))
    s.close()
# full run parity incl. history/analysis
for M in (0, 4):
    s = pv.Solver(size, size, 275, stream_rows=M)
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    s.run((100.0, 0, 90.0))
    r = s.results(); h = [s.history_plane(t) for t in (3, 100, 434)]
    if M == 0: r0, h0 = r, h
    else:
        print("run M=4: results same", same(r[0], r0[0]), "delay same", same(r[1], r0[1]), "hist same", all(same(a, b) for a, b in zip(h, h0)))
    s.close()
for n in (4096, 8192):
    size = (n + 0.5) * dx
    for M in (0, 2, 4, 8):
        s = pv.Solver(size, size, 275, stream_rows=M)
        s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
        s.run((5, 0, 4))
        t = []
        for _ in range(5):
            s.run((5, 0, 4)); t.append(s.timings().fdtdMs)
        cells = (s.gx + 1) * (s.gy + 1)
        print("n=%d M=%d fdtd min %.2f ms %.3e upd/s" % (n, M, min(t), cells * s.T / (min(t) * 1e-3)))
        s.close()
