#!/usr/bin/env python3
"""row-streaming air segments (pv_seg.h): bit-equivalence with the tile kernels and speed (development aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
dx = 343.21 / 275 / 3.5

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
def same(a, b): return bool(((bits(a) == bits(b)) | ((a == 0) & (b == 0))).all())

quick = "--quick" in sys.argv
KK = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 12
RR = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 36
K12 = dict(steps_per_launch=KK, tile_rows=RR, use_graph=2)
if "--no-parity" not in sys.argv:
    n = 900
    size = (n + 0.5) * dx
    rng = np.random.default_rng(0)
    init = [rng.standard_normal((n + 1, n + 1)).astype(np.float32) for _ in range(3)]
    ref = None
    for seg in (-1, 64, 256, 1024):
        s = pv.Solver(size, size, 275, no_free_grid=1, stream_rows=seg, **K12)
        s.add_geometry([60, 70, 20, 1, 0.9]); s.add_geometry([120, 40, 1, 30, 0.7])
        s.set_fields(*init)
        s.run_steps(3 * KK)
        f = s.fields()
        if ref is None: ref = f
        print("segments=%d raw 36 steps same as tiles:" % seg, [same(a, b) for a, b in zip(f, ref)], flush=True)
        if seg > 0 and not all(same(a, b) for a, b in zip(f, ref)):
            for a, b, nm in zip(f, ref, "pxy"):
                bad = np.argwhere(~((bits(a) == bits(b)) | ((a == 0) & (b == 0))))
                print(nm, len(bad), bad[:5], bad[-5:])
        s.close()
    # full run parity incl. history / analysis
    for seg in (-1, 256):
        s = pv.Solver(size, size, 275, stream_rows=seg, **K12)
        s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
        s.run((100.0, 0, 90.0))
        r = s.results(); h = [s.history_plane(t) for t in (3, 100, 434)]
        if seg < 0: r0, h0 = r, h
        else:
            print("run segments=%d: results same" % seg, same(r[0], r0[0]), "delay same", same(r[1], r0[1]), "hist same",
                  [same(a, b) for a, b in zip(h, h0)], flush=True)
        s.close()
    # open field: the pulse crosses many segments, all of which record
    for seg in (-1, 256):
        s = pv.Solver(size, size, 275, stream_rows=seg, **K12)
        s.run((150.0, 0, 170.0))
        r = s.results(); h = [s.history_plane(t) for t in (3, 100, 300, 434)]
        if seg < 0: r0, h0 = r, h
        else:
            print("open run segments=%d: results same" % seg, same(r[0], r0[0]), "delay same", same(r[1], r0[1]), "hist same",
                  [same(a, b) for a, b in zip(h, h0)], flush=True)
        s.close()
for n in ((4096,) if quick else (4096, 8192)):
    size = (n + 0.5) * dx
    for seg in (-1, 512, 768, 1024, 1536, 2048):
        s = pv.Solver(size, size, 275, stream_rows=seg, steps_per_launch=KK, tile_rows=RR)
        s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
        s.run((5, 0, 4))
        t = []
        for _ in range(5):
            s.run((5, 0, 4)); t.append(s.timings().fdtdMs)
        cells = (s.gx + 1) * (s.gy + 1)
        print("n=%d segments=%d fdtd min %.2f ms %.3e upd/s" % (n, seg, min(t), cells * s.T / (min(t) * 1e-3)), flush=True)
        s.close()
