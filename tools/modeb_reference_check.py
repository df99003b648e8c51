#!/usr/bin/env python3
"""One-off parity check of the sparse-emitter mode ABOVE the size the test suite can hold: the UNMODIFIED reference
(oracle/_ref/libpvref.so, one host core of the GPU box, Grid + FreeGrid cubes of (N+1)^2 x T x 16 B each) against the GPU's
streaming analysis in both forms (PVA_OPT_STREAM_FUSE = 0 / 1) on the Mode B scene: 25 m HugeRoom.pv at a resolution that
makes the grid N x N.  Every cell's onset; occlusion, low-pass and both directions of every cell whose windows lie inside
the response; wet gain and RT60 at the registered emitters -- bit for bit.

    python tools/modeb_reference_check.py [res=4017]      (4017 -> 1024^2, T = 6358: 2 x 107 GB of host memory)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import planeverb_amd.api as pv  # noqa: E402
from oracle import pvref  # noqa: E402  (checker only)

NAMES = ["occlusion", "wetGain", "rt60", "lowpass", "dirX", "dirY", "srcDirX", "srcDirY"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))


res = int(sys.argv[1]) if len(sys.argv) > 1 else 4017
scene = os.path.join(ROOT, "tests", "scenes", os.environ.get("SCENE", "HugeRoom.pv"))
E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
L = (5.0, 0.0, 4.0)

probe = pv.Solver(25.0, 25.0, res, streaming_analysis=1)
gx, gy, T = probe.gx, probe.gy, probe.T
probe.close()
need_gb = 2 * (gx + 1) * (gy + 1) * T * 16 / 1e9 + 8
avail_gb = 0.0
with open("/proc/meminfo") as f:
    for line in f:
        if line.startswith("MemAvailable"):
            avail_gb = int(line.split()[1]) / 1e6
print("grid %d^2, T = %d: the reference needs about %.0f GB of host memory, %.0f GB available" % (gx, T, need_gb, avail_gb), flush=True)
# (242 GB at 4097^2 ran fine; 934 GB at 8193^2 took the GPU box down although /proc/meminfo showed 3 TB available: the
# pool's boxes are not to be trusted beyond a few hundred GB)
if need_gb > 320 and not os.environ.get("PV_ALLOW_HUGE_REFERENCE"):
    raise SystemExit("refusing a reference run of %.0f GB (set PV_ALLOW_HUGE_REFERENCE=1 to override)" % need_gb)
if avail_gb < 1.3 * need_gb:
    raise SystemExit("not enough host memory for a safe run")

t0 = time.time()
ref = pvref.RefSolver(25.0, 25.0, res, pvref.load_pv(scene))
print("reference constructed in %.1f s (grid %.1f s, free grid %.1f s), efree %.9g" % (time.time() - t0, ref.ctor_grid_s, ref.ctor_free_s, ref.efree), flush=True)
assert (ref.gx, ref.gy, ref.T) == (gx, gy, T)
t0 = time.time()
ref.generate(L)
t1 = time.time()
ref.analyze(L)
t2 = time.time()
rres, rdelay = ref.results()
print("reference FDTD %.1f s (%.3e cell-updates/s on one core), analysis %.1f s" % (
    t1 - t0, (gx + 1) * (gy + 1) * T / (t1 - t0), t2 - t1), flush=True)
fs = ref.fs
ref.close()

n_dry = int(np.float32(0.01) * np.float32(fs))
valid = (rdelay < 1e30) & (rdelay + n_dry + 2 <= T - n_dry)
cells = [pv.host_cells(25.0, 25.0, res, e[0], e[2])[1] for e in E]
bad = 0
for fuse in (0, 1):
    s = pv.Solver(25.0, 25.0, res, streaming_analysis=1, stream_fuse=fuse)
    assert np.float32(s.efree) == np.float32(ref.efree), (s.efree, ref.efree)
    s.load_scene(scene)
    s.set_emitters(E)
    s.run(L)
    r, d = s.results()
    t = s.timings()
    s.close()
    msgs = []
    if not same_bits(d, rdelay).all():
        msgs.append("delay: %d cells" % (~same_bits(d, rdelay)).sum())
    for k in (0, 3, 6, 7):
        ne = ~same_bits(r[..., k][valid], rres[..., k][valid])
        if ne.any():
            msgs.append("%s: %d cells" % (NAMES[k], ne.sum()))
    for k in (4, 5):
        ne = ~same_bits(r[..., k], rres[..., k])
        if ne.any():
            msgs.append("%s: %d cells" % (NAMES[k], ne.sum()))
    for (cx, cy) in cells:
        if valid[cx, cy] and not same_bits(r[cx, cy, 1:3], rres[cx, cy, 1:3]).all():
            msgs.append("wet / RT60 at emitter cell (%d, %d): %r vs %r" % (cx, cy, r[cx, cy, 1:3], rres[cx, cy, 1:3]))
    bad += len(msgs)
    print("GPU streaming analysis, fuse %d: %.1f ms stencil + sums, %d of %d cells inside the Q5 mask, %d with an onset: %s" % (
        fuse, t.fdtdMs, valid.sum(), valid.size, (rdelay < 1e30).sum(), "bit-identical to the reference" if not msgs else "; ".join(msgs)),
        flush=True)
sys.exit(1 if bad else 0)
