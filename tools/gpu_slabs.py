#!/usr/bin/env python3
"""Single-grid decomposition (PvAmdCreateSlabs) on ONE device: time per run for S = 1 (plain solver), 2, 4, 8 slabs, with
the bytes exchanged -- what the decomposition costs before any second GPU helps (profiles/r02_slabs.txt).
    python tools/gpu_slabs.py [grid ...]
SLAB_DEVICES="0,1" (default "0") deals the slabs over several devices in turn -- slab i on device i mod n: the peer / xGMI path
of pv_slabs.cpp that no 1-GPU box of the pool can exercise (tools/first_contact.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    size = float((n + 0.5) * dx)
    want = None
    for S in [int(x) for x in os.environ.get("SLABS", "1,2,4,8").split(",")]:
        devs = [int(d) for d in os.environ.get("SLAB_DEVICES", "0").split(",")]
        s = pv.Solver(size, size, 275, slabs=None if S == 1 else [devs[i % len(devs)] for i in range(S)])
        s.load_scene(scene)
        for _ in range(2):
            s.run((5, 0, 4))
        reps = 10
        t0 = time.perf_counter()
        for r in range(reps):
            s.run((5, 0, 4))
        dt = (time.perf_counter() - t0) / reps
        o = s.get_output((5, 0, 6)).as_array()
        if want is None:
            want = o
        assert np.array_equal(o.view(np.uint32), want.view(np.uint32))
        cells = (s.gx + 1) * (s.gy + 1)
        extra = ""
        if S > 1:
            si = s.slab_info()
            extra = "; halo %.2f MB per launch x %d launches, %.1f MB of boundary histories + result blocks per run; HBM per slab %s MB" % (
                si.haloBytesPerLaunch / 1e6, s.timings().stepLaunches, si.exchangeBytesPerRun / 1e6,
                "/".join("%.0f" % (si.deviceBytes[k] / 1e6) for k in range(S)))
        print("grid %d slabs %d: %.3f ms per run (wall, incl. analysis), %.3e cell-updates/s%s" % (
            n, S, dt * 1e3, cells * s.T / dt, extra), flush=True)
        s.close()
