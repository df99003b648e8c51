import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import planeverb_amd.api as pv
scene = os.path.join("tests", "scenes", "SmallRoomScene.pv")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 275
outs = []
for fused in (0, 1):
    with pv.Solver(25.0, 25.0, res, fused_analysis=fused) as s:
        s.load_scene(scene)
        t0 = time.perf_counter()
        s.run((5.0, 0.0, 4.0))
        print("fused", fused, "first run s", time.perf_counter() - t0, flush=True)
        for _ in range(3): s.run((5.0, 0.0, 4.0))
        t = s.timings()
        print("  analysis ms", t.analysisMs, "fdtd", t.fdtdMs, "reached", t.reachedCells, "active", t.activeCells, flush=True)
        outs.append(s.results())
r0, d0 = outs[0]; r1, d1 = outs[1]
print("delay equal", np.array_equal(d0, d1), "res equal (bits)", np.array_equal(r0.view(np.uint32), r1.view(np.uint32)))
for k in range(8):
    a, b = r0[..., k], r1[..., k]
    ne = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b)) & ~((a == 0) & (b == 0))
    print(k, int(ne.sum()))
