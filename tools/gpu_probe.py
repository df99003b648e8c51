#!/usr/bin/env python3
"""Quick on-GPU diagnostic: parity of the HIP path against the golden fixtures, then stencil throughput.
Development aid (the judged checks are tests/ -m gpu and bench.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    with np.errstate(all="ignore"):
        e = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    e[(a == b) | (np.isnan(a) & np.isnan(b))] = 0
    return e


def check_scene(name, **opts):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    gx, gy, T, fs = g["dims"]
    s = pv.Solver(float(g["size"]), float(g["size"]), int(g["res"]), **opts)
    print("==", name, "dims", (s.gx, s.gy, s.T, s.fs), "ref", (gx, gy, T, fs), "K", s.info.stepsPerLaunch,
          "tile", s.info.tileRows, s.info.tileCols, "MB", s.info.deviceBytes >> 20)
    print("  efree", s.efree, float(g["efree"]), "same" if s.efree == float(g["efree"]) else "DIFF")
    print("  pulse same:", bool(same(s.pulse(), g["pulse"]).all()))
    for b in g["boxes"]:
        s.add_geometry(b)
    L = g["listener"]
    s.run(L)
    if "beta" in g.files:
        beta, R = s.material()
        print("  material same:", bool((beta == g["beta"]).all()), bool(same(R, g["R"]).all()))
    if "snap_ts" in g.files:
        for i, t in enumerate(g["snap_ts"]):
            p = s.history_plane(int(t))
            ok = same(p, g["snaps"][i][0])
            print("  pr plane t=%d same: %s (bad %d, maxabs %.3g)" % (t, bool(ok.all()), int((~ok).sum()),
                                                                        float(np.abs(p - g["snaps"][i][0]).max())))
    if "probe_cells" in g.files:
        bad = 0
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            mine = s.impulse_response(cx, cy)
            bad += int((~same(mine, ir)).sum())
        print("  IR probes (pr,vx,vy) mismatching samples:", bad)
    res, delay = s.results()
    if "results" in g.files:
        rres, rdelay = g["results"], g["delay"]
        nd, ndry, ncut = int(0.005 * fs), int(0.01 * fs), int(0.01 * fs)
        valid = (rdelay < 1e30) & (rdelay + ndry + 2 <= T - ncut)
        print("  delay same:", bool(same(delay, rdelay).all()), " valid cells", int(valid.sum()))
        for k, nm in enumerate(["occ", "wet", "rt60", "lowpass", "dirx", "diry", "sdx", "sdy"]):
            m = valid if k not in (4, 5) else np.ones_like(valid)
            e = relerr(res[..., k][m], rres[..., k][m])
            print("   %-8s bit-same %6d/%6d  max rel err %.3g" % (nm, int(same(res[..., k][m], rres[..., k][m]).sum()),
                                                                 int(m.sum()), float(e.max()) if e.size else 0))
    else:
        cells = g["cells"]
        mine = res[cells[:, 0], cells[:, 1]]
        rd = g["cell_delay"]
        nd, ndry, ncut = int(0.005 * fs), int(0.01 * fs), int(0.01 * fs)
        valid = (rd < 1e30) & (rd + ndry + 2 <= T - ncut)
        print("  sampled cells:", len(cells), "valid", int(valid.sum()), "delay same",
              bool(same(delay[cells[:, 0], cells[:, 1]], rd).all()))
        for k, nm in enumerate(["occ", "wet", "rt60", "lowpass", "dirx", "diry", "sdx", "sdy"]):
            e = relerr(mine[valid, k], g["cell_results"][valid, k])
            print("   %-8s max rel err %.3g" % (nm, float(e.max()) if e.size else 0))
    for e, ro in zip(g["emitters"], g["emitter_out"]):
        o = s.get_output(e).as_array()
        print("  emitter", e, "max rel err %.3g" % float(relerr(o, ro).max()))
    t = s.timings()
    print("  fdtd %.3f ms  analysis %.3f ms  launches %d" % (t.fdtdMs, t.analysisMs, t.stepLaunches))
    s.close()


def perf(n, K, rows, dense=0, steps=None):
    dx = 343.21 / 275 / 3.5
    size = (n + 0.5) * dx
    opts = dict(dense_history=dense)
    if K:
        opts.update(steps_per_launch=K, tile_rows=rows)  # 0 = the solver's own choice for the grid size
    if steps:
        opts["num_steps"] = steps
    try:
        s = pv.Solver(size, size, 275, **opts)
    except pv.PlaneverbError as e:
        print("perf n=%d K=%d rows=%d: %s" % (n, K, rows, e))
        return
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    L = (5, 0, 4)
    s.run(L)
    best = None
    for _ in range(3):
        s.run(L)
        t = s.timings()
        if best is None or t.fdtdMs < best.fdtdMs:
            best = t
    cells = (s.gx + 1) * (s.gy + 1)
    ups = cells * s.T / (best.fdtdMs * 1e-3)
    K, rows = s.info.stepsPerLaunch, s.info.tileRows
    print("perf n=%d K=%d rows=%d dense=%d T=%d: fdtd %.2f ms  analysis %.2f ms  %.3e upd/s  algorithmic %.2f TB/s "
          "(%.0f%% of 8 TB/s)  MB=%d" % (n, K, rows, dense, s.T, best.fdtdMs, best.analysisMs, ups, ups * 24 / 1e12,
                                         ups * 24 / 8e12 * 100, s.info.deviceBytes >> 20))
    # raw stencil (no history, no pulse) from random fields
    rng = np.random.default_rng(0)
    shp = (s.gx + 1, s.gy + 1)
    s.set_fields(*(rng.standard_normal(shp).astype(np.float32) * 1e-3 for _ in range(3)))
    s.run_steps(64)
    t0 = time.time()
    s.run_steps(256)
    t = s.timings()
    ups = cells * 256 / (t.fdtdMs * 1e-3)
    print("   raw stencil random fields: %.2f ms/256 steps %.3e upd/s algorithmic %.2f TB/s" % (
        t.fdtdMs, ups, ups * 24 / 1e12))
    s.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "perf"]
    print("devices:", pv.device_count())
    if "parity" in what:
        for nm in ["g71_smallroom", "g71_hugeroom", "g71_empty", "g71_floorplan", "g96_smallroom_res375"]:
            check_scene(nm)
        check_scene("g71_smallroom", steps_per_launch=2, tile_rows=28)
        check_scene("g71_smallroom", steps_per_launch=8, tile_rows=24)
        check_scene("g71_smallroom", dense_history=1)
        check_scene("g512A_shoebox")
    if "parityB" in what:
        check_scene("g512B_shoebox")
    if "perf" in what:
        perf(4096, 0, 0)
        perf(4096, 0, 0, dense=1)
        perf(2048, 0, 0); perf(1024, 0, 0); perf(512, 0, 0); perf(70, 0, 0)
        perf(8192, 0, 0)
        perf(4096, 8, 24)
