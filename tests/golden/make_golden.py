#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  The fixtures are data:
inputs (scene boxes, positions) and the reference's outputs (pulse table, material planes, field snapshots,
IR traces, result / delay maps).  No reference source text is stored.

    python tests/golden/make_golden.py small      # 25 m @ 275 Hz scenes (seconds)
    python tests/golden/make_golden.py modeA512   # Shoebox, 512^2 at 275 Hz   (~1 min, 4 GB)
    python tests/golden/make_golden.py modeB512   # Shoebox, 25 m at res 2009  (~3 min, 27 GB)
    python tests/golden/make_golden.py cfg4       # HugeRoom, the 8 listeners of BASELINE config 4 (seconds)
    python tests/golden/make_golden.py open_offset  # open 640^2 field, listener off-centre (~1 min, 3 GB)
    python tests/golden/make_golden.py cfg5       # 8192^2 open field: records of the 64 seeded listener cells (pinned oracle, ~5 min)
    python tests/golden/make_golden.py cells      # raw reference Cells of a few IRs (GetImpulseResponse layout)
    python tests/golden/make_golden.py findgain   # FindGainA/B/C table from PlaneverbDSP's compiled context file
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pvref  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

SMALL = {
    "smallroom": "DemoFiles/SmallRoomScene.pv",
    "shoebox": "Shoebox.pv",
    "bigroom": "BigRoom.pv",
    "hugeroom": "HugeRoom.pv",
    "floorplan": "DemoFiles/FloorPlanScene.pv",
    "direction": "DirectionTester.pv",
    "empty": None,
}


def probes(gx, gy, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, gx, n), rng.integers(0, gy, n)], 1).astype(np.int32)


def run(name, scene, size, res, listener, emitters, snap_ts, nprobe, full_map, sample_cells=0):
    boxes = pvref.load_pv(os.path.join(REF, scene)) if scene else np.zeros((0, 5), np.float32)
    r = pvref.RefSolver(size, size, res, boxes)
    t_f = r.generate(listener)
    t_a = r.analyze(listener)
    res8, delay = r.results()
    b, R = r.material()
    d = dict(
        boxes=boxes, size=np.float32(size), res=np.int32(res), listener=np.array(listener, np.float32),
        emitters=np.array(emitters, np.float32),
        dims=np.array([r.gx, r.gy, r.T, r.fs], np.int32), dx=np.float32(r.dx), dt=np.float32(r.dt),
        efree=np.float32(r.efree), pulse=r.pulse(),
        emitter_out=np.stack([r.output(e) for e in emitters]),
        ref_seconds=np.array([t_f, t_a, r.ctor_grid_s, r.ctor_free_s]),
    )
    if full_map:
        d.update(beta=b.astype(np.uint8), R=R, results=res8, delay=delay)
    else:
        rng = np.random.default_rng(7)
        cells = np.stack([rng.integers(0, r.gx, sample_cells), rng.integers(0, r.gy, sample_cells)], 1)
        # bias half of the sample towards the region the wave reaches
        lc = (int(listener[0] / r.dx), int(listener[2] / r.dx))
        near = np.stack([np.clip(lc[0] + rng.integers(-60, 60, sample_cells), 0, r.gx - 1),
                         np.clip(lc[1] + rng.integers(-60, 60, sample_cells), 0, r.gy - 1)], 1)
        cells = np.concatenate([cells, near]).astype(np.int32)
        d.update(cells=cells, cell_results=res8[cells[:, 0], cells[:, 1]],
                 cell_delay=delay[cells[:, 0], cells[:, 1]])
    if snap_ts:
        snaps = [np.stack(r.snapshot(t)) for t in snap_ts]
        d.update(snap_ts=np.array(snap_ts, np.int32), snaps=np.stack(snaps))
    if nprobe:
        pc = probes(r.gx, r.gy, nprobe, 3)
        d.update(probe_cells=pc, probe_ir=np.stack([r.ir(x, y) for x, y in pc]))
    r.close()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(name, "->", path, "%.1f kB" % (os.path.getsize(path) / 1e3), "fdtd %.2fs analysis %.2fs" % (t_f, t_a))


# 8 listener positions (x, z metres) inside HugeRoom.pv's 25 m room; emitter A = listener +
# (0, 2) m, emitter B = (5, 0, 6).  The room is closed, so the 71^2 (25 m) run equals the 4096^2 Mode A run bit for bit
# for every cell of the room (closed-room isolation) -- this file is what bench.py and the -m gpu tests compare with.
CFG4_LISTENERS = [(5, 4), (8, 8), (12, 6), (15, 15), (20, 5), (5, 20), (20, 20), (12.5, 18)]


def run_cfg4():
    boxes = pvref.load_pv(os.path.join(REF, "HugeRoom.pv"))
    r = pvref.RefSolver(25.0, 25.0, 275, boxes)
    listeners, emitters, outs, maps, delays = [], [], [], [], []
    for x, z in CFG4_LISTENERS:
        L = (float(x), 0.0, float(z))
        E = [(float(x), 0.0, float(z) + 2.0), (5.0, 0.0, 6.0)]
        # a fresh Analyzer pool per listener is what a fresh context gives; the harness re-zeroes it
        r.close()
        r = pvref.RefSolver(25.0, 25.0, 275, boxes)
        r.generate(L)
        r.analyze(L)
        res8, delay = r.results()
        listeners.append(L)
        emitters.append(E)
        outs.append(np.stack([r.output(e) for e in E]))
        maps.append(res8)
        delays.append(delay)
    b, R = r.material()
    d = dict(boxes=boxes, size=np.float32(25.0), res=np.int32(275), dims=np.array([r.gx, r.gy, r.T, r.fs], np.int32),
             efree=np.float32(r.efree), listeners=np.array(listeners, np.float32),
             emitters=np.array(emitters, np.float32), emitter_out=np.stack(outs), results=np.stack(maps),
             delay=np.stack(delays), beta=b.astype(np.uint8))
    # the "RT60 bucket" of every record: the reference's compiled FindGainA/B/C on (rt60, wetGain) (row 24)
    d["bus_gains"] = np.array([[pvref.find_gains(o[2], o[1]) for o in rec] for rec in d["emitter_out"]], np.float32)
    r.close()
    path = os.path.join(OUT, "g71_hugeroom_cfg4.npz")
    np.savez_compressed(path, **d)
    print("cfg4 ->", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


def run_open_offset():
    """Open field, listener off-centre in a 640^2 grid: the result / delay maps of the 141 x 141 cells around the
    listener, which no grid edge can have influenced within T steps.  Pins oracle.pvo_analyze_at (the oracle run on a
    513^2 window with a cell offset, used for BASELINE config 5 at 8192^2) against the compiled reference."""
    n, lc, R = 640, (352, 300), 70
    r = pvref.RefSolver(1, 1, 275, None, False)
    dx = np.float32(r.dx)
    r.close()
    size = float((n + 0.5) * dx)
    L = ((lc[0] + 0.5) * float(dx), 0.0, (lc[1] + 0.5) * float(dx))
    r = pvref.RefSolver(size, size, 275, None)
    assert (r.gx, r.gy) == (n, n)
    r.generate(L)
    r.analyze(L)
    res8, delay = r.results()
    sl = (slice(lc[0] - R, lc[0] + R + 1), slice(lc[1] - R, lc[1] + R + 1))
    d = dict(n=np.int32(n), size=np.float32(size), listener=np.array(L, np.float32), listener_cell=np.array(lc, np.int32),
             R=np.int32(R), efree=np.float32(r.efree), results=res8[sl], delay=delay[sl],
             dims=np.array([r.gx, r.gy, r.T, r.fs], np.int32))
    r.close()
    path = os.path.join(OUT, "g640_open_offset.npz")
    np.savez_compressed(path, **d)
    print("open_offset ->", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


def run_cells():
    """Planeverb::GetImpulseResponse hands out raw 16-byte Cells {pr, vx, vy, short b, short by} (PvTypes.h:106-121,
    FDTD.cpp:60-79,226-230).  Two cases: the static sandbox scene, and the same scene after a box over the x = 0 / y = 0
    corner was added and removed again (RemoveAABB restores `by` with its own rule, Grid.cpp:281-290)."""
    boxes = pvref.load_pv(os.path.join(REF, SMALL["smallroom"]))
    L = (5.0, 0.0, 4.0)
    cells = np.array([(14, 16), (14, 11), (0, 5), (5, 0), (0, 0), (3, 3), (70, 10), (10, 70), (33, 20), (1, 1), (2, 0),
                      (0, 2), (40, 40)], np.int32)
    corner = np.array([0.5, 0.5, 2.0, 2.0, 0.5], np.float32)
    d = dict(boxes=boxes, listener=np.array(L, np.float32), cells=cells, corner_box=corner)
    for tag in ("static", "removed"):
        r = pvref.RefSolver(25.0, 25.0, 275, boxes, with_free_grid=False)
        if tag == "removed":
            r.add_aabb(corner)
            r.remove_aabb(corner)
        r.generate(L)
        d["ir_" + tag] = np.stack([r.ir_cells(x, y) for x, y in cells])  # [ncells, T, 16] bytes
        r.close()
    path = os.path.join(OUT, "g71_smallroom_cells.npz")
    np.savez_compressed(path, **d)
    print("cells ->", path, "%.1f kB" % (os.path.getsize(path) / 1e3))


def run_cfg5():
    """BASELINE config 5 (8192^2 open field, Mode A): the two emitter records (listener + (16, 0) and + (0, 16) cells) of
    all 64 seeded listener cells.  No CPU run can hold an 8192^2 grid; the open field is translation-invariant, so the
    pinned C restatement runs ONE 513^2 window and analyses it with the large grid's position arithmetic
    (pvo_analyze_at, itself pinned against the compiled reference by g640_open_offset.npz).  bench.py --open-field
    --grid 8192 compares its timed runs with these."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pvoracle
    from test_oracle_golden import OpenFieldWindowOracle
    w = OpenFieldWindowOracle(pvoracle)
    cells = np.random.default_rng(0).integers(1024, 7168, size=(64, 2))
    out = np.zeros((64, 2, 8), np.float32)
    for i, (lx, ly) in enumerate(cells):
        res, _ = w.analyze((int(lx), int(ly)))
        out[i, 0] = res[w.c + 16, w.c]
        out[i, 1] = res[w.c, w.c + 16]
        print(i, out[i, 0, :4], flush=True)
    w.close()
    path = os.path.join(OUT, "g8192_open_cfg5.npz")
    np.savez_compressed(path, cells=cells.astype(np.int32), emitter_out=out, efree=np.float32(0.0447895788))
    print("cfg5 ->", path)


def findgain_inputs():
    """(rt60, wet) sweep for SURVEY.md 8a row 24: dense in rt60 incl. the 0.5 / 1.0 / 3.0 s bucket edges and their
    float neighbours, the analysis' degenerate values (0, negative, inf, NaN), a few wet gains"""
    f32 = np.float32
    rt = list(np.linspace(0.05, 4.0, 791, dtype=np.float32))
    for e in (0.5, 1.0, 3.0):
        rt += [np.nextafter(f32(e), f32(0)), f32(e), np.nextafter(f32(e), f32(10))]
    rt += [f32(0), f32(-1.5), f32(np.inf), f32(-np.inf), f32(np.nan), f32(1e-30), f32(1e30)]
    wet = [f32(0), f32(0.0924116895), f32(0.5), f32(0.700975895), f32(1.73174453), f32(25.0)]
    return np.array([(r, w) for r in rt for w in wet], np.float32)


def run_findgain():
    x = findgain_inputs()
    y = np.array([pvref.find_gains(r, w) for r, w in x], np.float32)
    path = os.path.join(OUT, "g_findgain.npz")
    np.savez_compressed(path, inputs=x, gains=y)
    print("findgain ->", path, len(x), "pairs, %.1f kB" % (os.path.getsize(path) / 1e3))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what == "findgain":
        return run_findgain()
    if what == "cells":
        return run_cells()
    if what == "cfg5":
        return run_cfg5()
    if what == "cfg4":
        return run_cfg4()
    if what == "open_offset":
        return run_open_offset()
    if what == "small":
        for name, scene in SMALL.items():
            run("g71_" + name, scene, 25.0, 275, (5, 0, 4), [(5, 0, 6), (12, 0, 9), (20.5, 0, 3.2)],
                [0, 1, 2, 9, 50, 200, 434], 16, True)
        # a second listener position and the mid resolution preset
        run("g71_smallroom_L2", SMALL["smallroom"], 25.0, 275, (9.3, 0, 8.1), [(5, 0, 6), (3, 0, 9)],
            [10, 434], 8, True)
        run("g96_smallroom_res375", SMALL["smallroom"], 25.0, 375, (5, 0, 4), [(5, 0, 6)], [100], 8, True)
    elif what == "modeA512":
        dx = pvref.RefSolver(1, 1, 275, None, False).dx
        run("g512A_shoebox", "Shoebox.pv", 182.748, 275, (91, 0, 91), [(95, 0, 97)], None, 8, False, 256)
    elif what == "modeB512":
        run("g512B_shoebox", "Shoebox.pv", 25.0, 2009, (5, 0, 4), [(5, 0, 6)], None, 4, False, 256)


if __name__ == "__main__":
    main()
