"""GPU (-m gpu): round 4's kernels through the C-ABI.

* the resident kernel (csrc/pv_resident.hip: one launch per run, tiles handed over between workgroups inside the launch)
  against the golden vectors of the unmodified reference, the pinned oracle and the replayed graph of tile-kernel launches;
* the forms of the decay-time pass (csrc/pv_rt60.hip) against each other and the golden vectors;
* two solvers taking turns on one sequence of iterations (PvAmdRunAsyncAfter: what the live module does with two iterations
  in flight) against one solver running the sequence;
* PvAmdTimings.reachedCells, PvAmdClockProbe.

Everything bit-exact float32 (modulo the sign of zero, NaN == NaN), as in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

from conftest import SCENES, golden, same_bits, valid_mask
from test_gpu_parity import NAMES, SMALL, compare_maps, compare_output

pytestmark = pytest.mark.gpu


def fields_equal(a, b):
    return all(same_bits(x, y).all() for x, y in zip(a.fields(), b.fields()))


def maps_equal(a, b, ctx=""):
    ra, da = a.results()
    rb, db = b.results()
    assert same_bits(da, db).all(), ctx + " delay"
    for k, nm in enumerate(NAMES):
        assert same_bits(ra[..., k], rb[..., k]).all(), "%s %s: %d cells differ" % (ctx, nm, int((~same_bits(ra[..., k], rb[..., k])).sum()))


@pytest.mark.parametrize("name", SMALL)
def test_resident_kernel_golden_small(pvlib, name):
    """the reference's vectors on the 71^2 / 96^2 grids, through the resident kernel (the default there)"""
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        assert s.info.residentKernel == 1, "the resident kernel must be what runs the reference's presets"
        for b in g["boxes"]:
            s.add_geometry(b)
        for rep in range(2):  # (the second run re-uses flags, planes and result maps)
            s.run(g["listener"])
            for i, t in enumerate(g["snap_ts"]):
                assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all(), "recorded pr, step %d" % t
            for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
                assert same_bits(s.impulse_response(cx, cy), ir).all(), "IR (pr,vx,vy) at %d,%d" % (cx, cy)
            res, delay = s.results()
            compare_maps(res, delay, g["results"], g["delay"], T, fs, name)
            for e, ref8 in zip(g["emitters"], g["emitter_out"]):
                compare_output(s.get_output(e), ref8, name)
        assert s.timings().stepLaunches == -(-T // 12)


@pytest.mark.parametrize("res,scene", [(275, "SmallRoomScene.pv"), (500, "FloorPlanScene.pv"), (750, "SmallRoomScene.pv"),
                                       (1000, "DirectionTester.pv"), (400, None)])
def test_resident_kernel_equals_tile_kernel_graph(pvlib, res, scene):
    """resident kernel vs the replayed graph of tile-kernel launches on the presets: fields, history planes, every map, for
    listeners in the open, in corners, on the grid's edges and inside a wall, incl. geometry changes between runs"""
    size = 25.0
    with pvlib.Solver(size, size, res) as r, pvlib.Solver(size, size, res, resident_kernel=2) as g:
        assert r.info.residentKernel == 1 and g.info.residentKernel == 0
        ids = []
        for s in (r, g):
            if scene:
                s.load_scene(os.path.join(SCENES, scene))
        listeners = [(5.0, 0.0, 4.0), (0.1, 0.0, 0.1), (size - 0.2, 0.0, size - 0.2), (12.5, 0.0, 0.2), (0.2, 0.0, 17.0),
                     (7.3, 0.0, 19.1)]
        for i, L in enumerate(listeners):
            if i == 3:
                ids = [(s.add_geometry([9.0, 9.0, 3.0, 5.0, 0.6]), s.add_geometry([16.0, 5.0, 1.0, 8.0, 0.95])) for s in (r, g)]
            if i == 5:
                for s, pair in zip((r, g), ids):
                    s.remove_geometry(pair[0])
            r.run(L)
            g.run(L)
            assert fields_equal(r, g), "final fields, listener %r" % (L,)
            for t in (0, 1, 11, 12, 13, r.T // 3, r.T - 1):
                assert same_bits(r.history_plane(t), g.history_plane(t)).all(), "recorded pr, step %d, listener %r" % (t, L)
            maps_equal(r, g, "listener %r" % (L,))


@pytest.mark.parametrize("env", [{"PLANEVERB_AMD_RESIDENT_XCD": "0"}, {"PLANEVERB_AMD_RESIDENT_XCD_TARGET": "9"}])
def test_resident_kernel_hand_off_modes(pvlib, monkeypatch, env):
    """Grids of up to 32 tiles hand their tiles over through ONE XCD's L2 (the blocks that run on the solver's XCD claim the
    tiles).  PLANEVERB_AMD_RESIDENT_XCD=0: the placement-independent hand-off at the same size.  ..._XCD_TARGET=9: an XCD that
    does not exist -- no block claims a tile, the host sees it, repeats the run in the placement-independent mode and stays
    there.  Same vectors either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = golden("g71_smallroom")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        assert s.info.residentKernel == 1
        for b in g["boxes"]:
            s.add_geometry(b)
        for rep in range(3):
            s.run(g["listener"])
            res, delay = s.results()
            compare_maps(res, delay, g["results"], g["delay"], T, fs, "%r run %d" % (env, rep))
            for i, t in enumerate(g["snap_ts"]):
                assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all(), "recorded pr, step %d" % t


def _random_scenes_vs_oracle(lib, oracle, **opts):
    rng = np.random.default_rng(2024)
    ran = 0
    for case in range(6):
        res = int(rng.choice([275, 300, 375, 420, 550]))
        size = float(rng.choice([25.0, 18.0, 31.0]))
        nb = int(rng.integers(0, 9))
        boxes = np.zeros((nb, 5), np.float32)
        for b in boxes:
            b[:] = [rng.uniform(0, size), rng.uniform(0, size), rng.uniform(0.3, 7), rng.uniform(0.3, 7),
                    rng.choice([0.0, 0.5, 0.9, 0.97, 1.0])]
        L = (float(rng.uniform(0.5, size - 0.5)), 0.0, float(rng.uniform(0.5, size - 0.5)))
        o = oracle.OracleGrid(size, size, res, boxes)
        f = o.fdtd(L, want_fields=True)
        if not all(np.isfinite(x).all() for x in f):  # (a layout on which the reference's own update diverges)
            o.close()
            continue
        ef = oracle.free_energy(size, size, res)
        rres, rdelay, _ = o.analyze(ef, L)
        hp, _, _ = o.history()
        with lib.Solver(size, size, res, **opts) as s:
            if not s.info.residentKernel:
                o.close()
                continue
            for b in boxes:
                s.add_geometry(b)
            s.run(L)
            for mine, ref in zip(s.fields(), f):
                assert same_bits(mine, ref).all(), "case %d final fields" % case
            for t in (0, 7, 12, 100, o.T - 1):
                assert same_bits(s.history_plane(t), hp[t]).all(), "case %d recorded pr, step %d" % (case, t)
            res8, delay = s.results()
            compare_maps(res8, delay, rres, rdelay, o.T, o.fs, "case %d" % case)
            ran += 1
        o.close()
    assert ran >= 3, ran


def test_resident_kernel_random_scenes_vs_oracle(pvlib, oracle):
    """seeded random box scenes at 70^2 ... 140^2, resident kernel against the pinned oracle"""
    _random_scenes_vs_oracle(pvlib, oracle)


def test_resident_kernel_concurrent_solvers_and_budget(pvlib):
    """several solvers' resident launches in flight at once (each waits for its own blocks only); a solver whose launch does
    not fit the device's budget of co-resident blocks falls back to the replayed graph by itself -- same results"""
    size, res = 25.0, 750  # 80 workgroups per run
    scene = os.path.join(SCENES, "SmallRoomScene.pv")
    L = [(5.0, 0.0, 4.0), (20.0, 0.0, 20.0), (12.0, 0.0, 7.0), (3.0, 0.0, 22.0), (18.0, 0.0, 3.0), (9.0, 0.0, 14.0)]
    solvers = [pvlib.Solver(size, size, res) for _ in L]
    try:
        for s in solvers:
            s.load_scene(scene)
        for rnd in range(3):
            for s, l in zip(solvers, L):
                s.run_async(l)
            for s in solvers:
                s.sync()
        with pvlib.Solver(size, size, res, resident_kernel=2) as g:
            g.load_scene(scene)
            for s, l in zip(solvers, L):
                g.run(l)
                maps_equal(s, g, "listener %r" % (l,))
                assert fields_equal(s, g)
    finally:
        for s in solvers:
            s.close()


@pytest.mark.parametrize("name", ["g71_smallroom", "g71_floorplan", "g96_smallroom_res375"])
@pytest.mark.parametrize("lanes", [16, 4, 1])
def test_decay_time_forms_golden(pvlib, name, lanes):
    """all forms of the wet gain / decay time pass reproduce the reference's vectors (the choice is made on the device by
    the number of reachable cells; PVA_OPT_RT60_LANES forces one)"""
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"]), rt60_lanes=lanes) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        res, delay = s.results()
        compare_maps(res, delay, g["results"], g["delay"], T, fs, "%s lanes %d" % (name, lanes))


def test_decay_time_forms_agree_512_mode_b(pvlib):
    """BASELINE config 2 / Mode B (512^2, T = 3179: every cell of the room is reached, ~100 000 impulse responses): the
    lane-per-cell form (the device's choice here), beside and behind the encode pass, against the reference's vectors and against
    the sixteen- and the four-lane form"""
    g = golden("g512B_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    maps = []
    for lanes, fork in ((0, 1), (16, 1), (4, 1), (1, 0)):
        with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"]), rt60_lanes=lanes, analysis_fork=fork) as s:
            for b in g["boxes"]:
                s.add_geometry(b)
            s.run(g["listener"])
            for e, ref8 in zip(g["emitters"], g["emitter_out"]):
                compare_output(s.get_output(e), ref8, "512 mode B lanes %d" % lanes)
            res, delay = s.results()
            c = g["cells"]  # the sample of result cells the fixture holds: (X, Y) pairs
            X, Y = c[:, 0], c[:, 1]
            assert same_bits(delay[X, Y], g["cell_delay"]).all()
            valid = valid_mask(g["cell_delay"], T, fs)
            for k, nm in enumerate(NAMES):
                m = valid if k not in (4, 5) else np.ones_like(valid)
                assert same_bits(res[X, Y, k][m], g["cell_results"][:, k][m]).all(), "512 mode B lanes %d %s" % (lanes, nm)
            maps.append((res, delay))
            if lanes == 0:
                assert s.timings().reachedCells == int((delay < 1e30).sum()) > 100000
    for other in maps[1:]:
        assert same_bits(maps[0][1], other[1]).all()
        for k, nm in enumerate(NAMES):
            assert same_bits(maps[0][0][..., k], other[0][..., k]).all(), nm


def test_two_solvers_taking_turns_equal_one_solver(pvlib):
    """PvAmdRunAsyncAfter: iterations alternate between two solvers, each run enqueued while the previous one is still in
    flight; what an iteration leaves untouched (cells without an onset: the reference's persistent m_results) is
    carried over on the device.  Every map after every iteration equals one solver running the whole sequence."""
    _taking_turns(pvlib)


def _taking_turns(pvlib, **pair_opts):
    size, res = 25.0, 375
    scene = os.path.join(SCENES, "FloorPlanScene.pv")
    # listeners that see different parts of the plan (so that cells lose their onset from one iteration to the next), and
    # geometry that comes and goes
    seq = [((3.0, 0.0, 3.0), None), ((22.0, 0.0, 22.0), None), ((3.0, 0.0, 22.0), ("add", [12.5, 12.5, 20.0, 1.0, 0.99])),
           ((22.0, 0.0, 3.0), None), ((12.5, 0.0, 5.0), ("add", [12.5, 8.0, 1.0, 12.0, 0.99])), ((12.5, 0.0, 20.0), None),
           ((3.0, 0.0, 3.0), ("remove", 0)), ((20.0, 0.0, 12.0), None)]
    with pvlib.Solver(size, size, res) as one, pvlib.Solver(size, size, res, **pair_opts) as a, \
            pvlib.Solver(size, size, res, **pair_opts) as b:
        for s in (one, a, b):
            s.load_scene(scene)
        added = {one: [], a: [], b: []}
        pair, prev = (a, b), None
        pending = []  # (solver, map of `one` after the same iteration)
        for i, (L, change) in enumerate(seq):
            if change:
                for s in (one, a, b):
                    if change[0] == "add":
                        added[s].append(s.add_geometry(change[1]))
                    else:
                        s.remove_geometry(added[s][change[1]])
            one.run(L)
            want = one.results()
            s = pair[i & 1]
            if prev is None:
                s.run_async(L)
            else:
                s.run_async_after(prev, L)  # (prev's run is still in flight)
            prev = s
            pending.append((s, want, i))
            if len(pending) == 2:  # collect the older of the two in flight
                t, w, j = pending.pop(0)
                t.sync()
                got = t.results()
                assert same_bits(got[1], w[1]).all(), "iteration %d delay" % j
                for k, nm in enumerate(NAMES):
                    assert same_bits(got[0][..., k], w[0][..., k]).all(), "iteration %d %s" % (j, nm)
        t, w, j = pending.pop(0)
        t.sync()
        got = t.results()
        for k, nm in enumerate(NAMES):
            assert same_bits(got[0][..., k], w[0][..., k]).all(), "iteration %d %s" % (j, nm)


def test_reached_cells_and_clock_probe(pvlib):
    g = golden("g71_smallroom")
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        assert s.timings().reachedCells == int((g["delay"] < 1e30).sum())
    mhz, by_memtime = pvlib.clock_probe(0)
    assert 500.0 < mhz < 4000.0 and by_memtime > 0.0


# ----------------------------------------------------------------------------------------------------------------------
# The one-launch analysis (csrc/pv_fused.hip, PVA_OPT_FUSED_ANALYSIS = 1): an arm of the EXPERIMENTAL build -- workers that draw
# items from a ticket counter, phases ordered by counters, agent-scope hand-offs inside the launch.  It re-schedules
# Analyzer::AnalyzeResponses (Analyzer.cpp:48-104: the cell loop, then the direction loop that needs every cell's delay and
# occlusion); its per-cell bodies are the separate kernels' own functions, so every map must equal theirs -- and the reference's
# vectors -- bit for bit.
# ----------------------------------------------------------------------------------------------------------------------
def test_fused_analysis_refused_by_the_product_build(pvlib):
    with pytest.raises(pvlib.PlaneverbError, match="experimental build"):
        with pvlib.Solver(25.0, 25.0, 275, fused_analysis=1) as s:
            s.run((5.0, 0.0, 4.0))


@pytest.mark.parametrize("name", SMALL)
def test_fused_analysis_golden_small(pvlib_exp, name):
    """the reference's vectors on the 71^2 / 96^2 grids: whole result and delay maps and the emitters' records, two runs back
    to back (the second finds the control words the first left)"""
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib_exp.Solver(float(g["size"]), float(g["size"]), int(g["res"]), fused_analysis=1) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        for rep in range(2):
            s.run(g["listener"])
            res, delay = s.results()
            compare_maps(res, delay, g["results"], g["delay"], T, fs, "%s run %d" % (name, rep))
            for e, ref8 in zip(g["emitters"], g["emitter_out"]):
                compare_output(s.get_output(e), ref8, name)
        assert s.timings().reachedCells == int((g["delay"] < 1e30).sum())


def test_fused_analysis_random_scenes_vs_oracle(pvlib_exp, oracle):
    """test_resident_kernel_random_scenes_vs_oracle's scenes (walls, closed pockets, several absorptions) with the analysis as one
    launch, against the pinned oracle"""
    _random_scenes_vs_oracle(pvlib_exp, oracle, fused_analysis=1)


def test_fused_analysis_two_iterations_in_flight(pvlib_exp):
    """two solvers taking turns, each run's carry pass inside the fused launch (FusedArgs::carrySrc: cells without an onset take
    the other solver's record), against ONE solver with the separate kernels running the whole sequence"""
    _taking_turns(pvlib_exp, fused_analysis=1)


@pytest.mark.parametrize("workers", ["1", "2", "3"])
@pytest.mark.parametrize("res,scene", [(275, "SmallRoomScene.pv"), (750, "FloorPlanScene.pv")])
def test_fused_analysis_any_number_of_workers(pvlib_exp, monkeypatch, workers, res, scene):
    """"no deadlock whatever the number of resident workgroups": the launch capped at 1, 2 and 3 workers (tickets are handed out
    in phase order, so whoever waits, waits for items that running workers hold) -- same maps as the separate kernels; 750 Hz
    takes the four-lane decay-time form (64 cells per item), 275 Hz the sixteen-lane one"""
    L = (5.0, 0.0, 4.0)
    with pvlib_exp.Solver(25.0, 25.0, res, fused_analysis=0) as ref:
        ref.load_scene(os.path.join(SCENES, scene))
        ref.run(L)
        want = ref.results()
    monkeypatch.setenv("PLANEVERB_AMD_FUSED_WORKERS", workers)
    with pvlib_exp.Solver(25.0, 25.0, res, fused_analysis=1) as s:
        s.load_scene(os.path.join(SCENES, scene))
        for rep in range(2):
            s.run(L)
            got = s.results()
            assert same_bits(got[1], want[1]).all(), "delay, run %d" % rep
            for k, nm in enumerate(NAMES):
                assert same_bits(got[0][..., k], want[0][..., k]).all(), "%s, run %d" % (nm, rep)
