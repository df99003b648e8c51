"""GPU (-m gpu): parity of the HIP path, called through the C-ABI of libplaneverb_amd.so, against
  (1) the golden vectors generated from the unmodified reference (tests/golden/),
  (2) the pinned oracle (oracle/pv_oracle.c) on seeded random scenes,
  (3) size-independent properties at BASELINE.json's full sizes.

Tolerance: BASELINE.json's north_star allows 1e-4 relative on the per-source outputs; the bar here is BIT-EXACT
float32 (modulo the sign of zero, NaN == NaN) for everything -- fields, pressure history, reconstructed vx/vy impulse
responses, onset delay, occlusion, wet gain, rt60, lowpass, source directivity, listener direction.  rt60 and lowpass
go through log10f / powf, which the device evaluates with glibc 2.35's own algorithms (planeverb_amd/csrc/pv_libm.h,
checked against the host libm for every float by tools/libm_check.cpp); RT60_TOL / LOWPASS_TOL stay as knobs for a
host whose libm is not glibc 2.35.
"""
import os

import numpy as np
import pytest

from conftest import SCENES, golden, needs_experimental, rel_err, same_bits, valid_mask

pytestmark = pytest.mark.gpu

RT60_TOL = 0.0
LOWPASS_TOL = 0.0
NAMES = ["occlusion", "wetGain", "rt60", "lowpass", "dirX", "dirY", "srcDirX", "srcDirY"]


def compare_maps(res, delay, rres, rdelay, T, fs, ctx=""):
    assert same_bits(delay, rdelay).all(), ctx + " delay map"
    valid = valid_mask(rdelay, T, fs)
    for k, nm in enumerate(NAMES):
        m = valid if k not in (4, 5) else np.ones_like(valid)
        a, b = res[..., k][m], rres[..., k][m]
        if k == 2:
            assert rel_err(a, b).max(initial=0) <= RT60_TOL, ctx + " rt60"
        elif k == 3:
            assert rel_err(a, b).max(initial=0) <= LOWPASS_TOL, ctx + " lowpass"
        else:
            assert same_bits(a, b).all(), "%s %s: %d cells differ" % (ctx, nm, int((~same_bits(a, b)).sum()))
    return int(valid.sum())


def compare_output(o, ref8, ctx=""):
    o = o.as_array()
    for k in (0, 1, 4, 5, 6, 7):
        assert same_bits(o[k], ref8[k]).all(), "%s %s %r vs %r" % (ctx, NAMES[k], o[k], ref8[k])
    assert rel_err(o[2], ref8[2]).max() <= RT60_TOL, ctx
    assert rel_err(o[3], ref8[3]).max() <= LOWPASS_TOL, ctx


SMALL = ["g71_smallroom", "g71_shoebox", "g71_bigroom", "g71_hugeroom", "g71_floorplan", "g71_direction",
         "g71_empty", "g71_smallroom_L2", "g96_smallroom_res375"]


@pytest.mark.parametrize("name", SMALL)
def test_golden_small(pvlib, name):
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        assert (s.gx, s.gy, s.T, s.fs) == (gx, gy, T, fs)
        assert np.float32(s.efree) == g["efree"], "FreeGrid energy"
        assert same_bits(s.pulse(), g["pulse"]).all()
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        beta, R = s.material()
        assert np.array_equal(beta, g["beta"]) and same_bits(R, g["R"]).all()
        for i, t in enumerate(g["snap_ts"]):
            assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all(), "recorded pr, step %d" % t
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all(), "IR (pr,vx,vy) at %d,%d" % (cx, cy)
        res, delay = s.results()
        nvalid = compare_maps(res, delay, g["results"], g["delay"], T, fs, name)
        assert nvalid > 100
        for e, ro in zip(g["emitters"], g["emitter_out"]):
            compare_output(s.get_output(e), ro, "%s emitter %s" % (name, e))
        # final fields == last recorded plane + the last pulse sample at the listener
        pr, vx, vy = s.fields()
        last = g["snaps"][list(g["snap_ts"]).index(T - 1)] if (T - 1) in list(g["snap_ts"]) else None
        if last is not None:
            lc = (int(np.float32(g["listener"][0]) / np.float32(s.dx)), int(np.float32(g["listener"][2]) / np.float32(s.dx)))
            exp = last[0].copy()
            exp[lc] += g["pulse"][T - 1]
            assert same_bits(pr, exp).all() and same_bits(vx, last[1]).all() and same_bits(vy, last[2]).all()


@pytest.mark.parametrize("opts", [dict(steps_per_launch=1, tile_rows=30), dict(steps_per_launch=2, tile_rows=28),
                                  dict(steps_per_launch=3, tile_rows=26), dict(steps_per_launch=4, tile_rows=24), dict(steps_per_launch=4, tile_rows=32),
                                  dict(steps_per_launch=6, tile_rows=28), dict(steps_per_launch=8, tile_rows=24),
                                  dict(steps_per_launch=8, tile_rows=40), dict(steps_per_launch=10, tile_rows=40),
                                  dict(steps_per_launch=10, tile_rows=36), dict(steps_per_launch=12, tile_rows=32),
                                  dict(steps_per_launch=8, tile_rows=44), dict(steps_per_launch=12, tile_rows=36),
                                  dict(steps_per_launch=9, tile_rows=42), dict(steps_per_launch=11, tile_rows=36),
                                  dict(steps_per_launch=9, tile_rows=40), dict(steps_per_launch=8, tile_rows=48),
                                  dict(steps_per_launch=12, tile_rows=40), dict(steps_per_launch=10, tile_rows=40, packed_math=0),
                                  dict(steps_per_launch=10, tile_rows=40, merged_launch=0),
                                  dict(dense_history=1), dict(use_graph=1), dict(use_graph=2),
                                  dict(small_grid_kernel=1), dict(small_grid_kernel=2), dict(small_grid_kernel=2, use_graph=2), dict(small_grid_kernel=2, packed_math=0), dict(small_grid_kernel=2, merged_launch=0),
                                  dict(small_grid_kernel=2, merged_launch=0, use_graph=2)])
def test_every_kernel_configuration(pvlib, request, opts):
    """every compiled (K, rows) instantiation and the dense-history mode produce the same bits (tiles outside the product
    library's set run on the experimental build, conftest.PRODUCT_TILES)"""
    g = golden("g71_smallroom")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    lib = request.getfixturevalue("pvlib_exp") if needs_experimental(opts) else pvlib
    with lib.Solver(25.0, 25.0, 275, **opts) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        for i, t in enumerate(g["snap_ts"]):
            assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all()
        res, delay = s.results()
        compare_maps(res, delay, g["results"], g["delay"], T, fs, str(opts))


def test_golden_512_mode_a(pvlib):
    """BASELINE config 2: Shoebox.pv at 512^2 (Mode A: 182.748 m at 275 Hz)"""
    g = golden("g512A_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), 275) as s:
        assert (s.gx, s.gy, s.T) == (512, 512, 435)
        assert np.float32(s.efree) == g["efree"]
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        res, delay = s.results()
        c = g["cells"]
        compare_maps(res[c[:, 0], c[:, 1]], delay[c[:, 0], c[:, 1]], g["cell_results"], g["cell_delay"], T, fs)
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all()
        compare_output(s.get_output(g["emitters"][0]), g["emitter_out"][0])
        want = np.array([1.04278421, 0.0924116895, 0.545365036, 16091.9189, 0.54074657, 0.84118551, -0.503068805,
                         -0.864246428], np.float32)  # SURVEY.md 8c anchor row
        compare_output(s.get_output((95, 0, 97)), want)


def test_golden_512_mode_b(pvlib):
    """BASELINE config 2, Mode B: the 25 m Shoebox at res 2009 (fs 10547, T = 3179): the only oracle-checkable
    case with T >> 435"""
    g = golden("g512B_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(25.0, 25.0, 2009) as s:
        assert (s.gx, s.gy, s.T, s.fs) == (512, 512, 3179, 10547)
        assert np.float32(s.efree) == g["efree"]
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        res, delay = s.results()
        c = g["cells"]
        compare_maps(res[c[:, 0], c[:, 1]], delay[c[:, 0], c[:, 1]], g["cell_results"], g["cell_delay"], T, fs)
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all()
        compare_output(s.get_output(g["emitters"][0]), g["emitter_out"][0])


STACKED = [dict(steps_per_launch=12, tile_rows=196), dict(steps_per_launch=12, tile_rows=210)]


@pytest.mark.parametrize("opts", STACKED)
def test_stacked_tiles_golden_512(pvlib_exp, opts):
    """stacked air tiles (4 waves share one tall tile, boundary faces exchanged through LDS every step) against the
    reference's vectors for BASELINE config 2 (Mode A): the pulse crosses several stacked tiles in x and y"""
    g = golden("g512A_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib_exp.Solver(float(g["size"]), float(g["size"]), 275, **opts) as s:
        assert s.info.tileRows == opts["tile_rows"]
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        res, delay = s.results()
        c = g["cells"]
        compare_maps(res[c[:, 0], c[:, 1]], delay[c[:, 0], c[:, 1]], g["cell_results"], g["cell_delay"], T, fs)
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all()
        compare_output(s.get_output(g["emitters"][0]), g["emitter_out"][0])


@pytest.mark.parametrize("opts", STACKED)
def test_stacked_tiles_match_single_wave_tiles_1024(pvlib_exp, opts):
    """same bits as the single-wave tile kernel on every cell of a 1024^2 grid: final fields, recorded planes, delay
    and result maps (walls inside the pulse's reach, listener near a stacked tile's corner)"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((1024 + 0.5) * dx)
    L = ((196 * 2 + 0.5) * float(dx), 0.0, (40 * 9 + 1.5) * float(dx))
    boxes = [[L[0] + 20.0, L[2] + 9.0, 30.0, 1.0, 0.85], [L[0] - 33.0, L[2] - 4.0, 1.2, 55.0, 0.5],
             [L[0] + 3.0, L[2] - 60.0, 44.0, 2.0, 0.969536]]
    with pvlib_exp.Solver(size, size, 275, steps_per_launch=8, tile_rows=24) as a, \
            pvlib_exp.Solver(size, size, 275, **opts) as b:
        for s in (a, b):
            for box in boxes:
                s.add_geometry(box)
            s.run(L)
        for fa, fb in zip(a.fields(), b.fields()):
            assert same_bits(fa, fb).all()
        for t in (0, 50, 211, 434):
            assert same_bits(a.history_plane(t), b.history_plane(t)).all(), "recorded pr, step %d" % t
        ra, da = a.results()
        rb, db = b.results()
        assert same_bits(da, db).all() and same_bits(ra, rb).all()
        assert (da < 1e30).sum() > 100000


@pytest.mark.parametrize("bands", [2, 3, 5])
def test_row_bands_match_single_launch_sweeps_1024(pvlib, bands):
    """PVA_OPT_ROW_BANDS: every sweep launched as bands of tile rows on their own streams, band b of sweep n+1 ordered
    only behind bands b-1, b, b+1 of sweep n.  Same bits as one launch per sweep on every cell: final fields, recorded
    planes, delay and result maps -- listener ON a band boundary row, walls crossing the boundaries, a second run with
    the listener elsewhere on the same solver"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((1024 + 0.5) * dx)
    with pvlib.Solver(size, size, 275, steps_per_launch=8, tile_rows=24, row_bands=1) as a, \
            pvlib.Solver(size, size, 275, steps_per_launch=8, tile_rows=24, row_bands=bands) as b:
        ntx = -(-1025 // 24)
        edge = (ntx * 1 // bands) * 24  # first row of band 1
        Ls = [((edge + 0.5) * float(dx), 0.0, 400.5 * float(dx)), ((edge - 3 + 0.5) * float(dx), 0.0, 700.5 * float(dx)),
              (150.5 * float(dx), 0.0, 150.5 * float(dx))]
        boxes = [[Ls[0][0] + 1.0, Ls[0][2] + 9.0, 40.0, 1.0, 0.85], [Ls[0][0] - 20.0, Ls[0][2] - 4.0, 1.2, 55.0, 0.5],
                 [Ls[0][0] + 3.0, Ls[0][2] - 30.0, 44.0, 2.0, 0.969536]]
        for s in (a, b):
            for box in boxes:
                s.add_geometry(box)
        for L in Ls:
            a.run(L)
            b.run(L)
            for fa, fb in zip(a.fields(), b.fields()):
                assert same_bits(fa, fb).all()
            for t in (0, 7, 8, 50, 211, 434):
                assert same_bits(a.history_plane(t), b.history_plane(t)).all(), "recorded pr, step %d" % t
            ra, da = a.results()
            rb, db = b.results()
            assert same_bits(da, db).all() and same_bits(ra, rb).all()
            assert (da < 1e30).sum() > 100000
        # the raw stencil on all-non-zero fields
        rng = np.random.default_rng(3)
        f = [(rng.random((1025, 1025), np.float32) - np.float32(0.5)) for _ in range(3)]
        for s in (a, b):
            s.set_fields(*f)
            s.run_steps(40)
        for fa, fb in zip(a.fields(), b.fields()):
            assert same_bits(fa, fb).all()


def random_scene(rng, size, nbox):
    boxes = []
    for _ in range(nbox):
        w, h = (rng.uniform(0.4, 8), rng.uniform(0.4, 1.2)) if rng.random() < 0.5 else (rng.uniform(0.4, 1.2),
                                                                                        rng.uniform(0.4, 8))
        boxes.append([rng.uniform(-1, size + 1), rng.uniform(-1, size + 1), w, h,
                      rng.choice([0.969536, 0.85, 0.5, 0.0, 0.999, rng.uniform(0.05, 0.99)])])
    return np.array(boxes, np.float32)


@pytest.mark.parametrize("seed,size,res", [(1, 25.0, 275), (2, 25.0, 275), (3, 31.7, 300), (4, 18.0, 375),
                                           (5, 25.0, 275), (6, 18.0, 375), (7, 40.0, 275)])
def test_random_scenes_vs_oracle(pvlib, oracle, seed, size, res):
    rng = np.random.default_rng(seed)
    boxes = random_scene(rng, size, 12)
    L = (rng.uniform(1, size - 1), 0.0, rng.uniform(1, size - 1))
    o = oracle.OracleGrid(size, size, res, boxes)
    o.fdtd(L)
    ef = oracle.free_energy(size, size, res)
    rres, rdelay, _ = o.analyze(ef, L)
    hp, hx, hy = o.history()
    with pvlib.Solver(size, size, res) as s:
        assert (s.gx, s.gy, s.T) == (o.gx, o.gy, o.T)
        assert np.float32(s.efree) == np.float32(ef)
        for b in boxes:
            s.add_geometry(b)
        s.run(L)
        # Some random wall layouts make the reference's boundary update diverge (|p| -> inf -> NaN, which the
        # reference then smears over the whole grid through 0 * NaN).  Parity is defined while the reference is
        # finite; a diverged scene is compared up to its last finite step only.
        flat = hp.reshape(o.T, -1)
        finite = np.isfinite(flat).all(1) & (np.abs(np.nan_to_num(flat, nan=np.inf)).max(1) < 1e30)
        tmax = o.T - 1 if finite.all() else int(np.argmin(finite)) - 1
        for t in (0, 3, 17, o.T // 3, o.T - 1):
            if t <= tmax:
                assert same_bits(s.history_plane(t), hp[t]).all(), "pr step %d" % t
        if tmax == o.T - 1:
            for cx, cy in rng.integers(0, o.gx, (6, 2)):
                ir = np.stack([hp[:, cx, cy], hx[:, cx, cy], hy[:, cx, cy]], 1)
                assert same_bits(s.impulse_response(int(cx), int(cy)), ir).all()
            res8, delay = s.results()
            compare_maps(res8, delay, rres, rdelay, o.T, o.fs, "seed %d" % seed)
    o.close()


@pytest.mark.parametrize("listener", [(0.1, 0, 0.1), (24.8, 0, 24.8), (12.5, 0, 0.2), (5.88, 0, 11.24)])
def test_listener_edge_positions_vs_oracle(pvlib, oracle, listener):
    """listener in the corner cells, on an edge, and INSIDE a wall (the pulse is swallowed: beta = 0)"""
    from oracle import pvref
    boxes = pvref.load_pv(os.path.join(SCENES, "SmallRoomScene.pv"))
    o = oracle.OracleGrid(25.0, 25.0, 275, boxes)
    o.fdtd(listener)
    hp, _, _ = o.history()
    rres, rdelay, _ = o.analyze(np.float32(0.0447895788), listener)
    with pvlib.Solver(25.0, 25.0, 275) as s:
        for b in boxes:
            s.add_geometry(b)
        s.run(listener)
        for t in (0, 1, 2, 50, 434):
            assert same_bits(s.history_plane(t), hp[t]).all()
        res8, delay = s.results()
        compare_maps(res8, delay, rres, rdelay, o.T, o.fs, str(listener))
    o.close()


def test_geometry_update_remove_sequence(pvlib, oracle):
    """Add / Update / Remove through the handle API follow GeometryManager + Grid semantics (ids recycled LIFO,
    Update = Remove(old) + Add(new), removing a box clears overlaps: SURVEY Q4)"""
    a = [10, 10, 6, 1, 0.9]
    b = [12, 10, 1, 6, 0.8]
    a2 = [10, 14, 6, 1, 0.7]
    L = (5, 0, 4)
    o = oracle.OracleGrid(25.0, 25.0, 275, None)
    with pvlib.Solver(25.0, 25.0, 275) as s:
        ia = s.add_geometry(a)
        ib = s.add_geometry(b)
        assert (ia, ib) == (0, 1)
        o.add_aabb(a), o.add_aabb(b)
        s.update_geometry(ia, a2)
        o.remove_aabb(a), o.add_aabb(a2)
        s.remove_geometry(ib)
        o.remove_aabb(b)
        assert s.add_geometry(b) == ib  # recycled id
        o.add_aabb(b)
        s.run(L)
        beta, R = s.material()
        ob, oR = o.material()
        assert np.array_equal(beta, ob.astype(np.uint8)) and same_bits(R, oR).all()
        o.fdtd(L)
        hp, _, _ = o.history()
        assert same_bits(s.history_plane(300), hp[300]).all()
        with pytest.raises(pvlib.PlaneverbError):
            s.update_geometry(99, a)
    o.close()


def test_output_sentinels(pvlib):
    with pvlib.Solver(25.0, 25.0, 275) as s:
        s.run((5, 0, 4))
        assert s.get_output((30, 0, 5)).occlusion == -1.0  # off grid: FDTD.cpp:43-47
        assert s.get_output((5, 0, -3)).occlusion == -1.0 or s.get_output((5, 0, -3)).occlusion >= 0
        o = s.get_output((5, 0, 6))
        assert o.occlusion > 0 and o.rt60 != 0


def test_run_survives_a_lost_graph_capture(pvlib, monkeypatch):
    """Small grids replay a captured graph of the run.  Another host thread's hipFree / device-wide synchronisation while the
    solver's stream is capturing invalidates the capture (seen with the live module's worker beside a test thread): the run
    then goes out as plain launches and the next run captures again -- same records either way."""
    g = golden("g71_smallroom")
    # (resident_kernel=2: the replayed graph of tile-kernel launches, which the resident kernel otherwise replaces at this size)
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"]), resident_kernel=2, debug_lose_first_capture=1) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        for attempt in ("capture lost: plain launches", "captured", "replayed"):
            s.run(g["listener"])
            res, delay = s.results()
            gx, gy, T, fs = (int(v) for v in g["dims"])
            compare_maps(res, delay, g["results"], g["delay"], T, fs, attempt)
            for e, ro in zip(g["emitters"], g["emitter_out"]):
                compare_output(s.get_output(e), ro, attempt)


def many_material_boxes(n, per, spacing, width, origin=0.6):
    """n boxes on a grid of `per` columns, every one with its own absorption (0.8 ... 0.99: reflective enough for the
    reference's wall update to stay bounded around small convex obstacles)"""
    i = np.arange(n)
    return np.stack([origin + (i % per) * spacing, origin + (i // per) * spacing, np.full(n, width), np.full(n, width),
                     0.8 + 0.19 * ((i * 37) % n) / n], 1).astype(np.float32)


def test_many_absorption_values_vs_oracle(pvlib, oracle):
    """The reference takes any number of absorption values; rounds 1-2 palettised them and failed a run with more than 127
    alive at once.  Face coefficients are stored as values now (csrc/pv_device.h FaceCoef).  One solver (70^2: the
    replayed-graph path) holds 5, then 400, then 90 distinct values; maps, recorded planes and final fields against the
    oracle each time."""
    boxes = many_material_boxes(400, 20, 1.2, 0.6)
    assert len(np.unique(boxes[:, 4])) == 400
    L = (12.0, 0.0, 12.0)
    ef = oracle.free_energy(25.0, 25.0, 275)

    def check(s, live, ctx):
        s.run(L)
        o = oracle.OracleGrid(25.0, 25.0, 275, live)
        f = o.fdtd(L, want_fields=True)
        assert all(np.isfinite(x).all() for x in f)
        hp, _, _ = o.history()
        for t in (40, 200, 434):
            assert same_bits(s.history_plane(t), hp[t]).all(), "%s: recorded pr, step %d" % (ctx, t)
        for mine, ref in zip(s.fields(), f):
            assert same_bits(mine, ref).all(), ctx + ": final fields"
        rres, rdelay, _ = o.analyze(ef, L)
        o.close()
        res, delay = s.results()
        assert same_bits(delay, rdelay).all(), ctx + ": delay map"
        m = valid_mask(rdelay, 435, 1443)
        for k in (0, 1, 6, 7):
            assert same_bits(res[..., k][m], rres[..., k][m]).all(), "%s: %s" % (ctx, NAMES[k])
        assert rel_err(res[..., 2][m], rres[..., 2][m]).max(initial=0) <= RT60_TOL, ctx + ": rt60"
        assert rel_err(res[..., 3][m], rres[..., 3][m]).max(initial=0) <= LOWPASS_TOL, ctx + ": lowpass"

    with pvlib.Solver(25.0, 25.0, 275) as s:
        ids = [s.add_geometry(b) for b in boxes[:5]]
        check(s, boxes[:5], "5 values")
        ids += [s.add_geometry(b) for b in boxes[5:]]
        check(s, boxes, "400 values")
        for i in ids[90:]:
            s.remove_geometry(i)
        check(s, boxes[:90], "back to 90 values")


def test_many_absorption_values_512_merged_kernel(pvlib, oracle):
    """the same through the merged step kernel's general tiles (512^2: plain launches, 4-wave general tiles in both their
    scalar and packed forms are covered by the tile configurations): 700 obstacles with 700 absorption values"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((512 + 0.5) * dx)
    boxes = many_material_boxes(700, 28, 6.2, 2.2, origin=4.0)
    L = (93.1, 0.0, 90.2)
    o = oracle.OracleGrid(size, size, 275, boxes)
    f = o.fdtd(L, want_fields=True)
    assert all(np.isfinite(x).all() for x in f) and np.abs(f[0]).max() < 1.0
    hist, _, _ = o.history()
    hp = {t: hist[t].copy() for t in (60, 300, 434)}  # (a view into the oracle's memory: gone with close())
    rres, rdelay, _ = o.analyze(oracle.free_energy(size, size, 275), L)
    o.close()
    for opts in ({}, {"steps_per_launch": 12, "tile_rows": 36}):
        with pvlib.Solver(size, size, 275, **opts) as s:
            for b in boxes:
                s.add_geometry(b)
            s.run(L)
            for t in (60, 300, 434):
                assert same_bits(s.history_plane(t), hp[t]).all(), "recorded pr, step %d (%r)" % (t, opts)
            for mine, ref in zip(s.fields(), f):
                assert same_bits(mine, ref).all(), "final fields (%r)" % (opts,)
            res, delay = s.results()
            compare_maps(res, delay, rres, rdelay, 435, 1443, "700 values %r" % (opts,))


def test_absorption_values_come_and_go(pvlib, oracle):
    """A long session that keeps changing one wall's absorption through UpdateGeometry passes 127 distinct values since
    creation with only a handful alive (rounds 1-2 rebuilt their palette here; there is none any more) (the reference
    accepts any number of values), and the results still equal the oracle's for the final scene"""
    g = golden("g71_smallroom")
    L = g["listener"]
    with pvlib.Solver(25.0, 25.0, 275) as s:
        ids = [s.add_geometry(b) for b in g["boxes"]]
        extra = s.add_geometry([18.0, 18.0, 3.0, 0.8, 0.5])
        for i in range(300):
            s.update_geometry(extra, [18.0, 18.0, 3.0, 0.8, 0.05 + i * 0.003])
            if i % 50 == 49:
                s.run(L)  # applies the queued rasterisation (and, eventually, the rebuild)
        s.run(L)
        res, delay = s.results()
    boxes = np.concatenate([g["boxes"], np.array([[18.0, 18.0, 3.0, 0.8, 0.05 + 299 * 0.003]], np.float32)])
    o = oracle.OracleGrid(25.0, 25.0, 275, boxes)
    o.fdtd(L)
    rres, rdelay, _ = o.analyze(oracle.free_energy(25.0, 25.0, 275), L)
    o.close()
    compare_maps(res, delay, rres, rdelay, 435, 1443, "after 300 absorption changes")


def test_dead_tiles_thick_walls_vs_oracle(pvlib, oracle):
    """Tiles whose interior is solid wall (thick walls: a 25 m scene at fine resolution, Mode B) are skipped by runs that
    start from zero fields -- their pr, vx, vy are identically zero.  A 256^2 grid with a 40 m solid block and a thick
    bar (several all-wall tiles), against the oracle: listener beside the block, listener INSIDE the block (the pulse is
    swallowed, the final field still shows its last sample), then the geometry changes on the same solver (block removed:
    sound enters where dead tiles were; a new block where sound was) -- whole maps, recorded planes and final fields."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((256 + 0.5) * dx)
    block = [30.0, 40.0, 40.0, 44.0, 0.9]
    bar = [70.0, 45.0, 12.0, 60.0, 0.5]
    late = [62.0, 18.0, 30.0, 26.0, 0.7]
    ef = oracle.free_energy(size, size, 275)

    def check(s, boxes, L):
        s.run(L)
        o = oracle.OracleGrid(size, size, 275, np.array(boxes, np.float32) if boxes else None)
        f = o.fdtd(L, want_fields=True)
        hp, _, _ = o.history()
        for t in (0, 30, 200, 434):
            assert same_bits(s.history_plane(t), hp[t]).all(), "recorded pr, step %d" % t
        for mine, ref in zip(s.fields(), f):
            assert same_bits(mine, ref).all(), "final fields"
        rres, rdelay, _ = o.analyze(ef, L)
        res, delay = s.results()
        o.close()
        # the solver is re-used: cells WITHOUT an onset keep the previous run's values and the direction walk
        # of such a cell looks at them, while the oracle call starts from a zeroed pool -- compare what a run defines
        assert same_bits(delay, rdelay).all(), "delay map"
        valid, onset = valid_mask(rdelay, 435, 1443), rdelay < 1e30
        for k, nm in enumerate(NAMES):
            m = onset if k in (4, 5) else valid
            assert same_bits(res[..., k][m], rres[..., k][m]).all(), "%s %s" % (L, nm)
        return int(valid.sum())

    with pvlib.Solver(size, size, 275, steps_per_launch=8, tile_rows=24) as s:
        assert np.float32(s.efree) == np.float32(ef)
        ids = [s.add_geometry(block), s.add_geometry(bar)]
        assert check(s, [block, bar], (55.0, 0.0, 30.0)) > 5000
        assert check(s, [block, bar], (30.0, 0.0, 40.0)) >= 0          # listener in the middle of the block
        s.remove_geometry(ids[0])
        assert check(s, [bar], (30.0, 0.0, 40.0)) > 5000              # the block's tiles are alive again
        s.add_geometry(late)
        assert check(s, [bar, late], (30.0, 0.0, 40.0)) > 5000        # a block lands where the field was non-zero
        # raw stepping from arbitrary fields does NOT skip (wall cells may hold anything): same bits as the two-kernel form
        rng = np.random.default_rng(4)
        f0 = [(rng.random((257, 257), np.float32) - np.float32(0.5)) for _ in range(3)]
        with pvlib.Solver(size, size, 275, steps_per_launch=8, tile_rows=24, merged_launch=0) as t:
            t.add_geometry(bar)
            t.add_geometry(late)
            for v in (s, t):
                v.set_fields(*f0)
                v.run_steps(24)
            for a, b in zip(s.fields(), t.fields()):
                assert same_bits(a, b).all()
        assert check(s, [bar, late], (30.0, 0.0, 70.0)) > 5000        # and a run after that starts clean again


@pytest.mark.parametrize("opts,packed", [(dict(steps_per_launch=8, tile_rows=24), None), (dict(steps_per_launch=12, tile_rows=12), "0"),
                                         (dict(steps_per_launch=12, tile_rows=12), "1")])
def test_solid_border_walls_dead_to_the_ghost_row_vs_oracle(pvlib, oracle, opts, packed):
    """A room whose four walls are thick boxes along the grid's borders: the tiles inside the right / bottom wall hold the grid's
    ghost row / column, whose face coefficient is 1, not 0 -- dead all the same since round 5 (the cell across the face is a wall cell
    too).  Whole maps, recorded planes and final fields against the oracle; at K = 12 with the scalar and with the packed general arm
    (PLANEVERB_AMD_GENERAL_PACKED: the second instantiation of the merged kernel that scenes with many wall tiles take)."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((256 + 0.5) * dx)
    w = 12.0
    walls = [[w / 2, size / 2, w, size, 0.8], [size - w / 2, size / 2, w, size, 0.6], [size / 2, w / 2, size, w, 0.7],
             [size / 2, size - w / 2, size, w, 0.9], [40.0, 50.0, 9.0, 14.0, 0.5]]
    ef = oracle.free_energy(size, size, 275)
    old = os.environ.get("PLANEVERB_AMD_GENERAL_PACKED")
    if packed is not None:
        os.environ["PLANEVERB_AMD_GENERAL_PACKED"] = packed
    try:
        with pvlib.Solver(size, size, 275, **opts) as s:
            for b in walls:
                s.add_geometry(b)
            for L in ((30.0, 0.0, 30.0), (size - w - 1.0, 0.0, size - w - 1.0), (size - 2.0, 0.0, 40.0)):  # mid-room, the far corner, INSIDE the wall
                s.run(L)
                o = oracle.OracleGrid(size, size, 275, np.array(walls, np.float32))
                f = o.fdtd(L, want_fields=True)
                hp, _, _ = o.history()
                for t in (0, 60, 300, 434):
                    assert same_bits(s.history_plane(t), hp[t]).all(), "recorded pr, step %d" % t
                for mine, ref in zip(s.fields(), f):
                    assert same_bits(mine, ref).all(), "final fields"
                rres, rdelay, _ = o.analyze(ef, L)
                res, delay = s.results()
                o.close()
                assert same_bits(delay, rdelay).all(), "delay map"
                valid, onset = valid_mask(rdelay, 435, 1443), rdelay < 1e30
                for k, nm in enumerate(NAMES):
                    m = onset if k in (4, 5) else valid
                    assert same_bits(res[..., k][m], rres[..., k][m]).all(), "%s %s" % (L, nm)
    finally:
        if old is None:
            os.environ.pop("PLANEVERB_AMD_GENERAL_PACKED", None)
        else:
            os.environ["PLANEVERB_AMD_GENERAL_PACKED"] = old


def test_step_composition_and_zero_fixed_point(pvlib, pvlib_exp):
    """raw stencil properties: 2n steps == n steps twice (any K), and an all-zero field stays all-zero"""
    rng = np.random.default_rng(11)
    boxes = random_scene(rng, 25.0, 8)
    init = [rng.standard_normal((71, 71)).astype(np.float32) for _ in range(3)]
    outs = []
    for opts in (dict(steps_per_launch=4, tile_rows=32), dict(steps_per_launch=1, tile_rows=30),
                 dict(steps_per_launch=8, tile_rows=24)):
        with (pvlib_exp if needs_experimental(opts) else pvlib).Solver(25.0, 25.0, 275, no_free_grid=1, **opts) as s:
            for b in boxes:
                s.add_geometry(b)
            s.set_fields(*init)
            s.run_steps(37)
            s.run_steps(37)
            a = s.fields()
            s.set_fields(*init)
            s.run_steps(74)
            b2 = s.fields()
            for x, y in zip(a, b2):
                assert same_bits(x, y).all()
            outs.append(a)
            s.set_fields(*[np.zeros((71, 71), np.float32)] * 3)
            s.run_steps(16)
            assert all((f == 0).all() for f in s.fields())
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert same_bits(x, y).all()


def test_single_steps_from_oracle_states(pvlib, oracle):
    """one FDTD.cpp:124-223 step started from the oracle's own dense mid-run states on a walled scene"""
    rng = np.random.default_rng(3)
    boxes = random_scene(rng, 25.0, 10)
    o = oracle.OracleGrid(25.0, 25.0, 275, boxes)
    L = (7.3, 0, 9.1)
    o.fdtd(L)
    hp, hx, hy = o.history()
    pulse = o.pulse()
    lc = o.listener_cell(np.float32(L[0]), np.float32(L[2]))
    with pvlib.Solver(25.0, 25.0, 275, no_free_grid=1) as s:
        for b in boxes:
            s.add_geometry(b)
        for t in (40, 100, 200, 433):
            p0 = hp[t].copy()
            p0[lc] += pulse[t]  # the state after step t includes the injected pulse (FDTD.cpp:234)
            s.set_fields(p0, hx[t], hy[t])
            s.run_steps(1)
            pr, vx, vy = s.fields()
            assert same_bits(pr, hp[t + 1]).all() and same_bits(vx, hx[t + 1]).all() and same_bits(vy, hy[t + 1]).all()
    o.close()


def test_closed_room_isolation_4096(pvlib):
    """BASELINE config 4 at full size: HugeRoom.pv in a 4096^2 grid (Mode A, 1460.737 m at 275 Hz).  The interior
    of a closed room is numerically decoupled from the outside (wall cells hold p = 0, no diagonal coupling), so the
    per-emitter outputs must equal the 71^2 reference run of the same room bit-for-bit."""
    g = golden("g71_hugeroom")
    with pvlib.Solver(1460.737, 1460.737, 275) as s:
        assert (s.gx, s.gy, s.T) == (4096, 4096, 435)
        assert np.float32(s.efree) == g["efree"]
        s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
        s.run(g["listener"])
        for e, ro in zip(g["emitters"], g["emitter_out"]):
            compare_output(s.get_output(e), ro, "4096^2 emitter %s" % e)
        res, delay = s.results()
        # the whole room interior: cells well inside the walls of the 25 m room
        sub, dsub = res[3:66, 3:66], delay[3:66, 3:66]
        compare_maps(sub, dsub, g["results"][3:66, 3:66], g["delay"][3:66, 3:66], 435, 1443, "room interior")
        # nothing outside the reach of the pulse has an onset
        assert (delay[1000:, :] > 1e30).all() and (delay[:, 1000:] > 1e30).all()


def test_second_run_reuses_solver(pvlib):
    """two consecutive runs with different listeners on one solver == fresh solvers (history window moves)"""
    ga, gb = golden("g71_smallroom"), golden("g71_smallroom_L2")
    with pvlib.Solver(25.0, 25.0, 275) as s:
        for b in ga["boxes"]:
            s.add_geometry(b)
        s.run(ga["listener"])
        s.run(gb["listener"])
        for i, t in enumerate(gb["snap_ts"]):
            assert same_bits(s.history_plane(int(t)), gb["snaps"][i][0]).all()
        # cells with an onset in this run are rewritten; cells without keep the previous run's values
        # (Analyzer.cpp:160-165) -- compare only where this run has an onset
        res, delay = s.results()
        assert same_bits(delay, gb["delay"]).all()
        m = valid_mask(gb["delay"], 435, 1443)
        assert same_bits(res[..., 0][m], gb["results"][..., 0][m]).all()
        assert same_bits(res[..., 1][m], gb["results"][..., 1][m]).all()


@pytest.mark.parametrize("size,res,want", [(40.0, 275, 0.028847147), (25.0, 275, 0.0447895788),
                                           (1460.737, 275, 0.0447895788), (25.0, 2009, 0.00686102314)])
def test_free_grid_energy(pvlib, size, res, want):
    """FreeGrid.cpp:71-110 incl. the centre-cell re-truncation quirk (40 m case) and the windowed evaluation on
    large grids; values are the unmodified reference's (tests/test_oracle_vs_ref.py)"""
    with pvlib.Solver(size, size, res) as s:
        assert np.float32(s.efree) == np.float32(want)


@pytest.mark.parametrize("n", [2048, 4096])
def test_exec_trapezoid_equals_product(pvlib, pvlib_exp, n):
    """The air arm with the y-trapezoid's dead lanes switched off (PV_EXEC_TRAPEZOID: EXEC narrowed to lanes [s, 63 - s] before step
    s; compiled into the EXPERIMENTAL build, measured and left out of the product: profiles/r06_exec_mask.txt) against the product
    kernel on seeded random fields -- every cell of every tile non-zero, so a lane switched off one step too early shows in the
    next tile column (the first mask, [s, 62 - s], passed every closed-room test) -- K-step launches plus a remainder launch"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((n + 0.5) * dx)
    rng = np.random.default_rng(7)
    f0 = [(rng.random((n + 1, n + 1), np.float32) - np.float32(0.5)) * np.float32(1e-3) for _ in range(3)]
    got = []
    for lib in (pvlib, pvlib_exp):
        with lib.Solver(size, size, 275) as s:
            s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
            s.set_fields(*f0)
            s.run_steps(2 * s.info.stepsPerLaunch + 5)
            got.append(s.fields())
    for a, b, nm in zip(got[0], got[1], ("pr", "vx", "vy")):
        assert np.isfinite(a).all()
        assert same_bits(a, b).all(), "%s: %d cells differ" % (nm, int((~same_bits(a, b)).sum()))


def test_open_field_8192_vs_oracle_window(pvlib, oracle):
    """BASELINE config 5 at full size (8192^2 open grid, Mode A): in the open field the pressure history and the onset
    map around the listener do not depend on where the listener sits, and -- inside the region the grid edges cannot
    have influenced within T steps -- they equal the oracle's on a 513^2 grid with the listener at its centre."""
    n_small = 512
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size_small = float((n_small + 0.5) * dx)
    c = 256
    Ls = ((c + 0.5) * float(dx), 0.0, (c + 0.5) * float(dx))
    o = oracle.OracleGrid(size_small, size_small, 275, None)
    assert o.listener_cell(np.float32(Ls[0]), np.float32(Ls[2])) == (c, c)
    o.fdtd(Ls)
    hp, _, _ = o.history()
    _, odelay, _ = o.analyze(np.float32(0.0447895788), Ls)
    R = 70  # edge effects of the 513^2 grid need 256 + (256 - R) > 434 steps to get back within R cells
    rng = np.random.default_rng(0)
    cells = rng.integers(1024, 7168, size=(64, 2))  # SURVEY.md 8d config 5 listener cells
    with pvlib.Solver(2921.297, 2921.297, 275) as s:
        assert (s.gx, s.gy, s.T) == (8192, 8192, 435)
        for lx, ly in (cells[0], cells[37]):
            L = ((lx + 0.5) * float(dx), 0.0, (ly + 0.5) * float(dx))
            s.run(L)
            for t in (0, 1, 2, 40, 150, 300, 434):
                plane = s.history_plane(t)
                assert same_bits(plane[lx - R:lx + R + 1, ly - R:ly + R + 1], hp[t][c - R:c + R + 1, c - R:c + R + 1]).all(), t
                far = plane.copy()
                far[max(lx - t - 2, 0):lx + t + 3, max(ly - t - 2, 0):ly + t + 3] = 0
                assert not far.any(), "pressure outside the causal reach of the pulse at step %d" % t
            _, delay = s.results()
            assert same_bits(delay[lx - R:lx + R + 1, ly - R:ly + R + 1], odelay[c - R:c + R + 1, c - R:c + R + 1]).all()
            # mirror symmetry of the onset map about the listener row and column (open field)
            w = delay[lx - 200:lx + 201, ly - 200:ly + 201]
            assert np.array_equal(w, w[::-1, :]) and np.array_equal(w, w[:, ::-1])
    o.close()


@pytest.mark.parametrize("cell", [(479, 1000), (1000, 479 + 48 * 3 - 1), (2030, 2040), (3, 5), (1007, 2047), (960, 960)])
def test_history_window_placement_2048(pvlib, cell):
    """listener at tile ends / grid corners: the (2T+3)-cell history window must hold everything the pulse reaches
    (a run fails loudly with 'history window overflow' otherwise) and results must not depend on the placement"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    lx, ly = cell
    L = ((lx + 0.5) * float(dx), 0.0, (ly + 0.5) * float(dx))
    size = float((2048 + 0.5) * dx)
    with pvlib.Solver(size, size, 275) as s, pvlib.Solver(size, size, 275, dense_history=1) as d:
        s.run(L)
        d.run(L)
        for t in (0, 100, 434):
            assert same_bits(s.history_plane(t), d.history_plane(t)).all()
        rs, ds = s.results()
        rd, dd = d.results()
        assert same_bits(ds, dd).all() and same_bits(rs, rd).all()
        assert (ds < 1e30).sum() > 1000


# ----------------------------------------------------------------------------------------------------------------
# streaming analysis (sparse-emitter mode)
# ----------------------------------------------------------------------------------------------------------------

def check_streaming_against(res, delay, rres, rdelay, T, fs, emitter_cells, ctx=""):
    """every cell: delay, occlusion, lowpass, source directivity, listener direction; emitter cells: all 8"""
    assert same_bits(delay, rdelay).all(), ctx + " delay"
    valid = valid_mask(rdelay, T, fs)
    for k in (0, 6, 7):
        assert same_bits(res[..., k][valid], rres[..., k][valid]).all(), ctx + " " + NAMES[k]
    assert rel_err(res[..., 3][valid], rres[..., 3][valid]).max(initial=0) <= LOWPASS_TOL
    for k in (4, 5):
        assert same_bits(res[..., k], rres[..., k]).all(), ctx + " " + NAMES[k]
    em = np.zeros_like(valid)
    for cx, cy in emitter_cells:
        em[cx, cy] = True
    # wet gain / RT60 exist only at the registered cells
    assert (res[..., 1][~em] == 0).all() and (res[..., 2][~em] == 0).all()
    for cx, cy in emitter_cells:
        if valid[cx, cy]:
            assert same_bits(res[cx, cy, 1], rres[cx, cy, 1]).all(), ctx + " wet at emitter"
            assert rel_err(res[cx, cy, 2], rres[cx, cy, 2]).max() <= RT60_TOL, ctx + " rt60 at emitter"


def fuse_opts(fuse):
    """stream_fuse = 1 needs a tile configuration that has the open-tile kernel (csrc/pv_kernels.hip PV_OPEN_CONFIGS); the default
    tile of small grids has none and the option would be ignored"""
    return dict(stream_fuse=1, steps_per_launch=8, tile_rows=24) if fuse else dict(stream_fuse=0)


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("name", ["g71_smallroom", "g71_hugeroom", "g71_floorplan", "g96_smallroom_res375"])
def test_streaming_mode_small(pvlib, name, fuse):
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    size, res = float(g["size"]), int(g["res"])
    rng = np.random.default_rng(1)
    extra = rng.uniform(0.5, size - 0.5, (12, 3)).astype(np.float32)
    emitters = np.concatenate([g["emitters"], extra])
    with pvlib.Solver(size, size, res, streaming_analysis=1, **fuse_opts(fuse)) as s:
        assert s.info.streamFuse == fuse
        for b in g["boxes"]:
            s.add_geometry(b)
        s.set_emitters(emitters)
        s.run(g["listener"])
        res8, delay = s.results()
        cells = [pvlib.host_cells(size, size, res, e[0], e[2])[1] for e in emitters]
        check_streaming_against(res8, delay, g["results"], g["delay"], T, fs, [c for c in cells if c], name)
        for e, ro in zip(g["emitters"], g["emitter_out"]):
            compare_output(s.get_output(e), ro, name)
        with pytest.raises(pvlib.PlaneverbError):
            s.impulse_response(3, 3)
        # a second run with another emitter set on the same solver
        s.set_emitters(g["emitters"][:1])
        s.run(g["listener"])
        compare_output(s.get_output(g["emitters"][0]), g["emitter_out"][0], name + " rerun")


@pytest.mark.parametrize("fuse", [1, 0])
def test_streaming_mode_512_mode_b(pvlib, fuse):
    """T = 3179: 50 ring passes; checked against the reference vectors of BASELINE config 2 / Mode B (fuse: see
    test_streaming_equals_full_history_1024)"""
    g = golden("g512B_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    c = g["cells"]
    dx = np.float32(343.21) / np.float32(2009) / np.float32(3.5)
    em_pos = np.stack([(c[:, 0] + 0.5) * float(dx), np.zeros(len(c)), (c[:, 1] + 0.5) * float(dx)], 1)
    with pvlib.Solver(25.0, 25.0, 2009, streaming_analysis=1, **fuse_opts(fuse)) as s:
        assert s.info.streamFuse == fuse
        for b in g["boxes"]:
            s.add_geometry(b)
        s.set_emitters(np.concatenate([g["emitters"], em_pos]))
        s.run(g["listener"])
        res, delay = s.results()
        compare_maps(res[c[:, 0], c[:, 1]], delay[c[:, 0], c[:, 1]], g["cell_results"], g["cell_delay"], T, fs, "512B")
        compare_output(s.get_output(g["emitters"][0]), g["emitter_out"][0])


@pytest.mark.parametrize("fuse", [1, 0])
def test_streaming_equals_full_history_1024(pvlib, fuse):
    """fuse = 1: the forward sums of air tiles advance inside the step kernel (PVA_OPT_STREAM_FUSE, csrc/pv_stream.h);
    0: ring + accumulate pass for every tile (round 2's form)"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((1024 + 0.5) * dx)
    L, E = (91.0, 0.0, 91.0), [(95.0, 0.0, 97.0), (100.0, 0.0, 80.0), (60.3, 0.0, 110.9)]
    with pvlib.Solver(size, size, 275) as full, pvlib.Solver(size, size, 275, streaming_analysis=1, **fuse_opts(fuse)) as st:
        assert st.info.streamFuse == fuse
        for s in (full, st):
            s.load_scene(os.path.join(SCENES, "Shoebox.pv"))
        st.set_emitters(E)
        full.run(L)
        st.run(L)
        rf, df = full.results()
        rs, ds = st.results()
        cells = [pvlib.host_cells(size, size, 275, e[0], e[2])[1] for e in E]
        check_streaming_against(rs, ds, rf, df, 435, 1443, cells, "1024")
        for e in E:
            assert same_bits(st.get_output(e).as_array(), full.get_output(e).as_array()).all()


def test_lazy_far_cells_equal_the_full_rewrite_1024(pvlib):
    """Far cells (outside the history window: no onset, default listener direction) are no longer rewritten for the whole map
    on every run (PVA_OPT_LAZY_FAR_CELLS).  A sequence of runs whose window MOVES across the grid -- and comes back -- must
    leave the same maps, records and blocks as the full rewrite of rounds 1-2, after every run: whole-map read-backs, a
    block that straddles the window, single outputs and output queries of cells inside, just outside and far from it."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    n = 1400  # the 873-cell window fits 1.6 times: consecutive windows overlap partly or not at all
    size = float((n + 0.5) * dx)
    cell = lambda cx, cy: ((cx + 0.5) * float(dx), 0.0, (cy + 0.5) * float(dx))
    Ls = [cell(300, 350), cell(1100, 1000), cell(700, 700), cell(300, 350), cell(1390, 5), cell(1100, 1000)]
    probes = [cell(310, 360), cell(5, 5), cell(1399, 1399), cell(760, 300), cell(1100, 990), cell(739, 738), cell(20, 1380)]
    boxes = [[120.0, 130.0, 3.0, 40.0, 0.9], [390.0, 360.0, 40.0, 2.0, 0.8]]
    with pvlib.Solver(size, size, 275, lazy_far_cells=1) as lazy, pvlib.Solver(size, size, 275, lazy_far_cells=0) as full:
        for s in (lazy, full):
            for b in boxes:
                s.add_geometry(b)
        for k, L in enumerate(Ls):
            for s in (lazy, full):
                s.set_output_queries(probes)
                s.run(L)
            ql, qf = lazy.queried_outputs(), full.queried_outputs()
            assert np.array_equal(ql.view(np.uint32), qf.view(np.uint32)), "queries, run %d" % k
            for p_ in probes:
                a, b = lazy.get_output(p_).as_array(), full.get_output(p_).as_array()
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "get_output, run %d" % k
            if k % 2 == 0:  # a block across the window's edge, BEFORE any whole-map read-back of this run
                lcx, lcy = pvlib.host_cells(size, size, 275, L[0], L[2])[0]
                r0, c0 = max(0, min(lcx - 600, n - 301)), max(0, min(lcy + 300, n - 401))
                bl, bf = lazy.results_block(r0, c0, 300, 400), full.results_block(r0, c0, 300, 400)
                assert np.array_equal(bl[0].view(np.uint32), bf[0].view(np.uint32)), "block, run %d" % k
                assert np.array_equal(bl[1].view(np.uint32), bf[1].view(np.uint32)), "delay block, run %d" % k
            if k != 1:  # (run 1 is followed by run 2 without a whole-map read-back in between)
                rl, dl = lazy.results()
                rf, df = full.results()
                assert np.array_equal(dl.view(np.uint32), df.view(np.uint32)), "delay map, run %d" % k
                assert np.array_equal(rl.view(np.uint32), rf.view(np.uint32)), "result map, run %d" % k


def test_near_box_closed_room_2048(pvlib, monkeypatch):
    """Round 6: the analysis' window-wide far frame and direction passes are replaced by passes over the bounding box of the cells a
    run REACHED (a device-side box left by the onset kernel; every other cell of the window is a far cell whose direction its readers
    compute).  A closed room in a 2048^2 grid -- a box of ~75^2 cells in a 873^2-cell window -- with the listener moving inside the room,
    then out into the open grid (a box that is the whole window, far from the previous one), then back: after every run the maps,
    blocks across the room's wall and the window's edge, single outputs and output queries must equal the window-wide passes'
    (PLANEVERB_AMD_NEAR_BOX=0), bit for bit."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    n = 2048
    size = float((n + 0.5) * dx)
    scene = os.path.join(SCENES, "HugeRoom.pv")
    cell = lambda cx, cy: ((cx + 0.5) * float(dx), 0.0, (cy + 0.5) * float(dx))
    Ls = [(5.0, 0.0, 4.0), (20.0, 0.0, 20.0), cell(1500, 1200), (12.0, 0.0, 6.0), cell(900, 40), (5.0, 0.0, 20.0)]
    probes = [(5.0, 0.0, 6.0), (24.0, 0.0, 24.0), (26.5, 0.0, 3.0), cell(1500, 1210), cell(1000, 1000), cell(2047, 2047), cell(80, 80)]
    with pvlib.Solver(size, size, 275) as box:
        monkeypatch.setenv("PLANEVERB_AMD_NEAR_BOX", "0")
        with pvlib.Solver(size, size, 275) as wide:
            monkeypatch.delenv("PLANEVERB_AMD_NEAR_BOX")
            for s in (box, wide):
                s.load_scene(scene)
            for k, L in enumerate(Ls):
                for s in (box, wide):
                    s.set_output_queries(probes)
                    s.run(L)
                qa, qb = box.queried_outputs(), wide.queried_outputs()
                assert np.array_equal(qa.view(np.uint32), qb.view(np.uint32)), "queries, run %d" % k
                for p_ in probes[:4]:
                    a, b = box.get_output(p_).as_array(), wide.get_output(p_).as_array()
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "get_output, run %d" % k
                # a block across the room's wall (the room is the grid's first ~71 x 71 cells) and one across the window's edge,
                # BEFORE any whole-map read-back of this run
                lcx, lcy = pvlib.host_cells(size, size, 275, L[0], L[2])[0]
                for r0, c0 in ((40, 30), (max(0, min(lcx - 500, n - 301)), max(0, min(lcy + 300, n - 401)))):
                    ba, bb = box.results_block(r0, c0, 300, 400), wide.results_block(r0, c0, 300, 400)
                    assert np.array_equal(ba[0].view(np.uint32), bb[0].view(np.uint32)), "block at %d,%d, run %d" % (r0, c0, k)
                    assert np.array_equal(ba[1].view(np.uint32), bb[1].view(np.uint32)), "delay block, run %d" % k
                assert box.timings().reachedCells == wide.timings().reachedCells
                if k != 1:  # (run 1 is followed by run 2 without a whole-map read-back in between)
                    ra, da = box.results()
                    rb, db = wide.results()
                    assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), "delay map, run %d" % k
                    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), "result map, run %d" % k


@pytest.mark.parametrize("fuse", [1, 0])
def test_streaming_equals_full_history_2048_long(pvlib, fuse):
    """The regime the sparse-emitter mode lives in (parity-breadth note): fields non-zero everywhere and a response
    far longer than the ring.  2048^2 with PVA_OPT_NUM_STEPS = 2000 (31 ring passes; the full-history solver keeps all 2000
    planes of the whole grid, 33.6 GB), scattered reflectors around an off-centre listener, three emitters: every cell's
    onset, occlusion, lowpass and both directions, and wet gain / RT60 at the emitters, bit for bit."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((2048 + 0.5) * dx)
    L = (300.0, 0.0, 330.0)
    E = [(310.0, 0.0, 338.0), (420.0, 0.0, 250.0), (150.3, 0.0, 600.9)]
    walls = [[330.0, 340.0, 4.0, 60.0, 0.9], [280.0, 300.0, 80.0, 3.0, 0.7], [500.0, 500.0, 30.0, 30.0, 0.95],
             [120.0, 580.0, 50.0, 5.0, 0.5], [360.0, 200.0, 6.0, 90.0, 0.85]]
    T = 2000
    with pvlib.Solver(size, size, 275, num_steps=T) as full, \
            pvlib.Solver(size, size, 275, num_steps=T, streaming_analysis=1, stream_fuse=fuse) as st:
        assert st.info.streamFuse == fuse
        for s in (full, st):
            for wbox in walls:
                s.add_geometry(wbox)
        st.set_emitters(E)
        full.run(L)
        st.run(L)
        pr, vx, vy = full.fields()
        assert (pr != 0).mean() > 0.9, "the field is meant to be non-zero (almost) everywhere by the end"
        for a, b in zip((pr, vx, vy), st.fields()):
            assert same_bits(a, b).all()
        rf, df = full.results()
        rs, ds = st.results()
        cells = [pvlib.host_cells(size, size, 275, e[0], e[2])[1] for e in E]
        check_streaming_against(rs, ds, rf, df, T, 1443, cells, "2048 long")
        for e in E:
            assert same_bits(st.get_output(e).as_array(), full.get_output(e).as_array()).all()


def test_streaming_open_tiles_beside_or_behind_the_merged_launch(pvlib, monkeypatch):
    """Round 4: the open half tiles of a sweep go out on a stream of their own beside the merged launch (Solver::joinOpen);
    PLANEVERB_AMD_OPEN_STREAM=0 keeps them behind it.  Same fields, maps and emitter records, and the same as the ring form."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((1024 + 0.5) * dx)
    L = (150.0, 0.0, 170.0)
    E = [(160.0, 0.0, 180.0), (240.0, 0.0, 90.0)]
    walls = [[170.0, 175.0, 3.0, 40.0, 0.9], [120.0, 150.0, 50.0, 2.0, 0.7]]
    T = 1500
    got = {}
    for key, fuse, env in (("beside", 1, "1"), ("behind", 1, "0"), ("ring", 0, "1")):
        monkeypatch.setenv("PLANEVERB_AMD_OPEN_STREAM", env)
        with pvlib.Solver(size, size, 275, num_steps=T, streaming_analysis=1, **fuse_opts(fuse)) as st:
            assert st.info.streamFuse == fuse
            for wbox in walls:
                st.add_geometry(wbox)
            st.set_emitters(E)
            for _ in range(2):  # (twice: the events and the pending flag are per run)
                st.run(L)
            r, d = st.results()
            got[key] = (r.copy(), d.copy(), [f.copy() for f in st.fields()], [st.get_output(e).as_array().copy() for e in E])
    for key in ("behind", "ring"):
        assert same_bits(got["beside"][1], got[key][1]).all(), "delay map vs " + key
        for k in range(8):
            assert same_bits(got["beside"][0][..., k], got[key][0][..., k]).all(), "result plane %d vs %s" % (k, key)
        for fa, fb in zip(got["beside"][2], got[key][2]):
            assert same_bits(fa, fb).all(), "final fields vs " + key
        for oa, ob in zip(got["beside"][3], got[key][3]):
            assert same_bits(oa, ob).all(), "emitter record vs " + key


def test_encode_prescan_hint_changes_no_bit(pvlib):
    """Round 4: when a run finds silent cells -- air cells of active tiles whose whole history stays below the audible threshold --
    the NEXT run's pv_encode_kernel looks for an audible sample before anything else, also outside the room regime (the hint is
    written by the run's last kernel).  BASELINE config 2 / Mode B (Shoebox.pv 25 m at 512^2, T = 3179: 132 000 active cells,
    5 470 of them silent): run 1 (no hint) and runs 2, 3 (hint) give the same maps, the reference's vectors at the fixture's
    cells, and the timings report the cells."""
    g = golden("g512B_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    c = g["cells"]
    X, Y = c[:, 0], c[:, 1]
    valid = valid_mask(g["cell_delay"], T, fs)
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        maps = []
        for i in range(3):
            s.run(g["listener"])
            r, d = s.results()
            maps.append((r.copy(), d.copy()))
            t = s.timings()
            assert t.reachedCells == int((d < 1e30).sum())
            assert t.activeCells >= 65536, "outside the room regime: only the hint switches the pre-scan on"
            assert t.activeCells >= t.reachedCells + t.silentCells
            assert t.silentCells * 64 > t.reachedCells, "the scene is meant to raise the hint"
            assert same_bits(d[X, Y], g["cell_delay"]).all()
            for k, nm in enumerate(NAMES):
                m = valid if k not in (4, 5) else np.ones_like(valid)
                assert same_bits(r[X, Y, k][m], g["cell_results"][:, k][m]).all(), "run %d %s" % (i, nm)
        for r, d in maps[1:]:
            assert same_bits(maps[0][1], d).all()
            for k in range(8):
                assert same_bits(maps[0][0][..., k], r[..., k]).all()


@pytest.mark.parametrize("size,res,scene", [(10.0, 500, "ExampleProject.pv"), (10.0, 750, "SmallRoom.pv"),
                                            (3.0, 275, None), (6.5, 375, "SmallRoom.pv")])
def test_resolution_presets_and_tiny_grids_vs_oracle(pvlib, oracle, size, res, scene):
    """pv_HighResolution / pv_ExtremeResolution (PvTypes.h:22-30) and grids of a few cells"""
    from oracle import pvref
    boxes = pvref.load_pv(os.path.join(SCENES, scene)) if scene else np.zeros((0, 5), np.float32)
    L = (size * 0.45, 0.0, size * 0.3)
    o = oracle.OracleGrid(size, size, res, boxes)
    o.fdtd(L)
    ef = oracle.free_energy(size, size, res)
    rres, rdelay, _ = o.analyze(ef, L)
    hp, _, _ = o.history()
    with pvlib.Solver(size, size, res) as s:
        assert (s.gx, s.gy, s.T, s.fs) == (o.gx, o.gy, o.T, o.fs)
        assert np.float32(s.efree) == np.float32(ef)
        for b in boxes:
            s.add_geometry(b)
        s.run(L)
        for t in (0, 5, o.T // 2, o.T - 1):
            assert same_bits(s.history_plane(t), hp[t]).all()
        res8, delay = s.results()
        compare_maps(res8, delay, rres, rdelay, o.T, o.fs, "%g m @ %d" % (size, res))
    o.close()


def test_non_square_grid_closed_room(pvlib):
    """Non-square grids are inconsistent in the reference and are implemented here with stride gy+1
    throughout.  Property: a closed room is decoupled from the outside, so the same room gives identical per-emitter
    outputs on a 25 x 25 m and on a 25 x 14.6 m / 14.6 x 25 m grid."""
    room = golden("g71_bigroom")  # 10 m closed room in the corner of the 25 m grid
    outs = []
    for sx, sy in ((25.0, 25.0), (25.0, 14.6), (14.6, 25.0)):
        with pvlib.Solver(sx, sy, 275) as s:
            assert s.gx == int(np.float32(sx) * (np.float32(1) / np.float32(s.dx)))
            for b in room["boxes"]:
                s.add_geometry(b)
            s.run(room["listener"])
            outs.append(np.stack([s.get_output(e).as_array() for e in room["emitters"][:1]]))
            res, delay = s.results()
            assert res.shape == (s.gx, s.gy, 8)
    compare_output_arrays = lambda a, b: [same_bits(a[:, k], b[:, k]).all() for k in (0, 1, 4, 5, 6, 7)]
    assert all(compare_output_arrays(outs[0], outs[1])) and all(compare_output_arrays(outs[0], outs[2]))
    assert rel_err(outs[0][:, 2], outs[1][:, 2]).max() <= RT60_TOL
    compare_output(type("O", (), {"as_array": lambda self: outs[0][0]})(), room["emitter_out"][0], "square")


def test_listener_outside_grid_and_api_misc(pvlib):
    """a listener off the grid injects nothing (the reference would write outside its array); RunAsync/Sync;
    PlaneverbCreateGrid alias; scene save through the handle API"""
    import ctypes as C
    import tempfile
    with pvlib.Solver(25.0, 25.0, 275) as s:
        s.load_scene(os.path.join(SCENES, "SmallRoomScene.pv"))
        s.run((5, 0, 4))
        before, d0 = s.results()
        s.run_async((-7.0, 0.0, 300.0))
        s.sync()
        after, d1 = s.results()
        assert (d1 > 1e30).all()                      # no onset anywhere
        keep = [0, 1, 2, 3, 6, 7]                     # results untouched (Analyzer.cpp:160-165) ...
        assert same_bits(before[..., keep], after[..., keep]).all()
        assert not same_bits(before[..., 4:6], after[..., 4:6]).all()  # ... but direction is re-encoded for every cell
        assert (s.history_plane(400) == 0).all()
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "scene.pv")
            s.save_scene(p)
            assert np.array_equal(pvlib.load_pv(p), pvlib.load_pv(os.path.join(SCENES, "SmallRoomScene.pv")))
    L = pvlib.lib()
    h = L.PlaneverbCreateGrid(25.0, 25.0, 275, 0)
    assert h
    info = pvlib.PvAmdInfo()
    assert L.PvAmdGetInfo(h, info) == 0 and (info.gx, info.T) == (70, 435)
    assert L.PvAmdSetOption(h, pvlib.PVA_OPT_DENSE_HISTORY, 1) != 0  # options only before first use
    L.PvAmdDestroy(h)
    assert L.PvAmdCreate(25.0, 25.0, 275, 99) is None and "device" in pvlib.last_error()


@pytest.mark.parametrize("K,rows,nseg", [(8, 40, 40), (8, 40, 300), (12, 36, 64), (12, 36, 1000)])
def test_row_streaming_segments_equivalence(pvlib_exp, K, rows, nseg):
    """PVA_OPT_STREAM_ROWS (row-streaming air segments, pv_seg.h): same bits as the tile kernels -- the raw stencil from
    dense random fields with walls, a closed-room run and an open-field run whose pulse crosses many segments (history
    planes, activity flags, every result member)"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    n = 900
    size = float((n + 0.5) * dx)
    rng = np.random.default_rng(0)
    init = [rng.standard_normal((n + 1, n + 1)).astype(np.float32) for _ in range(3)]
    walls = [[60, 70, 20, 1, 0.9], [120, 40, 1, 30, 0.7]]
    cfg = dict(steps_per_launch=K, tile_rows=rows, use_graph=2)
    outs = []
    for m in (0, nseg):
        with pvlib_exp.Solver(size, size, 275, no_free_grid=1, stream_rows=m, **cfg) as s:
            for w in walls:
                s.add_geometry(w)
            s.set_fields(*init)
            s.run_steps(3 * K + 1)  # (the short last launch takes the tile kernel)
            outs.append(s.fields())
    assert all(same_bits(a, b).all() for a, b in zip(*outs))
    for scene, listener in (("HugeRoom.pv", (100.0, 0.0, 90.0)), (None, (150.0, 0.0, 170.0))):
        res = []
        for m in (0, nseg):
            with pvlib_exp.Solver(size, size, 275, stream_rows=m, **cfg) as s:
                if scene:
                    s.load_scene(os.path.join(SCENES, scene))
                s.run(listener)
                s.run(listener)  # twice: the per-tile flags of the first run must not leak into the second
                res.append((s.results(), [s.history_plane(t) for t in (3, 100, 300, 434)]))
        (r0, h0), (r1, h1) = res
        assert same_bits(r0[0], r1[0]).all() and same_bits(r0[1], r1[1]).all()
        assert all(same_bits(a, b).all() for a, b in zip(h0, h1))


@pytest.mark.parametrize("n,strip", [(900, 3), (1250, 1), (1250, 5)])
def test_patch_kernel_equivalence(pvlib_exp, n, strip):
    """PVA_OPT_PATCH_KERNEL (persistent per-CU air-tile kernel with LDS-DMA run-ahead, pv_patch.h): same bits as the
    one-wave-per-tile kernel -- the raw stencil from dense random fields with walls (every tile non-zero, ragged last
    patch: tile columns not a multiple of 4), a closed-room run and an open-field run (history planes, activity flags,
    every result member), two runs in a row (the zero-extent input descriptors of a run's first launch go through the
    DMA path too)"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((n + 0.5) * dx)
    rng = np.random.default_rng(n)
    init = [rng.standard_normal((n + 1, n + 1)).astype(np.float32) for _ in range(3)]
    walls = [[60, 70, 20, 1, 0.9], [120, 40, 1, 30, 0.7]]
    cfg = dict(steps_per_launch=12, tile_rows=36, use_graph=2)
    outs = []
    for m in (0, 1):
        with pvlib_exp.Solver(size, size, 275, no_free_grid=1, patch_kernel=m, patch_strip=strip, **cfg) as s:
            for w in walls:
                s.add_geometry(w)
            s.set_fields(*init)
            s.run_steps(3 * 12 + 5)  # three full launches and a short one
            outs.append(s.fields())
    assert all(same_bits(a, b).all() for a, b in zip(*outs))
    for scene, listener in (("HugeRoom.pv", (100.0, 0.0, 90.0)), (None, (150.0, 0.0, 170.0))):
        res = []
        for m in (0, 1):
            with pvlib_exp.Solver(size, size, 275, patch_kernel=m, patch_strip=strip, **cfg) as s:
                if scene:
                    s.load_scene(os.path.join(SCENES, scene))
                s.run(listener)
                s.run(listener)  # twice: the per-tile flags of the first run must not leak into the second
                res.append((s.results(), [s.history_plane(t) for t in (3, 100, 300, 434)], s.fields()))
        (r0, h0, f0), (r1, h1, f1) = res
        assert same_bits(r0[0], r1[0]).all() and same_bits(r0[1], r1[1]).all()
        assert all(same_bits(a, b).all() for a, b in zip(h0, h1))
        assert all(same_bits(a, b).all() for a, b in zip(f0, f1))


def test_tile_orders_cover_every_tile(pvlib):
    """PVA_OPT_TILE_ORDER only changes which workgroup advances which air tile: linear, XCD bands of tile rows (row- /
    column-major / sub-bands) and XCD strips of tile columns must give the same fields, on a grid whose tile counts are
    not multiples of 8 in either direction"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    n = 1250
    size = float((n + 0.5) * dx)
    rng = np.random.default_rng(3)
    init = [rng.standard_normal((n + 1, n + 1)).astype(np.float32) for _ in range(3)]
    ref = None
    # (order, PVA_OPT_ALTERNATE_SWEEPS): order 3 with its odd launches walking the strips backwards (the default) and not
    # PVA_OPT_XCD_REGIONS): order 3 as strips and as 2 x 4 regions
    for order, alt, reg in ((1, -1, -1), (0, -1, -1), (2, -1, -1), (3, 0, 0), (3, 1, 0), (3, 0, 1), (3, 1, 1), (3, -1, -1), (5, -1, -1)):
        with pvlib.Solver(size, size, 275, no_free_grid=1, steps_per_launch=12, tile_rows=36, use_graph=2,
                          tile_order=order, alternate_sweeps=alt, xcd_regions=reg) as s:
            s.add_geometry([200, 170, 30, 2, 0.8])
            s.set_fields(*init)
            s.run_steps(40)
            f = s.fields()
        if ref is None:
            ref = f
        else:
            assert all(same_bits(a, b).all() for a, b in zip(f, ref)), (order, alt, reg)


def test_history_that_cannot_fit_fails_loudly(pvlib):
    """a 25 m scene at 4096^2 needs T = 25 432 history planes (1.7 TB): refused with a pointer to the streaming mode"""
    with pytest.raises(pvlib.PlaneverbError, match="sparse-emitter mode"):
        pvlib.Solver(25.0, 25.0, 16067)


_GRAPH_SYNC_SCRIPT = r"""
import os, sys
import numpy as np
import torch                      # first, as in bench.py: the library then binds to torch's bundled HIP runtime
torch.cuda.set_device(0)
sys.path.insert(0, sys.argv[1])
import planeverb_amd.api as pv
dx = 343.21 / 275 / 3.5
size = (2048 + 0.5) * dx
listeners = [(5, 4), (8, 8), (5, 4), (8, 8), (12, 6), (15, 15), (20, 5), (5, 20)]
outs = {}
for name, ug in (("graph", 1), ("plain", 2)):
    with pv.Solver(size, size, 275, use_graph=ug) as s:
        s.load_scene(os.path.join(sys.argv[1], "tests", "scenes", "HugeRoom.pv"))
        o = []
        for n, (x, z) in enumerate(listeners):
            if n == 2:
                torch.cuda.synchronize()
            s.run((float(x), 0.0, float(z)))
            o.append(np.stack([s.get_output(e).as_array() for e in [(x, 0.0, z + 2.0), (5.0, 0.0, 6.0)]]))
        outs[name] = np.stack(o).view(np.uint32)
assert (outs["graph"] == outs["plain"]).all()
print("REPLAY-OK")
"""


def test_graph_replay_survives_device_sync(pvlib):
    """Regression: a run replayed from the captured hipGraph must start from zero fields every time.  The reset used
    to be three hipMemsetAsync nodes; after a device-wide synchronize (which torch.cuda.synchronize() issues in
    bench.py, on torch's bundled ROCm 7.0 runtime) later replays skipped them, the previous run's wave kept spreading
    and left the history window.  The run now has no memset / copy nodes at all (zero-extent input descriptors on
    the first launch, parameters uploaded by a kernel); graph replays must equal plain launches for a moving
    listener.  Own process: torch has to be imported before the library, as bench.py does."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GRAPH_SYNC_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REPLAY-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ----------------------------------------------------------------------------------------------------------------
# runs in flight (independent runs per GPU)
# ----------------------------------------------------------------------------------------------------------------

def test_runs_in_flight_match_sequential_runs(pvlib):
    """dist.run_sharded keeps two independent runs in flight on one GPU (two solver instances, two streams); the
    per-emitter outputs must be the bits of the same runs made one after the other"""
    from planeverb_amd import dist as pvd
    scene = os.path.join(SCENES, "HugeRoom.pv")
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((512 + 0.5) * dx)
    listeners = [(5.0, 0.0, 4.0), (8.0, 0.0, 8.0), (12.0, 0.0, 6.0), (15.0, 0.0, 15.0), (20.0, 0.0, 5.0)]

    def emitters_for(k):
        x, _, z = listeners[k]
        return [(x, 0.0, z + 2.0), (5.0, 0.0, 6.0)]

    def make():
        s = pvlib.Solver(size, size, 275)
        s.load_scene(scene)
        return s

    seq = pvd.run_sharded(make, listeners, emitters_for, inflight=1)
    con = pvd.run_sharded(make, listeners, emitters_for, inflight=2)
    tri = pvd.run_sharded(make, listeners, emitters_for, inflight=3)
    assert seq.shape == (5, 2, 8) and np.isfinite(seq[:, :, 0]).all() and (seq[:, :, 0] > 0).all()
    assert same_bits(seq, con).all() and same_bits(seq, tri).all()


def test_output_queries_equal_get_output(pvlib):
    """PvAmdSetOutputQueries / PvAmdGetQueriedOutputs (outputs gathered behind the run's analysis into pinned memory)
    against PvAmdGetOutput, emitter by emitter -- incl. a position outside the grid (the reference's sentinel), a
    wall cell, a second run with other queries, and the golden emitters of the 71^2 scene"""
    g = golden("g71_smallroom")
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"])) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        ems = [tuple(e) for e in g["emitters"]] + [(-3.0, 0.0, 2.0), (12.5, 0.0, 12.5), (24.9, 0.0, 0.1)]
        s.set_output_queries(ems)
        s.run_async(g["listener"])
        s.sync()
        q = s.queried_outputs()
        assert q.shape == (len(ems), 8)
        for i, e in enumerate(ems):
            assert same_bits(q[i], s.get_output(e).as_array()).all(), e
        for i, ro in enumerate(g["emitter_out"]):
            assert same_bits(q[i], np.asarray(ro, np.float32)).all()
        assert q[len(g["emitters"])][0] == -1.0 and not q[len(g["emitters"])][1:].any()
        s.set_output_queries(ems[:2])
        s.run((10.0, 0.0, 10.0))
        q2 = s.queried_outputs()
        for i, e in enumerate(ems[:2]):
            assert same_bits(q2[i], s.get_output(e).as_array()).all()
        s.set_output_queries([])
        s.run(g["listener"])
        assert s.queried_outputs().shape == (0, 8)


def test_batched_runs_match_sequential_runs(pvlib):
    """PvAmdRunBatch: B runs advanced by one launch per K steps (blockIdx.y = run).  Per-emitter outputs through
    dist.run_sharded(batch=...) and, solver by solver, final fields, recorded planes and whole result maps must be
    the bits of the same runs made one at a time -- with DIFFERENT scenes in the solvers of one batch."""
    from planeverb_amd import dist as pvd
    scene = os.path.join(SCENES, "HugeRoom.pv")
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((512 + 0.5) * dx)
    listeners = [(5.0, 0.0, 4.0), (8.0, 0.0, 8.0), (12.0, 0.0, 6.0), (15.0, 0.0, 15.0), (20.0, 0.0, 5.0),
                 (5.0, 0.0, 20.0), (20.0, 0.0, 20.0)]

    def emitters_for(k):
        x, _, z = listeners[k]
        return [(x, 0.0, z + 2.0), (5.0, 0.0, 6.0)]

    def make():
        s = pvlib.Solver(size, size, 275)
        s.load_scene(scene)
        return s

    seq = pvd.run_sharded(make, listeners, emitters_for, inflight=1)
    b3 = pvd.run_sharded(make, listeners, emitters_for, inflight=1, batch=3)
    b4x2 = pvd.run_sharded(make, listeners, emitters_for, inflight=2, batch=4)
    assert seq.shape == (7, 2, 8) and (seq[:, :, 0] > 0).all()
    assert same_bits(seq, b3).all() and same_bits(seq, b4x2).all()

    scenes = ["HugeRoom.pv", "Shoebox.pv", None, "BigRoom.pv"]
    Ls = [(12.0, 0.0, 6.0), (5.0, 0.0, 4.0), (90.0, 0.0, 100.0), (5.0, 0.0, 4.0)]
    # (edge_tiles: the batch's solvers send their grid-border tiles down the air path, the single ones do not)
    for opts, edge in ((dict(), 0), (dict(steps_per_launch=12, tile_rows=36), 0),
                       (dict(steps_per_launch=10, tile_rows=36), 1), (dict(steps_per_launch=8, tile_rows=40), 1),
                       (pvlib.batch_solver_options(512), 0)):
        batch, single = [], []
        for sc in scenes:
            for lst in (batch, single):
                o = dict(opts)
                if lst is batch and edge:
                    o["edge_tiles"] = 1
                elif lst is single:
                    o.pop("edge_tiles", None)
                sv = pvlib.Solver(size, size, 275, **o)
                if sc:
                    sv.load_scene(os.path.join(SCENES, sc))
                lst.append(sv)
        pvlib.run_batch(batch, Ls)
        pvlib.run_batch(batch, Ls)  # a second batch on the same solvers: nothing may be left over from the first
        for sv, L in zip(single, Ls):
            sv.run(L)
        for a, b in zip(batch, single):
            for fa, fb in zip(a.fields(), b.fields()):
                assert same_bits(fa, fb).all()
            for t in (0, 9, 100, 300, 434):
                assert same_bits(a.history_plane(t), b.history_plane(t)).all(), "recorded pr, step %d" % t
            ra, da = a.results()
            rb, db = b.results()
            assert same_bits(da, db).all() and same_bits(ra, rb).all()
        # a solver of the batch keeps working alone afterwards
        batch[1].run(Ls[0])
        single[1].run(Ls[0])
        assert same_bits(batch[1].results()[0], single[1].results()[0]).all()
        for sv in batch + single:
            sv.close()
    # mismatched configurations are refused
    with pvlib.Solver(size, size, 275) as a, pvlib.Solver(size * 2, size * 2, 275) as b:
        with pytest.raises(RuntimeError):
            pvlib.run_batch([a, b], Ls[:2])


@pytest.mark.parametrize("n,cell", [(1024, (512, 512)), (1024, (3, 1000)), (2048, (1500, 600))])
def test_open_field_analysis_window_vs_dense(pvlib, n, cell):
    """open field (walks of hundreds of steps): the windowed analysis -- far cells by formula, window cells by pointer
    jumping -- against the dense-history mode, whose analysis visits every cell with the plain walk"""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((n + 0.5) * dx)
    L = ((cell[0] + 0.5) * float(dx), 0.0, (cell[1] + 0.5) * float(dx))
    with pvlib.Solver(size, size, 275) as s, pvlib.Solver(size, size, 275, dense_history=1) as d:
        s.run(L)
        d.run(L)
        rs, ds = s.results()
        rd, dd = d.results()
        assert same_bits(ds, dd).all() and same_bits(rs, rd).all()
        assert (ds < 1e30).sum() > 50000
        # a second listener on the same solvers: stale results of the first run must be handled alike
        L2 = (L[0] + 37.0 * float(dx), 0.0, max(0.5, L[2] - 211.0 * float(dx)))
        s.run(L2)
        d.run(L2)
        rs, ds = s.results()
        rd, dd = d.results()
        assert same_bits(ds, dd).all() and same_bits(rs, rd).all()


# ----------------------------------------------------------------------------------------------------------------
# edge tiles (tile class 2): grid-edge tiles of empty regions on the air path + edge overrides.  The arm lives in the
# batched kernel only (it slows the air arm of a kernel that contains it); a solver created with
# edge_tiles=1 runs everything through that kernel (a batch of one), which is what these tests exercise.
# ----------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("opts", [dict(steps_per_launch=12, tile_rows=36), dict(steps_per_launch=10, tile_rows=36),
                                  dict(steps_per_launch=8, tile_rows=40)])
@pytest.mark.parametrize("name", ["g71_empty", "g71_direction"])
def test_edge_tiles_golden(pvlib, name, opts):
    """71^2 scenes whose border tiles are edge tiles (x = 0, y = 0, y = gy; the ghost row stays general) against the
    reference's vectors: recorded planes incl. the ghost column, IRs, all eight outputs"""
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), int(g["res"]), edge_tiles=1, **opts) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        for i, t in enumerate(g["snap_ts"]):
            assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all(), "recorded pr, step %d" % t
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all()
        res, delay = s.results()
        compare_maps(res, delay, g["results"], g["delay"], T, fs, name)
        pr, vx, vy = s.fields()
        if (T - 1) in list(g["snap_ts"]):
            last = g["snaps"][list(g["snap_ts"]).index(T - 1)]
            lc = (int(np.float32(g["listener"][0]) / np.float32(s.dx)), int(np.float32(g["listener"][2]) / np.float32(s.dx)))
            exp = last[0].copy()
            exp[lc] += g["pulse"][T - 1]
            assert same_bits(pr, exp).all() and same_bits(vx, last[1]).all() and same_bits(vy, last[2]).all()


@pytest.mark.parametrize("cell", [(30, 40), (20, 1000), (500, 10), (100, 100)])
def test_edge_tiles_match_general_path_1024(pvlib, cell):
    """open 1024^2 field, listener near the x = 0 / y = 0 / y = gy edges and a corner: the wave runs along and into
    the absorbing edges for hundreds of steps.  Edge tiles on / off must give the same bits everywhere: final fields
    (ghost row and column included), recorded planes, delay and result maps."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((1024 + 0.5) * dx)
    L = ((cell[0] + 0.5) * float(dx), 0.0, (cell[1] + 0.5) * float(dx))
    with pvlib.Solver(size, size, 275, steps_per_launch=12, tile_rows=36, edge_tiles=1) as a, \
            pvlib.Solver(size, size, 275, steps_per_launch=12, tile_rows=36, edge_tiles=0) as b:
        a.run(L)
        b.run(L)
        for fa, fb in zip(a.fields(), b.fields()):
            assert same_bits(fa, fb).all()
        for t in (0, 40, 150, 300, 434):
            assert same_bits(a.history_plane(t), b.history_plane(t)).all(), "recorded pr, step %d" % t
        ra, da = a.results()
        rb, db = b.results()
        assert same_bits(da, db).all() and same_bits(ra, rb).all()
        assert (da < 1e30).sum() > 50000


def test_listener_tiles_of_the_small_tile_in_a_replayed_graph(pvlib, oracle):
    """The default tile of the reference's presets (12 rows at 12 steps per launch) has the listener inside the loaded regions
    of 3 x 2 tiles, not 2 x 2: the replayed run graph sized its general-tile launch for four and nobody advanced the other two
    (tools/gpu_fuzz.py seeds 30051 / 30098 / 30172: open scenes, where those tiles are air otherwise).  An open 187^2 grid:
    recorded planes, final fields and maps against the oracle."""
    size, res, L = 61.22308529983864, 300, (45.2920940650875, 0.0, 26.158050733506652)
    o = oracle.OracleGrid(size, size, res, None)
    f = o.fdtd(L, want_fields=True)
    hist, _, _ = o.history()
    hp = {t: hist[t].copy() for t in (3, 11, 12, 17, 40, o.T - 1)}
    rres, rdelay, _ = o.analyze(oracle.free_energy(size, size, res), L)
    T, fs = o.T, o.fs
    o.close()
    with pvlib.Solver(size, size, res) as s:
        assert (s.info.stepsPerLaunch, s.info.tileRows) == (12, 12)
        for rep in range(2):  # captured, then replayed
            s.run(L)
            for t, want in hp.items():
                assert same_bits(s.history_plane(t), want).all(), "recorded pr, step %d (run %d)" % (t, rep)
            for mine, ref in zip(s.fields(), f):
                assert same_bits(mine, ref).all(), "final fields"
            res8, delay = s.results()
            compare_maps(res8, delay, rres, rdelay, T, fs, "open 187^2")
