"""CPU: the plain-C restatement (oracle/pv_oracle.c) against the golden vectors generated from the unmodified
reference (tests/golden/make_golden.py).  Bit-exact: this is what pins the oracle."""
import numpy as np
import pytest

from conftest import SCENES, golden, same_bits, valid_mask

SMALL = ["g71_smallroom", "g71_shoebox", "g71_bigroom", "g71_hugeroom", "g71_floorplan", "g71_direction",
         "g71_empty", "g71_smallroom_L2", "g96_smallroom_res375"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_reproduces_reference(oracle, name):
    g = golden(name)
    gx, gy, T, fs = (int(v) for v in g["dims"])
    o = oracle.OracleGrid(float(g["size"]), float(g["size"]), int(g["res"]), g["boxes"])
    assert (o.gx, o.gy, o.T, o.fs) == (gx, gy, T, fs)
    assert np.float32(o.dx) == g["dx"] and np.float32(o.dt) == g["dt"]
    assert same_bits(o.pulse(), g["pulse"]).all()
    b, R = o.material()
    assert np.array_equal(b.astype(np.uint8), g["beta"]) and same_bits(R, g["R"]).all()

    L = g["listener"]
    o.fdtd(L)
    hp, hx, hy = o.history()
    for i, t in enumerate(g["snap_ts"]):
        for k, h in enumerate((hp, hx, hy)):
            assert same_bits(h[int(t)], g["snaps"][i][k]).all(), (name, int(t), k)
    for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
        mine = np.stack([hp[:, cx, cy], hx[:, cx, cy], hy[:, cx, cy]], 1)
        assert same_bits(mine, ir).all()

    efree = oracle.free_energy(float(g["size"]), float(g["size"]), int(g["res"]))
    assert np.float32(efree) == g["efree"]
    res, delay, valid = o.analyze(efree, L)
    assert same_bits(delay, g["delay"]).all()
    assert np.array_equal(valid, valid_mask(g["delay"], T, fs))
    for k in range(8):
        m = valid if k not in (4, 5) else np.ones_like(valid)
        assert same_bits(res[..., k][m], g["results"][..., k][m]).all(), (name, k)
    for e, ro in zip(g["emitters"], g["emitter_out"]):
        idx = o.result_index(e)
        assert idx >= 0
        assert same_bits(res.reshape(-1, 8)[idx], ro).all()
    o.close()


def test_known_answers_from_survey(oracle):
    """SURVEY.md section 8c anchors (strict-IEEE reference, 25 m @ 275 Hz, L=(5,0,4), E=(5,0,6))."""
    g = golden("g71_smallroom")
    want = np.array([0.282664806, 0.494537175, 0.789843261, 13214.0439, 0.819445133, 0.573157668, 0.915881634,
                     -0.401448548], np.float32)
    assert same_bits(g["emitter_out"][0], want).all()
    assert g["efree"] == np.float32(0.0447895788)


def test_grid_parameters_table(oracle):
    """SURVEY.md section 8 derived-size table"""
    for res, fs, T in [(275, 1443, 435), (375, 1968, 593), (500, 2625, 791), (750, 3937, 1187), (2009, 10547, 3179)]:
        dx, dt, f = oracle.grid_params(res)
        assert f == fs
        assert oracle.lib().pvo_response_length(f) == T


def test_remove_aabb_restores_air(oracle):
    box = np.array([5, 5, 5, 0.5, 0.97], np.float32)
    o = oracle.OracleGrid(25.0, 25.0, 275, None, with_history=False)
    b0, R0 = o.material()
    o.add_aabb(box)
    b1, _ = o.material()
    assert (b1 != b0).any()
    o.remove_aabb(box)
    b2, R2 = o.material()
    assert np.array_equal(b2, b0) and np.array_equal(R2, R0)
    o.close()


def test_reverb_bus_split(oracle):
    """PvDSPContext.cpp:165-228: buses A/B/C partition the wet gain inside (0.5, 3) s"""
    for rt in (0.6, 0.79, 0.99):
        a, b, c = oracle.find_gains(rt, 0.5)
        assert c == 0.0 and abs((a + b) - 0.5) < 1e-6 and 0 <= a <= 0.5
    for rt in (1.2, 2.0, 2.9):
        a, b, c = oracle.find_gains(rt, 0.5)
        assert a == 0.0 and abs((b + c) - 0.5) < 1e-6
    assert oracle.find_gains(0.3, 0.5) == (1.0, 0.0, 0.0)
    a, b, c = oracle.find_gains(3.5, 0.5)
    assert a == 0.0 and c == 1.0


def test_find_gains_golden(oracle):
    """row 24 pinned: the table generated from the reference's compiled FindGainA/B/C (make_golden.py findgain)"""
    g = golden("g_findgain")
    for (rt, w), want in zip(g["inputs"], g["gains"]):
        assert same_bits(np.array(oracle.find_gains(rt, w), np.float32), want).all(), (rt, w)


def test_pv_scene_files_parse():
    import os
    from oracle import pvref
    for f in sorted(os.listdir(SCENES)):
        boxes = pvref.load_pv(os.path.join(SCENES, f))
        assert boxes.shape[1] == 5 and len(boxes) >= 1


class OpenFieldWindowOracle:
    """Oracle for an open field too large to simulate (BASELINE config 5, 8192^2): a 513^2 window with the listener at
    its centre cell c = 256 (the open field is translation-invariant, so ONE FDTD run serves every listener cell),
    analysed with the LARGE grid's position arithmetic (pvo_analyze_at).  Valid for the (2R+1)^2 cells around the
    listener that no window edge can have influenced within T steps (256 + (256 - R) > 434)."""

    def __init__(self, oracle, n_small=512):
        self.dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
        size_small = float((n_small + 0.5) * self.dx)
        self.c, self.R = n_small // 2, 70
        Ls = ((self.c + 0.5) * float(self.dx), 0.0, (self.c + 0.5) * float(self.dx))
        self.o = oracle.OracleGrid(size_small, size_small, 275, None)
        assert self.o.listener_cell(np.float32(Ls[0]), np.float32(Ls[2])) == (self.c, self.c)
        self.o.fdtd(Ls)
        self.hist_pr = self.o.history()[0]

    def listener_metres(self, cell):
        return ((cell[0] + 0.5) * float(self.dx), 0.0, (cell[1] + 0.5) * float(self.dx))

    def analyze(self, listener_cell, efree=np.float32(0.0447895788)):
        """(res8, delay) of the window for a listener at `listener_cell` of the large grid"""
        res, delay, _ = self.o.analyze(efree, self.listener_metres(listener_cell),
                                       offset=(listener_cell[0] - self.c, listener_cell[1] - self.c))
        return res, delay

    def close(self):
        self.o.close()


def test_window_oracle_with_offset_reproduces_reference(oracle):
    """pvo_analyze_at pinned: the reference itself on an open 640^2 grid with the listener off-centre at (352, 300)
    against the oracle's 513^2 window + cell offset, on all 141 x 141 cells both can vouch for, all 8 result members"""
    g = golden("g640_open_offset")
    lc = tuple(int(v) for v in g["listener_cell"])
    w = OpenFieldWindowOracle(oracle)
    res, delay = w.analyze(lc, g["efree"])
    c, R = w.c, w.R
    w.close()
    assert R == int(g["R"])
    sl = (slice(c - R, c + R + 1), slice(c - R, c + R + 1))
    assert same_bits(delay[sl], g["delay"]).all()
    T, fs = int(g["dims"][2]), int(g["dims"][3])
    valid = valid_mask(g["delay"], T, fs)
    assert valid.sum() > 15000
    for k in range(8):
        m = valid if k not in (4, 5) else np.ones_like(valid)
        assert same_bits(res[sl][..., k][m], g["results"][..., k][m]).all(), k


def test_config5_records_fixture_matches_the_window_oracle(oracle):
    """tests/golden/g8192_open_cfg5.npz (what bench.py --open-field --grid 8192 verifies its runs against) is what the
    pinned window oracle gives: re-derived here for three of the 64 listener cells"""
    g = golden("g8192_open_cfg5")
    assert np.array_equal(g["cells"], np.random.default_rng(0).integers(1024, 7168, size=(64, 2)))
    w = OpenFieldWindowOracle(oracle)
    for i in (0, 17, 63):
        res, _ = w.analyze(tuple(int(v) for v in g["cells"][i]), g["efree"])
        assert same_bits(res[w.c + 16, w.c], g["emitter_out"][i, 0]).all()
        assert same_bits(res[w.c, w.c + 16], g["emitter_out"][i, 1]).all()
    w.close()
