"""GPU (-m gpu): SURVEY.md 8f N4 -- ONE grid decomposed into row slabs (PvAmdCreateSlabs, planeverb_amd/csrc/pv_slabs.*):
S slab solvers (here on one device) that exchange K halo rows of pr, vx, vy per launch, the boundary rows' pressure
histories per run and gather the window block of their per-slab analysis into whole-grid maps for the listener-
direction descent.  The bar: BIT-IDENTICAL to one solver on the whole grid -- final fields, recorded planes, impulse
responses across a slab boundary, delay and all 8 result maps -- and to the reference's golden vectors."""
import os

import numpy as np
import pytest

from conftest import SCENES, golden, same_bits
from test_gpu_parity import compare_maps, compare_output

pytestmark = pytest.mark.gpu

DX = np.float32(343.21) / np.float32(275) / np.float32(3.5)


def size_of(n):
    return float((n + 0.5) * DX)


def cell(cx, cy):
    return ((cx + 0.5) * float(DX), 0.0, (cy + 0.5) * float(DX))


@pytest.mark.parametrize("nslabs", [2, 4])
def test_slabs_match_single_solver_1024(pvlib, nslabs):
    """listener ON the first row of a slab, one row above a boundary, and far from any; walls crossing the boundaries;
    three consecutive runs on the same objects (the history window moves between slabs)"""
    n = 1024
    opts = dict(steps_per_launch=8, tile_rows=24)
    with pvlib.Solver(size_of(n), size_of(n), 275, **opts) as a, \
            pvlib.Solver(size_of(n), size_of(n), 275, slabs=[0] * nslabs, **opts) as b:
        si = b.slab_info()
        assert si.nslabs == nslabs and si.row0[0] == 0 and sum(si.rows[:nslabs]) == n + 1
        edge = si.row0[1]
        assert edge % 24 == 0
        assert np.float32(a.efree) == np.float32(b.efree)
        Ls = [cell(edge, 400), cell(edge - 1, 700), cell(150, 150)]
        boxes = [[Ls[0][0] + 1.0, Ls[0][2] + 9.0, 40.0, 1.0, 0.85], [Ls[0][0] - 20.0, Ls[0][2] - 4.0, 1.2, 55.0, 0.5],
                 [Ls[0][0] + 3.0, Ls[0][2] - 30.0, 44.0, 2.0, 0.969536]]
        for s in (a, b):
            for box in boxes:
                s.add_geometry(box)
        for L in Ls:
            a.run(L)
            b.run(L)
            for fa, fb in zip(a.fields(), b.fields()):
                assert same_bits(fa, fb).all(), "final fields"
            for t in (0, 7, 8, 50, 211, 434):
                assert same_bits(a.history_plane(t), b.history_plane(t)).all(), "recorded pr, step %d" % t
            ra, da = a.results()
            rb, db = b.results()
            assert same_bits(da, db).all(), "delay map"
            for k in range(8):
                assert same_bits(ra[..., k], rb[..., k]).all(), "result plane %d" % k
            assert (da < 1e30).sum() > 100000
            # impulse responses (pr, vx, vy) on both sides of the first slab boundary: vx of the slab's first row is
            # re-derived from the neighbour's last row's pressure history
            lc = int(np.float32(L[2]) / DX)
            for cx in (edge - 1, edge, edge + 1):
                assert same_bits(a.impulse_response(cx, lc + 5), b.impulse_response(cx, lc + 5)).all(), cx
            e = (L[0] + 3.0, 0.0, L[2] + 2.0)
            assert same_bits(a.get_output(e).as_array(), b.get_output(e).as_array()).all()
        assert b.slab_info().haloBytesPerLaunch == (nslabs - 1) * 2 * 3 * 8 * b.info.pitch * 4


def test_slabs_golden_96(pvlib):
    """the reference's own vectors through a 2-slab decomposition (25 m at 375 Hz: 96^2 cells = 5 tile rows of 24, the
    smallest grid two slabs of two tile rows fit in); the room, its walls and the direction walks cross the boundary"""
    g = golden("g96_smallroom_res375")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(25.0, 25.0, 375, slabs=[0, 0], steps_per_launch=8, tile_rows=24) as s:
        assert np.float32(s.efree) == g["efree"]
        assert s.slab_info().row0[1] == 48
        for b in g["boxes"]:
            s.add_geometry(b)
        s.run(g["listener"])
        for i, t in enumerate(g["snap_ts"]):
            assert same_bits(s.history_plane(int(t)), g["snaps"][i][0]).all(), "recorded pr, step %d" % t
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all(), "IR at %d,%d" % (cx, cy)
        res, delay = s.results()
        assert compare_maps(res, delay, g["results"], g["delay"], T, fs, "96^2, 2 slabs") > 100
        for e, ro in zip(g["emitters"], g["emitter_out"]):
            compare_output(s.get_output(e), ro, "emitter %s" % (e,))


def test_slabs_golden_512_mode_a(pvlib):
    """BASELINE config 2 (Shoebox.pv at 512^2, Mode A) through 3 slabs against the reference's vectors"""
    g = golden("g512A_shoebox")
    gx, gy, T, fs = (int(v) for v in g["dims"])
    with pvlib.Solver(float(g["size"]), float(g["size"]), 275, slabs=[0, 0, 0]) as s:
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


@pytest.mark.parametrize("nslabs", [2, 4])
def test_slabs_config4_hugeroom_4096(pvlib, nslabs):
    """BASELINE config 4 at full size through the decomposition: the default 4096^2 tile (K = 12, 36-row tiles), two of the
    eight listeners, records and the 25 m block against the reference's closed-room vectors; slab memory ~ 1/S"""
    g = golden("g71_hugeroom_cfg4")
    with pvlib.Solver(size_of(4096), size_of(4096), 275, slabs=[0] * nslabs) as s:
        assert (s.gx, s.gy, s.T) == (4096, 4096, 435)
        assert np.float32(s.efree) == g["efree"]
        s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
        for i in (0, 6):
            s.run(g["listeners"][i])
            for j in range(2):
                compare_output(s.get_output(g["emitters"][i, j]), g["emitter_out"][i, j], "listener %d" % i)
            res, delay = s.results()
            assert compare_maps(res[:70, :70], delay[:70, :70], g["results"][i], g["delay"][i], 435, 1443) > 3500
        # planes, codes and result maps of a slab cover its own rows only; the pressure-history window does not shrink
        # (any slab may hold all of it): 1.6 of the whole solver's 2.8 GB at this size
        si = s.slab_info()
        assert max(si.deviceBytes[k] for k in range(nslabs)) < 1.7e9 + 1.3e9 / nslabs


def test_slabs_open_field_4096_matches_single_solver(pvlib):
    """open field, listener two rows below a slab boundary of a 3-slab decomposition: the pulse crosses it at once, the
    history window spans two slabs, the direction walks (pointer jumping) cross it"""
    n = 4096
    with pvlib.Solver(size_of(n), size_of(n), 275) as a, pvlib.Solver(size_of(n), size_of(n), 275, slabs=[0, 0, 0]) as b:
        edge = b.slab_info().row0[1]
        L = cell(edge + 2, 2000)
        a.run(L)
        b.run(L)
        for fa, fb in zip(a.fields(), b.fields()):
            assert same_bits(fa, fb).all()
        ra, da = a.results()
        rb, db = b.results()
        assert same_bits(da, db).all()
        for k in range(8):
            assert same_bits(ra[..., k], rb[..., k]).all(), k
        assert (da < 1e30).sum() > 200000


@pytest.mark.parametrize("world,on_device", [(2, False), (3, False), (2, True), (3, True)])
def test_slab_ranks_with_host_exchange_match_single_solver_1024(pvlib, world, on_device):
    """the decomposition with one slab per RANK (PvAmdCreateSlabRank + PvAmdSlab* primitives, whole-grid maps in
    PvAmdSlabRoot*): the exchange schedule of planeverb_amd.dist_slabs (the one that runs over torch.distributed / RCCL with
    one process per GPU; tests/test_dist_cpu.py runs it over gloo) driven in lock-step inside this process, halos, boundary
    histories and result blocks passing through host buffers -- or (on_device) through DEVICE tensors handed over by address,
    the way TorchTransport moves them between RCCL ranks.  Bit-identical to one solver on the whole grid."""
    from planeverb_amd import dist_slabs
    device = None
    if on_device:
        import torch
        device = torch.device("cuda", 0)
    n = 1024
    opts = dict(steps_per_launch=8, tile_rows=24)
    size = size_of(n)
    efree = pvlib.compute_efree(size, size, 275)
    slabs = [pvlib.SlabRank(size, size, 275, 0, r, world, efree, **opts) for r in range(world)]
    root = pvlib.SlabRoot(slabs[0], 0)
    try:
        with pvlib.Solver(size, size, 275, **opts) as a:
            assert np.float32(a.efree) == np.float32(efree)
            edge = (-(-(n + 1) // 24) * 1 // world) * 24
            Ls = [cell(edge, 400), cell(edge - 1, 640)]
            boxes = [[Ls[0][0] + 1.0, Ls[0][2] + 9.0, 40.0, 1.0, 0.85], [Ls[0][0] - 20.0, Ls[0][2] - 4.0, 1.2, 55.0, 0.5]]
            for box in boxes:
                a.add_geometry(box)
                for s in slabs:
                    s.add_geometry(box)
            for L in Ls:
                a.run(L)
                dist_slabs.run_local(slabs, root, L, device=device)
                ra, da = a.results()
                rb, db = root.results()
                assert same_bits(da, db).all(), "delay map"
                for k in range(8):
                    assert same_bits(ra[..., k], rb[..., k]).all(), "result plane %d" % k
                assert (da < 1e30).sum() > 100000
                e = (L[0] + 3.0, 0.0, L[2] + 2.0)
                assert same_bits(a.get_output(e).as_array(), root.get_output(e).as_array()).all()
                fa = a.fields()
                fb = [np.concatenate(p) for p in zip(*[s.solver.fields_local() for s in slabs])]
                for x, y in zip(fa, fb):
                    assert same_bits(x, y).all(), "final fields"
    finally:
        root.close()
        for s in slabs:
            s.close()


@pytest.mark.parametrize("nslabs", [2, 3, 5])
def test_slab_handoff_words_equal_events(pvlib, nslabs, monkeypatch):
    """Round 4: slabs of one device hand over through words in device memory inside pv_halo_push_kernel (write-through rows, block
    count, bounded wait) instead of cross-queue events.  Same bits as the event form (PLANEVERB_AMD_SLAB_HANDOFF=0), as one solver,
    over consecutive runs with a moving listener (the words are cleared per run).  Mode 2 forces the words on groups with more
    slab streams than hardware queues, where a wait may time out: the run is then repeated with events and -- thanks to the abort
    word, which keeps the failed run's analysis away from the result maps -- gives the same bits too."""
    n = 1024
    Ls = [cell(300, 400), cell(511, 700), cell(800, 90)]
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("PLANEVERB_AMD_SLAB_HANDOFF", mode)
        with pvlib.Solver(size_of(n), size_of(n), 275, slabs=[0] * nslabs) as b:
            b.add_geometry([Ls[0][0] + 2.0, Ls[0][2] + 7.0, 60.0, 1.5, 0.8])
            res = []
            for L in Ls:
                b.run(L)
                r, d = b.results()
                res.append((r.copy(), d.copy(), [f.copy() for f in b.fields()]))
            out[mode] = res
    with pvlib.Solver(size_of(n), size_of(n), 275) as a:
        a.add_geometry([Ls[0][0] + 2.0, Ls[0][2] + 7.0, 60.0, 1.5, 0.8])
        for i, L in enumerate(Ls):
            a.run(L)
            ra, da = a.results()
            for mode in ("2", "0"):
                r, d, f = out[mode][i]
                assert same_bits(da, d).all(), "delay map, hand-off %s, run %d" % (mode, i)
                for k in range(8):
                    assert same_bits(ra[..., k], r[..., k]).all(), "result plane %d, hand-off %s, run %d" % (k, mode, i)
                for fa, fb in zip(a.fields(), f):
                    assert same_bits(fa, fb).all(), "final fields, hand-off %s, run %d" % (mode, i)


def test_slab_streams_redealt_at_creation(pvlib, monkeypatch):
    """Round 6: a slab group whose hand-off's dry run times out OR comes through slowly (its slabs' streams take turns instead of running
    beside each other: 5.0-7.3 instead of 2.6-6.2 ms per run, profiles/r06_slabs.txt) gives every slab another stream and tries
    again (QueueClaim::replace).  PLANEVERB_AMD_SLAB_REDEAL=1 forces one such re-deal: the group must come up with the hand-off words
    on, report the re-deal, and give one solver's bits over consecutive runs."""
    n = 1024
    Ls = [cell(300, 400), cell(511, 700), cell(800, 90)]
    monkeypatch.setenv("PLANEVERB_AMD_SLAB_REDEAL", "1")
    with pvlib.Solver(size_of(n), size_of(n), 275) as a, pvlib.Solver(size_of(n), size_of(n), 275, slabs=[0, 0]) as b:
        si = b.slab_info()
        assert si.handoffWords == 1 and si.streamRedeals >= 1 and 0.0 < si.dryRunUsPerSweep < 1000.0, (si.handoffWords, si.streamRedeals, si.dryRunUsPerSweep)
        for s in (a, b):
            s.add_geometry([Ls[0][0] + 2.0, Ls[0][2] + 7.0, 60.0, 1.5, 0.8])
        for i, L in enumerate(Ls):
            a.run(L)
            b.run(L)
            ra, da = a.results()
            rb, db = b.results()
            assert same_bits(da, db).all(), "delay map, run %d" % i
            for k in range(8):
                assert same_bits(ra[..., k], rb[..., k]).all(), "result plane %d, run %d" % (k, i)
            for fa, fb in zip(a.fields(), b.fields()):
                assert same_bits(fa, fb).all(), "final fields, run %d" % i
