"""GPU (-m gpu): BASELINE.json's configs 3, 4 and 5 at their FULL sizes, compared with the reference bit for bit.

* configs 3 / 4 (BigRoom.pv at 2048^2, HugeRoom.pv at 4096^2, Mode A): the interior of a closed room is numerically
  decoupled from everything outside it (wall cells hold p = 0 and the stencil has no diagonal coupling), so the
  reference's 71^2 (25 m) run of the same room -- tests/golden/g71_bigroom.npz, g71_hugeroom_cfg4.npz, generated from
  the compiled reference by tests/golden/make_golden.py -- gives the expected values of every cell of the 25 m block
  at any Mode A size.  All 8 listener positions of config 4, both emitters each.
* config 5 (8192^2 open field): the open field is translation-invariant; the oracle runs a 513^2 window and analyses
  it with the large grid's position arithmetic (pvo_analyze_at, pinned against the reference by
  tests/test_oracle_golden.py::test_window_oracle_with_offset_reproduces_reference).  8 of the 64 listener cells.
"""
import os

import numpy as np
import pytest

from conftest import SCENES, golden, same_bits
from test_gpu_parity import compare_maps, compare_output

pytestmark = pytest.mark.gpu

DX = np.float32(343.21) / np.float32(275) / np.float32(3.5)


def mode_a_size(n):
    return float((n + 0.5) * DX)


def test_config4_hugeroom_4096_all_listeners(pvlib):
    """BASELINE config 4: the 8 listener positions, 2 emitters each (the 16 records bench.py gathers), plus the whole
    25 m block of the result and delay maps for every listener, on ONE re-used solver (as the bench re-uses it)"""
    g = golden("g71_hugeroom_cfg4")
    with pvlib.Solver(mode_a_size(4096), mode_a_size(4096), 275) as s:
        assert (s.gx, s.gy, s.T) == (4096, 4096, 435)
        assert np.float32(s.efree) == g["efree"]
        s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
        for i, L in enumerate(g["listeners"]):
            s.set_output_queries(g["emitters"][i])
            s.run(L)
            q = s.queried_outputs()
            for j in range(2):
                assert same_bits(q[j], g["emitter_out"][i, j]).all(), "listener %d emitter %d: %r vs %r" % (
                    i, j, q[j], g["emitter_out"][i, j])
                compare_output(s.get_output(g["emitters"][i, j]), g["emitter_out"][i, j], "listener %d" % i)
                # row 24: the reverb-bus split ("RT60 bucket") of the GPU's record vs the reference's compiled FindGain*
                bus = np.array(pvlib.reverb_bus_gains(float(q[j][2]), float(q[j][1])), np.float32)
                assert same_bits(bus, g["bus_gains"][i, j]).all(), (i, j, bus, g["bus_gains"][i, j])
            res, delay = s.results()
            n = compare_maps(res[:70, :70], delay[:70, :70], g["results"][i], g["delay"][i], 435, 1443,
                             "listener %d, 25 m block" % i)
            assert n > 3500
            assert (delay[80:, :] > 1e30).all() and (delay[:, 80:] > 1e30).all()  # nothing leaves the closed room


@pytest.mark.parametrize("K,rows,nseg", [(8, 40, 2048), (12, 36, 1024)])
def test_config4_hugeroom_4096_row_streaming_segments(pvlib_exp, K, rows, nseg):
    """BASELINE config 4 at full size through the row-streaming segment kernels (PVA_OPT_STREAM_ROWS, pv_seg.h): three of
    the listeners, both emitters each and the 25 m block of every map against the reference's vectors"""
    g = golden("g71_hugeroom_cfg4")
    with pvlib_exp.Solver(mode_a_size(4096), mode_a_size(4096), 275, steps_per_launch=K, tile_rows=rows,
                      stream_rows=nseg) as s:
        s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
        for i in (0, 3, 7):
            s.set_output_queries(g["emitters"][i])
            s.run(g["listeners"][i])
            q = s.queried_outputs()
            for j in range(2):
                assert same_bits(q[j], g["emitter_out"][i, j]).all(), (i, j)
            res, delay = s.results()
            n = compare_maps(res[:70, :70], delay[:70, :70], g["results"][i], g["delay"][i], 435, 1443,
                             "listener %d, 25 m block" % i)
            assert n > 3500
            assert (delay[80:, :] > 1e30).all() and (delay[:, 80:] > 1e30).all()


def test_config3_bigroom_2048(pvlib):
    """BASELINE config 3: BigRoom.pv (closed 10 m room) at 2048^2 Mode A (730.458 m), L (5,0,4), E (5,0,6) + two
    emitters outside the room (no onset: untouched zero records), and the whole 25 m block of the maps"""
    g = golden("g71_bigroom")
    with pvlib.Solver(mode_a_size(2048), mode_a_size(2048), 275) as s:
        assert (s.gx, s.gy, s.T) == (2048, 2048, 435)
        assert abs(mode_a_size(2048) - 730.458) < 1e-2
        assert np.float32(s.efree) == g["efree"]
        s.load_scene(os.path.join(SCENES, "BigRoom.pv"))
        s.run(g["listener"])
        for e, ro in zip(g["emitters"], g["emitter_out"]):
            compare_output(s.get_output(e), ro, "2048^2 emitter %s" % e)
        res, delay = s.results()
        n = compare_maps(res[:70, :70], delay[:70, :70], g["results"], g["delay"], 435, 1443, "25 m block")
        assert n > 500
        assert (delay[40:, :] > 1e30).all() and (delay[:, 40:] > 1e30).all()
        for (cx, cy), ir in zip(g["probe_cells"], g["probe_ir"]):
            assert same_bits(s.impulse_response(cx, cy), ir).all(), "IR at %d,%d" % (cx, cy)


def test_config5_open_8192_all_64_listeners(pvlib, oracle):
    """BASELINE config 5: ALL 64 seeded listener cells of SURVEY.md 8d.  For every one: onset map and ALL EIGHT result members
    of the 141 x 141 cells around the listener -- which hold both emitters, listener + (16, 0) and + (0, 16) cells -- bit for
    bit against the pinned oracle's window analysis, and the two emitter records against the committed vectors
    (tests/golden/g8192_open_cfg5.npz, what bench.py --open-field --grid 8192 verifies its timed runs with); for the first 8
    (the runs that bench makes first) also four planes of the pressure history."""
    from test_oracle_golden import OpenFieldWindowOracle
    w = OpenFieldWindowOracle(oracle)
    c, R = w.c, w.R
    cells = np.random.default_rng(0).integers(1024, 7168, size=(64, 2))
    want = golden("g8192_open_cfg5")["emitter_out"]
    assert want.shape == (64, 2, 8)
    with pvlib.Solver(mode_a_size(8192), mode_a_size(8192), 275) as s:
        assert (s.gx, s.gy, s.T) == (8192, 8192, 435)
        for k, (lx, ly) in enumerate(cells):
            lx, ly = int(lx), int(ly)
            L = w.listener_metres((lx, ly))
            E = [(L[0] + 16 * float(DX), 0.0, L[2]), (L[0], 0.0, L[2] + 16 * float(DX))]
            s.set_output_queries(E)
            s.run(L)
            ores, odelay = w.analyze((lx, ly), np.float32(s.efree))
            if k < 8:
                for t in (0, 3, 150, 434):
                    plane = s.history_plane(t)
                    assert same_bits(plane[lx - R:lx + R + 1, ly - R:ly + R + 1],
                                     w.hist_pr[t][c - R:c + R + 1, c - R:c + R + 1]).all(), t
            res, delay = s.results_block(lx - R, ly - R, 2 * R + 1, 2 * R + 1)
            if k == 0:  # the block read-back against the whole-map one
                fr, fd = s.results()
                assert np.array_equal(fr[lx - R:lx + R + 1, ly - R:ly + R + 1].view(np.uint32), res.view(np.uint32))
                assert np.array_equal(fd[lx - R:lx + R + 1, ly - R:ly + R + 1].view(np.uint32), delay.view(np.uint32))
            osub = (slice(c - R, c + R + 1), slice(c - R, c + R + 1))
            n = compare_maps(res, delay, ores[osub], odelay[osub], 435, 1443, "listener cell %d,%d" % (lx, ly))
            assert n > 15000
            q = s.queried_outputs()
            assert same_bits(q[0], ores[c + 16, c]).all() and same_bits(q[1], ores[c, c + 16]).all()
            assert same_bits(q, want[k]).all(), "run %d differs from the committed config-5 vectors" % k
    w.close()


def test_run_sharded_cpp_and_rccl_gather_single_rank(pvlib):
    """PvAmdRunSharded (C++: this rank's runs round-robin over two solvers kept in flight) + the C++ side's own RCCL
    communicator at world size 1 (ncclCommInitRank / ncclAllGather bound at run time): the 8 config-4 listeners at
    1024^2, all 16 records against the reference's closed-room vectors; N > 1 ranks are the driver's scaling run"""
    g = golden("g71_hugeroom_cfg4")
    size = mode_a_size(1024)
    solvers = [pvlib.Solver(size, size, 275) for _ in range(2)]
    try:
        for s in solvers:
            s.load_scene(os.path.join(SCENES, "HugeRoom.pv"))
        out = pvlib.run_sharded(solvers, g["listeners"], g["emitters"])
        assert out.shape == (8, 2, 8)
        assert same_bits(out, g["emitter_out"]).all()
        comm = pvlib.Comm(pvlib.Comm.unique_id(), 0, 1, 0)
        try:
            x = np.arange(24, dtype=np.float32)
            assert np.array_equal(comm.all_gather(x), x[None])
            out2 = pvlib.run_sharded(solvers, g["listeners"], g["emitters"], 0, 1, comm)
            assert same_bits(out2, g["emitter_out"]).all()
            from planeverb_amd import dist as pvd
            local = {k: out[k] for k in range(8)}
            assert same_bits(pvd.gather_outputs_native(local, 8, comm), out).all()
        finally:
            comm.close()
    finally:
        for s in solvers:
            s.close()
