"""GPU (-m gpu): the live module behind the reference's flat C-ABI (PlaneverbInit ... PlaneverbGetOutput), driven the
way the Sandbox / Unity drive it (PlaneverbSandbox/src/main.cpp:14-21, Editor.cpp:245-281, PlaneverbEmitter.cs:49-60)."""
import os

import numpy as np
import pytest

from conftest import SCENES, golden, same_cells, valid_mask
from test_gpu_parity import compare_output

pytestmark = pytest.mark.gpu


@pytest.fixture
def module(pvlib):
    pvlib.Init(pvlib.Config((25.0, 25.0), 275, 0, ".", 0, pvlib.pv_GPU))
    yield pvlib
    pvlib.Exit()


def settle(pv, n=3):
    """geometry queued now is rasterised after the current iteration, the listener is latched after it too
    (PvContext.cpp:80-90): results are current from the second completed iteration on"""
    target = pv.IterationCount() + n
    assert pv.WaitIterations(target, 60000) >= target


def test_sandbox_defaults(module):
    pv = module
    g = golden("g71_smallroom")
    pv.SetListenerPosition(g["listener"])
    assert pv.LoadScene(os.path.join(SCENES, "SmallRoomScene.pv")) == 5
    ids = [pv.Emit(e) for e in g["emitters"]]
    assert ids == [0, 1, 2]
    settle(pv)
    for i, ro in zip(ids, g["emitter_out"]):
        compare_output(pv.GetOutput(i), ro, "emitter %d" % i)
    # an emitter moved off the grid -> PV_INVALID_DRY_GAIN, rest zero (FDTD.cpp:43-47)
    pv.UpdateEmission(ids[0], (40.0, 0.0, 5.0))
    o = pv.GetOutput(ids[0])
    assert o.occlusion == -1.0 and o.wetGain == 0.0 and o.rt60 == 0.0
    # never-issued id
    assert pv.GetOutput(17).occlusion == -1.0
    # ended ids are recycled LIFO (EmissionManager.cpp:40-46)
    pv.EndEmission(ids[1])
    pv.EndEmission(ids[2])
    assert pv.Emit((1, 0, 1)) == ids[2]
    assert pv.Emit((1, 0, 1)) == ids[1]
    assert pv.Emit((1, 0, 1)) == 3


def test_listener_and_geometry_updates(module):
    pv = module
    ga, gb = golden("g71_smallroom"), golden("g71_shoebox")
    pv.SetListenerPosition(ga["listener"])
    gids = [pv.AddGeometry(b) for b in ga["boxes"]]
    assert gids == list(range(len(gids)))
    e = pv.Emit(ga["emitters"][0])
    settle(pv)
    compare_output(pv.GetOutput(e), ga["emitter_out"][0], "scene A")
    # swap the scene: remove everything, add the shoebox (ids are recycled LIFO, GeometryManager.cpp:81-85)
    for i in gids:
        pv.RemoveGeometry(i)
    new = [pv.AddGeometry(b) for b in gb["boxes"]]
    assert sorted(new) == sorted(gids[:len(new)]) or len(new) != len(gids)
    settle(pv)
    compare_output(pv.GetOutput(e), gb["emitter_out"][0], "scene B")
    # UpdateGeometry = Remove(old) + Add(new) (GeometryManager.cpp:112-121): move one wall away and back
    b0 = gb["boxes"][0]
    moved = b0.copy()
    moved[1] += 3.0
    pv.UpdateGeometry(new[0], moved)
    settle(pv)
    o_moved = pv.GetOutput(e).as_array()
    pv.UpdateGeometry(new[0], b0)
    # Remove(moved) also clears whatever the moved wall overlapped -- re-add the side walls
    for i, b in zip(new[1:], gb["boxes"][1:]):
        pv.UpdateGeometry(i, b)
    settle(pv)
    compare_output(pv.GetOutput(e), gb["emitter_out"][0], "scene B restored")
    assert not np.array_equal(o_moved, pv.GetOutput(e).as_array())


def test_live_publish_window_only_1024(pvlib):
    """The live module copies only the history-window block of the result map per iteration; cells outside it are
    answered on the host: earlier values (the reference leaves m_results untouched where an iteration finds
    no onset) + the closed-form listener direction.  Checked against a batch solver that makes the same sequence of runs
    and reads its full device map: emitters inside the window, far outside it, and -- after the listener (and with it the
    window) has moved 600 cells away -- an emitter whose cell was inside the FIRST window only."""
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((2048 + 0.5) * dx)
    cell = lambda cx, cy: ((cx + 0.5) * float(dx), 0.0, (cy + 0.5) * float(dx))
    L1, L2 = cell(700, 800), cell(1500, 1200)
    boxes = [[L1[0] + 8.0, L1[2] + 1.0, 1.0, 30.0, 0.9], [L2[0] - 9.0, L2[2], 1.0, 25.0, 0.8]]
    emitters = [cell(712, 806), cell(700, 800), cell(705, 700), cell(1900, 100), cell(30, 2000), cell(1490, 1215),
                cell(1100, 1000)]
    pvlib.Init(pvlib.Config((size, size), 275, 0, ".", 0, pvlib.pv_GPU))
    try:
        with pvlib.Solver(size, size, 275) as ref:
            ref.run((0.0, 0.0, 0.0))  # the module's first iteration: listener still (0,0,0), no geometry yet
            for b in boxes:
                pvlib.AddGeometry(b)
                ref.add_geometry(b)
            pvlib.SetListenerPosition(L1)
            ids = [pvlib.Emit(e) for e in emitters]
            settle(pvlib, 3)
            ref.run(L1)
            ref.run(L1)  # (the module has run L1 more than once: same values, the map is idempotent for one listener)
            for i, e in zip(ids, emitters):
                compare_output(pvlib.GetOutput(i), ref.get_output(e).as_array(), "listener 1, emitter %s" % (e,))
            assert pvlib.GetOutput(ids[0]).occlusion > 0 and pvlib.GetOutput(ids[3]).occlusion == 0
            # move the listener: the window follows it, emitter 0's cell keeps its values from the first window
            pvlib.SetListenerPosition(L2)
            settle(pvlib, 3)
            ref.run(L2)
            for i, e in zip(ids, emitters):
                compare_output(pvlib.GetOutput(i), ref.get_output(e).as_array(), "listener 2, emitter %s" % (e,))
            o0 = pvlib.GetOutput(ids[0])
            assert o0.occlusion > 0 and pvlib.GetOutput(ids[5]).occlusion > 0
            # ... and an emitter moved between iterations is looked up at its new cell at once (FDTD.cpp:16-58)
            pvlib.UpdateEmission(ids[3], emitters[5])
            compare_output(pvlib.GetOutput(ids[3]), ref.get_output(emitters[5]).as_array(), "moved emitter")
    finally:
        pvlib.Exit()


def test_get_impulse_response_live(module):
    """Planeverb::GetImpulseResponse (Planeverb.h:47, FDTD.cpp:60-79) through the live module: raw reference Cells
    {pr, vx, vy, b, by} of the last completed iteration, bit-identical to the reference's cube (incl. b / by of wall,
    ghost, x = 0 and y = 0 cells)"""
    pv = module
    g = golden("g71_smallroom_cells")
    pv.SetListenerPosition(g["listener"])
    for b in g["boxes"]:
        pv.AddGeometry(b)
    settle(pv)
    dx = float(pv.host_grid_info(25.0, 25.0, 275).dx)
    for (cx, cy), want in zip(g["cells"], g["ir_static"]):
        got = pv.GetImpulseResponse(((cx + 0.5) * dx, 0.0, (cy + 0.5) * dx))
        assert got.shape == (435,)
        assert same_cells(got, want), "cell %d,%d" % (cx, cy)
    assert len(pv.GetImpulseResponse((40.0, 0.0, 3.0))) == 0  # outside the cell array
    assert pv.IsRunning()


def test_impulse_response_cells_after_remove(pvlib):
    """b / by of the recorded Cells follow Grid::RemoveAABB's own restore rule (Grid.cpp:281-290: `by` comes back as 0
    on the x = 0 row, not on the y = 0 column the constructor cleared)"""
    g = golden("g71_smallroom_cells")
    with pvlib.Solver(25.0, 25.0, 275) as s:
        for b in g["boxes"]:
            s.add_geometry(b)
        gid = s.add_geometry(g["corner_box"])
        s.remove_geometry(gid)
        s.run(g["listener"])
        for (cx, cy), want in zip(g["cells"], g["ir_removed"]):
            got = s.impulse_response_cells(cx, cy)
            assert same_cells(got, want), "cell %d,%d" % (cx, cy)


def test_sandbox_probe_linked_against_binding(pvlib):
    """A C++ program written against the reference's Planeverb.h ONLY (tests/host/sandbox_probe.cpp: the Sandbox's
    config, LoadGeometry, Emit, GetOutput, GetImpulseResponse + the other 7 namespace functions), linked with
    bindings/PlaneverbAmdBinding.cpp + libplaneverb_amd.so in the build container (`make -C oracle ref`; it needs the
    reference's headers, which do not exist on the GPU box) and run here as its own process"""
    import json
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "sandbox_probe")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sandbox_probe not built (needs the reference's headers: make -C oracle ref)")
    g, gc = golden("g71_smallroom"), golden("g71_smallroom_cells")
    dx = float(pvlib.host_grid_info(25.0, 25.0, 275).dx)
    args = [exe, os.path.join(SCENES, "SmallRoomScene.pv"), "5", "4", "5", "6"]
    cells = gc["cells"][:6]
    for cx, cy in cells:
        args += [repr(float((cx + 0.5) * dx)), repr(float((cy + 0.5) * dx))]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["emitter_id"] == 0 and out["geometry_id"] == 5 and out["off_grid_occlusion"] == -1.0
    got = np.array(out["output_bits"], np.uint32).view(np.float32)
    want = g["emitter_out"][0]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
    for ir, want_cells in zip(out["irs"], gc["ir_static"]):
        assert ir["n"] == 435
        assert same_cells(np.frombuffer(bytes.fromhex(ir["cells_hex"]), np.uint8), want_cells)


def test_reinit_while_running(pvlib):
    """Init while running == Exit + Init (PvContext.cpp:27-31); Exit twice is harmless"""
    pvlib.Init(pvlib.Config((25.0, 25.0), 275, 0, ".", 0, 1))
    pvlib.Init(pvlib.Config((10.0, 10.0), 375, 0, ".", 0, 1))
    assert pvlib.WaitIterations(2, 60000) >= 2
    pvlib.Exit()
    pvlib.Exit()
    assert pvlib.GetOutput(0).occlusion == -1.0


def test_headless_cli(pvlib):
    """python -m planeverb_amd scene.pv ... prints the per-emitter parameters"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    g = golden("g71_smallroom")
    r = subprocess.run([sys.executable, "-m", "planeverb_amd", os.path.join(SCENES, "SmallRoomScene.pv"), "--listener",
                        "5,0,4", "--emitter", "5,0,6", "--emitter", "12,0,9"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["grid"] == [70, 70] and out["T"] == 435
    for e, ro in zip(out["emitters"], g["emitter_out"][:2]):
        assert np.float32(e["occlusion"]) == ro[0] and np.float32(e["wetGain"]) == ro[1]
        assert abs(e["rt60"] - ro[2]) <= 1e-4 * abs(ro[2])
        assert np.float32(e["direction"][0]) == ro[4] and np.float32(e["sourceDirectivity"][1]) == ro[7]
        assert len(e["reverbBusGains"]) == 3


def test_live_module_sparse_emitter_mode_512_mode_b(pvlib, monkeypatch):
    """The live module in the sparse-emitter mode, through the UNCHANGED flat ABI: BASELINE config 2 /
    Mode B (25 m at res 2009: 512^2, T = 3179) with the mode forced by PLANEVERB_AMD_LIVE_STREAMING=1.  The worker registers
    the emitters alive at the start of every iteration (PlaneverbEmit / UpdateEmission / EndEmission -> setEmitters before
    run, EmissionManager.cpp:37-75): records bit-equal to the reference's vectors, an emitter added between iterations is
    picked up by the next one, an ended one is dropped."""
    g = golden("g512B_shoebox")
    monkeypatch.setenv("PLANEVERB_AMD_LIVE_STREAMING", "1")
    pvlib.Init(pvlib.Config((25.0, 25.0), 2009, 0, ".", 0, pvlib.pv_GPU))
    try:
        assert pvlib.lib().PlaneverbIsStreaming() == 1
        pvlib.SetListenerPosition(g["listener"])
        for b in g["boxes"]:
            pvlib.AddGeometry(b)
        e0 = pvlib.Emit(g["emitters"][0])
        settle(pvlib, 3)
        compare_output(pvlib.GetOutput(e0), g["emitter_out"][0], "emitter present from the start")
        # a second emitter, added between iterations, on one of the reference's sampled cells
        c = g["cells"]
        dx = np.float32(343.21) / np.float32(2009) / np.float32(3.5)
        gx, gy, T, fs = (int(v) for v in g["dims"])
        valid = np.flatnonzero(valid_mask(g["cell_delay"], T, fs) & (g["cell_results"][:, 1] > 0))
        k = int(valid[len(valid) // 2])
        pos = ((c[k, 0] + 0.5) * float(dx), 0.0, (c[k, 1] + 0.5) * float(dx))
        e1 = pvlib.Emit(pos)
        settle(pvlib, 2)
        compare_output(pvlib.GetOutput(e1), g["cell_results"][k], "emitter added between iterations")
        compare_output(pvlib.GetOutput(e0), g["emitter_out"][0], "first emitter, later iteration")
        # moving an emitter: the forward quantities of the new cell are there at once, wet gain / RT60 one iteration later
        k2 = int(valid[len(valid) // 3])
        pos2 = ((c[k2, 0] + 0.5) * float(dx), 0.0, (c[k2, 1] + 0.5) * float(dx))
        pvlib.UpdateEmission(e1, pos2)
        o = pvlib.GetOutput(e1).as_array()
        for idx in (0, 4, 5, 6, 7):
            assert np.float32(o[idx]).view(np.uint32) == np.float32(g["cell_results"][k2][idx]).view(np.uint32) or \
                (o[idx] == 0 and g["cell_results"][k2][idx] == 0)
        settle(pvlib, 2)
        compare_output(pvlib.GetOutput(e1), g["cell_results"][k2], "moved emitter, next iteration")
        pvlib.EndEmission(e1)
        settle(pvlib, 2)
        compare_output(pvlib.GetOutput(e0), g["emitter_out"][0], "after EndEmission of the other one")
        # the IR cube does not exist in this mode: the call fails cleanly
        with pytest.raises(pvlib.PlaneverbError):
            pvlib.GetImpulseResponse(g["emitters"][0])
    finally:
        pvlib.Exit()


def test_live_module_falls_back_to_sparse_emitter_mode_by_itself(pvlib, monkeypatch):
    """PlaneverbInit(25, 25, 16067, ...) -- a 4096^2 grid with T = 25 432, whose history window would take 1.7 TB --
    comes up by itself in the sparse-emitter mode (Planeverb::Init accepts any resolution >= 275, PvContext.cpp:101-107),
    at this size with the forward sums of the air tiles inside the stencil (csrc/pv_stream.h).  The real 25 m room
    (HugeRoom.pv) with two emitters: the records of the module's iterations equal those of a batch solver in the same mode."""
    monkeypatch.delenv("PLANEVERB_AMD_LIVE_STREAMING", raising=False)
    L, E = (5.0, 0.0, 4.0), [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0)]
    boxes = pvlib.load_pv(os.path.join(SCENES, "HugeRoom.pv"))
    pvlib.Init(pvlib.Config((25.0, 25.0), 16067, 0, ".", 0, pvlib.pv_GPU))
    try:
        assert pvlib.IsRunning()
        assert pvlib.lib().PlaneverbIsStreaming() == 1
        pvlib.SetListenerPosition(L)
        for b in boxes:
            pvlib.AddGeometry(b)
        ids = [pvlib.Emit(e) for e in E]
        assert ids == [0, 1]
        settle(pvlib, 2)
        assert pvlib.IsRunning(), pvlib.last_error()
        live = [pvlib.GetOutput(i).as_array() for i in ids]
    finally:
        pvlib.Exit()
    with pvlib.Solver(25.0, 25.0, 16067, streaming_analysis=1) as s:
        for b in boxes:
            s.add_geometry(b)
        s.set_emitters(E)
        s.run(L)
        for e, o in zip(E, live):
            want = s.get_output(e).as_array()
            assert want[1] > 0 and want[2] > 0
            assert np.array_equal(o.view(np.uint32), want.view(np.uint32)), (e, o, want)
    # a config whose history fits stays in the full-history mode
    pvlib.Init(pvlib.Config((25.0, 25.0), 275, 0, ".", 0, pvlib.pv_GPU))
    try:
        assert pvlib.lib().PlaneverbIsStreaming() == 0
    finally:
        pvlib.Exit()


def test_solvers_on_concurrent_host_threads(pvlib):
    """Several host threads creating, running and destroying solvers on one device at the same time -- what the live module's
    worker and a caller's own batch solver do.  Small grids replay a captured run graph; a synchronous hipMemcpy of ANOTHER thread
    runs on the legacy stream, which the runtime refuses while a capture is open anywhere ("operation would make the legacy stream
    depend on a capturing blocking stream") and which invalidates that capture: 4 of 80 solvers failed with two threads, 40 of
    100 with four, until every copy went through the solver's own stream (tools/capture_stress.py)."""
    import threading
    g = golden("g71_smallroom")
    fails, outs = [], []

    def work(tid):
        for i in range(6):
            try:
                with pvlib.Solver(25.0, 25.0, 275) as s:
                    for b in g["boxes"]:
                        s.add_geometry(b)
                    for _ in range(2):
                        s.run(g["listener"])
                    outs.append(s.get_output(g["emitters"][0]).as_array())
            except Exception as e:  # noqa: BLE001
                fails.append((tid, i, str(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not fails, fails[:3]
    assert len(outs) == 24
    for o in outs:
        assert np.array_equal(o.view(np.uint32), np.asarray(g["emitter_out"][0], np.float32).view(np.uint32))
