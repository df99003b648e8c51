#!/usr/bin/env python3
"""Iterations per second of the live module (PlaneverbInit ... background loop: FDTD + analysis + publish of the
history-window block) against the batch API's time for the same run, at 71^2 (the Sandbox's grid), 513^2, 2049^2 and
4097^2 cells (Mode A, HugeRoom.pv, one emitter polled like a game thread would).  Writes what it prints to
profiles/r02_live_rate.txt when given --out.

    python tools/gpu_live_rate.py [--out profiles/r02_live_rate.txt]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import planeverb_amd.api as pv  # noqa: E402

out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else None


def say(s):
    print(s)
    if out:
        out.write(s + "\n")


dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
scene = os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv")
say("# live module (PlaneverbInit, worker loop) vs batch API (PvAmdRun), HugeRoom.pv, Mode A 275 Hz, T = 435, L (5,0,4)")
say("# cells   batch run ms (wall, incl. analysis + output fetch)   live ms/iteration   live/batch   published block   GetOutput polls/s")
for n in (70, 512, 2048, 4096):
    size = 25.0 if n == 70 else float((n + 0.5) * dx)
    with pv.Solver(size, size, 275) as s:
        s.load_scene(scene)
        s.set_output_queries([(5.0, 0.0, 6.0)])
        for _ in range(3):
            s.run((5.0, 0.0, 4.0))
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            s.run((5.0, 0.0, 4.0))
            s.queried_outputs()
        batch_ms = (time.perf_counter() - t0) / reps * 1e3
        want = s.get_output((5.0, 0.0, 6.0)).as_array()
        block = min(s.info.histRows, s.gx) * min(s.info.histPitch, s.gy)
    pv.Init(pv.Config((size, size), 275, 0, ".", 0, pv.pv_GPU))
    pv.SetListenerPosition((5.0, 0.0, 4.0))
    pv.LoadScene(scene)
    e = pv.Emit((5.0, 0.0, 6.0))
    pv.WaitIterations(pv.IterationCount() + 10, 60000)
    n0, t0 = pv.IterationCount(), time.perf_counter()
    polls = 0
    while time.perf_counter() - t0 < 2.0:
        pv.GetOutput(e)
        polls += 1
    n1, t1 = pv.IterationCount(), time.perf_counter()
    got = pv.GetOutput(e).as_array()
    pv.Exit()
    live_ms = 1e3 * (t1 - t0) / max(1, n1 - n0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
    say("%5d^2   %8.3f   %8.3f   %5.2f   <= %d records (%.1f MB)   %.0f" % (
        n + 1, batch_ms, live_ms, live_ms / batch_ms, block, block * 32 / 1e6, polls / (t1 - t0)))
say("# full result map for comparison: 4096^2 x 32 B = 537 MB per iteration (round 1 copied that: ~20 ms)")
if out:
    out.close()
