#!/usr/bin/env python3
"""Iterations per second of the live module (PlaneverbInit ... background loop) at the reference's resolution presets,
sandbox scene, one emitter polled like a game thread would.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv
for res in (275, 375, 500, 750):
    pv.Init(pv.Config((25.0, 25.0), res, 0, ".", 0, pv.pv_GPU))
    pv.SetListenerPosition((5.0, 0.0, 4.0))
    pv.LoadScene(os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv"))
    e = pv.Emit((5.0, 0.0, 6.0))
    pv.WaitIterations(pv.IterationCount() + 20, 60000)
    n0, t0 = pv.IterationCount(), time.time()
    polls = 0
    while time.time() - t0 < 1.0:
        pv.GetOutput(e)
        polls += 1
    n1, t1 = pv.IterationCount(), time.time()
    o = pv.GetOutput(e)
    print("res %d: %.0f iterations/s (%.2f ms per iteration), %.0f GetOutput polls/s, occlusion %.6f" % (
        res, (n1 - n0) / (t1 - t0), 1e3 * (t1 - t0) / max(1, n1 - n0), polls / (t1 - t0), o.occlusion))
    pv.Exit()
