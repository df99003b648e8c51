#!/usr/bin/env python3
"""Phase stamps of the resident kernel's middle tile (development aid).  Needs the trace build of the library:
    make -C planeverb_amd/csrc BUILD=build_trace OUT=../libplaneverb_amd_trace.so EXTRA=-DPV_RESIDENT_TRACE
    PLANEVERB_AMD_LIB=$PWD/planeverb_amd/libplaneverb_amd_trace.so python tools/gpu_resident_trace.py 275 750"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402

scene = os.path.join(ROOT, "tests", "scenes", "SmallRoomScene.pv")
for res in [int(a) for a in sys.argv[1:]] or [275]:
    sys.stderr.write("## res %d\n" % res)
    with pv.Solver(25.0, 25.0, res, no_free_grid=1) as s:
        s.load_scene(scene)
        for _ in range(3):
            s.run((5.0, 0.0, 4.0))
        t = s.timings()
        sys.stderr.write("## fdtd %.3f ms, analysis %.3f ms, resident %d\n" % (t.fdtdMs, t.analysisMs, s.info.residentKernel))
