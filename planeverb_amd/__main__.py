"""Headless command line for the solver: load a .pv scene, simulate one listener position,
print the acoustic parameters of each emitter as JSON -- the numbers the Sandbox shows in its "Analyzer Output" panel
(PlaneverbSandbox/src/Editor/Editor.cpp:396-434) without the GUI.

    python -m planeverb_amd tests/scenes/SmallRoomScene.pv --listener 5,0,4 --emitter 5,0,6 --emitter 12,0,9
    python -m planeverb_amd tests/scenes/HugeRoom.pv --cells 4096 --listener 5,0,4 --emitter 5,0,6
    python -m planeverb_amd scene.pv --save copy.pv          # .pv round trip (Editor::SaveGeometry format)
"""
import argparse
import json
import sys

import numpy as np

from . import api


def vec3(s):
    v = [float(x) for x in s.split(",")]
    if len(v) != 3:
        raise argparse.ArgumentTypeError("expected x,y,z")
    return tuple(v)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m planeverb_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", help=".pv scene file")
    ap.add_argument("--size", type=float, default=25.0, help="grid size in metres (square), default 25 (Sandbox)")
    ap.add_argument("--cells", type=int, default=0, help="instead of --size: N cells per side at --res (Mode A)")
    ap.add_argument("--res", type=int, default=275, help="grid resolution in Hz (>= 275)")
    ap.add_argument("--listener", type=vec3, default=(5.0, 0.0, 4.0))
    ap.add_argument("--emitter", type=vec3, action="append", default=[])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--save", help="write the loaded boxes back as a .pv file and exit (no GPU needed)")
    a = ap.parse_args(argv)

    boxes = api.load_pv(a.scene)
    if a.save:
        api.save_pv(a.save, boxes)
        print(json.dumps({"saved": a.save, "boxes": len(boxes)}))
        return 0
    size = a.size
    if a.cells:
        dx = np.float32(343.21) / np.float32(a.res) / np.float32(3.5)
        size = float((a.cells + 0.5) * dx)
    emitters = a.emitter or [(5.0, 0.0, 6.0)]
    with api.Solver(size, size, a.res, device=a.device) as s:
        for b in boxes:
            s.add_geometry(b)
        s.run(a.listener)
        t = s.timings()
        out = {"scene": a.scene, "grid": [s.gx, s.gy], "T": s.T, "res": a.res, "dx": s.dx, "efree": s.efree,
               "listener": a.listener, "fdtd_ms": t.fdtdMs, "analysis_ms": t.analysisMs, "emitters": []}
        for e in emitters:
            o = s.get_output(e)
            ga, gb, gc = api.reverb_bus_gains(o.rt60, o.wetGain)
            out["emitters"].append({
                "position": e, "occlusion": o.occlusion, "wetGain": o.wetGain, "rt60": o.rt60, "lowpass": o.lowpass,
                "direction": [o.directionX, o.directionY], "sourceDirectivity": [o.sourceDirectionX, o.sourceDirectionY],
                "reverbBusGains": [ga, gb, gc]})
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
