import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planeverb_amd.api as pv
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dx = 343.21 / 275 / 3.5
ets = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [None]
for n, scene in ((4096, "HugeRoom.pv"), (2048, "BigRoom.pv"), (8192, None)):
    size = (n + 0.5) * dx
    for et in ets:
        kw = {} if et is None else {"edge_tiles": et}
        s = pv.Solver(size, size, 275, **kw)
        if scene:
            s.load_scene(R + "/tests/scenes/" + scene)
            L = (5, 0, 4)
        else:
            L = (size / 2, 0, size / 2)
        s.run(L)
        f = []
        for _ in range(6):
            s.run(L); f.append(s.timings().fdtdMs)
        print("%s n=%d edge_tiles=%s fdtd min %.3f ms" % (os.path.basename(os.environ.get("PLANEVERB_AMD_LIB", "tree")), n, et, min(f)), flush=True)
        s.close()
