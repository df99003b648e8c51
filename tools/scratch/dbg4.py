import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import planeverb_amd.api as pv
E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
L = (5.0, 0.0, 4.0)
res = int(sys.argv[1])
s = pv.Solver(25.0, 25.0, res, streaming_analysis=1, stream_fuse=0)
s.load_scene("/root/repo/tests/scenes/HugeRoom.pv")
s.set_emitters(E)
s.run(L)
t = s.timings()
a = np.stack([s.get_output(e).as_array() for e in E])
r, d = s.results()
cells = [pv.host_cells(25.0, 25.0, res, e[0], e[2])[1] for e in E]
print("res", res, "finalize %.2f ms" % t.analysisMs, "efree", s.efree, "T", s.T)
for (cx, cy), o in zip(cells, a):
    print("  cell", cx, cy, "delay", d[cx, cy], "map", r[cx, cy, :4], "get_output", o[:4])
