import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import planeverb_amd.api as pv
E = [(5.0, 0.0, 6.0), (12.0, 0.0, 9.0), (20.5, 0.0, 3.2), (7.0, 0.0, 4.0)]
L = (5.0, 0.0, 4.0)
for res in (2009, 16067):
  for fuse in (1, 0):
    s = pv.Solver(25.0, 25.0, res, streaming_analysis=1, stream_fuse=fuse)
    s.load_scene("/root/repo/tests/scenes/HugeRoom.pv")
    s.set_emitters(E)
    s.run(L)
    a = np.stack([s.get_output(e).as_array() for e in E])
    r, d = s.results()
    b = np.stack([s.get_output(e).as_array() for e in E])
    print(res, fuse, "before results():", a[:, 1:3].ravel(), "after:", b[:, 1:3].ravel())
    s.close()
