#!/usr/bin/env python3
"""One-off: the UNMODIFIED reference (oracle/_ref/libpvref.so) on ONE host core of the GPU box at a BASELINE grid size --
2049 x 2049 cells (Mode A, 275 Hz, T = 435: a 29 GB impulse-response cube per Grid, the reference builds two), same scene
and listener as bench.py's cpu_baseline sample; records checked against the committed config-4 vectors.  Too slow and too
large for the driver's bench run (bench.py keeps its bounded 1025^2 sample); kept as profiles/r03_cpu_reference_2049.txt.

    python tools/cpu_reference_big.py [cells=2049]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
need_gb = 2 * cells * cells * 435 * 16 / 1e9 + 8
avail_gb = 0.0
with open("/proc/meminfo") as f:
    for line in f:
        if line.startswith("MemAvailable"):
            avail_gb = int(line.split()[1]) / 1e6
print("host memory available %.0f GB, the reference needs about %.0f GB at %d^2 cells" % (avail_gb, need_gb, cells))
# (242 GB at 4097^2 ran fine; 934 GB at 8193^2 took the GPU box down although /proc/meminfo showed 3 TB available: the
# pool's boxes are not to be trusted beyond a few hundred GB)
if need_gb > 320 and not os.environ.get("PV_ALLOW_HUGE_REFERENCE"):
    raise SystemExit("refusing a reference run of %.0f GB (set PV_ALLOW_HUGE_REFERENCE=1 to override)" % need_gb)
if avail_gb < 1.5 * need_gb:
    raise SystemExit("not enough host memory for a safe run")
out = bench.cpu_baseline(cells)
g = np.load(os.path.join(ROOT, "tests", "golden", "g71_hugeroom_cfg4.npz"))
print(json.dumps(out, indent=1))
