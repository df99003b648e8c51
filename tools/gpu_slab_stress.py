#!/usr/bin/env python3
"""Stress of the slab decomposition on ONE device (hand-off words between the slabs' push kernels, pv_halo_push_kernel): random
scenes, random slab counts and listeners, many consecutive runs per group -- fields, delay map and all eight result planes of
every run against one solver on the whole grid, bit for bit.  usage: gpu_slab_stress.py [first_seed] [scenes] [runs per scene]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import planeverb_amd.api as pv  # noqa: E402
from conftest import same_bits  # noqa: E402
from test_gpu_parity import random_scene  # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nscenes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nruns = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
bad = 0
t0 = time.time()
for seed in range(seed0, seed0 + nscenes):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([384, 512, 768, 1024, 1536, 2048]))
    S = int(rng.choice([2, 2, 3, 4, 5, 8]))
    size = float((n + 0.5) * dx)
    boxes = random_scene(rng, size, int(rng.integers(0, 20)))
    try:
        with pv.Solver(size, size, 275) as a, pv.Solver(size, size, 275, slabs=[0] * S) as b:
            for s in (a, b):
                for box in boxes:
                    s.add_geometry(box)
            for r in range(nruns):
                L = (float(rng.uniform(1, size - 1)), 0.0, float(rng.uniform(1, size - 1)))
                a.run(L)
                b.run(L)
                ra, da = a.results()
                rb, db = b.results()
                ok = same_bits(da, db).all() and all(same_bits(ra[..., k], rb[..., k]).all() for k in range(8))
                ok = ok and all(same_bits(fa, fb).all() for fa, fb in zip(a.fields(), b.fields()))
                if not ok:
                    bad += 1
                    print("seed %d run %d: MISMATCH (grid %d, %d slabs)" % (seed, r, n, S), flush=True)
        print("seed %d: grid %d^2, %d slabs, %d boxes, %d runs: %s" % (seed, n, S, len(boxes), nruns, "identical" if not bad else "see above"), flush=True)
    except pv.PlaneverbError as e:
        print("seed %d: grid %d^2, %d slabs: %s" % (seed, n, S, e), flush=True)
        bad += 1
print("%d scenes x %d runs, %d mismatches, %.0f s" % (nscenes, nruns, bad, time.time() - t0))
sys.exit(1 if bad else 0)
