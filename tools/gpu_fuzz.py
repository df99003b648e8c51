#!/usr/bin/env python3
"""Parity fuzz on the GPU box: seeded random scenes / sizes / resolutions / listener positions / kernel
configurations, HIP path (through the C-ABI) against the pinned oracle (checker only).  Bit-exact comparison of
recorded pressure planes, sampled impulse responses, the delay map and all eight result planes (SURVEY Q5 mask).
Development aid beside tests/test_gpu_parity.py::test_random_scenes_vs_oracle, which holds 7 fixed seeds.

usage: gpu_fuzz.py [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import planeverb_amd.api as pv  # noqa: E402
from oracle import pvoracle  # noqa: E402  (checker)
from conftest import same_bits  # noqa: E402
from test_gpu_parity import check_streaming_against, compare_maps, random_scene  # noqa: E402

CONFIGS = [dict(), dict(), dict(steps_per_launch=12, tile_rows=36), dict(steps_per_launch=10, tile_rows=36),
           dict(steps_per_launch=8, tile_rows=24), dict(steps_per_launch=4, tile_rows=32),
           dict(steps_per_launch=8, tile_rows=40), dict(steps_per_launch=12, tile_rows=12), dict(steps_per_launch=10, tile_rows=20),
           dict(small_grid_kernel=2), dict(small_grid_kernel=2, use_graph=2),
           dict(steps_per_launch=12, tile_rows=36, merged_launch=0), dict(steps_per_launch=1, tile_rows=30),
           dict(streaming_analysis=1), dict(streaming_analysis=1, steps_per_launch=12, tile_rows=36),
           dict(streaming_analysis=1, steps_per_launch=4, tile_rows=32),
           # round 3: forward sums of air tiles inside the stencil (pv_stream.h), forced on (auto: from 6000 tiles)
           dict(streaming_analysis=1, stream_fuse=1, steps_per_launch=8, tile_rows=24), dict(streaming_analysis=1, stream_fuse=1, steps_per_launch=12, tile_rows=36),
           dict(streaming_analysis=1, stream_fuse=1, steps_per_launch=10, tile_rows=36),
           dict(steps_per_launch=8, tile_rows=40, edge_tiles=1), dict(steps_per_launch=10, tile_rows=36, edge_tiles=1),
           dict(steps_per_launch=12, tile_rows=36, edge_tiles=1),
           # round 2: row bands, single-grid decomposition into slabs (falls back to a plain solver where the grid has
           # fewer than 4 tile rows), thick walls (dead tiles) come from random_scene's larger boxes
           dict(row_bands=2), dict(steps_per_launch=8, tile_rows=24, row_bands=3),
           dict(steps_per_launch=8, tile_rows=24, slabs=[0, 0]), dict(steps_per_launch=8, tile_rows=24, slabs=[0, 0, 0]),
           dict(steps_per_launch=8, tile_rows=24, slabs=[0, 0]),
           # row-streaming air segments (pv_seg.h)
           dict(steps_per_launch=8, tile_rows=40, stream_rows=6, use_graph=2),
           dict(steps_per_launch=12, tile_rows=36, stream_rows=30, use_graph=2)]
SEG_CONFIGS = [dict(steps_per_launch=k, tile_rows=r, stream_rows=n, use_graph=2)
               for k, r in ((8, 40), (12, 36)) for n in (1, 7, 40, 400)]
if os.environ.get("PV_FUZZ_SEG"):  # a campaign on the segment kernels only
    CONFIGS = SEG_CONFIGS
if os.environ.get("PV_FUZZ_CONFIG"):  # a campaign on one configuration: PV_FUZZ_CONFIG="steps_per_launch=16,tile_rows=12"
    CONFIGS = [dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ["PV_FUZZ_CONFIG"].split(",")))]


def one(seed):
    rng = np.random.default_rng(1000 + seed)
    res = int(rng.choice([275, 275, 300, 375, 500]))
    size = float(rng.uniform(6.0, 62.0 if res <= 300 else 30.0))
    if res == 275 and rng.random() < 0.2:
        size = float(rng.uniform(95.0, 135.0))  # > 256 cells: analysis windows wide enough for pointer jumping
    boxes = random_scene(rng, size, int(rng.integers(0, 26)))
    L = (rng.uniform(0.2, size - 0.2), 0.0, rng.uniform(0.2, size - 0.2))
    opts = CONFIGS[int(rng.integers(0, len(CONFIGS)))]
    o = pvoracle.OracleGrid(size, size, res, boxes)
    o.fdtd(L)
    ef = pvoracle.free_energy(size, size, res)
    rres, rdelay, _ = o.analyze(ef, L)
    hp, hx, hy = o.history()
    flat = hp.reshape(o.T, -1)
    finite = np.isfinite(flat).all(1) & (np.abs(np.nan_to_num(flat, nan=np.inf)).max(1) < 1e30)
    tmax = o.T - 1 if finite.all() else int(np.argmin(finite)) - 1
    if "slabs" in opts:
        try:
            pv.Solver(size, size, res, **opts).close()
        except pv.PlaneverbError:  # too few tile rows for that many slabs
            opts = {k: v for k, v in opts.items() if k != "slabs"}
    with pv.Solver(size, size, res, **opts) as s:
        assert (s.gx, s.gy, s.T) == (o.gx, o.gy, o.T)
        assert np.float32(s.efree) == np.float32(ef), "EFree"
        for b in boxes:
            s.add_geometry(b)
        emitters = rng.uniform(0.3, size - 0.3, (10, 3)).astype(np.float32)
        if opts.get("streaming_analysis"):
            s.set_emitters(emitters)  # sparse-emitter mode keeps a ring, not the whole history
        s.run(L)
        if not opts.get("streaming_analysis"):
            for t in sorted(set([0, 1, 2, 3, 17, o.T // 3, o.T // 2, o.T - 2, o.T - 1])):
                if t <= tmax:
                    ok = same_bits(s.history_plane(t), hp[t])
                    if not ok.all():
                        idx = np.argwhere(~ok)
                        raise AssertionError("pr step %d: %d cells (first %s, rows %d..%d, cols %d..%d) grid %dx%d res %d boxes %d K %d rows %d %r L %r" % (
                            t, (~ok).sum(), idx[0], idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(), o.gx, o.gy, res,
                            len(boxes), s.info.stepsPerLaunch, s.info.tileRows, opts, L))
        nvalid = -1
        if tmax == o.T - 1 and not opts.get("streaming_analysis"):
            for cx, cy in rng.integers(0, o.gx, (4, 2)):
                ir = np.stack([hp[:, cx, cy], hx[:, cx, cy], hy[:, cx, cy]], 1)
                assert same_bits(s.impulse_response(int(cx), int(cy)), ir).all(), "IR"
            res8, delay = s.results()
            nvalid = compare_maps(res8, delay, rres, rdelay, o.T, o.fs, "seed %d" % seed)
        elif tmax == o.T - 1:
            res8, delay = s.results()
            cells = [pv.host_cells(size, size, res, e[0], e[2])[1] for e in emitters]
            check_streaming_against(res8, delay, rres, rdelay, o.T, o.fs, [c for c in cells if c], "seed %d" % seed)
            nvalid = int(((rdelay < 1e30)).sum())
        k, rows = s.info.stepsPerLaunch, s.info.tileRows
    o.close()
    return "seed %3d: %3dx%-3d res %d T %4d boxes %2d K %2d rows %2d %-40s %s" % (
        seed, o.gx, o.gy, res, o.T, len(boxes), k, rows, opts, "diverged at t=%d (compared up to there)" % (tmax + 1)
        if nvalid < 0 else "%d valid cells, all 8 outputs + delay + planes + IRs bit-identical" % nvalid)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t0 = time.time()
    bad = 0
    for seed in range(first, first + count):
        try:
            print(one(seed), flush=True)
        except AssertionError as e:
            bad += 1
            print("seed %3d: MISMATCH %s" % (seed, e), flush=True)
    print("%d scenes, %d mismatches, %.0f s" % (count, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
