#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FDTD + IR-analysis hot path on MI355X.

Workload (BASELINE.json configs[3], SURVEY.md 8d "config 4"): HugeRoom.pv in a 4096 x 4096 grid (Mode A: 275 Hz,
dx = 0.3566 m, 1460.737 m side, T = 435 steps), one independent simulation run (= one listener position) per step
per GPU; listener positions cycle through the eight of SURVEY.md 8d.  A "step" is one full pass of the hot path:
field reset + T fused leapfrog steps incl. pressure-history record + per-cell IR analysis (what one iteration of the
reference's background loop does, PvContext.cpp:80-83).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--grid 4096] [--inflight B] [--no-cpu-baseline]

The block of K steps -- barrier + device sync on both sides, max over ranks -- is timed R times back to back (default 5) in
one invocation; `ms_per_step` and `value` are the MEDIAN block, `spread` holds every block, min / max and their relative
spread: one 0.07 s block cannot tell a 3 % change from the box's noise.  `device` says what the line was measured on
(name, CUs, partition modes, shader clock sampled beside the in-flight warm-up runs and right after the timed blocks).

Runs in flight: independent runs share nothing, so a GPU can work on B of them at once (B solver instances, each with
its own planes and HIP stream; default B = 2).  One launch of the step kernel fills the chip for ~130 us and then
drains; a second run's launches fill the start-up and drain gaps of the first (+22 % cell-updates/s at 4096^2, +40 %
at 2048^2).  A step is then one batch of B listener positions per GPU; --inflight 1 runs them one at a time.

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL); runs are sharded round-robin
with no data-path collective and one all-gather of the per-emitter outputs at the end ("scaling": "weak").
Prints ONE JSON line on rank 0.

PV_BENCH_BACKEND (default "nccl" = RCCL) names the torch.distributed backend.  tests/test_dist_cpu.py runs this file's
N = 2 orchestration on CPU with PV_BENCH_BACKEND=gloo and a stand-in solver (main(argv, hooks=...)): process group,
communicator fail-over on every rank at once, barriers, max-over-ranks timing, the JSON line on rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 8 listener positions inside the 25 m room (x, z metres); emitter A = listener + (0, 2),
# emitter B = (5, 0, 6)
LISTENERS = [(5, 4), (8, 8), (12, 6), (15, 15), (20, 5), (5, 20), (20, 20), (12.5, 18)]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_CELL_STEP = 24  # read + write pr, vx, vy once


def mode_a_size(n, res=275):
    dx = np.float32(343.21) / np.float32(res) / np.float32(3.5)
    return float((n + 0.5) * dx)


def same_bits(a, b):
    """bit-identical float32, modulo the sign of zero (NaN == NaN)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))


def verify_records(gathered, scene, open_field, listener_index, grid=4096):
    """Compare EVERY gathered per-emitter record of the timed runs with the reference, bit for bit.
    HugeRoom.pv is a closed room: at any Mode A grid size the records equal the reference's 71^2 (25 m) run of the same
    listener / emitters (closed-room isolation); tests/golden/g71_hugeroom_cfg4.npz holds those 8 x 2
    records, generated from the compiled reference (tests/golden/make_golden.py cfg4).  Returns (verified_runs, how);
    raises if any record differs -- a fast run with wrong results is not a result."""
    if open_field and grid == 8192:
        # BASELINE config 5: records of the 64 seeded listener cells from the pinned oracle's 513^2 window analysed with the
        # 8192^2 grid's position arithmetic (tests/golden/make_golden.py cfg5; the open field is translation-invariant)
        g = np.load(os.path.join(ROOT, "tests", "golden", "g8192_open_cfg5.npz"))
        want = g["emitter_out"]
        bad = [k for k in range(gathered.shape[0]) if not same_bits(gathered[k], want[k % len(want)]).all()]
        if bad:
            raise AssertionError("bench: %d of %d timed runs differ from the reference records, first run %d: got %r "
                                 "want %r" % (len(bad), gathered.shape[0], bad[0], gathered[bad[0]], want[bad[0] % len(want)]))
        return int(gathered.shape[0]), ("every timed run's 2 emitter records bit-identical to the pinned oracle's (513^2 "
                                        "window, 8192^2 position arithmetic: tests/golden/g8192_open_cfg5.npz)")
    if open_field or scene != "HugeRoom.pv":
        return None, "no reference records for this workload (tests/test_gpu_configs.py covers configs 3 and 5)"
    g = np.load(os.path.join(ROOT, "tests", "golden", "g71_hugeroom_cfg4.npz"))
    want = g["emitter_out"]
    assert [tuple(l) for l in g["listeners"][:, [0, 2]].tolist()] == [tuple(map(float, l)) for l in LISTENERS]
    bad = []
    for k in range(gathered.shape[0]):
        if not same_bits(gathered[k], want[listener_index(k)]).all():
            bad.append(k)
    if bad:
        k = bad[0]
        raise AssertionError("bench: %d of %d timed runs differ from the reference records, first run %d: got %r want "
                             "%r" % (len(bad), gathered.shape[0], k, gathered[k], want[listener_index(k)]))
    return int(gathered.shape[0]), ("every timed run's 2 emitter records (8 floats each) bit-identical to the "
                                    "reference's 71^2 closed-room run of the same listener "
                                    "(tests/golden/g71_hugeroom_cfg4.npz)")


def cpu_baseline(grid_cells=1025, scene="HugeRoom.pv"):
    """Reference algorithm on ONE host core (the reference is single-threaded: SURVEY.md 6): the unmodified
    reference compiled into oracle/_ref/libpvref.so when present ("reference"), else the C restatement ("port").
    Bounded sample: the same scene / dx / T on a (grid_cells)^2 cell array."""
    from oracle import pvref, pvoracle
    scene_path = os.path.join(ROOT, "tests", "scenes", scene)
    size = mode_a_size(grid_cells - 1)
    L = (5.0, 0.0, 4.0)
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    if pvref.available():
        boxes = pvref.load_pv(scene_path)
        r = pvref.RefSolver(size, size, 275, boxes)
        tf = r.generate(L)
        ta = r.analyze(L)
        cells = (r.gx + 1) * (r.gy + 1)
        out = dict(value=cells * r.T / tf, unit="cell-updates/s", cores=1, kind="reference",
                   sample="%s, %dx%d cells (Mode A, 275 Hz), T=%d, 1 listener: FDTD %.2f s, analysis %.2f s "
                          "(grid ctor %.1f s + FreeGrid ctor %.1f s not counted)" % (
                              scene, r.gx + 1, r.gy + 1, r.T, tf, ta, r.ctor_grid_s, r.ctor_free_s),
                   ir_per_s=r.gx * r.gy / ta, fdtd_s=tf, analysis_s=ta)
        r.close()
    else:
        boxes = pvref.load_pv(scene_path)
        o = pvoracle.OracleGrid(size, size, 275, boxes)
        t0 = time.perf_counter()
        o.fdtd(L)
        tf = time.perf_counter() - t0
        t0 = time.perf_counter()
        o.analyze(np.float32(0.0447895788), L)
        ta = time.perf_counter() - t0
        out = dict(value=o.ncell * o.T / tf, unit="cell-updates/s", cores=1, kind="port",
                   sample="%s, %dx%d cells, T=%d (C restatement, SoA history)" % (scene, o.gx + 1, o.gy + 1, o.T),
                   ir_per_s=o.gx * o.gy / ta, fdtd_s=tf, analysis_s=ta)
        o.close()
    try:
        os.sched_setaffinity(0, set(range(os.cpu_count())))
    except Exception:
        pass
    out["host_cpus"] = os.cpu_count()
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        out["cpu_model"] = None
    # (the bounded sample is a 1025^2 grid so that the default run stays short.  --cpu-baseline-cells 4097 times the same
    # reference at the headline's full size in this very run: two impulse-response cubes of 117 GB each in host memory and
    # ~65 s of FDTD on one core; measured that way once, on a GPU box's host: profiles/r03_cpu_reference_4097.txt.)
    out["sample_cells"] = grid_cells
    out["full_size_how"] = ("--cpu-baseline-cells 4097 (needs ~240 GB of host memory for the reference's two T-step cubes, ~65 s of FDTD); "
                            "a record of such a run: profiles/r03_cpu_reference_4097.txt")
    return out


class GpuHooks:
    """what main() needs from the machine: the real thing.  (tests/_bench_worker.py substitutes a CPU stand-in to run the
    multi-rank orchestration under gloo.)"""
    backend_default = "nccl"

    def __init__(self, local_rank):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs a HIP device (no CPU path)")
        # PV_BENCH_DEVICES="0,0": device of every local rank (development: the multi-rank orchestration on a one-GPU box --
        # with PV_BENCH_BACKEND=gloo, since RCCL refuses two ranks on one device; the collectives then carry host tensors)
        devmap = os.environ.get("PV_BENCH_DEVICES")
        self.rank_index = local_rank
        if devmap:
            local_rank = int(devmap.split(",")[local_rank])
        torch.cuda.set_device(local_rank)
        self.torch = torch
        self.local_rank = local_rank
        host = os.environ.get("PV_BENCH_BACKEND", self.backend_default) != "nccl"
        self.device = torch.device("cpu") if host else torch.device("cuda", local_rank)

    def init_process_group(self, dist, backend, rank, world):
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=self.torch.device("cuda", self.local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def make_solver(self, size, res, **opts):
        from planeverb_amd import api
        return api.Solver(size, size, res, device=self.local_rank, **opts)

    def batch_solver_options(self, grid):
        from planeverb_amd import api
        return api.batch_solver_options(grid)

    def run_batch(self, solvers, listeners, wait):
        from planeverb_amd import api
        return api.run_batch(solvers, listeners, wait=wait)

    def make_comm(self, dist):
        from planeverb_amd import dist as pvd
        return pvd.make_comm(dist, self.local_rank)

    def device_sync(self):
        self.torch.cuda.synchronize()

    def clock_probe(self):
        from planeverb_amd import api
        return api.clock_probe(self.local_rank)[0]

    def bandwidth_probe(self):
        """the box's own streaming bandwidth (PvAmdBandwidthProbe: device-to-device copy, read, write; ~60 ms, idle device)"""
        from planeverb_amd import api
        return api.bandwidth_probe(self.local_rank)

    def device_record(self):
        p = self.torch.cuda.get_device_properties(self.local_rank)
        rec = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
               "hbm_bytes": int(p.total_memory), "max_clock_mhz": getattr(p, "clock_rate", 0) / 1e3 or None,
               "torch_hip": self.torch.version.hip}

        def sysfs(name):  # amdgpu exposes the partition modes per card
            import glob
            vals = []
            for f in sorted(glob.glob("/sys/class/drm/card*/device/" + name)):
                try:
                    vals.append(open(f).read().strip())
                except Exception:
                    pass
            return vals[self.local_rank] if len(vals) > self.local_rank else (vals[0] if vals else None)
        rec["compute_partition"] = sysfs("current_compute_partition")
        rec["memory_partition"] = sysfs("current_memory_partition")
        return rec

    def barrier(self, dist, backend):
        if backend == "nccl":
            dist.barrier(device_ids=[self.local_rank])
        else:
            dist.barrier()

    def power_watts(self):
        """socket power of this GPU right now (amdgpu hwmon, microwatts), or None"""
        import glob
        for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
            fs = sorted(glob.glob(pat))
            if fs:
                try:
                    return float(open(fs[min(self.local_rank, len(fs) - 1)]).read().strip()) / 1e6
                except Exception:
                    return None
        return None

    def dense_leg(self, solvers_in_flight, K, T, cells, seconds=1.2):
        """roofline.dense: the raw stencil (PvAmdRunSteps: the dominant kernel and nothing else) on SEEDED
        RANDOM fields -- every cell non-zero -- at the bench's grid, tile and number of runs in flight, behind the timed region.
        The headline scene keeps the field at zero outside a 70^2-cell room (the closed-room workload BASELINE names): same
        instructions, but operands that toggle no bits.  Reports the wall rate, the launch durations (HIP events around each
        call's back-to-back launches: p50 / p90 over the calls), the shader clock and the socket power beside it, and the same
        for ZERO fields measured the same way right after, so that the two are comparable call for call."""
        import threading
        rng = np.random.default_rng(1)
        launches = max(1, T // K)
        out = {"fields": "uniform random in (-5e-4, 5e-4) for pr, vx, vy of every cell (numpy default_rng(1)), scene geometry in place",
               "steps_per_call": launches * K, "runs_in_flight": len(solvers_in_flight)}

        def leg(tag):
            per_call, clocks, power = [], [], []
            stop = [False]

            def work(sv):
                while not stop[0]:
                    sv.run_steps(launches * K)
                    per_call.append(sv.timings().fdtdMs / launches)

            def watch():
                while not stop[0]:
                    clocks.append(self.clock_probe())
                    pw = self.power_watts()
                    if pw is not None:
                        power.append(pw)
                    time.sleep(0.05)
            for sv in solvers_in_flight:  # warm (first launch from these fields)
                sv.run_steps(K)
            self.device_sync()
            per_call.clear()
            th = [threading.Thread(target=work, args=(sv,)) for sv in solvers_in_flight] + [threading.Thread(target=watch)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            time.sleep(seconds)
            stop[0] = True
            for t in th:
                t.join()
            self.device_sync()
            dt = time.perf_counter() - t0
            n = len(per_call)
            pc = np.sort(np.asarray(per_call)) if n else np.zeros(1)
            out[tag] = {"value": n * launches * K * cells / dt, "unit": "cell-updates/s (wall)", "calls": n,
                        "launch_ms_p50": float(pc[len(pc) // 2]), "launch_ms_p90": float(pc[min(len(pc) - 1, int(0.9 * len(pc)))]),
                        "clock_mhz_median": float(np.median(clocks)) if clocks else None,
                        "clock_mhz_min": float(np.min(clocks)) if clocks else None,
                        "power_w_median": float(np.median(power)) if power else None,
                        "power_w_max": float(np.max(power)) if power else None}
        try:
            for sv in solvers_in_flight:
                shp = (sv.gx + 1, sv.gy + 1)
                sv.set_fields(*[(rng.random(shp, np.float32) - np.float32(0.5)) * np.float32(1e-3) for _ in range(3)])
            leg("random")
            for sv in solvers_in_flight:
                z = np.zeros((sv.gx + 1, sv.gy + 1), np.float32)
                sv.set_fields(z, z, z)
            leg("zero")
            out["random_over_zero"] = out["random"]["value"] / out["zero"]["value"]
        except Exception as e:  # noqa: BLE001  (e.g. a tile configuration without stencil-only stepping)
            out["skipped"] = str(e)
        return out


def main(argv=None, hooks=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the block of --steps steps is timed this many times back to back; value = the median block")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--scene", default="HugeRoom.pv", help="a .pv file under tests/scenes, or 'none' (empty grid)")
    ap.add_argument("--open-field", action="store_true",
                    help="SURVEY.md 8d config 5: empty scene, listener cells from numpy.random.default_rng(0)"
                         ".integers(N/8, 7N/8, (64, 2)), emitters = listener + (16, 0) and + (0, 16) cells")
    ap.add_argument("--steps-per-launch", type=int, default=0)
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--dense-history", type=int, default=0)
    ap.add_argument("--edge-tiles", type=int, default=-1,
                    help="PVA_OPT_EDGE_TILES: 1 = grid-border tiles on the air path (every run then goes through the "
                         "batched kernel, also with --batch 1)")
    ap.add_argument("--tile-order", type=int, default=-1, help="PVA_OPT_TILE_ORDER (development: block -> tile map)")
    ap.add_argument("--xcd-regions", type=int, default=-1, help="PVA_OPT_XCD_REGIONS (development: 2 x 4 regions / 8 strips)")
    ap.add_argument("--alternate-sweeps", type=int, default=-1,
                    help="PVA_OPT_ALTERNATE_SWEEPS (development: odd launches walk the tiles backwards)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip roofline.dense (the raw stencil on random fields, ~3 s)")
    ap.add_argument("--cpu-baseline-cells", type=int, default=1025,
                    help="side of the cell array of the bounded CPU-baseline sample (same scene, dx and T)")
    ap.add_argument("--stream-priority", type=int, default=0,
                    help="1: every other in-flight group's solvers get PVA_OPT_STREAM_PRIORITY (a hardware queue of another "
                         "priority; measured 6 %% SLOWER at 4096^2: the two runs then take turns instead of sharing the chip)")
    ap.add_argument("--placement-tries", type=int, default=0,
                    help="(accepted and ignored: stream placement moved into the library in round 5, Solver::claimOwnQueue)")
    ap.add_argument("--aux-streams", type=int, default=-1,
                    help="PVA_OPT_AUX_STREAMS of every solver (idle streams that shift which hardware queue the next solver's "
                         "streams are dealt); -1: the library's / batch_solver_options' default")
    ap.add_argument("--inflight", type=int, default=2,
                    help="independent runs a GPU works on concurrently (one solver instance + stream each); a step "
                         "is one batch of this many listener positions per GPU")
    ap.add_argument("--batch", type=int, default=1,
                    help="runs per batched launch (PvAmdRunBatch, <= 8): every one of the --inflight groups is a "
                         "batch of this many solvers advanced by ONE launch per K steps; the lever for launch-bound "
                         "grids (<= 1024^2), no gain at 4096^2; 0 = by grid size")
    ap.add_argument("--use-graph", type=int, default=0, help="0 auto (grids of <= 4096 tiles), 1 always, 2 never")
    ap.add_argument("--time-kernels", type=int, default=0,
                    help="N > 0: HIP events around every Nth step-kernel launch instead of around the whole launch loop "
                         "(the extra events cost 0.2-0.5 ms per run)")
    args = ap.parse_args(argv)

    import torch  # first: libplaneverb_amd.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    from planeverb_amd import dist as pvd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if hooks is None:
        hooks = GpuHooks(local_rank)
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one rank per GPU: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." % (
                             args.gpus, args.gpus, args.gpus))
    dev = hooks.device
    backend = os.environ.get("PV_BENCH_BACKEND", hooks.backend_default)
    use_dist = world > 1 or os.environ.get("PV_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank RCCL self-test
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        hooks.init_process_group(dist, backend, rank, world)

    size = mode_a_size(args.grid)
    # roofline launch duration: HIP events on the solver's stream around the back-to-back step launches of every timed
    # run (PvAmdTimings.stepLoopMs), or around every Nth single launch with --time-kernels N
    opts = dict(time_kernels=args.time_kernels, use_graph=args.use_graph)
    if args.steps_per_launch:
        opts["steps_per_launch"] = args.steps_per_launch
    if args.tile_rows:
        opts["tile_rows"] = args.tile_rows
    if args.dense_history:
        opts["dense_history"] = 1
    if args.tile_order >= 0:
        opts["tile_order"] = args.tile_order
    if args.edge_tiles >= 0:
        opts["edge_tiles"] = args.edge_tiles
    if args.alternate_sweeps >= 0:
        opts["alternate_sweeps"] = args.alternate_sweeps
    if args.xcd_regions >= 0:
        opts["xcd_regions"] = args.xcd_regions
    if args.batch == 0:  # auto: batched launches pay for launch-bound grids only
        args.batch = 8 if args.grid <= 1536 else 1
    NB = max(1, min(args.batch, 8))  # runs per batched launch
    if NB > 1 and not args.steps_per_launch and not args.tile_rows:
        opts.update(hooks.batch_solver_options(args.grid))  # mirror-pair tile + edge tiles (batched kernel only)
    G = max(1, args.inflight)         # groups in flight (one stream each)
    B = G * NB                        # runs per step and GPU
    # every other group's streams on the device's highest priority: streams of different priorities never share a hardware queue
    # (PVA_OPT_STREAM_PRIORITY; two groups whose step loops land on one queue run their launches one after the other)
    if args.aux_streams >= 0:
        opts["aux_streams"] = args.aux_streams
    if args.open_field:
        args.scene = "none"

    def new_solver(i):
        sv = hooks.make_solver(size, 275, **dict(opts, stream_priority=((i // NB) & 1) if args.stream_priority else 0))
        if args.scene != "none":
            sv.load_scene(os.path.join(ROOT, "tests", "scenes", args.scene))
        return sv

    solvers = [new_solver(i) for i in range(B)]
    s = solvers[0]
    of_cells = np.random.default_rng(0).integers(args.grid // 8, 7 * args.grid // 8, size=(64, 2))
    cells = (s.gx + 1) * (s.gy + 1)
    T = s.T

    def run_id(step, b):  # global index of the run solver b of this rank simulates in `step`
        return (step * B + b) * world + rank

    def listener(step, b):
        if args.open_field:  # cell centres, so that the metre -> cell truncation cannot land on a neighbour
            cx, cz = of_cells[run_id(step, b) % len(of_cells)]
            return ((cx + 0.5) * float(s.dx), 0.0, (cz + 0.5) * float(s.dx))
        x, z = LISTENERS[run_id(step, b) % len(LISTENERS)]
        return (float(x), 0.0, float(z))

    def emitters(step, b):
        x, _, z = listener(step, b)
        if args.open_field:
            return [(x + 16 * float(s.dx), 0.0, z), (x, 0.0, z + 16 * float(s.dx))]
        return [(x, 0.0, z + 2.0), (5.0, 0.0, 6.0)]

    def start_group(step, g):  # enqueue the NB runs of group g: solvers g*NB .. g*NB+NB-1
        for b in range(g * NB, (g + 1) * NB):
            solvers[b].set_output_queries(emitters(step, b))
        if NB == 1:
            solvers[g].run_async(listener(step, g))
        else:
            hooks.run_batch(solvers[g * NB:(g + 1) * NB], [listener(step, g * NB + j) for j in range(NB)], False)

    def sync():
        hooks.device_sync()
        if use_dist:
            hooks.barrier(dist, backend)

    # warm-up: the first round one run at a time, which also gives the step kernel's duration with a single run in
    # flight; further rounds in flight together like the timed steps
    single_loop_ms, clock_loaded = [], []
    t_setup0 = time.perf_counter()
    for w in range(args.warmup):
        if w == 0:
            for b, sv in enumerate(solvers):
                sv.run(listener(w, b))
                single_loop_ms.append(sv.timings().stepLoopMs or sv.timings().fdtdMs)
        else:
            for g in range(G):
                start_group(w, g)
            if hasattr(hooks, "clock_probe"):
                clock_loaded.append(hooks.clock_probe())  # (a stream of its own: beside the runs in flight)
            for sv in solvers:
                sv.sync()
    # Stream placement is the LIBRARY's business since round 5: a solver's main stream is checked at creation against the other
    # live solvers' (Solver::claimOwnQueue: do the two take turns on one dispatch pipe?) and re-dealt if so -- what rounds 3-4 did
    # here, for this program only, by timing the groups together and one after the other and re-creating solvers.
    placement = {"in": "libplaneverb_amd.so (Solver::claimOwnQueue, PLANEVERB_AMD_QUEUE_PROBE)"}
    # The one collective of the data path: the C++ side's own RCCL communicator (PvAmdComm: ncclCommInitRank /
    # ncclAllGather inside libplaneverb_amd.so; torch.distributed only carries the 128-byte id to the ranks).  Should
    # RCCL not bind there, torch.distributed's all_gather does the same job and the JSON line says so.
    comm, gather_how = None, "single process: no collective"
    warmup_s = time.perf_counter() - t_setup0
    t_comm0 = time.perf_counter()
    if use_dist and os.environ.get("PV_BENCH_GATHER") == "torch":
        gather_how = "torch.distributed.all_gather_into_tensor (PV_BENCH_GATHER=torch)"
        pvd.gather_outputs({run_id(0, b): np.zeros((2, 8), np.float32) for b in range(B)}, B * world, dist, dev)
    elif use_dist:
        try:
            comm = hooks.make_comm(dist)
            gather_how = "ncclAllGather in libplaneverb_amd.so (PvAmdCommAllGather), id bootstrapped over torch.distributed"
        except Exception as e:  # noqa: BLE001
            gather_how = "torch.distributed.all_gather_into_tensor (C++ RCCL communicator unavailable: %s)" % e
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:  # all ranks or none
            comm.close()
            comm = None
            gather_how = "torch.distributed.all_gather_into_tensor (C++ RCCL communicator failed on another rank)"
        # first use sets up RCCL's channels: not part of the timed steps
        warm = {run_id(0, b): np.zeros((2, 8), np.float32) for b in range(B)}
        if comm is not None:
            pvd.gather_outputs_native(warm, B * world, comm, n_em=2)
        else:
            pvd.gather_outputs(warm, B * world, dist, dev)
    comm_init_s = time.perf_counter() - t_comm0  # communicator + first (channel set-up) gather
    n_runs = args.steps * B * world
    local = {}
    fdtd_ms, ana_ms, air_ms, gen_ms, loop_ms, reached = [], [], [], [], [], []
    pending = [None] * B

    def collect(b):  # wait for solver b's run, fetch its per-emitter outputs and timings
        sv, k = solvers[b], pending[b]
        sv.sync()
        local[run_id(k, b)] = sv.queried_outputs()  # gathered behind the run's analysis (PvAmdSetOutputQueries)
        t = sv.timings()
        fdtd_ms.append(t.fdtdMs)
        ana_ms.append(t.analysisMs)
        air_ms.append(t.airKernelMs)
        gen_ms.append(t.generalKernelMs)
        loop_ms.append(t.stepLoopMs)
        reached.append(getattr(t, "reachedCells", 0))
        pending[b] = None

    def timed_block():  # EXACTLY args.steps steps between barrier + device sync, max over ranks
        local.clear()
        sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            for g in range(G):
                for b in range(g * NB, (g + 1) * NB):
                    if pending[b] is not None:
                        collect(b)  # the other groups' runs keep the GPU busy meanwhile
                start_group(k, g)
                for b in range(g * NB, (g + 1) * NB):
                    pending[b] = k
        for b in range(B):
            if pending[b] is not None:
                collect(b)
        if comm is not None:
            got = pvd.gather_outputs_native(local, n_runs, comm, n_em=2)  # the one RCCL gather, in C++
        else:
            got = pvd.gather_outputs(local, n_runs, dist if use_dist else None, dev)
        sync()
        mine = time.perf_counter() - t0
        el = mine
        if use_dist:
            tt = torch.tensor([mine], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, mine, got

    blocks, mine_blocks, gathered_all = [], [], []
    for _ in range(max(1, args.repeats)):
        el, mine, got = timed_block()
        blocks.append(el)
        mine_blocks.append(mine)
        gathered_all.append(got)
    clock_after = hooks.clock_probe() if hasattr(hooks, "clock_probe") else None
    elapsed = float(np.median(blocks))
    gathered = gathered_all[0]
    # per-rank record (a first multi-GPU run must be diagnosable from the one line rank 0 prints)
    rank_rec = {"rank": rank, "local_rank": local_rank, "block_s": [round(x, 6) for x in mine_blocks],
                "warmup_s": round(warmup_s, 3), "comm_init_and_first_gather_s": round(comm_init_s, 3),
                "gather": gather_how, "clock_mhz_loaded": clock_loaded, "clock_mhz_after": clock_after,
                "stream_placement": placement}
    ranks = [rank_rec]
    if use_dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_rec)

    if rank == 0:
        verified_runs, verified_how = 0, None
        for got in gathered_all:  # every timed block's records
            assert got.shape == (n_runs, 2, 8)
            # run k of the gathered array = global run index k (gather_outputs orders by run index) = listener k mod 8
            v, verified_how = verify_records(got, args.scene, args.open_field, lambda k: k % len(LISTENERS), args.grid)
            verified_runs = None if v is None else verified_runs + v
        info = s.info
        K = info.stepsPerLaunch
        launches = s.timings().stepLaunches
        steps_per_launch_avg = K
        if args.time_kernels > 0:
            # sampled single launches (only full K-step launches are sampled)
            air = float(np.mean(air_ms))
            how = "HIP events around every %d. launch" % args.time_kernels
        else:
            # mean over the timed runs of (launch-loop time) / (T / K): the duration of one full K-step launch with
            # the short remainder launch counted by its share of steps
            air = float(np.mean(loop_ms)) * K / T
            how = "HIP events around the %d back-to-back launches of each timed run, x K/T" % launches
            if air == 0.0 and NB > 1:  # batched: one launch advances NB runs; events around the batch's launch loop
                air = float(np.mean(fdtd_ms)) * K / T
                how = "HIP events around the %d batched launches (%d runs each) of every batch, x K/T" % (launches, NB)
            elif air == 0.0:  # the run was replayed from a hipGraph (<= 4096 tiles): events sit around the whole graph
                air = float(np.mean(fdtd_ms)) * K / T
                how = "HIP events around the graph replay (field reset + %d launches), x K/T" % launches
        alg_bytes = ALG_BYTES_PER_CELL_STEP * cells * steps_per_launch_avg
        # B runs are in flight: B launches of the kernel overlap, each taking `air` ms, so the chip retires B launches'
        # algorithmic bytes per `air`.  The single-run figures come from the warm-up runs (one run at a time).
        achieved = B * alg_bytes / (air * 1e-3) / 1e9
        single = None
        if single_loop_ms and args.time_kernels == 0:
            sl = float(np.mean(single_loop_ms)) * K / T
            single = {"launch_ms": sl, "achieved": alg_bytes / (sl * 1e-3) / 1e9,
                      "frac": alg_bytes / (sl * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "from": "the first %d warm-up runs, made one at a time" % len(single_loop_ms)}
        # Counter profile of the dominant kernel (profiles/hbm_traffic.json, written by tools/summarize_profiles.py from
        # rocprofv3 --pmc passes): quoted only if it was collected on THIS device code (kernel_source_hash) and tile
        from planeverb_amd.build import kernel_source_hash
        traffic, traffic_note, prof, sqprof, allp = None, None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc):
            try:
                allp = json.load(open(pmc))
                prof = allp.get("%d" % args.grid)
                sqprof = allp.get("sq_%d" % args.grid)
            except Exception:
                prof = None
        for pr in (prof, sqprof):
            if pr is not None and pr.get("kernel_source_hash") != kernel_source_hash():
                traffic_note = ("profiles/hbm_traffic.json is stale: collected on kernel sources %s at %s, this build is "
                                "%s -- counters not quoted" % (pr.get("kernel_source_hash"), pr.get("collected_at_head"),
                                                               kernel_source_hash()))
                prof = sqprof = None
                break
        if prof is not None and ("<%d, %d," % (K, info.tileRows)) not in prof.get("kernel", ""):
            traffic_note = "profiles/hbm_traffic.json holds another tile configuration (%s)" % prof.get("kernel")
            prof = None
        if prof is not None:
            traffic = prof.get("bytes_per_launch")
        # executed vs useful cell-steps of one air tile and launch: step s advances rows [s, ROWS - s - 1) of 64 lanes
        ROWS = info.tileRows + 2 * K
        executed = sum((ROWS - 2 * st - 1) * 64 for st in range(K)) if ROWS % 2 == 0 else ROWS * 64 * K
        useful_lane_fraction = info.tileRows * info.tileCols * K / executed
        hbm = None
        if traffic:
            hbm = {"bytes_per_launch": traffic, "achieved": G * traffic / (air * 1e-3) / 1e9, "unit": "GB/s",
                   "frac": G * traffic / (air * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "vs_algorithmic": traffic / (alg_bytes * NB),
                   "collected_at_head": prof.get("collected_at_head"), "source": "profiles/hbm_traffic.json (rocprofv3 "
                   "--pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections calibrated on the box)"}
        valu = None
        if sqprof and "zero1" in sqprof:
            z = sqprof["zero1"]
            q = z["counters"].get("SQ_ACTIVE_INST_VALU")
            clk = z.get("effective_clock_ghz")
            if q and clk:
                valu = {"issue_utilisation": q * 4 / 1024 / ((air / G) * 1e-3 * clk * 1e9),
                        "issue_utilisation_single_launch_measured": z.get("valu_busy_single_launch"),
                        "valu_quad_cycles_per_launch": q, "effective_clock_ghz": clk,
                        "how": "SQ_ACTIVE_INST_VALU (quad-cycles per launch, rocprofv3 --pmc) x 4 / 1024 SIMDs / (live "
                               "launch_ms / concurrent_launches x effective clock = GRBM_GUI_ACTIVE / 8 / duration)",
                        "collected_at_head": sqprof.get("collected_at_head"),
                        "source": "profiles/%s_sq_pmc.md" % sqprof.get("round", "r02")}
        value = world * B * cells * T * args.steps / elapsed
        fd = float(np.mean(fdtd_ms)) * 1e-3
        ms_blocks = [b / args.steps * 1e3 for b in blocks]
        spread = {"repeats": len(blocks), "ms_per_step_all": [round(x, 4) for x in ms_blocks],
                  "ms_per_step_min": min(ms_blocks), "ms_per_step_max": max(ms_blocks),
                  "ms_per_step_median": float(np.median(ms_blocks)),
                  "rel_spread": (max(ms_blocks) - min(ms_blocks)) / float(np.median(ms_blocks)),
                  "value_best": world * B * cells * T * args.steps / min(blocks),
                  "value_worst": world * B * cells * T * args.steps / max(blocks),
                  "how": "the block of %d steps timed %d times back to back in this invocation (barrier + device sync "
                         "around each, max over ranks); value / ms_per_step = the median block" % (args.steps, len(blocks))}
        device = hooks.device_record() if hasattr(hooks, "device_record") else {}
        device["effective_clock_ghz"] = (float(np.median(clock_loaded)) / 1e3) if clock_loaded else None
        device["effective_clock_how"] = ("one wave's timed s_sleep (812 800 shader-clock cycles against the 100 MHz counter, "
                                         "PvAmdClockProbe) beside the in-flight warm-up runs: median of %d samples; "
                                         "clock_mhz_after = one more sample right behind the last timed block" % len(clock_loaded))
        device["clock_mhz_after"] = clock_after
        # the analysis half of the metric: impulse responses per second, counted on the cells that HAVE one (an onset),
        # and the bytes their analysis must read at least: every recorded pressure sample of a reached cell once
        # (4 B x T), + 36 B of results per cell.  The other cells of the map leave the analysis at once.
        n_reached = float(np.mean(reached)) if reached else 0.0
        ana_s = float(np.mean(ana_ms)) * 1e-3
        ana_alg = n_reached * (4.0 * T + 36.0)
        ana_prof = None
        try:
            ana_prof = (allp or {}).get("analysis_%d" % args.grid)
            if ana_prof is not None and ana_prof.get("kernel_source_hash") != kernel_source_hash():
                ana_prof = None
        except Exception:
            ana_prof = None
        analysis = {
            "reached_cells_per_run": n_reached, "cells_per_run": s.gx * s.gy,
            "reached_ir_per_s": world * B * n_reached * args.steps / elapsed,
            "analysis_only_reached_ir_per_s": (n_reached / ana_s) if ana_s > 0 else None,
            "analysis_ms": ana_s * 1e3, "bound": "latency" if n_reached < 65536 else "hbm",
            "algorithmic_bytes_per_run": ana_alg,
            "achieved": (ana_alg / ana_s / 1e9) if ana_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (ana_alg / ana_s / 1e9 / HBM_PEAK_GBS) if ana_s > 0 else None,
            "traffic": (ana_prof or {}).get("bytes_per_run"),
            "traffic_source": (ana_prof or {}).get("source"),
            "note": "kernels: pv_far_frame + pv_encode (onset, dry gain, flux, lowpass) + pv_rt60_wave / pv_rt60_blocked (wet "
                    "gain, decay time) + listener direction; analysis_ms = HIP events around the chain of one run, beside the "
                    "other runs in flight.  A closed room in a large grid reaches a few thousand cells: the chain is bound by "
                    "its dependent launches and memory round trips, not by bytes; profiles/r04_analysis_pmc.md has the "
                    "all-cells-reached workload (Shoebox 25 m at 512^2, T = 3179)"}
        out = {
            "metric": "grid_cell_updates_per_s", "value": value, "unit": "cell-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "repeats": len(blocks), "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s in a %dx%d grid (Mode A: 275 Hz, dx=%.4f m, %.3f m), T=%d; 1 run = reset + T "
                                   "leapfrog steps with pr-history record + per-cell IR analysis for one listener "
                                   "position; 1 step = one batch of %d independent run(s) per GPU, in flight "
                                   "together; %d run(s) per step across %d GPU(s)" % (
                                       "empty scene (open field, SURVEY.md 8d config 5 listener cells)"
                                       if args.open_field else args.scene, s.gx, s.gy, s.dx, size, T, B, B * world,
                                       world),
                       "grid": [s.gx, s.gy], "T": T, "res": 275, "mode": "A", "steps_per_launch": K,
                       "tile": [info.tileRows, info.tileCols], "dense_history": bool(args.dense_history),
                       "runs_in_flight_per_gpu": B, "runs_per_batched_launch": NB,
                       "parallelism": "runs sharded round-robin, 1 all-gather of outputs", "gather": gather_how,
                       "backend": backend if use_dist else None},
            "fdtd_cell_updates_per_s": world * B * cells * T / fd,
            "impulse_responses_per_s": world * B * s.gx * s.gy * args.steps / elapsed,
            "impulse_responses_per_s_note": "all cells of the map incl. the ones without an onset (early-out); "
                                            "roofline.analysis.reached_ir_per_s counts the analysed ones",
            "spread": spread, "device": device, "ranks": ranks,
            "fdtd_ms": fd * 1e3, "analysis_ms": float(np.mean(ana_ms)),
            "hbm_bytes_held": int(info.deviceBytes) * B,
            "verified_runs": verified_runs, "timed_runs": n_runs * len(blocks), "verified_how": verified_how,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "basis": "achieved / peak / frac are on SURVEY.md 8d's basis: ALGORITHMIC bytes (24 B per "
                                  "cell-step) against the 8 TB/s HBM peak.  K-step temporal blocking in registers moves "
                                  "9-10x fewer bytes than that, so frac > 1: what binds is named in `bound` from the two measured "
                                  "fractions, VALU issue (valu.frac) and the counter-HBM bytes against the box's own copy rate "
                                  "(hbm.frac_of_measured_copy); useful_lane_fraction says how much of the issued arithmetic is halo",
                         "valu": valu, "hbm": hbm, "useful_lane_fraction": useful_lane_fraction,
                         "kernel": "pv_step_merged_kernel<K=%d,rows=%d> (air tiles + general tiles, one launch per K "
                                   "steps)" % (K, info.tileRows),
                         "launch_ms": air, "launch_ms_from": how, "launches_per_run": launches,
                         "concurrent_launches": G, "runs_per_launch": NB,
                         "algorithmic_bytes_per_launch": alg_bytes * NB,
                         "single_run": single, "analysis": analysis,
                         "note": "algorithmic = 24 B per cell-step x cells x K fused steps; K-step temporal "
                                 "blocking makes frac > 1 possible (SURVEY.md 8d).  achieved = concurrent_launches x "
                                 "algorithmic_bytes_per_launch / launch_ms: the launches of the runs in flight "
                                 "overlap, each lasting launch_ms; single_run = the same kernel with one run in "
                                 "flight"},
        }
        # rank 0 only, after the timed region (the other ranks wait at the final barrier)
        # "confirm on the box with a device-to-device copy micro-benchmark and report against both" -- the box's own
        # copy / read / write rates, measured now (idle device), and the kernel's fractions of them beside the fractions of the spec
        rl = out["roofline"]
        if hasattr(hooks, "bandwidth_probe"):
            try:
                bw = hooks.bandwidth_probe()
                copy = max(bw["copy_x4"], bw["copy_dword"])
                rl["measured_bandwidth"] = dict(bw, unit="GB/s", copy=copy,
                                                how="PvAmdBandwidthProbe behind the timed region: 1 GiB buffers, best of 5 launches per leg "
                                                    "(copy = bytes read + written per second; read / write = 4 B per lane in 256-byte rows per "
                                                    "wave, the stencil kernels' pattern)")
                rl["frac_of_measured_copy"] = achieved / copy  # (algorithmic bytes, like frac)
                if hbm:
                    hbm["frac_of_measured_copy"] = hbm["achieved"] / copy
                    # reads against the read-only rate and writes against the write-only rate, summed: the time the kernel's own
                    # mix of bytes would take at the box's one-directional rates, over the time it took
                    if prof.get("read_bytes") and prof.get("write_bytes") and bw["read_dword"] > 0 and bw["write_dword"] > 0:
                        t_mix = (prof["read_bytes"] / bw["read_dword"] + prof["write_bytes"] / bw["write_dword"]) / 1e9
                        hbm["frac_of_measured_read_write_mix"] = G * t_mix / (air * 1e-3)
            except Exception as e:  # noqa: BLE001
                rl["measured_bandwidth"] = {"skipped": str(e)}
        if valu:
            valu["frac"] = valu["issue_utilisation"]
        fr_valu = (valu or {}).get("frac")
        fr_hbm = (hbm or {}).get("frac_of_measured_copy")
        if fr_valu is not None and fr_hbm is not None:
            rl["bound"] = "valu+hbm-path" if min(fr_valu, fr_hbm) >= 0.75 else ("valu" if fr_valu >= fr_hbm else "hbm-path")
            rl["bound_how"] = ("VALU issue %.2f of the SIMDs' rate, counter-HBM bytes at %.2f of the box's measured copy rate: named "
                               "together when both are >= 0.75" % (fr_valu, fr_hbm))
        else:
            rl["bound_how"] = "counter profile not quoted (see traffic_note): the bound named is round 5's finding"
        if not args.no_dense_leg and hasattr(hooks, "dense_leg") and NB == 1:
            dense = hooks.dense_leg(solvers[:G], K, T, cells)
            if "random" in dense:
                dense["random"]["frac"] = dense["random"]["value"] * ALG_BYTES_PER_CELL_STEP / 1e9 / HBM_PEAK_GBS
                dense["random"]["vs_headline_fdtd_rate"] = dense["random"]["value"] / (world * B * cells * T / fd) * world
            out["roofline"]["dense"] = dense
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_baseline_cells)
        line = json.dumps(out)
    for sv in solvers:
        sv.close()
    if comm is not None:
        comm.close()
    if use_dist:
        hooks.barrier(dist, backend)
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner (NCCL_DEBUG=VERSION in this image) through C stdio, which is flushed at
        # exit when stdout is a pipe: flush it now so that the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
