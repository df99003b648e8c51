"""Worker process of tests/test_dist_cpu.py::test_bench_orchestration_gloo_world2: ONE rank of bench.py's N = 2 run on CPU.

TEST INFRASTRUCTURE.  bench.py's multi-GPU code path (process group, C++ communicator with fail-over to the torch gather on
every rank at once, warm-up gather, barrier + sync around the timed steps, max-over-ranks timing, verification of every
gathered record, the JSON line on rank 0 only with cpu_baseline kept) has only ever met world = 1 on the GPU pool.  Here it
runs with world = 2 under gloo (PV_BENCH_BACKEND=gloo): the machine-facing hooks are replaced by CPU stand-ins, the
orchestration is bench.py's own.  The stand-in solver answers with the reference's own records for its listener
(tests/golden/g71_hugeroom_cfg4.npz), so bench.py's bit-for-bit verification of the gathered array also checks that the
gather puts every run where its index says."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "g71_hugeroom_cfg4.npz"))


class FakeBenchSolver:
    """the slice of planeverb_amd.api.Solver that bench.py uses"""
    made = 0

    def __init__(self, size, res, **opts):
        FakeBenchSolver.made += 1
        self.opts = opts
        self.gx = self.gy = 4096
        self.T = 435
        self.dx = float(np.float32(343.21) / np.float32(res) / np.float32(3.5))
        self.listener, self.done, self.q = None, True, None
        self.info = types.SimpleNamespace(stepsPerLaunch=12, tileRows=36, tileCols=40, deviceBytes=1 << 20)

    def load_scene(self, path):
        assert os.path.exists(path)

    def _record(self):
        L = [tuple(l) for l in GOLD["listeners"][:, [0, 2]].tolist()]
        return GOLD["emitter_out"][L.index((float(self.listener[0]), float(self.listener[2])))]

    def set_output_queries(self, emitters):
        assert self.done, "queries changed while a run is in flight"
        self.q = list(emitters)

    def run_async(self, listener):
        assert self.done, "run started before the previous one was collected"
        self.listener, self.done = listener, False

    def run(self, listener):
        self.run_async(listener)
        self.sync()

    def sync(self):
        self.done = True

    def queried_outputs(self):
        assert self.done and len(self.q) == 2
        return np.array(self._record(), np.float32)

    def timings(self):
        return types.SimpleNamespace(fdtdMs=5.0, analysisMs=0.2, airKernelMs=0.0, generalKernelMs=0.0, stepLoopMs=4.8,
                                     stepLaunches=37)

    def close(self):
        pass


class CpuHooks:
    backend_default = "gloo"
    device = None  # torch tensors on the CPU

    def __init__(self, comm_mode):
        self.comm_mode = comm_mode
        self.barriers = 0

    def init_process_group(self, dist, backend, rank, world):
        assert backend == "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)

    def make_solver(self, size, res, **opts):
        return FakeBenchSolver(size, res, **opts)

    def batch_solver_options(self, grid):
        return {}

    def run_batch(self, solvers, listeners, wait):
        for s, l in zip(solvers, listeners):
            s.run_async(l)

    def make_comm(self, dist):
        # "fail": the C++ RCCL communicator cannot be made on ANY rank (no GPU here): every rank must fall back together.
        # "fail-rank1": it fails on rank 1 only; rank 0's communicator must be closed and both use the torch gather.
        if self.comm_mode == "gloo-comm":  # the native gather's code path with the real world size: a stand-in communicator
            from _dist_worker import GlooComm

            class _G(GlooComm):
                def close(self_inner):
                    pass
            return _G(dist)
        if self.comm_mode == "fail" or dist.get_rank() == 1:
            raise RuntimeError("no RCCL on a CPU host")

        class _Comm:
            closed = False

            def close(self_inner):
                _Comm.closed = True
        self.comm = _Comm()
        return self.comm

    def device_sync(self):
        pass

    def barrier(self, dist, backend):
        self.barriers += 1
        dist.barrier()


def main():
    comm_mode, out_path = sys.argv[1], sys.argv[2]
    hooks = CpuHooks(comm_mode)
    world = os.environ.get("WORLD_SIZE", "2")
    shape = os.environ.get("PV_BENCH_WORKER_SHAPE", "")
    argv = ["--gpus", world, "--steps", "3", "--warmup", "2", "--cpu-baseline-cells", "97"]
    if shape == "config4":  # one run per GPU and step
        argv += ["--inflight", "1"]
    elif shape == "config5":  # two in flight, 4 steps x 2 x 8 ranks = 64 runs
        argv[3] = "4"
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main(argv, hooks=hooks)
    with open(out_path, "w") as f:
        f.write(buf.getvalue())
    assert hooks.barriers >= 3, hooks.barriers
    if comm_mode == "fail-rank1" and int(os.environ["RANK"]) == 0:
        assert hooks.comm.closed, "rank 0 kept a communicator the other rank does not have"


if __name__ == "__main__":
    main()
