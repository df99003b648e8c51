"""Worker process of tests/test_dist_cpu.py: one rank of a gloo group (started with plain subprocess)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_result(k, n_em=3):
    return np.array([[k, e, k * 10 + e, 0.5, -1, 1, 0, k + e / 8] for e in range(n_em)], np.float32)


class FakeSolver:
    """stands in for planeverb_amd.api.Solver in run_sharded: run k's listener is (k, 0, 0); an output can only be
    fetched between sync() and the next run_async(), like the real one's result map"""
    made = 0

    def __init__(self):
        FakeSolver.made += 1
        self.k, self.done = None, False

    def run_async(self, listener):
        assert self.k is None or self.done, "run started before the previous one was collected"
        self.k, self.done = int(listener[0]), False

    def sync(self):
        self.done = True

    def set_output_queries(self, emitters):
        assert self.k is None or self.done, "queries changed while a run is in flight"
        self.q = [int(e) for e in emitters]

    def queried_outputs(self):
        assert self.done
        return fake_result(self.k)[self.q]

    def close(self):
        pass


class GlooComm:
    """stands in for planeverb_amd.api.Comm (PvAmdComm: ncclAllGather inside libplaneverb_amd.so): the same interface over
    torch.distributed, so that gather_outputs_native's packing and ordering run with the real world size on CPU"""

    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self.calls = 0

    def all_gather(self, mine):
        import torch
        mine = np.ascontiguousarray(mine, np.float32).ravel()
        out = torch.empty(self.world * mine.size, dtype=torch.float32)
        self.dist.all_gather_into_tensor(out, torch.from_numpy(mine))
        self.calls += 1
        return out.numpy().reshape(self.world, -1)


def main():
    rank, world, port, n_runs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    import torch.distributed as dist
    from planeverb_amd import dist as pvd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = pvd.shard_runs(n_runs, world, rank)
    assert mine == [k for k in range(n_runs) if k % world == rank]  # run k -> rank k mod W
    local = {k: fake_result(k) for k in mine}
    res = pvd.gather_outputs(local, n_runs, dist)
    # the native gather's packing (one all-gather of per_rank x n_em x 8 floats, also from ranks that own no run) against it
    comm = GlooComm(dist)
    res_native = pvd.gather_outputs_native(local, n_runs, comm, n_em=3)
    assert np.array_equal(res, res_native), "gather_outputs_native differs from gather_outputs"
    assert comm.calls == 1, "exactly one collective on the data path"
    assert np.array_equal(pvd.gather_outputs_native(local, n_runs, comm), res)  # (n_em agreed on by one more 4-byte gather)
    # the same through run_sharded with two runs in flight per rank
    res2 = pvd.run_sharded(FakeSolver, [(k, 0, 0) for k in range(n_runs)], lambda k: [0, 1, 2], dist, inflight=2)
    assert np.array_equal(res, res2), "run_sharded differs from gather_outputs"
    assert FakeSolver.made == min(2, len(mine))
    # batched: groups of 3 runs started together by one call (PvAmdRunBatch on the GPU), two groups in flight
    calls = []

    def fake_batch(solvers, listeners, wait=True):
        assert len(solvers) == len(listeners) <= 3 and not wait
        calls.append(len(solvers))
        for sv, L in zip(solvers, listeners):
            sv.run_async(L)

    res3 = pvd.run_sharded(FakeSolver, [(k, 0, 0) for k in range(n_runs)], lambda k: [0, 1, 2], dist, inflight=2,
                           batch=3, run_batch=fake_batch)
    assert np.array_equal(res, res3), "batched run_sharded differs"
    assert sum(calls) == len(mine) and all(c == 3 for c in calls[:-1])
    np.savez(out, mine=np.array(mine, np.int64), out=res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
