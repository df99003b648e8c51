"""Multi-GPU layer: independent simulation runs sharded over the GPUs of one node, one process per GPU.

A "run" is one listener position on one scene (one pass of the reference's background loop,
ProjectPlaneverb/src/Context/PvContext.cpp:74-93).  Runs share nothing, so the data path has no collective: rank r
simulates runs r, r + W, r + 2W, ...  The only exchange is the final gather of the per-emitter PlaneverbOutput
records (8 floats each, PvTypes.h:63-71): one all-gather over RCCL/xGMI on GPU ranks (backend "nccl"), or over gloo
in the CPU tests.  Latency-bound: n_emitters * 32 B per rank.
"""
import numpy as np


def shard_runs(n_runs, world_size, rank):
    """indices of the runs rank `rank` simulates (round-robin: run k -> rank k mod W)"""
    return list(range(rank, n_runs, world_size))


def gather_outputs(local, n_runs, dist=None, device=None):
    """local: {run index: float32 array [n_emitters, 8]} computed by this rank.
    Returns float32 [n_runs, n_emitters, 8] on every rank (one all-gather).  Every run must have the same number of
    emitters.  `dist` is torch.distributed (already initialised) or None for a single process."""
    n_em = next(iter(local.values())).shape[0] if local else 0
    if dist is None or not dist.is_initialized():
        out = np.zeros((n_runs, n_em, 8), np.float32)
        for k, v in local.items():
            out[k] = v
        return out
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_runs + world - 1) // world
    # every rank must agree on n_em even if it owns no run
    t = torch.tensor([n_em], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n_em = int(t.item())
    send = torch.zeros((per_rank, n_em, 8), dtype=torch.float32, device=device)
    for j, k in enumerate(shard_runs(n_runs, world, rank)):
        send[j] = torch.from_numpy(np.ascontiguousarray(local[k], np.float32)).to(send.device)
    recv = torch.empty((world * per_rank, n_em, 8), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(recv, send)  # concatenation along dim 0, rank-major
    recv = recv.cpu().numpy().reshape(world, per_rank, n_em, 8)
    out = np.zeros((n_runs, n_em, 8), np.float32)
    for r in range(world):
        for j, k in enumerate(shard_runs(n_runs, world, r)):
            out[k] = recv[r, j]
    return out


def make_comm(dist, device_index):
    """The C++ side's RCCL communicator (api.Comm = PvAmdComm: ncclCommInitRank inside libplaneverb_amd.so) for the
    ranks of an initialised torch.distributed group: rank 0 creates the 128-byte id, torch.distributed only carries
    it to the other ranks (bootstrap); the data path's one collective then runs in C++ (PvAmdCommAllGather)."""
    from . import api
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [None]
    if rank == 0:
        try:
            box = [api.Comm.unique_id()]
        except Exception as e:  # noqa: BLE001 -- every rank must still leave the broadcast below
            box = [RuntimeError("PvAmdCommUniqueId failed: %s" % e)]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], (bytes, bytearray)):
        raise RuntimeError(str(box[0]))
    return api.Comm(box[0], rank, world, device_index)


def gather_outputs_native(local, n_runs, comm, n_em=None):
    """gather_outputs through the C++ RCCL communicator (one ncclAllGather): same result.
    Every rank enters the collective with the same count, also a rank that owns no run (n_runs < world): pass the
    number of emitters per run as `n_em`, or leave it None and the ranks agree on it first (one extra 4-byte gather,
    the counterpart of gather_outputs' all_reduce(MAX))."""
    world, rank = comm.world, comm.rank
    if n_em is None:
        mine = next(iter(local.values())).shape[0] if local else 0
        n_em = int(comm.all_gather(np.array([mine], np.float32)).max())
    per_rank = (n_runs + world - 1) // world
    if per_rank * n_em == 0:
        return np.zeros((n_runs, n_em, 8), np.float32)  # nothing to exchange, on every rank alike
    send = np.zeros((per_rank, n_em, 8), np.float32)
    for j, k in enumerate(shard_runs(n_runs, world, rank)):
        send[j] = local[k]
    recv = comm.all_gather(send).reshape(world, per_rank, n_em, 8)
    out = np.zeros((n_runs, n_em, 8), np.float32)
    for r in range(world):
        for j, k in enumerate(shard_runs(n_runs, world, r)):
            out[k] = recv[r, j]
    return out


def run_sharded_native(solvers, listeners, emitters, comm=None):
    """The whole sharded job in C++ (PvAmdRunSharded): this rank's runs round-robin over `solvers` (two per GPU keep two
    runs in flight), one RCCL all-gather of the records.  listeners [n, 3], emitters [n, E, 3] -> [n, E, 8]."""
    from . import api
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    return api.run_sharded(solvers, listeners, emitters, rank, world, comm)


def default_inflight(gx, gy):
    """runs a GPU should keep in flight for a gx x gy grid: 2 fill the launch gaps of large grids (a third adds
    1 % at 4096^2); launch-latency-bound grids take 4 (+58 % at 1024^2, +73 % at 512^2)"""
    return 2 if gx * gy > 1536 * 1536 else 4


def default_batch(gx, gy):
    """runs per batched launch for a gx x gy grid: 8 for launch-bound grids (+60 % at 512^2, +25 % at 1024^2 with
    api.batch_solver_options), 1 (= plain runs in flight) from 2048^2 up"""
    return 8 if gx * gy <= 1536 * 1536 else 1


def run_sharded(make_solver, listeners, emitters_for, dist=None, device=None, inflight=2, batch=1, run_batch=None):
    """Simulate `listeners` (list of (x, y, z)) sharded over the ranks and gather all per-emitter outputs.
    make_solver() -> planeverb_amd.api.Solver bound to this rank's GPU; emitters_for(k) -> list of emitter positions
    of run k.  Returns [n_runs, n_emitters, 8].

    inflight: groups of runs a rank works on concurrently (one HIP stream per group).  A K-step launch fills the
    chip and then drains; a second group's launches fill those gaps (+22 % cell-updates/s at 4096^2, +40 % at
    2048^2, measured on MI355X), a third brings nothing more.
    batch: runs per group, advanced together by ONE launch per K steps (PvAmdRunBatch, <= 8): the lever for
    launch-bound grids, where a run is a chain of short dependent launches.  (run_batch: test hook, default
    planeverb_amd.api.run_batch.)"""
    if run_batch is None and batch > 1:
        from . import api
        run_batch = api.run_batch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = shard_runs(len(listeners), world, rank)
    local = {}
    if mine:
        batch = max(1, min(int(batch), 8))
        chunks = [mine[i:i + batch] for i in range(0, len(mine), batch)]
        n_groups = max(1, min(inflight, len(chunks)))
        groups = [[make_solver() for _ in range(min(batch, len(mine)))] for _ in range(n_groups)]
        pending = [None] * n_groups

        def collect(g):
            for sv, k in zip(groups[g], pending[g]):
                sv.sync()
                local[k] = sv.queried_outputs()  # gathered behind the run's analysis: no further GPU work
            pending[g] = None

        for j, chunk in enumerate(chunks):
            g = j % n_groups
            if pending[g] is not None:
                collect(g)  # the other groups' runs keep the GPU busy meanwhile
            for sv, k in zip(groups[g], chunk):
                sv.set_output_queries(emitters_for(k))
            if batch == 1:
                groups[g][0].run_async(listeners[chunk[0]])
            else:
                run_batch(groups[g][:len(chunk)], [listeners[k] for k in chunk], wait=False)
            pending[g] = chunk
        for g in range(n_groups):
            if pending[g] is not None:
                collect(g)
        for grp in groups:
            for s in grp:
                s.close()
    return gather_outputs(local, len(listeners), dist, device)
