#!/usr/bin/env python3
"""ONE grid decomposed into row slabs with one slab per PROCESS (planeverb_amd.dist_slabs.run_rank over torch.distributed), on
real slab solvers: every rank of this script owns one api.SlabRank on the box's GPU (PV_SLAB_DEVICES maps ranks to devices; the
test pool has one GPU, so all ranks share device 0) and the halos / boundary histories / result blocks travel through
TorchTransport -- gloo with host buffers here; with backend nccl and one GPU per rank the same schedule moves device tensors
over RCCL.  Rank 0 holds the whole-grid maps (api.SlabRoot) and compares them with one solver on the whole grid, bit for bit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/slab_ranks_two_procs.py [cells=1024]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planeverb_amd.api as pv  # noqa: E402
from planeverb_amd import dist_slabs  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
devs = os.environ.get("PV_SLAB_DEVICES", ",".join(["0"] * world)).split(",")
device = int(devs[int(os.environ.get("LOCAL_RANK", "0"))])
backend = os.environ.get("PV_SLAB_BACKEND", "gloo")
dist.init_process_group(backend, rank=rank, world_size=world)
transport = dist_slabs.TorchTransport(dist, device=torch.device("cuda", device) if backend == "nccl" else None)

dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
size = float((n + 0.5) * dx)
cell = lambda cx, cy: ((cx + 0.5) * float(dx), 0.0, (cy + 0.5) * float(dx))
opts = dict(steps_per_launch=12, tile_rows=36) if n >= 2048 else dict(steps_per_launch=8, tile_rows=24)
rows = opts["tile_rows"]
edge = (-(-(n + 1) // rows) // world) * rows  # first row of slab 1: the listener sits right on the boundary
Ls = [cell(edge, int(0.4 * n)), cell(edge - 1, int(0.62 * n))]
boxes = [[Ls[0][0] + 1.0, Ls[0][2] + 9.0, 40.0, 1.0, 0.85], [Ls[0][0] - 20.0, Ls[0][2] - 4.0, 1.2, 55.0, 0.5]]

efree = pv.compute_efree(size, size, 275, device=device)
slab = pv.SlabRank(size, size, 275, device, rank, world, efree, **opts)
root = pv.SlabRoot(slab, device) if rank == 0 else None
for b in boxes:
    slab.add_geometry(b)
bad = []
ms = []
# (one whole-grid solver for all the runs, like the slab root's maps: cells a run finds no onset for keep the previous run's
# values)
whole = None
if rank == 0:
    whole = pv.Solver(size, size, 275, device=device, **opts)
    for b in boxes:
        whole.add_geometry(b)
for L in Ls:
    dist.barrier()
    t0 = time.perf_counter()
    dist_slabs.run_rank(slab, root, L, transport)
    dist.barrier()
    ms.append((time.perf_counter() - t0) * 1e3)
    if rank == 0:
        a = whole
        a.run(L)
        ra, da = a.results()
        rb, db = root.results()
        if not same_bits(da, db).all():
            bad.append("delay map, listener %r" % (L,))
        for k in range(8):
            if not same_bits(ra[..., k], rb[..., k]).all():
                bad.append("result plane %d, listener %r" % (k, L))
        e = (L[0] + 3.0, 0.0, L[2] + 2.0)
        if not same_bits(a.get_output(e).as_array(), root.get_output(e).as_array()).all():
            bad.append("emitter record, listener %r" % (L,))
        onsets = int((da < 1e30).sum())
if rank == 0:
    print("%d x %d cells in %d slabs, one process each (%s, device(s) %s): %s ms per run incl. the exchange through %s; %d cells "
          "with an onset: %s" % (n, n, world, backend, ",".join(devs), " / ".join("%.1f" % m for m in ms),
                                 "host buffers" if backend != "nccl" else "device tensors", onsets,
                                 "every map and record bit-identical to one solver on the whole grid" if not bad else "; ".join(bad)),
          flush=True)
if root is not None:
    root.close()
    whole.close()
slab.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
