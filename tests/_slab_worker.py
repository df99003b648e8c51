"""Worker process of tests/test_dist_cpu.py::test_slab_exchange_schedule_gloo: one rank of a gloo group running
planeverb_amd.dist_slabs.run_rank on a toy slab."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch.distributed as dist
    from planeverb_amd import dist_slabs
    from _slab_toy import ToyRoot, ToySlab
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NX, cols, T, K = 60, 17, 23, 4
    tr = dist_slabs.TorchTransport(dist)
    root = ToyRoot(NX, cols) if rank == 0 else None
    finals = {}
    for src in [(NX // world, 5), (NX // world - 1, 9), (3, 2)]:  # the source ON a boundary row, just above it, far away
        slab = ToySlab(NX, cols, T, K, rank, world, src)
        dist_slabs.run_rank(slab, root, (0, 0, 0), tr)
        finals[src] = (slab.u[K:K + slab.n].copy(), root.maps.copy() if root is not None else None)
    np.savez(out, **{"f%d" % i: v[0] for i, v in enumerate(finals.values())},
             **({"m%d" % i: v[1] for i, v in enumerate(finals.values())} if rank == 0 else {}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
