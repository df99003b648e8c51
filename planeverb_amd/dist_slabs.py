"""ONE grid decomposed into row slabs with the slabs in DIFFERENT PROCESSES (the same-process form is
PvAmdCreateSlabs / csrc/pv_slabs.cpp).  One rank per GPU owns one slab (api.SlabRank); this module is the exchange
schedule on top of a transport:

  per K-step launch   every rank advances its rows, then swaps K boundary rows of pr, vx, vy with its neighbours
                      (rank r sends its first rows to r - 1 and its last rows to r + 1: 3 x K x pitch floats each);
  per run             rank r's last row's pressure history goes to rank r + 1 (T x histPitch floats), every rank analyses
                      its own cells, the window blocks of the per-rank maps go to rank 0, which holds the whole-grid maps
                      (api.SlabRoot) and runs the listener-direction descent.

Transports: `TorchTransport` (torch.distributed point-to-point: gloo with host buffers in the CPU tests; with a `device`
-- backend "nccl" = RCCL over xGMI on GPU ranks -- the halos and boundary histories travel between DEVICE tensors: the slab
exports into the send tensor and imports from the receive tensor by address, nothing is staged through the host) and
`LocalTransport` (all ranks inside one process, driven in lock-step: how the one-GPU box of the test pool runs it, with host
or device buffers).  All give the same bits as one solver on the whole grid."""
import numpy as np


class TorchTransport:
    """point-to-point over an initialised torch.distributed group; tensors live on `device` (None = CPU / gloo)"""

    def __init__(self, dist, device=None):
        import torch
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, sends, recv_sizes):
        """sends {peer: float32 array}, recv_sizes {peer: n floats} -> {peer: float32 array}; all transfers of one
        call are posted together (batch_isend_irecv), so neighbours that send to each other cannot deadlock"""
        torch, dist = self.torch, self.dist
        ops, bufs = [], {}
        for peer, a in sorted(sends.items()):
            t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
            if self.device is not None:
                t = t.to(self.device)
            ops.append(dist.P2POp(dist.isend, t, peer))
        for peer, n in sorted(recv_sizes.items()):
            bufs[peer] = torch.empty(int(n), dtype=torch.float32, device=self.device if self.device is not None else "cpu")
            ops.append(dist.P2POp(dist.irecv, bufs[peer], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {p: b.cpu().numpy() for p, b in bufs.items()}

    def exchange_device(self, fill_sends, recv_sizes):
        """the same between device tensors.  fill_sends {peer: (n floats, fill(ptr))}: `fill` writes the peer's payload
        to the device address it is given (SlabRank.export_*_to); returns {peer: device tensor} -- read them by
        .data_ptr() (SlabRank.import_*_from).  The solver's exports synchronise their own stream; the transfers run on
        torch's streams, which are drained before the tensors are handed out."""
        torch, dist = self.torch, self.dist
        ops, keep, bufs = [], [], {}
        for peer, (n, fill) in sorted(fill_sends.items()):
            t = torch.empty(int(n), dtype=torch.float32, device=self.device)
            fill(t.data_ptr())
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
        for peer, n in sorted(recv_sizes.items()):
            bufs[peer] = torch.empty(int(n), dtype=torch.float32, device=self.device)
            ops.append(dist.P2POp(dist.irecv, bufs[peer], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            torch.cuda.synchronize(self.device)
        return bufs

    def gather_to_root(self, obj):
        """rank 0 gets [obj of rank 0, ..., obj of rank W-1], the others None"""
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out


def run_rank(slab, root, listener, transport):
    """One run on this rank's slab (api.SlabRank or anything with its interface); `root` = api.SlabRoot on rank 0, else
    None.  Collective: every rank of the transport calls it with the same listener."""
    r, W = transport.rank, transport.world
    up, down = (r - 1 if r > 0 else None), (r + 1 if r + 1 < W else None)
    slab.begin(listener)
    if root is not None:
        root.begin(listener)
    on_device = getattr(transport, "device", None) is not None and hasattr(slab, "export_halo_to")
    for li in range(slab.num_launches):
        slab.launch(li)
        recvs = {}
        if up is not None:
            recvs[up] = slab.halo_floats
        if down is not None:
            recvs[down] = slab.halo_floats
        if on_device:  # device tensors end to end
            fills = {}
            if up is not None:
                fills[up] = (slab.halo_floats, lambda p: slab.export_halo_to(0, p))
            if down is not None:
                fills[down] = (slab.halo_floats, lambda p: slab.export_halo_to(1, p))
            got = transport.exchange_device(fills, recvs)
            if up is not None:
                slab.import_halo_from(0, got[up].data_ptr())
            if down is not None:
                slab.import_halo_from(1, got[down].data_ptr())
            continue
        sends = {}
        if up is not None:
            sends[up] = slab.export_halo(0)     # my first K rows: the guard rows below the slab above
        if down is not None:
            sends[down] = slab.export_halo(1)   # my last K rows: the guard rows above the slab below
        got = transport.exchange(sends, recvs)
        if up is not None:
            slab.import_halo(0, got[up])
        if down is not None:
            slab.import_halo(1, got[down])
    # the vx recurrence of my first row needs the pressure history of the row above it
    recvs = {up: slab.history_floats} if up is not None else {}
    if on_device:
        fills = {down: (slab.history_floats, slab.export_edge_history_to)} if down is not None else {}
        got = transport.exchange_device(fills, recvs)
        if up is not None:
            slab.import_above_history_from(got[up].data_ptr())
    else:
        sends = {down: slab.export_edge_history()} if down is not None else {}
        got = transport.exchange(sends, recvs)
        if up is not None:
            slab.import_above_history(got[up])
    slab.analyze()
    blocks = transport.gather_to_root(slab.window_block())
    if root is not None:
        for info, data in blocks:
            root.import_block(info, data)
        root.finish()


def _run_local_device(slabs, root, listener, device):
    import torch
    W = len(slabs)
    buf = lambda n: torch.empty(int(n), dtype=torch.float32, device=device)
    for s in slabs:
        s.begin(listener)
    root.begin(listener)
    for li in range(slabs[0].num_launches):
        for s in slabs:
            s.launch(li)
        first = [buf(s.halo_floats) if r > 0 else None for r, s in enumerate(slabs)]
        last = [buf(s.halo_floats) if r + 1 < W else None for r, s in enumerate(slabs)]
        for r, s in enumerate(slabs):
            if r > 0:
                s.export_halo_to(0, first[r].data_ptr())
            if r + 1 < W:
                s.export_halo_to(1, last[r].data_ptr())
        for r, s in enumerate(slabs):
            if r > 0:
                s.import_halo_from(0, last[r - 1].data_ptr())
            if r + 1 < W:
                s.import_halo_from(1, first[r + 1].data_ptr())
    edge = [buf(s.history_floats) if r + 1 < W else None for r, s in enumerate(slabs)]
    for r, s in enumerate(slabs):
        if r + 1 < W:
            s.export_edge_history_to(edge[r].data_ptr())
    for r, s in enumerate(slabs):
        if r > 0:
            s.import_above_history_from(edge[r - 1].data_ptr())
    for s in slabs:
        s.analyze()
    for s in slabs:
        root.import_block(*s.window_block())
    root.finish()


class LocalTransport:
    """W ranks inside one process (one GPU holding all slabs): run_local drives them in lock-step with the same
    per-rank calls and the same buffers run_rank moves"""

    def __init__(self, world):
        self.world = world


def run_local(slabs, root, listener, device=None):
    """the schedule of run_rank for all ranks at once (slabs[r] = rank r's slab); returns nothing: results are in `root`.
    device = a torch device: the halos and boundary histories move between DEVICE tensors by address, the way
    TorchTransport.exchange_device moves them between ranks (tested on the one-GPU box this way)"""
    W = len(slabs)
    if device is not None:
        return _run_local_device(slabs, root, listener, device)
    for s in slabs:
        s.begin(listener)
    root.begin(listener)
    for li in range(slabs[0].num_launches):
        for s in slabs:
            s.launch(li)
        first = [s.export_halo(0) if r > 0 else None for r, s in enumerate(slabs)]
        last = [s.export_halo(1) if r + 1 < W else None for r, s in enumerate(slabs)]
        for r, s in enumerate(slabs):
            if r > 0:
                s.import_halo(0, last[r - 1])
            if r + 1 < W:
                s.import_halo(1, first[r + 1])
    edge = [s.export_edge_history() if r + 1 < W else None for r, s in enumerate(slabs)]
    for r, s in enumerate(slabs):
        if r > 0:
            s.import_above_history(edge[r - 1])
    for s in slabs:
        s.analyze()
    for s in slabs:
        root.import_block(*s.window_block())
    root.finish()
