"""Sharding of the block-transform path over the GPUs of one node.

Blocks -- and therefore planes and block rows -- are independent units (no DC prediction, no
inter-block state anywhere in steps 4-6; pipeline/__init__.py:104-106 processes bands one after
another with nothing shared), so the data path needs NO collective: every rank transforms its
own contiguous range.  The only exchange is the final gather of the int16 coefficient stream to
one rank, done with ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo"
in the CPU tests).  torch is plumbing here (process group + device tensors); the kernels are
reached through libjpegx as everywhere else.
"""


def shard_range(n_units, world, rank):
    """Contiguous [lo, hi) of ``n_units`` for ``rank``: sizes differ by at most one, low ranks first."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank %r / world %r" % (rank, world))
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_planes(n_planes, world, rank):
    """Plane ids owned by ``rank`` when a batch of independent planes is split over the ranks."""
    return shard_range(n_planes, world, rank)


def shard_block_rows(height, world, rank):
    """Row range [y0, y1) (multiples of 8) of ONE tall plane owned by ``rank``: contiguous block rows,
    so that both the rank's input rows and its slice of the zigzag stream are single spans."""
    if height % 8:
        raise ValueError("plane height must be a multiple of 8")
    lo, hi = shard_range(height // 8, world, rank)
    return lo * 8, hi * 8


def gather_stream(local, dst=0, group=None):
    """Gather every rank's coefficient stream (a 1-D/N-D tensor, sizes may differ) on ``dst``.

    Returns the list of per-rank tensors on ``dst`` (in rank order, so concatenating them yields the
    stream of the un-sharded job) and None elsewhere.  One size all-gather + one data gather.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dtype = local.dtype
    # ship raw bytes: neither RCCL/NCCL nor gloo has an int16 datatype
    flat = local.contiguous().reshape(-1).view(torch.uint8)
    sizes = torch.zeros(world, dtype=torch.int64, device=flat.device)
    sizes[rank] = flat.numel()
    dist.all_reduce(sizes, group=group)
    sizes = [int(s) for s in sizes.tolist()]
    cap = max(sizes)
    if flat.numel() < cap:                                  # gather wants equal shapes: pad the short ones
        flat = torch.cat([flat, flat.new_zeros(cap - flat.numel())])
    bufs = [torch.empty(cap, dtype=flat.dtype, device=flat.device) for _ in range(world)] if rank == dst else None
    dist.gather(flat, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[:n].view(dtype) for b, n in zip(bufs, sizes)]


class NativeComm:
    """RCCL communicator owned by libjpegx (jpegx_comm_*): the PyTorch-free form of the gather.

    ``exchange_id`` is any callable that takes rank 0's 128-byte id (bytes on rank 0, None elsewhere)
    and returns the id on every rank -- e.g. an MPI bcast, a shared file, or
    ``NativeComm.exchange_via_torch`` when a torch.distributed group (gloo is enough) exists.
    """

    def __init__(self, nranks, rank, exchange_id):
        import ctypes
        import jpegx
        self._jpegx = jpegx
        L = jpegx.lib()
        ident = None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            jpegx.check(L.jpegx_comm_unique_id(buf), "jpegx_comm_unique_id")
            ident = buf.raw
        ident = exchange_id(ident)
        handle = ctypes.c_void_p()
        jpegx.check(L.jpegx_comm_create(ctypes.byref(handle), nranks, rank, ident), "jpegx_comm_create")
        self.handle, self.nranks, self.rank = handle.value, nranks, rank

    @staticmethod
    def exchange_via_torch(ident):
        import torch.distributed as dist
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather_bytes(self, send_ptr, send_bytes, recv_ptr=None, recv_bytes=None, root=0, stream=None):
        """Every rank sends ``send_bytes`` from ``send_ptr``; the root lays rank r's bytes out at the
        running offset of ``recv_bytes`` inside ``recv_ptr``.  Enqueues on ``stream``."""
        import ctypes
        n = self.nranks
        sizes = (ctypes.c_size_t * n)(*([0] * n))
        offs = (ctypes.c_size_t * n)(*([0] * n))
        if self.rank == root:
            run = 0
            for r in range(n):
                sizes[r], offs[r] = int(recv_bytes[r]), run
                run += int(recv_bytes[r])
        self._jpegx.check(self._jpegx.lib().jpegx_comm_gather_bytes(self.handle, send_ptr, int(send_bytes), recv_ptr, sizes, offs,
                                                                      root, stream), "jpegx_comm_gather_bytes")

    def close(self):
        if self.handle:
            self._jpegx.lib().jpegx_comm_destroy(self.handle)
            self.handle = None
