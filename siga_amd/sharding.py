"""Read sharding and the one exchange step of the multi-GPU path.

Every read's result depends only on the immutable index, so the index is replicated on every GPU and the reads
are cut into `world` contiguous ranges (contiguous keeps VT order and hits order trivial).  The only exchange is
a variable-length gather of fixed-size edge records to rank 0: an all_gather of the per-rank counts, then one
padded gather of the records (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_reads, rank, world):
    """Contiguous, balanced [lo, hi) of reads for `rank`; the first n_reads % world ranks get one more."""
    q, r = divmod(n_reads, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class PendingGather:
    """An edge gather in flight: `wait()` returns what `gather_edges` returns."""

    def __init__(self, work, bufs, counts, is_dst, local=None):
        self._work, self._bufs, self.counts, self._is_dst, self._local = work, bufs, counts, is_dst, local

    def wait(self):
        if self._work is not None:
            self._work.wait()
        if self._local is not None:
            return self._local, self.counts
        if not self._is_dst:
            return None, self.counts
        return torch.cat([self._bufs[r][: self.counts[r]] for r in range(len(self._bufs))], dim=0), self.counts


def gather_edges_async(local_edges, group=None, dst=0):
    """Like gather_edges, but the record gather itself is asynchronous so that it overlaps the next batch's kernels
    (the caller must not touch `local_edges` until wait()).  The tiny count exchange is synchronous."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return PendingGather(None, None, [int(local_edges.shape[0])], True, local=local_edges)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_edges.device
    cnt = torch.tensor([local_edges.shape[0]], dtype=torch.int64, device=dev)
    allc = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, cnt, group=group)
    counts = [int(c) for c in allc.tolist()]
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, 4), dtype=torch.int32, device=dev)
    padded[: local_edges.shape[0]] = local_edges
    if rank == dst:
        bufs = [torch.empty((mx, 4), dtype=torch.int32, device=dev) for _ in range(world)]
        work = dist.gather(padded, gather_list=bufs, dst=dst, group=group, async_op=True)
        return PendingGather(work, bufs, counts, True)
    work = dist.gather(padded, gather_list=None, dst=dst, group=group, async_op=True)
    pg = PendingGather(work, None, counts, False)
    pg._keep = padded
    return pg


def gather_edges(local_edges, group=None, dst=0):
    """local_edges: int32 tensor [k, 4] (query, target, length, af) on this rank's device.

    Returns on `dst` the concatenation over ranks in rank order (= read order, so the ED order of a single-GPU
    run is preserved) and the per-rank counts; on other ranks (None, counts)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_edges, [int(local_edges.shape[0])]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_edges.device
    cnt = torch.tensor([local_edges.shape[0]], dtype=torch.int64, device=dev)
    allc = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, cnt, group=group)
    counts = [int(c) for c in allc.tolist()]  # one device->host sync for all ranks' counts
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, 4), dtype=torch.int32, device=dev)
    padded[: local_edges.shape[0]] = local_edges
    if rank == dst:
        bufs = [torch.empty((mx, 4), dtype=torch.int32, device=dev) for _ in range(world)]
        dist.gather(padded, gather_list=bufs, dst=dst, group=group)
        return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0), counts
    dist.gather(padded, gather_list=None, dst=dst, group=group)
    return None, counts
