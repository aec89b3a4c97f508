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


# ------------------------------------------------------------------------------------------------------------------
# Key-range sharding: which reads go to which rank
# ------------------------------------------------------------------------------------------------------------------
LOCALITY_K = 16  # bases of the minimizer


def locality_keys(reads, device=None, chunk=1 << 20):
    """reads: uint8 array [n, L] of ASCII bases -> int64 array [n] of locality keys, computed from the sequences alone.

    key = min over the read's 16-mers of  hash(canonical 16-mer) << 16 | (L - 16 - off): the read's minimizer -- the canonical
    16-mer (the smaller of a 16-mer and its reverse complement) with the smallest hash, so a read and the reverse complement of
    its neighbour on the genome share it -- above, and below it where the read starts before the minimizer on the canonical
    strand (off), turned so that an earlier start sorts first.  Reads that share a minimizer cover the same 2L-wide stretch of
    the genome: sorted by key they are neighbours, and a contiguous slice of that order (one rank's share: `key_order` /
    `shard_range`) covers its part of the genome as deep as the whole read set covers the whole genome -- the backward searches
    of a rank then walk a fraction of the index's rows (a non-ACGT base counts as 'A' here; the key only decides placement,
    never a result).

    On a GPU (`device` a cuda device) the keys come from the library's kernel (sigax_locality_keys, csrc/sigax_keys.hip); on
    the CPU from the torch restatement below, which the tests hold the kernel against."""
    import numpy as np
    n, L = reads.shape
    out = np.empty(n, dtype=np.int64)
    if L < LOCALITY_K:
        out[:] = 0
        return out
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type == "cuda":
        from . import _lib
        lib = _lib.lib()
        d_offs = torch.arange(0, (min(chunk, n) + 1) * L, L, dtype=torch.int64, device=dev)
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            r = torch.from_numpy(np.ascontiguousarray(reads[lo:lo + m]).reshape(-1)).to(dev)
            k = torch.empty(m, dtype=torch.int64, device=dev)
            torch.cuda.current_stream(dev).synchronize()  # (the library launches on the default stream)
            rc = lib.sigax_locality_keys(dev.index or 0, r.data_ptr(), d_offs.data_ptr(), m, k.data_ptr(), None)
            if rc != 0:
                raise RuntimeError("sigax_locality_keys: " + _lib.last_error())
            out[lo:lo + m] = k.cpu().numpy()
        return out
    return _locality_keys_torch(reads, dev, chunk)


def _locality_keys_torch(reads, dev, chunk=1 << 20):
    """the restatement of csrc/sigax_keys.hip in torch ops (any device)"""
    import numpy as np
    n, L = reads.shape
    k = LOCALITY_K
    out = np.empty(n, dtype=np.int64)
    lut = torch.zeros(256, dtype=torch.int64, device=dev)
    for ch, v in ((b"C", 1), (b"G", 2), (b"T", 3), (b"c", 1), (b"g", 2), (b"t", 3)):
        lut[ch[0]] = v
    at = torch.arange(L - k + 1, dtype=torch.int64, device=dev)[None, :]
    for lo in range(0, n, chunk):
        r = torch.from_numpy(np.ascontiguousarray(reads[lo:lo + chunk])).to(dev)
        f = lut[r.long()]          # [m, L] codes, most significant base first in a k-mer
        g = 3 - f                  # complement; the reverse complement reads it the other way round
        w = 1
        while w < k:               # k-mers by doubling: f[i] = f[i] << 2w | f[i + w]; g[i] = g[i] | g[i + w] << 2w
            f = (f[:, :-w] << (2 * w)) | f[:, w:]
            g = g[:, :-w] | (g[:, w:] << (2 * w))
            w *= 2
        fw = f <= g
        c = torch.where(fw, f, g)  # canonical 16-mer (32 bits)
        h = (c * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF   # wraps like uint64 arithmetic, sign bit dropped
        h = (h >> 31) & 0xFFFFFFFF                             # bits 31..62 of the product: 32-bit hash
        # the read starts `at` bases before the 16-mer on its own strand; seen from the canonical strand a read whose 16-mer
        # is the larger one of the pair starts L - k - at before it
        off = torch.where(fw, at, (L - k) - at)
        key = ((h << 16) | ((L - k) - off)).min(dim=1).values  # earlier start on the canonical strand = smaller key
        out[lo:lo + r.shape[0]] = key.cpu().numpy()
    return out


def key_order(keys, device=None):
    """stable order of the reads by locality key -> int64 permutation (read ids in key order); sorted on `device` if given"""
    import numpy as np
    if device is not None and torch.device(device).type == "cuda":
        return torch.sort(torch.from_numpy(np.ascontiguousarray(keys)).to(device), stable=True).indices.cpu().numpy()
    return np.argsort(keys, kind="stable")
