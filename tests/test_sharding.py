"""N>1 path on CPU: world_size-2 gloo test of the read sharding and the edge gather (siga_amd/sharding.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from siga_amd.sharding import gather_edges, gather_edges_async, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _edges_of(rank):
    rng = np.random.default_rng(100 + rank)
    k = 5 + 7 * rank
    return torch.from_numpy(rng.integers(0, 1000, size=(k, 4)).astype(np.int32))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allv, counts = gather_edges(_edges_of(rank))
        if rank == 0:
            want = torch.cat([_edges_of(r) for r in range(world)])
            out.put((bool(torch.equal(allv, want)), counts))
        else:
            assert allv is None
        pend = [gather_edges_async(_edges_of(rank) + 7 * i) for i in range(3)]  # several gathers in flight
        for i, pg in enumerate(pend):
            got, cs = pg.wait()
            if rank == 0:
                assert torch.equal(got, torch.cat([_edges_of(r) + 7 * i for r in range(world)])) and cs == [5, 12]
            else:
                assert got is None
        empty, c2 = gather_edges(torch.zeros((0, 4), dtype=torch.int32))  # ragged: nobody has edges
        if rank == 0:
            out.put((empty.shape[0] == 0, c2))
    finally:
        dist.destroy_process_group()


def test_gather_edges_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, counts = q.get(timeout=10)
    assert ok and counts == [5, 12]
    ok2, c2 = q.get(timeout=10)
    assert ok2 and c2 == [0, 0]


def _locus_reads(G, L, N, seed):
    """reads of both strands at known places of a random genome -> (ASCII array [N, L], start of each on the forward strand)"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    pos = rng.integers(0, G - L + 1, size=N)
    rc = rng.random(N) < 0.5
    codes = np.lib.stride_tricks.sliding_window_view(genome, L)[pos]
    codes[rc] = (3 - codes[rc])[:, ::-1]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes], pos


def test_locality_keys_put_neighbours_on_one_rank():
    """Key-range sharding (bench.py --shard key): the key comes from a read's sequence alone, a read and its reverse
    complement get the same minimizer, reads next to each other in key order lie next to each other on the genome, and a
    rank's slice of the order covers a fraction of the genome deeply where a slice of the file covers all of it thinly.
    Every read is on exactly one rank."""
    from siga_amd.sharding import key_order, locality_keys
    G, L, N, W = 100000, 100, 30000, 8
    reads, pos = _locus_reads(G, L, N, 11)
    keys = locality_keys(reads, chunk=7000)  # several chunks
    assert keys.shape == (N,) and keys.dtype == np.int64 and (keys >= 0).all()
    comp = np.zeros(256, dtype=np.uint8)
    comp[np.frombuffer(b"ACGT", dtype=np.uint8)] = np.frombuffer(b"TGCA", dtype=np.uint8)
    krc = locality_keys(comp[reads[:, ::-1]])
    assert np.array_equal(keys >> 16, krc >> 16)
    order = key_order(keys)
    assert np.array_equal(np.sort(order), np.arange(N))
    near = np.abs(np.diff(pos[order])) < L
    assert near.mean() > 0.85
    def covered(ids):
        c = np.zeros(G + 1, dtype=np.int64)
        np.add.at(c, pos[ids], 1)
        np.add.at(c, pos[ids] + L, -1)
        d = np.cumsum(c)[:G]
        return (d > 0).mean(), d[d > 0].mean()
    seen = np.zeros(N, dtype=np.int64)
    for r in range(W):
        lo, hi = shard_range(N, r, W)
        seen[order[lo:hi]] += 1
        frac_key, depth_key = covered(order[lo:hi])
        frac_file, depth_file = covered(np.arange(lo, hi))
        assert frac_key < 0.45 and frac_file > 0.9          # (1/8 of the genome would be perfect)
        assert depth_key > 2.5 * depth_file
    assert (seen == 1).all()
    # shorter than a minimizer: no key, still a valid order
    assert (locality_keys(reads[:5, :10]) == 0).all()


def test_fast_reads_subset_by_ids():
    from tests.golden.make_reads import fast_reads
    whole, _ = fast_reads(20000, 100, 3000, 3)
    ids = np.random.default_rng(1).permutation(3000)[:500].astype(np.uint32)
    part, _ = fast_reads(20000, 100, 3000, 3, subset=ids)
    assert np.array_equal(part, whole[ids])
    part, _ = fast_reads(20000, 100, 3000, 3, subset=(100, 200))
    assert np.array_equal(part, whole[100:200])


def _key_by_definition(seq):
    """csrc/sigax_keys.hip's definition, base by base"""
    k = 16
    L = len(seq)
    if L < k:
        return 0
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}
    c = [code.get(ch, 0) for ch in seq]
    best = None
    for i in range(L - k + 1):
        f = 0
        for j in range(k):
            f = (f << 2) | c[i + j]
        g = 0
        for j in range(k):
            g |= (3 - c[i + j]) << (2 * j)
        fw = f <= g
        canon = f if fw else g
        h = ((canon * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF) >> 31 & 0xFFFFFFFF
        off = i if fw else (L - k) - i
        key = (h << 16) | ((L - k) - off)
        best = key if best is None or key < best else best
    return best


def test_locality_keys_follow_their_definition():
    """the torch restatement (what the GPU kernel is held against in tests/test_gpu_keys.py) against the definition, base by
    base: mixed case, non-ACGT bytes, palindromic 16-mers (f == g), reads of exactly 16 bases"""
    from siga_amd.sharding import locality_keys
    rng = np.random.default_rng(4)
    for L in (16, 17, 40, 151):
        reads = np.frombuffer(b"ACGTacgtNn.", dtype=np.uint8)[rng.choice([0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10], size=(60, L))]
        reads[0, :16] = np.frombuffer(b"ACGTACGTACGTACGT", dtype=np.uint8)  # its own reverse complement
        got = locality_keys(reads, chunk=25)
        want = [_key_by_definition(bytes(r).decode("latin1")) for r in reads]
        assert got.tolist() == want, L
