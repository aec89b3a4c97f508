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
