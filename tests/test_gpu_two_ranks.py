"""Two ranks of the REAL path (torch.distributed, gloo; both ranks on the one GPU of the test box): each rank runs its
contiguous shard of the reads through the HIP kernels with read_base = shard start, the 16-byte edge records are gathered
to rank 0, and rank 0's ASQG text built from the gathered records is byte for byte the one-GPU ASQG and the oracle's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.fixtures import fixture

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, m, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import siga_amd
        from siga_amd.overlap import format_asqg, name_ranks, read_sequences
        from siga_amd.sharding import gather_edges, shard_range
        fx = fixture(name)
        reads = read_sequences(fx.fa)
        seqs = [r[2] for r in reads]
        pair = siga_amd.FMIndexPair.load(fx.prefix, device=0)
        pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
        lo, hi = shard_range(len(seqs), rank, world)
        res = siga_amd.OverlapBuilder(pair).overlap(seqs[lo:hi], m, read_base=lo, edges=True)
        e = res["edges"]
        local = torch.from_numpy(np.stack([e["query"], e["target"], e["length"], e["af"]], axis=1).astype(np.int32).reshape(-1, 4))
        allv, counts = gather_edges(local)
        subs = [None] * world
        dist.gather_object(res["substring"].tolist(), subs if rank == 0 else None, dst=0)
        if rank == 0:
            ed = np.zeros(allv.shape[0], dtype=siga_amd.overlap.EDGE_DTYPE)
            a = allv.numpy()
            ed["query"], ed["target"], ed["length"], ed["af"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
            sub = np.array([x for part in subs for x in part], dtype=np.uint8)
            text = format_asqg(reads, {"substring": sub, "edges": ed}, m)
            one, _ = siga_amd.OverlapBuilder(pair, fx.prefix).build(fx.fa, m)
            want, _, _ = fx.oracle_asqg(m)
            out.put((text == one, text == want, counts))
        pair.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,m", [("mid", 45), ("dup", 8)])
def test_two_ranks_gathered_edges_give_the_one_gpu_asqg(name, m):
    fixture(name)  # build the fixture files before the ranks start
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    same_one, same_oracle, counts = q.get(timeout=10)
    assert same_one and same_oracle and len(counts) == 2 and min(counts) > 0
