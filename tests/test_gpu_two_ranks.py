"""Two ranks of the REAL path (torch.distributed, gloo; both ranks on the one GPU of the test box): each rank runs its
contiguous shard of the reads through the HIP kernels with read_base = shard start (the bench rehearsal below: its slice of
the locality-key order, every read under its own id), the 16-byte edge records are gathered
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


def test_bench_self_launch_two_ranks_gloo_counts_the_one_rank_edges(tmp_path):
    """`python bench.py --gpus 2` as the driver starts it for N > 1 rehearsed on this one-GPU box: bench.py launches its two
    ranks itself (fresh torchrun children, before anything touches the GPU), `--backend gloo` gathers the edge records through
    the host, and rank 0 prints ONE JSON line with n_gpus 2 whose edge count is the one-rank run's on the same reads."""
    import json
    import subprocess
    import sys
    from tests.fixtures import ROOT
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--seed", "2", "--read-len", "100", "--min-overlap", "40",
              "--workdir", str(tmp_path / "job")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)

    def run(extra):
        r = subprocess.run([sys.executable, bench] + extra + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    two = run(["--gpus", "2", "--backend", "gloo", "--reads-per-gpu", "20000", "--genome-per-gpu", "100000"])
    one = run(["--gpus", "1", "--reads-per-gpu", "40000", "--genome-per-gpu", "200000"])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["edges"] == one["config"]["edges"] > 20000
    # the ranks' reads were slices of the locality-key order (the default for N > 1; each read under its own id); slices of
    # the file give the same records
    assert two["config"]["sharding"]["by"].startswith("locality key") and one["config"]["sharding"] is None
    assert os.path.exists(str(tmp_path / "job" / "reads.keyorder.npy"))
    by_file = run(["--gpus", "2", "--backend", "gloo", "--shard", "contiguous", "--reads-per-gpu", "20000", "--genome-per-gpu", "100000"])
    assert by_file["config"]["sharding"] == {"by": "file position"} and by_file["config"]["edges"] == one["config"]["edges"]
    assert two["config"]["reads_per_gpu"] == 20000 and two["value"] > 0
