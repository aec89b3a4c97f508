"""Multi-GPU readiness.  The pool's test boxes have ONE GPU, so what needs two is guarded by the device count
and runs the day a larger box appears; what can run on one does: the library's RCCL exchange step with a world of one rank
(communicator, count all-gather, the root's own share), and its error paths.  The guarded tests go through the real thing:
sigax_index_clone device 0 -> 1 (hipMemcpyPeer), `siga overlap --gpus 2` without a device map, sigax_gather_edges between
two processes on two GPUs, and bench.py --gpus 2 over RCCL (fresh torchrun children: nothing that has touched the GPU is
ever re-executed)."""
import ctypes as C
import gzip
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.fixtures import ROOT, fixture

pytestmark = pytest.mark.gpu
N_GPUS = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
two_gpus = pytest.mark.skipif(N_GPUS < 2, reason="needs two GPUs (this box has %d)" % N_GPUS)


def _edges(rank, n):
    from siga_amd.overlap import EDGE_DTYPE
    e = np.zeros(n, dtype=EDGE_DTYPE)
    e["query"] = np.arange(n) + 1000000 * rank
    e["target"] = 7 * np.arange(n) + rank
    e["length"] = 45 + (np.arange(n) % 100)
    e["af"] = (np.arange(n) + rank) % 8
    return e


def test_rccl_exchange_step_with_one_rank():
    """sigax_comm_* / sigax_gather_*: RCCL bound at run time, a communicator of one rank, counts and records of the root's
    own share (a device copy on the stream), twice on the same communicator, empty share included."""
    from siga_amd import _lib
    from siga_amd.overlap import EDGE_DTYPE
    L = _lib.lib()
    idb = (C.c_uint8 * 128)()
    assert L.sigax_comm_unique_id(idb) == 0, _lib.last_error()
    comm = C.c_void_p()
    assert L.sigax_comm_create(0, 0, 1, idb, C.byref(comm)) == 0, _lib.last_error()
    try:
        for n in (12345, 0, 7):
            e = _edges(0, n)
            d_in = torch.from_numpy(e.view(np.int32).reshape(-1, 4).copy()).cuda()
            d_out = torch.zeros((n + 5, 4), dtype=torch.int32, device="cuda")
            cnt = (C.c_uint64 * 1)()
            st = torch.cuda.current_stream().cuda_stream
            assert L.sigax_gather_counts(comm, n, cnt, C.c_void_p(st)) == 0, _lib.last_error()
            assert int(cnt[0]) == n
            assert L.sigax_gather_edges(comm, C.c_void_p(d_in.data_ptr()), cnt, 0, C.c_void_p(d_out.data_ptr()), C.c_void_p(st)) == 0, _lib.last_error()
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            assert got[:n].tobytes() == e.tobytes() and not got[n:].any()
        # argument errors come back as codes, before anything is posted
        cnt = (C.c_uint64 * 1)(3)
        assert L.sigax_gather_edges(comm, None, cnt, 0, None, None) == -1
        assert L.sigax_gather_edges(comm, None, cnt, 5, None, None) == -1
    finally:
        L.sigax_comm_destroy(comm)
    assert L.sigax_comm_create(0, 3, 2, idb, C.byref(comm)) == -1  # rank outside the world


def test_bench_takes_the_multi_rank_path_with_a_world_of_one(tmp_path):
    """bench.py's N > 1 code -- RCCL process group, the library's communicator made from an id broadcast over it, every step's
    edge records through sigax_gather_counts / sigax_gather_edges into rank 0's device buffer and on to pinned host memory,
    timing reduced over ranks, `config.ranks` -- rehearsed with a world of ONE rank on the one GPU of this box
    (SIGA_BENCH_REHEARSE_RANKS=1), so that the driver's 2/4/8-GPU runs are not its first execution.  Same edge count as the
    plain one-GPU run."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--seed", "2", "--read-len", "100", "--min-overlap", "40",
              "--workdir", str(tmp_path / "job"), "--no-e2e", "--upload-steps", "0", "--reads-per-gpu", "40000", "--genome-per-gpu", "200000"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)

    def run(extra_env):
        r = subprocess.run([sys.executable, bench] + common, cwd=ROOT, env=dict(env, **extra_env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    plain = run({})
    multi = run({"SIGA_BENCH_REHEARSE_RANKS": "1"})
    assert "ranks" not in plain["config"]
    assert multi["config"]["ranks"]["world_size"] == 1 and "sigax_gather_edges" in multi["config"]["ranks"]["edge_gather"]
    assert multi["config"]["edges"] == plain["config"]["edges"] > 20000
    fallback = run({"SIGA_BENCH_REHEARSE_RANKS": "1", "SIGA_BENCH_TORCH_GATHER": "1"})
    assert "torch.distributed" in fallback["config"]["ranks"]["edge_gather"] and fallback["config"]["edges"] == plain["config"]["edges"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, idfile, out):
    """one process per GPU: rank r sends 1000 * (r + 1) records to rank 0 through sigax_gather_edges"""
    from siga_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(rank)
    idb = (C.c_uint8 * 128)()
    if rank == 0:
        assert L.sigax_comm_unique_id(idb) == 0, _lib.last_error()
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(idb))
        os.rename(idfile + ".tmp", idfile)
    else:
        import time
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        raw = open(idfile, "rb").read()
        for i in range(128):
            idb[i] = raw[i]
    comm = C.c_void_p()
    assert L.sigax_comm_create(rank, rank, world, idb, C.byref(comm)) == 0, _lib.last_error()
    n = 1000 * (rank + 1)
    e = _edges(rank, n)
    d_in = torch.from_numpy(e.view(np.int32).reshape(-1, 4).copy()).cuda()
    cnt = (C.c_uint64 * world)()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.sigax_gather_counts(comm, n, cnt, st) == 0, _lib.last_error()
    counts = [int(c) for c in cnt]
    d_out = torch.zeros((sum(counts), 4), dtype=torch.int32, device="cuda") if rank == 0 else None
    assert L.sigax_gather_edges(comm, C.c_void_p(d_in.data_ptr()), cnt, 0, C.c_void_p(d_out.data_ptr()) if rank == 0 else None, st) == 0, _lib.last_error()
    torch.cuda.synchronize()
    if rank == 0:
        want = np.concatenate([_edges(r, 1000 * (r + 1)) for r in range(world)])
        out.put((counts, d_out.cpu().numpy().tobytes() == want.tobytes()))
    L.sigax_comm_destroy(comm)


@two_gpus
def test_rccl_gather_between_two_gpus(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / "nccl.id")
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, idfile, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    counts, same = q.get(timeout=10)
    assert counts == [1000, 2000] and same


@two_gpus
def test_index_clone_on_the_second_gpu_answers_like_the_first():
    """sigax_index_clone 0 -> 1: tables copied device to device (hipMemcpyPeer), the replica builds its own row and deep start
    tables on ITS GPU and gives the same blocks and edges."""
    import siga_amd
    from siga_amd import _lib
    from siga_amd.overlap import FMIndexPair, name_ranks, read_sequences
    fx = fixture("mid")
    reads = read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    pair = FMIndexPair.load(fx.prefix, device=0)
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    h = C.c_void_p()
    assert _lib.lib().sigax_index_clone(pair.handle, 1, C.byref(h)) == 0, _lib.last_error()
    twin = FMIndexPair(h.value)
    twin._resident = True
    assert twin.info()["device"] == 1
    pos = np.arange(0, len(fx.fwd), 97, dtype=np.uint64)
    for which in (0, 1):
        assert np.array_equal(pair.occ(pos, which), twin.occ(pos, which))
    a = siga_amd.OverlapBuilder(pair).overlap(seqs, 45, edges=True)
    b = siga_amd.OverlapBuilder(twin).overlap(seqs, 45, edges=True)
    for k in ("blocks", "block_offs", "edges", "substring"):
        assert a[k].tobytes() == b[k].tobytes(), k
    twin.close()
    pair.close()


@two_gpus
def test_cli_overlap_on_two_real_gpus_writes_the_one_gpu_file(tmp_path, monkeypatch):
    from siga_amd import host
    monkeypatch.delenv("SIGA_DEVICE_MAP", raising=False)
    fx = fixture("mid")
    want, _, _ = fx.oracle_asqg(45)
    outs = []
    for gpus in (1, 2):
        monkeypatch.setenv("SIGA_BATCH_READS", str(max(len(fx.reads) // 6, 1)))
        out = str(tmp_path / ("g%d.asqg.gz" % gpus))
        host.overlap_file(fx.fa, fx.prefix, 45, out, gpus=gpus)
        outs.append(open(out, "rb").read())
    assert gzip.decompress(outs[0]).decode() == want
    assert outs[0] == outs[1]


@two_gpus
def test_bench_two_ranks_over_rccl(tmp_path):
    """bench.py --gpus 2 --backend nccl as the driver starts N > 1: two fresh ranks, one per GPU, the edge records gathered
    through the C-ABI's RCCL step; ONE JSON line with n_gpus 2, the one-rank edge count, and what RCCL saw."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--seed", "2", "--read-len", "100", "--min-overlap", "40",
              "--workdir", str(tmp_path / "job"), "--no-e2e", "--upload-steps", "0"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)

    def run(extra):
        r = subprocess.run([sys.executable, bench] + extra + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    two = run(["--gpus", "2", "--backend", "nccl", "--reads-per-gpu", "20000", "--genome-per-gpu", "100000"])
    one = run(["--gpus", "1", "--reads-per-gpu", "40000", "--genome-per-gpu", "200000"])
    assert two["n_gpus"] == 2 and two["config"]["ranks"]["world_size"] == 2 and two["config"]["ranks"]["devices_visible"] >= 2
    assert "sigax_gather_edges" in two["config"]["ranks"]["edge_gather"]
    assert two["config"]["edges"] == one["config"]["edges"] > 20000
