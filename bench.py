#!/usr/bin/env python3
"""bench.py -- reads/s overlapped by the MI355X-native `siga overlap` path (BASELINE.json metric).

A "step" is one pass of the whole hot path (block finder -> sub-maximal filter / irreducible extraction ->
ordered compaction -> edge records) over this rank's shard of reads, inputs already resident in HBM, followed (for
N > 1) by the RCCL gather of the edge records to rank 0.  Workload at N = 1: BASELINE configs[1], synthetic
1M x 150 bp reads from a 5 Mb genome, min-overlap 45, irreducible, both strands.  For N > 1 the per-GPU share is
kept (weak scaling): N x 1M reads from an N x 5 Mb genome, index replicated on every GPU, reads sharded.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
KERNELS = ["k_find", "k_filter_extract_fast", "k_filter_extract", "k_order", "k_edges"]


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-gpu", type=int, default=1000000)
    ap.add_argument("--genome-per-gpu", type=int, default=5000000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-overlap", type=int, default=45)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads timed on the CPU restatement (0 = skip)")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--subbatches", type=int, default=2,
                    help="sub-batches per step (0 = library default for one batch at a time: 4 at this size); sub-batch i's "
                         "filter/extract kernels run beside sub-batch i+1's finder")
    ap.add_argument("--depth", type=int, default=2,
                    help="batches in flight on the index (2 = batch k+1's finder starts beside batch k's filter/extract tail)")
    ap.add_argument("--isolated", action="store_true",
                    help="after the timed steps, also time 2 steps with sub-batching off and report them under roofline.isolated")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="measurement aid: build the index of an N-GPU job (N x reads, N x genome) but run only rank 0's shard "
                         "in this single process, to see what one GPU of that job achieves")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 flow on fewer GPUs than ranks (edge records gathered via host)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the overlap path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60),
                                    device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60))

    from siga_amd import _lib, host
    from siga_amd import build as sbuild
    from siga_amd.overlap import FMIndexPair
    from siga_amd.sharding import gather_edges_async, shard_range
    from tests.golden.make_reads import fast_reads

    job_world = args.emulate_world if (args.emulate_world and world == 1) else world
    n_total = args.reads_per_gpu * job_world
    G = args.genome_per_gpu * job_world
    L = args.read_len
    workdir = args.workdir or os.path.join(tempfile.gettempdir(), "siga_bench_%d_%d_%d_%d" % (n_total, G, L, args.seed))
    prefix = os.path.join(workdir, "reads")

    # ---- synthetic reads (every rank draws the same set) and the index (rank 0 builds, everyone loads) ----
    t0 = time.time()
    reads, _ = fast_reads(G, L, n_total, args.seed)  # uint8 [n_total, L]
    if rank == 0:
        sbuild.build_all()
        os.makedirs(workdir, exist_ok=True)
        if not all(os.path.exists(prefix + e) for e in (".bwt", ".rbwt", ".sai", ".rsai")):
            offs_all = np.arange(0, (n_total + 1) * L, L, dtype=np.uint64)
            host.index_build(reads.reshape(-1), offs_all, prefix, threads=max(2, min(os.cpu_count() or 2, 96)))
        log("reads + index ready in %.1f s (%d reads, %d symbols per strand)" % (time.time() - t0, n_total, n_total * (L + 1)))
    if world > 1:
        dist.barrier()
    pair = FMIndexPair.load(prefix, device=dev_index)
    info = pair.info()
    # ReadInfo{name,length}: names r<i>; rank of a name under std::string operator<
    names = np.char.add("r", np.arange(n_total).astype(str))
    order = np.argsort(names, kind="stable")
    name_rank = np.empty(n_total, dtype=np.uint32)
    name_rank[order] = np.arange(n_total, dtype=np.uint32)
    pair.set_reads(np.full(n_total, L, dtype=np.uint32), name_rank)
    log("index on GPU: %.1f MB, wide=%d (%.1f s since start)" % (info["device_bytes"] / 1e6, info["wide"], time.time() - t0))

    lo, hi = shard_range(n_total, rank, job_world)
    n_local = hi - lo
    d_seqs = torch.from_numpy(reads[lo:hi].reshape(-1).copy()).to(dev)
    d_offs = torch.arange(0, (n_local + 1) * L, L, dtype=torch.int64, device=dev)
    lib = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    # --depth batches in flight on one index: the library queues every batch's finder launches on one stream and its
    # filter/extract launches on another, so batch k+1's first finder launch runs beside batch k's last filter/extract
    # launch.  Each batch has its own workspace and is driven from its own stream; a step = one batch run to completion.
    depth = max(1, args.depth)
    batches, streams = [], []
    for _ in range(depth):
        bt = C.c_void_p()
        rc = lib.sigax_batch_create(pair.handle, n_local, n_local * L, L, C.byref(bt))
        if rc != 0:
            raise SystemExit("sigax_batch_create: " + _lib.last_error())
        rc = lib.sigax_batch_set_device_reads(bt, d_seqs.data_ptr(), d_offs.data_ptr(), n_local, n_local * L, L)
        assert rc == 0, _lib.last_error()
        assert lib.sigax_batch_set_subbatches(bt, args.subbatches) == 0, _lib.last_error()
        batches.append(bt)
        streams.append(torch.cuda.Stream(device=dev) if depth > 1 else torch.cuda.current_stream(dev))
    batch = batches[0]
    torch.cuda.synchronize(dev)

    class _EdgeView:  # zero-copy view of the library's device edge buffer for torch.distributed
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<i4", "data": (ptr, False), "version": 2}

    stats = _lib.Stats()
    kms = (C.c_float * 5)()
    nsub = C.c_uint32(1)
    ksum = np.zeros(5)
    inflight = [False] * depth
    pending = []  # edge gathers in flight: step k's gather runs beside step k+1's kernels
    last_edges = [0]

    def drain():
        tot = None
        while pending:
            _, counts = pending.pop(0).wait()
            tot = sum(counts)
        return tot

    def complete(i):
        """finish batch i's run: stats, kernel times, and (N > 1) start the gather of its edge records"""
        if not inflight[i]:
            return
        inflight[i] = False
        sp = C.c_void_p(streams[i].cuda_stream)
        rc = lib.sigax_batch_finish(batches[i], sp, C.byref(stats))
        if rc != 0:
            raise SystemExit("overlap step failed: " + _lib.last_error())
        lib.sigax_batch_kernel_ms(batches[i], C.byref(kms), C.byref(nsub))  # HIP events on the streams the kernels run on
        ksum[:] += np.array(list(kms))
        last_edges[0] = int(stats.n_edges)
        if world > 1:
            d_edges = C.c_void_p()
            lib.sigax_batch_device_outputs(batches[i], None, None, None, C.byref(d_edges))
            ne = int(stats.n_edges)
            with torch.cuda.stream(streams[i]):
                # copy out of the batch's buffer (its next run overwrites it), then gather asynchronously
                local = torch.as_tensor(_EdgeView(d_edges.value, ne), device=dev).clone() if ne else torch.zeros((0, 4), dtype=torch.int32, device=dev)
                if args.backend == "gloo":
                    local = local.cpu()
                if len(pending) >= 2:
                    pending.pop(0).wait()
                pending.append(gather_edges_async(local))

    def submit(k):
        i = k % depth
        complete(i)  # the batch's previous run must be done before its workspace is reused
        rc = lib.sigax_batch_run(batches[i], lo, args.min_overlap, flags, C.c_void_p(streams[i].cuda_stream))
        if rc != 0:
            raise SystemExit("overlap step failed: " + _lib.last_error())
        inflight[i] = True

    def run_steps(n, k0=0):
        for k in range(k0, k0 + n):
            submit(k)
        for k in range(k0 + n, k0 + n + depth):  # complete in submission order
            complete(k % depth)

    # every batch object runs once before anything is counted (its first run allocates its device workspace), then the
    # W warm-up steps
    run_steps(depth)
    run_steps(args.warmup, depth)
    drain()
    ksum[:] = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    run_steps(args.steps, depth + args.warmup)
    total_edges = last_edges[0]
    if world > 1:
        total_edges = drain()  # every step's edge records have reached rank 0 before the clock stops
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kavg = ksum / max(args.steps, 1)  # per step, summed over the step's sub-batch launches
    st = stats.as_dict()
    launches = int(nsub.value)

    # the same kernels with sub-batching off (no overlap between find and filter/extract): untimed extra steps
    iso = None
    if launches > 1 and args.isolated:
        assert lib.sigax_batch_set_subbatches(batch, 1) == 0
        ksum[:] = 0
        for k in range(2):
            submit(k * depth)  # always batch 0, one run at a time
            complete(0)
        drain()
        iso = ksum / 2

    out = None
    if rank == 0:
        reads_per_s = (n_total if job_world == world else n_local) * args.steps / elapsed
        # algorithmic bytes: 64 B per distinct Occ evaluation + L per read + 64 B per block out
        n_occ = st["n_occ_find"] + st["n_occ_extract"]
        bytes_find = 64 * st["n_occ_find"] + n_local * L + 64 * st["n_candidate_blocks"]
        bytes_read = (64 * n_occ + n_local * L + 64 * st["n_blocks"]) / max(n_local, 1)
        find_ms = float(kavg[0]) / launches          # average duration of one k_find launch in the timed region
        bytes_launch = bytes_find / launches
        achieved = bytes_launch / (find_ms * 1e-3) / 1e9 if find_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "k_find/%d/%d/%d/%d" % (args.reads_per_gpu, args.genome_per_gpu, L, launches)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "reads/sec overlapped (ASQG bit-exact)", "value": reads_per_s, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d bp reads from %d bp genome, min-overlap %d, irreducible, both strands; "
                                   "FM-index (both strands) resident in HBM, reads sharded over %d GPU(s)" % (
                                       n_total, L, G, args.min_overlap, world),
                       "reads_per_gpu": n_local, "edges": total_edges, "blocks_per_read": st["n_blocks"] / max(n_local, 1),
                       "n_occ_min_per_read": n_occ / max(n_local, 1), "algorithmic_bytes_per_read": bytes_read,
                       "slow_path_reads": st["n_slow_reads"], "batches_in_flight": depth},
            "kernel_ms_per_step": {k: float(v) for k, v in zip(KERNELS, kavg)},
            "launches_per_step": launches,
            "roofline": {"bound": "hbm", "kernel": "k_find", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": find_ms,
                         "note": "launch durations in the timed region, where sub-batch i+1's k_find runs beside "
                                 "sub-batch i's filter/extract kernels" if launches > 1 else "kernels run back to back",
                         "whole_path_achieved": bytes_read * n_local / (elapsed / args.steps) / 1e9},
        }
        if iso is not None:
            ims = float(iso[0])
            out["roofline"]["isolated"] = {
                "what": "same kernels, sub-batching off (one launch per step, nothing beside it), 2 untimed steps",
                "k_find_ms": ims, "achieved": bytes_find / (ims * 1e-3) / 1e9, "frac": bytes_find / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel_ms_per_step": {k: float(v) for k, v in zip(KERNELS, iso)}}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(prefix, reads, min(args.cpu_sample, n_total), args.min_overlap, st, lib, batch)

    for bt in batches:
        lib.sigax_batch_destroy(bt)
    pair.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline(prefix, reads, sample, min_overlap, st, lib, batch):
    """The oracle (CPU restatement of the reference's OverlapBuilder::overlap, same RL-BWT + marker structures)
    timed on the first `sample` reads of the workload against the full index, OpenMP over reads on all host
    cores.  Checker/baseline only: nothing here feeds the GPU path."""
    from oracle import pyoracle as po
    po.build()
    t0 = time.time()
    fwd = po.Index.load(prefix + ".bwt", prefix + ".sai")
    rev = po.Index.load(prefix + ".rbwt", prefix + ".rsai")
    log("oracle index loaded in %.1f s" % (time.time() - t0))
    seqs = [bytes(r) for r in reads[:sample]]
    threads = po.max_threads()
    sec, o = po.overlap_batch_timed(fwd, rev, seqs, min_overlap, True, True, threads)
    sec1, _ = po.overlap_batch_timed(fwd, rev, seqs[: max(sample // 20, 1)], min_overlap, True, True, 1)
    return {"value": sample / sec, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": "first %d reads of the workload against the full index; OverlapBuilder::overlap only (no I/O); "
                      "OpenMP over reads" % sample,
            "seconds": sec, "single_thread_reads_per_s": max(sample // 20, 1) / sec1,
            "blocks_per_read": o["blocks"] / sample, "n_occ_min_per_read": o["n_occ_min"] / sample}


if __name__ == "__main__":
    main()
