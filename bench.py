#!/usr/bin/env python3
"""bench.py -- reads/s overlapped by the MI355X-native `siga overlap` path (BASELINE.json metric).

A "step" is one pass of the whole hot path (locality order of the batch -> block finder -> sub-maximal filter /
irreducible extraction -> ordered compaction -> edge records) over this rank's shard of reads, inputs already resident
in HBM, ending with the edge records in pinned host memory on rank 0 (for N > 1 after the RCCL gather to rank 0).  Every
step hands the batch object its reads anew, so everything a product batch pays per upload -- the ordering included -- is
inside the timed region.  Workload at N = 1: BASELINE configs[1], synthetic 1M x 150 bp reads from a 5 Mb genome, seed 1,
min-overlap 45, irreducible, both strands.  N = 8: BASELINE configs[2] exactly, 20M x 150 bp reads from a 100 Mb genome,
seed 2, 2.5M reads per rank, index replicated on every GPU; N = 2, 4 keep that per-GPU share (weak scaling: N x 2.5M
reads from an N x 12.5 Mb genome, seed 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload correct                          (BASELINE configs[3]: `siga correct` k-mer path)

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CACHE_ROWS_GBS = 7650.0     # random rows of a 151 MB table: 7.4-7.9 TB/s (MI355X_MICROARCH.md, "uniformly random rows"); midpoint
INFINITY_CACHE_BYTES = 256 << 20
GATHER_CEILING_GLINES = 55.0  # dependency-free random line reads this chip sustains (profiles/r01_gather_probe*.txt): 55-57 G lines/s
KERNELS = ["k_find", "k_filter_extract_fast", "k_filter_extract", "k_order", "k_edges"]


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def kernels_sha():
    """sha1 of the kernel sources: a PMC figure in profiles/traffic.json only counts for the kernels it was measured on"""
    import hashlib
    h = hashlib.sha1()
    for f in ("sigax_kernels.hip", "fm_layout.h"):
        h.update(open(os.path.join(ROOT, "siga_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def traffic_entry(key):
    """hbm_bytes_per_launch of profiles/traffic.json for `key`, or None when there is none or it was measured on other
    kernel sources than the ones in the tree (entries carry the sha of the sources they were collected with)"""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:
        return None
    if not e or e.get("kernels_sha") != kernels_sha():
        return None
    return e.get("hbm_bytes_per_launch")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="overlap", choices=["overlap", "correct"])
    ap.add_argument("--reads-per-gpu", type=int, default=None, help="default: 1 000 000 at N = 1 (configs[1]), 2 500 000 at N > 1 (configs[2]'s share)")
    ap.add_argument("--genome-per-gpu", type=int, default=None, help="default: 5 000 000 at N = 1, 12 500 000 at N > 1")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-overlap", type=int, default=45)
    ap.add_argument("--seed", type=int, default=None, help="default: 1 at N = 1 (configs[1]), 2 at N > 1 (configs[2])")
    ap.add_argument("--cpu-sample", type=int, default=400000, help="reads timed on the CPU restatement (0 = skip); 400 k = 4 s on 128 threads")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--subbatches", type=int, default=1,
                    help="sub-batches per step (0 = library default for one batch at a time: 4 at this size); sub-batch i's "
                         "filter/extract kernels run beside sub-batch i+1's finder.  With three batches in flight the overlap "
                         "comes from the next batch and one launch chain per step is fastest (tools/sweep_pipeline.sh)")
    ap.add_argument("--depth", type=int, default=3,
                    help="batches in flight on the index (batch k+1's finder starts beside batch k's filter/extract tail; the third "
                         "hides batch k-1's scan / scatter / edge tail and its copy to the host)")
    ap.add_argument("--isolated", action="store_true",
                    help="after the timed steps, also time 2 steps with sub-batching off and report them under roofline.isolated")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="measurement aid: build the index of an N-GPU job (N x reads, N x genome) but run only rank 0's shard "
                         "in this single process, to see what one GPU of that job achieves")
    ap.add_argument("--max-local-reads", type=int, default=0,
                    help="with --emulate-world: run only the first so many reads of rank 0's shard per step (a shard of BASELINE "
                         "configs[4] is 6.25 M reads of 250 bp: more than three batch objects in flight hold beside its index)")
    ap.add_argument("--shard", default="key", choices=["key", "contiguous"],
                    help="N > 1: which reads a rank gets.  key (default): a contiguous range of the reads' locality-key order "
                         "(siga_amd.sharding.locality_keys -> sigax_locality_keys: minimizer hash, from the sequences alone; computed once per read set, "
                         "outside the step; the reads keep their ids: sigax_batch_set_device_read_ids); contiguous: a contiguous "
                         "range of the file, as rounds 1-3 did")
    ap.add_argument("--key-order", action="store_true",
                    help="N = 1: hand the batch its reads in locality-key order (the one rank's slice of --shard key is the whole order; "
                         "ids kept).  Off by default: the headline keeps the file's order")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 flow on fewer GPUs than ranks (edge records gathered via host)")
    ap.add_argument("--reuse-order", action="store_true",
                    help="A/B aid: set the reads once, so that the locality order is computed in the warm-up only (what round 2 timed)")
    ap.add_argument("--kmer", type=int, default=31, help="--workload correct: k-mer size (31 = code default, 41 = example script)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end_to_end leg (N = 1, error-free reads of at most 2 M: the reads as FASTA through the `siga overlap` "
                         "CLI in a child process, after the timed region)")
    ap.add_argument("--upload-steps", type=int, default=30,
                    help="steps of the upload_inclusive leg after the timed region (every step's reads come from pinned host memory "
                         "through sigax_batch_upload); 0 = skip")
    ap.add_argument("--error-rate", type=float, default=None,
                    help="substitutions per base: --workload correct default 0.01; --workload overlap default 0 (the BASELINE configs are "
                         "error-free), > 0 shows what uncorrected reads cost (branches in the extraction)")
    args = ap.parse_args()
    multi = args.gpus > 1 or args.emulate_world > 1
    if args.reads_per_gpu is None:
        args.reads_per_gpu = 2500000 if multi and args.workload == "overlap" else 1000000
    if args.genome_per_gpu is None:
        args.genome_per_gpu = 12500000 if multi and args.workload == "overlap" else 5000000
    if args.seed is None:
        args.seed = 2 if multi and args.workload == "overlap" else 1
    return args


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child processes (torchrun, one
    per GPU) BEFORE anything in this process touches the GPU, and leave with their exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd)


def make_sigax_comm(lib, rank, world, dev_index, dev):
    """The library's own communicator for the edge gather (include/sigax.h: sigax_comm_*, RCCL bound at run time): rank 0's
    unique id travels over the torch process group; whether every rank got its communicator is agreed on before anyone uses
    it (MIN all-reduce), so a rank that could not falls back together with the others.  Returns the handle or None."""
    import torch
    import torch.distributed as dist
    idbuf = (C.c_uint8 * 128)()
    ok = 1
    if rank == 0:
        ok = 1 if lib.sigax_comm_unique_id(idbuf) == 0 else 0
        if not ok:
            log("sigax_comm_unique_id: " + lib.sigax_last_error().decode(errors="replace"))
    t = torch.tensor(list(idbuf) + [ok], dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    vals = t.cpu().tolist()
    if vals[128] != 1:
        return None
    for i in range(128):
        idbuf[i] = vals[i]
    h = C.c_void_p()
    rc = lib.sigax_comm_create(dev_index, rank, world, idbuf, C.byref(h))
    if rc != 0:
        log("rank %d sigax_comm_create: %s" % (rank, lib.sigax_last_error().decode(errors="replace")))
    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if rc == 0:
            lib.sigax_comm_destroy(h)
        return None
    return h


def rank_of_r_names(n):
    from tests.golden.make_reads import rank_of_r_names as f
    return f(n)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    if args.workload == "correct":
        return bench_correct(args)
    return bench_overlap(args)


def bench_overlap(args):
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
    # SIGA_BENCH_REHEARSE_RANKS=1 (tests, one-GPU boxes): a world of ONE rank takes the N > 1 path -- process group, the library's
    # RCCL communicator, counts + gather of every step's edge records, timing reduced over ranks -- so that the code the
    # driver's multi-GPU runs execute has run before it gets there
    multi_path = world > 1 or os.environ.get("SIGA_BENCH_REHEARSE_RANKS") == "1"
    if multi_path:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
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

    # ---- synthetic reads (every rank draws from the same set: its own shard, and rank 0 all of them when the index has to
    # be built) and the index (rank 0 builds on its GPU, everyone loads) ----
    t0 = time.time()
    by_pos = os.environ.get("SIGA_BENCH_BY_POSITION") == "1"  # measurement aid: reads in genome order
    if by_pos:
        workdir += "_bypos"
    if args.error_rate:
        workdir += "_e%g" % args.error_rate
    prefix = os.path.join(workdir, "reads")
    lo, hi = shard_range(n_total, rank, job_world)
    if args.max_local_reads:
        hi = min(hi, lo + args.max_local_reads)
    have_index = all(os.path.exists(prefix + e) for e in (".bwt", ".rbwt", ".sai", ".rsai"))
    # key-range sharding (N > 1): rank r runs reads order[lo:hi] of the locality-key order instead of reads lo..hi-1 of the
    # file; rank 0 computes the order once per read set (from the sequences alone; kept beside the index files)
    by_key = ((job_world > 1 and args.shard == "key") or args.key_order) and not by_pos
    order_path = prefix + ".keyorder.npy"
    whole = (rank == 0 and not have_index) or bool(args.error_rate) or (args.cpu_sample > 0 and world == 1) or \
            (by_key and rank == 0 and not os.path.exists(order_path))

    def draw(subset):
        r, _ = fast_reads(G, L, n_total, args.seed, by_position=by_pos, subset=subset)  # uint8 [., L]
        if args.error_rate:
            from tests.golden.make_reads import substitute
            r = substitute(r, args.error_rate, args.seed + 100)
        return r

    reads = draw(None) if whole else None
    if rank == 0:
        sbuild.build_all()
        os.makedirs(workdir, exist_ok=True)
        if not have_index:
            offs_all = np.arange(0, (n_total + 1) * L, L, dtype=np.uint64)
            host.index_build_gpu(reads.reshape(-1), offs_all, prefix, device=dev_index)
        log("reads + index ready in %.1f s (%d reads, %d symbols per strand)" % (time.time() - t0, n_total, n_total * (L + 1)))
    shard_keys_s = None
    if by_key and rank == 0 and not os.path.exists(order_path):
        try:
            from siga_amd.sharding import locality_keys, key_order
            locality_keys(reads[:1024], device=dev)  # (the library's code object is loaded at its first launch: not the keys' time)
            tk = time.time()
            order = key_order(locality_keys(reads, device=dev), device=dev).astype(np.uint32)
            shard_keys_s = time.time() - tk
            np.save(order_path + ".tmp.npy", order)
            os.replace(order_path + ".tmp.npy", order_path)
            with open(order_path + ".seconds", "w") as f:
                f.write("%.3f\n" % shard_keys_s)
            log("locality keys + order of %d reads in %.1f s (once per read set)" % (n_total, shard_keys_s))
            del order
            torch.cuda.empty_cache()
        except Exception as e:  # the sharding policy must not cost the run: every rank then takes its file range (below)
            log("locality keys failed (%s: %s); ranks take file ranges" % (type(e).__name__, e))
    if multi_path:
        dist.barrier()
    ids = None
    if by_key:
        # every rank looks at the same file: no order (or not this read set's) = file ranges for all of them
        try:
            by_key = np.load(order_path, mmap_mode="r").shape == (n_total,)
        except Exception:
            by_key = False
        if not by_key:
            log("rank %d: no locality-key order for this read set, taking the file range" % rank)
    if by_key:
        ids = np.ascontiguousarray(np.load(order_path, mmap_mode="r")[lo:hi]).astype(np.uint32)
        if shard_keys_s is None and os.path.exists(order_path + ".seconds"):
            shard_keys_s = float(open(order_path + ".seconds").read())
        shard = reads[ids] if whole else draw(ids)
    else:
        shard = reads[lo:hi] if whole else draw((lo, hi))
    pair = FMIndexPair.load(prefix, device=dev_index)
    # ReadInfo{name,length}: names r<i>; rank of a name under std::string operator<
    pair.set_reads(np.full(n_total, L, dtype=np.uint32), rank_of_r_names(n_total))
    # the index stays open for every step: row tables (set_reads did that) and the finder's deep start table for this -m
    pair.prepare_overlap(args.min_overlap)
    info = pair.info()  # (device bytes with every table in place)
    log("index on GPU: %.1f MB, wide=%d (%.1f s since start)" % (info["device_bytes"] / 1e6, info["wide"], time.time() - t0))

    # N > 1 over RCCL: the edge gather goes through the library's own exchange step (sigax_gather_counts + sigax_gather_edges:
    # ncclAllGather of counts, grouped ncclSend/ncclRecv of records); torch.distributed's gather stays as the fallback
    # (gloo rehearsals, SIGA_BENCH_TORCH_GATHER=1, or a rank without its communicator)
    comm = None
    if multi_path and args.backend == "nccl" and os.environ.get("SIGA_BENCH_TORCH_GATHER") != "1":
        comm = make_sigax_comm(_lib.lib(), rank, world, dev_index, dev)
        log("rank %d: edge gather through %s" % (rank, "sigax_gather_edges (RCCL, C-ABI)" if comm else "torch.distributed"))
    n_local = hi - lo
    d_seqs = torch.from_numpy(np.ascontiguousarray(shard).reshape(-1)).to(dev)
    d_offs = torch.arange(0, (n_local + 1) * L, L, dtype=torch.int64, device=dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev) if ids is not None else None
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
        if d_ids is not None:  # the reads' places in the index's read table stay with the batch object
            assert lib.sigax_batch_set_device_read_ids(bt, d_ids.data_ptr(), n_local) == 0, _lib.last_error()
        batches.append(bt)
        streams.append(torch.cuda.Stream(device=dev) if depth > 1 else torch.cuda.current_stream(dev))
    batch = batches[0]
    torch.cuda.synchronize(dev)

    class _EdgeView:  # zero-copy view of the library's device edge buffer for torch
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<i4", "data": (ptr, False), "version": 2}

    stats = _lib.Stats()
    rinfo = _lib.RunInfo()
    order_ms = [0.0]
    kms = (C.c_float * 5)()
    nsub = C.c_uint32(1)
    ksum = np.zeros(5)
    inflight = [False] * depth
    pending = []  # edge gathers in flight: step k's gather runs beside step k+1's kernels
    last_edges = [0]
    gather_buf = [None] * depth   # rank 0, sigax_gather_edges: device buffers the gathered records land in
    gathered_total = [0]
    host_edges = [None] * depth   # pinned host buffers: where a step's edge records end up (rank 0)
    host_cap = [0]

    def host_buf(i, need):
        if host_edges[i] is None or host_edges[i].shape[0] < need:
            host_cap[0] = max(host_cap[0], int(need * 1.25) + 1024)
            host_edges[i] = torch.empty((host_cap[0], 4), dtype=torch.int32).pin_memory()
        return host_edges[i]

    def land(i, t):
        """rank 0: the gathered records of one step -> pinned host memory (asynchronous copy on the batch's stream)"""
        if t is not None and t.shape[0]:
            host_buf(i, t.shape[0])[: t.shape[0]].copy_(t, non_blocking=True)

    def drain():
        tot = None
        while pending:
            i, pg = pending.pop(0)
            t, counts = pg.wait()
            with torch.cuda.stream(streams[i]):
                land(i, t if (t is None or t.is_cuda) else None)
            tot = sum(counts)
        return tot

    def complete(i):
        """finish batch i's run: stats, kernel times, and send its edge records on their way (N > 1: gather to rank 0)"""
        if not inflight[i]:
            return
        inflight[i] = False
        sp = C.c_void_p(streams[i].cuda_stream)
        rc = lib.sigax_batch_finish(batches[i], sp, C.byref(stats))
        if rc != 0:
            raise SystemExit("overlap step failed: " + _lib.last_error())
        lib.sigax_batch_kernel_ms(batches[i], C.byref(kms), C.byref(nsub))  # HIP events on the streams the kernels run on
        ksum[:] += np.array(list(kms))
        lib.sigax_batch_run_info(batches[i], C.byref(rinfo))
        order_ms[0] += rinfo.order_ms
        last_edges[0] = int(stats.n_edges)
        d_edges = C.c_void_p()
        lib.sigax_batch_device_outputs(batches[i], None, None, None, C.byref(d_edges))
        ne = int(stats.n_edges)
        with torch.cuda.stream(streams[i]):
            if not multi_path:
                # the records leave the device here, inside the timed region; the copy runs beside the next step's kernels
                # and is waited for when this batch object is finished the next time (or by the final synchronize)
                if ne:
                    land(i, torch.as_tensor(_EdgeView(d_edges.value, ne), device=dev))
            elif comm is not None:
                # the library's exchange step on the batch's own stream: counts (one host wait), then the records straight
                # out of the batch's edge buffer (the batch's next run is ordered behind them on this stream) into rank 0's
                # device buffer, and from there to pinned host memory
                cnts = (C.c_uint64 * world)()
                rc = lib.sigax_gather_counts(comm, ne, cnts, sp)
                if rc != 0:
                    raise SystemExit("sigax_gather_counts: " + _lib.last_error())
                tot = sum(int(c) for c in cnts)
                gathered_total[0] = tot
                dout = None
                if rank == 0:
                    if gather_buf[i] is None or gather_buf[i].shape[0] < tot:
                        gather_buf[i] = torch.empty((int(tot * 1.25) + 1024, 4), dtype=torch.int32, device=dev)
                    dout = C.c_void_p(gather_buf[i].data_ptr())
                rc = lib.sigax_gather_edges(comm, d_edges, cnts, 0, dout, sp)
                if rc != 0:
                    raise SystemExit("sigax_gather_edges: " + _lib.last_error())
                if rank == 0 and tot:
                    land(i, gather_buf[i][:tot])
            else:
                # copy out of the batch's buffer (its next run overwrites it), then gather asynchronously
                local = torch.as_tensor(_EdgeView(d_edges.value, ne), device=dev).clone() if ne else torch.zeros((0, 4), dtype=torch.int32, device=dev)
                if args.backend == "gloo":
                    local = local.cpu()
                while len(pending) >= 2:
                    j, pg = pending.pop(0)
                    t, _ = pg.wait()
                    land(j, t if (t is None or t.is_cuda) else None)
                pending.append((i, gather_edges_async(local)))

    h_seqs = h_offs = None  # upload_inclusive leg: the shard in pinned host memory

    def submit(k, upload=False):
        i = k % depth
        complete(i)  # the batch's previous run must be done before its workspace is reused
        if upload:
            # what a product batch pays: the reads leave (pinned) host memory inside the step
            rc = lib.sigax_batch_upload(batches[i], C.c_char_p(h_seqs.data_ptr()), h_offs.ctypes.data, n_local, C.c_void_p(streams[i].cuda_stream))
            assert rc == 0, _lib.last_error()
        elif not args.reuse_order:
            # a new set of reads as far as the library knows: the step pays for its own locality order, like a product batch
            rc = lib.sigax_batch_set_device_reads(batches[i], d_seqs.data_ptr(), d_offs.data_ptr(), n_local, n_local * L, L)
            assert rc == 0, _lib.last_error()
        rc = lib.sigax_batch_run(batches[i], lo, args.min_overlap, flags, C.c_void_p(streams[i].cuda_stream))
        if rc != 0:
            raise SystemExit("overlap step failed: " + _lib.last_error())
        inflight[i] = True

    def run_steps(n, k0=0, upload=False):
        for k in range(k0, k0 + n):
            submit(k, upload)
        for k in range(k0 + n, k0 + n + depth):  # complete in submission order
            complete(k % depth)

    # every batch object runs once before anything is counted (its first run allocates its device workspace), then the
    # W warm-up steps
    run_steps(depth)
    run_steps(args.warmup, depth)
    drain()
    ksum[:] = 0
    order_ms[0] = 0.0
    if multi_path:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    run_steps(args.steps, depth + args.warmup)
    total_edges = last_edges[0]
    if multi_path and comm is not None:
        total_edges = gathered_total[0]  # (the copies to pinned host memory are waited for by the synchronize below)
    elif multi_path:
        total_edges = drain()  # every step's edge records have reached rank 0 before the clock stops
    torch.cuda.synchronize(dev)  # ... and its pinned host buffer
    if multi_path:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if multi_path:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kavg = ksum / max(args.steps, 1)  # per step, summed over the step's sub-batch launches
    order_avg = order_ms[0] / max(args.steps, 1)
    st = stats.as_dict()
    launches = int(nsub.value)
    # how the library ran the finder (sigax_batch_run_info): launches per sub-batch, two-step lines or granules, cooperative
    ri = rinfo.as_dict()
    nsub_step = max(1, ri["n_sub"])

    # upload_inclusive: the same steps with every step's reads going up from pinned host memory through sigax_batch_upload
    # (the PCIe-inclusive rate of the device path; the headline keeps the reads resident, as the metric's contract says)
    upl = None
    if world == 1 and not multi_path and args.upload_steps > 0:
        h_seqs = torch.from_numpy(np.ascontiguousarray(shard).reshape(-1)).pin_memory()
        h_offs = np.arange(0, (n_local + 1) * L, L, dtype=np.uint64)
        k0 = depth + args.warmup + args.steps
        run_steps(depth, k0, upload=True)  # the batch objects allocate their own read buffers here
        torch.cuda.synchronize(dev)
        t_u = time.perf_counter()
        run_steps(args.upload_steps, k0 + depth, upload=True)
        torch.cuda.synchronize(dev)
        t_u = time.perf_counter() - t_u
        upl = {"value": n_local * args.upload_steps / t_u, "unit": "reads/s", "steps": args.upload_steps, "ms_per_step": t_u / args.upload_steps * 1e3,
               "bytes_up_per_step": int(n_local * L + 8 * (n_local + 1)),
               "what": "the timed step with its reads uploaded from pinned host memory (sigax_batch_upload) inside the step, "
                       "edge records back to pinned host memory as in the headline"}
        for bt in batches:  # back to the resident reads for the legs below
            assert lib.sigax_batch_set_device_reads(bt, d_seqs.data_ptr(), d_offs.data_ptr(), n_local, n_local * L, L) == 0

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
        done_per_step = n_total if (job_world == world and not args.max_local_reads) else n_local * world
        reads_per_s = done_per_step * args.steps / elapsed
        step_s = elapsed / args.steps
        two_step = ri["two_step"]
        cand_rec = 64 if info["wide"] else 32
        # Algorithmic bytes of THIS formulation: the distinct 64-byte sectors of the rank tables each step /
        # round asks for, counted by the kernels themselves (a 128-byte line of the two-step table = 2 sectors), plus
        # the reads, the candidate records and the final records as they are written and read once.
        sec_f, sec_x = st["n_sectors_find"], st["n_sectors_extract"]
        bytes_find = 64 * sec_f + n_local * L + cand_rec * st["n_candidate_blocks"] + 16 * n_local
        bytes_fx = 64 * sec_x + cand_rec * st["n_candidate_blocks"] + 16 * n_local + 48 * st["n_blocks"] + 16 * 2 * n_local
        bytes_order = (48 + cand_rec + 80) * st["n_blocks"] + 20 * 2 * n_local
        bytes_edges = 2 * (80 * st["n_blocks"] + 4 * st["n_edges"] + 12 * st["n_blocks"]) + 16 * st["n_edges"]
        bytes_step = bytes_find + bytes_fx + bytes_order + bytes_edges
        find_ms = float(kavg[0]) / launches          # average duration of one k_find launch in the timed region
        fx_ms = float(kavg[1]) / nsub_step           # ... of one sub-batch's 32-lane + 64-lane launch pair
        ach_find = bytes_find / launches / (find_ms * 1e-3) / 1e9 if find_ms > 0 else 0.0
        ach_fx = bytes_fx / nsub_step / (fx_ms * 1e-3) / 1e9 if fx_ms > 0 else 0.0
        lines_find = sec_f / 2 if two_step else sec_f  # memory requests: 128-byte lines / 64-byte granules
        glines = lines_find / launches / (find_ms * 1e-3) / 1e9 if find_ms > 0 else 0.0
        key = "%d/%d/%d/%d" % (n_local, G, L, launches) + ("/key" if by_key else "")  # (a key-range shard misses the caches less)
        traffic = traffic_entry("k_find/" + key)
        fx_traffic = traffic_entry("k_filter_extract_fast/" + key)
        # What bounds the finder's gathers: the table one launch gathers from (one strand's two-step lines, or both strands'
        # granules) either stays in the 256 MiB Infinity Cache -- then the PMC "traffic" is L2-miss traffic and the ceiling
        # is what random rows of a cache-resident table read at -- or it does not, and the ceiling is HBM.
        per_strand = (info["n_symbols"] // 64 + 1) * 128 if two_step else (info["n_symbols"] // 128 + 1) * 64
        table_bytes = per_strand * (1 if ri["find_per_sub"] == 2 else 2)
        # `frac` is always against the HBM spec peak.  (Round 3 priced a 302 MB table against the guide's 7.65 TB/s for random
        # rows of a cache-resident table; the table is larger than the 256 MiB cache and tools/ic_fit.sh shows the finder does
        # not gain when it fits, so that ceiling was unsupported.)  The cache-rows figure is printed as a secondary field only
        # when the table really fits.
        cache_resident = table_bytes <= INFINITY_CACHE_BYTES
        bound = "hbm"
        peak = HBM_PEAK_GBS
        if job_world == 8 and args.reads_per_gpu == 2500000 and args.genome_per_gpu == 12500000 and L == 150 and args.seed == 2:
            named = "BASELINE configs[2]: "
        elif job_world == 1 and n_total == 1000000 and G == 5000000 and L == 150 and args.seed == 1:
            named = "BASELINE configs[1]: "
        elif job_world == 8 and args.reads_per_gpu == 6250000 and args.genome_per_gpu == 28750000 and L == 250 and args.seed == 3:
            named = "BASELINE configs[4] (chr1 stand-in: 50M x 250 bp from 230 Mb, 1.255e10 symbols, 64-bit positions)%s: " % (
                ", first %d reads of the rank's 6.25 M per step" % args.max_local_reads if args.max_local_reads else "")
        else:
            named = ""
        out = {
            "metric": "reads/sec overlapped (ASQG bit-exact)", "value": reads_per_s, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": named + "synthetic %dx%d bp reads from %d bp genome, seed %d, min-overlap %d, irreducible, both strands; "
                                   "FM-index (both strands) resident in HBM, reads sharded over %d GPU(s)%s; every step hands "
                                   "the batch its reads anew (what a product batch pays per upload%s is inside the timed region) and "
                                   "ends with the edge records in pinned host memory on rank 0" % (
                                       n_total, L, G, args.seed, args.min_overlap, job_world,
                                       (" -- handed over in locality-key order, ids kept (--key-order)" if by_key and job_world == 1 else " by locality-key range (each rank a contiguous slice of the reads' minimizer-key order, ids kept)" if by_key else " by file position" if job_world > 1 else "") +
                                       (" (this process: rank 0's shard only)" if job_world != world else ""),
                                       ", the locality ordering included," if ri["read_order"] else ""),
                       "sharding": ({"by": "locality key (siga_amd.sharding.locality_keys)", "keys_and_order_seconds": shard_keys_s,
                                     "note": "computed from the sequences alone, once per read set, outside the step"} if by_key
                                    else {"by": "file position"} if job_world > 1 else None),
                       "reads_per_gpu": n_local, "genome_bp": G, "read_len": L, "kernels_sha": kernels_sha(), "edges": total_edges, "blocks_per_read": st["n_blocks"] / max(n_local, 1),
                       "n_occ_min_per_read": (st["n_occ_find"] + st["n_occ_extract"]) / max(n_local, 1),
                       "sectors_per_read": {"find": sec_f / max(n_local, 1), "extract": sec_x / max(n_local, 1)},
                       "algorithmic_bytes_per_read": bytes_step / max(n_local, 1),
                       "reference_formulation_bytes_per_read": (64 * (st["n_occ_find"] + st["n_occ_extract"]) + n_local * L + 64 * st["n_blocks"]) / max(n_local, 1),
                       "slow_path_reads": st["n_slow_reads"], "batches_in_flight": depth, "two_step_table": two_step,
                       "finder": ("cooperative (lines through LDS)" if ri["coop"] else "per lane") + (", locality order" if ri["read_order"] else "") +
                                 (", chains start %d symbols in (deep start table)" % ri["deep_k"] if ri["deep_k"] else ", chains start 12 symbols in"),
                       "order_in_timed_region": not args.reuse_order,
                       "candidate_slots_per_chain": ri["cap"], "worst_case_slots_per_chain": ri["worst_cap"],
                       "candidate_arena_bytes": ri["arena_bytes"], "batch_workspace_bytes": ri["workspace_bytes"], "reruns": ri["reruns"],
                       "row_table": {"bits_per_row": ri["row_bits"], "symbols_per_entry": ri["row_syms"], "text": bool(ri["row_text"]),
                                     "direct_maps": bool(ri["row_direct"])},
                       "index_device_bytes": info["device_bytes"]},
            "kernel_ms_per_step": dict({k: float(v) for k, v in zip(KERNELS, kavg)}, order_reads=order_avg),
            "launches_per_step": launches,
            "roofline": {"bound": bound, "kernel": "k_find", "achieved": ach_find, "peak": peak, "unit": "GB/s",
                         "frac": ach_find / peak, "traffic": traffic,
                         "frac_of_cache_rows_ceiling": (ach_find / CACHE_ROWS_GBS) if cache_resident else None,
                         "table_bytes_per_launch": table_bytes,
                         "algorithmic_bytes_per_launch": bytes_find / launches, "avg_launch_ms": find_ms,
                         "request_rate": {"achieved": glines, "ceiling": GATHER_CEILING_GLINES, "unit": "G lines/s",
                                          "frac": glines / GATHER_CEILING_GLINES,
                                          "note": "random line requests per second against what dependency-free random "
                                                  "reads sustain on this chip (tools/gather_probe*.hip)"},
                         "note": ("launch durations in the timed region, where a k_find launch runs beside the filter/extract "
                                  "kernels of the previous sub-batch or batch" if (nsub_step > 1 or depth > 1)
                                  else "kernels run back to back")},
            "roofline_filter_extract": {"bound": "hbm", "kernel": "k_filter_extract_fast launch chain (k_fx_route + strict 32 + branching 32 + branching 64 + full 64)", "achieved": ach_fx,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_fx / HBM_PEAK_GBS, "traffic": fx_traffic,
                                        "algorithmic_bytes_per_launch": bytes_fx / nsub_step, "avg_launch_ms": fx_ms,
                                        "note": "latency-bound: a chain of dependent lookups per (read, side)"},
            "whole_path": {"achieved": bytes_step / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_step / step_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": bytes_step,
                           "traffic_find_plus_filter_extract": (launches * traffic + nsub_step * fx_traffic) if (traffic and fx_traffic) else None,
                           "note": "algorithmic bytes count filter/extract's lines per lane; L2 absorbs most of those (compare traffic)"},
        }
        if iso is not None:
            ims = float(iso[0])
            out["roofline"]["isolated"] = {
                "what": "same kernels, sub-batching off (one launch per step, nothing beside it), 2 untimed steps",
                "k_find_ms": ims, "achieved": bytes_find / (ims * 1e-3) / 1e9, "frac": bytes_find / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "request_rate_glines": lines_find / (ims * 1e-3) / 1e9,
                "kernel_ms_per_step": {k: float(v) for k, v in zip(KERNELS, iso)}}
        if upl is not None:
            out["upload_inclusive"] = upl
        if multi_path:
            # what RCCL saw (the driver's SCALE record can be checked against it)
            out["config"]["ranks"] = {"backend": args.backend, "world_size": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
                                      "device_of_rank0": dev_index,
                                      "edge_gather": "sigax_gather_counts + sigax_gather_edges (C-ABI; ncclAllGather + grouped ncclSend/ncclRecv)"
                                                     if comm is not None else "torch.distributed all_gather + gather (%s)" % args.backend}
        if world == 1 and not multi_path and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(prefix, reads, min(args.cpu_sample, n_total), args.min_overlap, st, lib, batch)

    for bt in batches:
        lib.sigax_batch_destroy(bt)
    if comm is not None:
        torch.cuda.synchronize(dev)
        lib.sigax_comm_destroy(comm)
    pair.close()
    if rank == 0 and world == 1 and not multi_path and job_world == 1 and not args.no_e2e and not args.error_rate and not by_pos and n_total <= 2000000:
        if reads is None:
            reads = draw(None)
        out["end_to_end"] = end_to_end_cli(workdir, reads, args.min_overlap, out.get("cpu_baseline"),
                                           named.startswith("BASELINE configs[1]") and args.min_overlap == 45)
    if multi_path:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


E2E_MD5_CONFIGS1 = "f47c833bc3fd5cc5f91daace91c5a345"  # ASQG text of BASELINE configs[1] at -m 45: the oracle's, rounds 1-3's


def end_to_end_cli(workdir, reads, min_overlap, cpu, is_configs1):
    """The product end to end ("incl. parse + ASQG write"): the workload's reads as FASTA (written here,
    outside the clock) through `siga overlap -m M` -- the host C++ of siga_amd/host over the C-ABI -- as a child process,
    one GPU: FMIndex::load x2 + .sai x2, parse, device batches, VT/ED text, gzip (src/overlap.cpp:41-47).  Wall time of the
    child, its own phase times (SIGA_TIMING), md5 of the decompressed ASQG."""
    import gzip
    import hashlib
    import re
    from siga_amd import host
    n = len(reads)
    fa = os.path.join(workdir, "reads.fa")
    with open(fa, "wb") as f:
        for lo in range(0, n, 100000):
            f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in zip(range(lo, min(n, lo + 100000)), reads[lo:lo + 100000])))
    out = os.path.join(workdir, "reads.asqg.gz")
    if os.path.exists(out):
        os.remove(out)
    env = dict(os.environ, SIGA_TIMING="1")
    cmd = [host.CLI_PATH, "overlap", "-m", str(min_overlap), "-t", "8", "reads.fa"]
    best = None
    for _ in range(2):  # the first run pays the page-in of the binary and its libraries on a fresh box; report the second
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=workdir, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "siga overlap exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        first = best is None
        best = (dt, r.stderr)
        if first:
            first_dt = dt
    dt, err = best
    phases = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[siga\]\s+(.*?)\s+([0-9.]+) s", err)}
    h = hashlib.md5()
    n_text = 0
    with gzip.open(out, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
            n_text += len(b)
    md5 = h.hexdigest()
    res = {"command": "siga overlap -m %d -t 8 reads.fa  (child process, 1 GPU, %d reads as FASTA)" % (min_overlap, n),
           "seconds": dt, "first_run_seconds": first_dt, "value": n / dt, "unit": "reads/s", "phases_s": phases,
           "asqg_gz_bytes": os.path.getsize(out), "asqg_text_bytes": n_text, "asqg_md5": md5,
           "asqg_md5_expected": E2E_MD5_CONFIGS1 if is_configs1 else None,
           "asqg_md5_ok": (md5 == E2E_MD5_CONFIGS1) if is_configs1 else None}
    if cpu:
        res["vs_cpu_overlap_only"] = {"ratio": (n / dt) / cpu["value"],
                                      "note": "end-to-end GPU reads/s (parse, index load, text and gzip included) over the CPU restatement's "
                                              "OverlapBuilder::overlap-only reads/s on %d threads (no I/O on its side)" % cpu["cores"]}
    return res


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


def correct_traffic(N, G, L, k):
    return traffic_entry("k_correct/%d/%d/%d/%d" % (N, G, L, k))


def bench_correct(args):
    """BASELINE configs[3]: the `siga correct` k-mer path (KmerCorrector::process, src/correct_processor.cpp:72-229) on
    configs[1]-shaped reads with substitution errors, one GPU.  A step = k_correct over all reads, everything resident."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the correction path has no CPU fallback")
    from siga_amd import _lib, host
    from siga_amd import build as sbuild
    from siga_amd.overlap import FMIndexPair
    from tests.golden.make_reads import fast_reads, substitute
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, G, L, k = args.reads_per_gpu, args.genome_per_gpu, args.read_len, args.kmer
    t0 = time.time()
    clean, _ = fast_reads(G, L, N, args.seed)
    err_rate = 0.01 if args.error_rate is None else args.error_rate
    reads = substitute(clean, err_rate, args.seed + 100)
    workdir = args.workdir or os.path.join(tempfile.gettempdir(), "siga_bench_correct_%d_%d_%d_%d" % (N, G, L, args.seed))
    os.makedirs(workdir, exist_ok=True)
    prefix = os.path.join(workdir, "reads")
    sbuild.build_all()
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    if not all(os.path.exists(prefix + e) for e in (".bwt", ".rbwt", ".sai", ".rsai")):
        host.index_build_gpu(reads.reshape(-1), offs, prefix)
    pair = FMIndexPair.load(prefix, device=0, with_sai=False)
    log("reads + index ready in %.1f s" % (time.time() - t0))
    lib = _lib.lib()
    d_seqs = torch.from_numpy(reads.reshape(-1).copy()).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_out = torch.empty_like(d_seqs)
    d_valid = torch.empty(N + 16, dtype=torch.uint8, device=dev)
    d_stat = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        rc = lib.sigax_correct_device(pair.handle, d_seqs.data_ptr(), None, d_offs.data_ptr(), N, k, 3, 10, 1, d_out.data_ptr(),
                                      d_valid.data_ptr(), d_stat.data_ptr(), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise SystemExit("sigax_correct_device: " + _lib.last_error())

    steps, warm = args.steps, args.warmup
    for _ in range(warm):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    e0.record(stream)
    for _ in range(steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    kernel_ms = e0.elapsed_time(e1) / steps  # memset + kernel on the stream the kernel is launched on
    stat = d_stat.cpu().numpy()
    valid = d_valid[:N].cpu().numpy()
    fixed = d_out.cpu().numpy().reshape(N, L)
    n_valid = int((valid == 1).sum())
    restored = int(((fixed == clean).all(axis=1) & (valid == 1)).sum())
    bytes_step = 64 * int(stat[1]) + 2 * N * L + N
    out = {
        "metric": "reads/sec corrected (siga correct k-mer path, output bit-exact)", "value": N * steps / elapsed, "unit": "reads/s",
        "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] (`siga correct` k-mer path; configs[1]-shaped reads with substitutions stand in for the E. coli "
                               "MiSeq set): siga correct -k %d -x 3 -i 10 -O 1 on synthetic %dx%d bp reads from %d bp genome with %.2g "
                               "substitutions per base; FM-index of those reads resident in HBM" % (k, N, L, G, err_rate),
                   "kernels_sha": kernels_sha(),
                   "reads_written": n_valid, "reads_equal_to_truth": restored, "kmer_lookups_per_read": int(stat[2]) / N,
                   "sectors_per_read": int(stat[1]) / N, "algorithmic_bytes_per_read": bytes_step / N},
        "roofline": None,
    }
    # k_correct's rank lines are asked for again and again (a 5 Mb genome at 30x holds ~10 M distinct k-mers; the index is
    # 0.4 GB): three quarters of its requests hit L2, so bytes REQUESTED over time says nothing about HBM (it exceeds the
    # peak).  `achieved` is therefore what went past L2 (PMC: TCC_EA0_RDREQ by size class, profiles/traffic.json, per
    # launch) over the launch time; the requested bytes are printed beside it.
    traffic = correct_traffic(N, G, L, k)
    moved = traffic if traffic else None
    out["roofline"] = {"bound": "hbm", "kernel": "k_correct",
                       "achieved": (moved / (kernel_ms * 1e-3) / 1e9) if moved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": (moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if moved else None, "traffic": traffic,
                       "requested_bytes_per_launch": bytes_step, "requested_over_time_GBs": bytes_step / (kernel_ms * 1e-3) / 1e9,
                       "avg_launch_ms": kernel_ms,
                       "note": "latency kernel on cache-resident tables: `achieved` = bytes past L2 (PMC) / launch time; "
                               "`requested_*` = 64 B x the distinct rank-table sectors the k-mer lookups ask for, mostly L2 hits"}
    if args.cpu_sample > 0:
        from oracle import pyoracle as po
        po.build()
        sample = min(args.cpu_sample, N, 50000)  # one thread, about 2.5 k reads/s: 20 s
        fa = os.path.join(workdir, "sample.fa")
        with open(fa, "wb") as f:
            f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads[:sample])))
        fwd = po.Index.load(prefix + ".bwt", "")
        t1 = time.time()
        st = po.correct(fwd, fa, os.path.join(workdir, "sample.out.fa"), k, 3, 10, 1)
        sec = time.time() - t1
        want = open(os.path.join(workdir, "sample.out.fa"), "rb").read()
        got = b"".join(b">r%d\n%s\n" % (i, bytes(fixed[i])) for i in range(sample) if valid[i] == 1)
        out["cpu_baseline"] = {"value": sample / sec, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads through the oracle's CorrectProcessor (file in, file out, one thread) "
                                         "against the full index" % sample,
                               "seconds": sec, "reads_written": st["written"], "gpu_output_identical": got == want}
    pair.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
