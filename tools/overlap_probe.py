"""Do the find (memory-request-bound) and filter/extract (VALU-bound) kernels overlap when two batches run on two
HIP streams?  Times two half-size batches back to back on one stream vs concurrently on two streams."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import _lib, host
from siga_amd.overlap import FMIndexPair
from tests.golden.make_reads import fast_reads

N, G, L = 1000000, 5000000, 150
reads, _ = fast_reads(G, L, N, 1)
wd = "/tmp/siga_overlap_probe"; os.makedirs(wd, exist_ok=True); prefix = wd + "/reads"
if not os.path.exists(prefix + ".rsai"):
    host.index_build(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64), prefix, threads=2)
pair = FMIndexPair.load(prefix)
names = np.char.add("r", np.arange(N).astype(str)); order = np.argsort(names, kind="stable")
rank = np.empty(N, dtype=np.uint32); rank[order] = np.arange(N, dtype=np.uint32)
pair.set_reads(np.full(N, L, dtype=np.uint32), rank)
lib = _lib.lib(); dev = torch.device("cuda", 0)
flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cuts = [N * i // parts for i in range(parts + 1)]
batches, keep, streams = [], [], []
for i in range(parts):
    lo, hi = cuts[i], cuts[i + 1]
    d_seqs = torch.from_numpy(reads[lo:hi].reshape(-1).copy()).to(dev)
    d_offs = torch.arange(0, (hi - lo + 1) * L, L, dtype=torch.int64, device=dev)
    b = C.c_void_p(); assert lib.sigax_batch_create(pair.handle, hi - lo, (hi - lo) * L, L, C.byref(b)) == 0
    assert lib.sigax_batch_set_device_reads(b, d_seqs.data_ptr(), d_offs.data_ptr(), hi - lo, (hi - lo) * L, L) == 0
    batches.append((b, lo)); keep.append((d_seqs, d_offs)); streams.append(torch.cuda.Stream(dev))
st = _lib.Stats()
def run(concurrent):
    for i, (b, lo) in enumerate(batches):
        s = streams[i if concurrent else 0]
        assert lib.sigax_batch_run(b, lo, 45, flags, C.c_void_p(s.cuda_stream)) == 0, _lib.last_error()
    tot = 0
    for i, (b, lo) in enumerate(batches):
        s = streams[i if concurrent else 0]
        assert lib.sigax_batch_finish(b, C.c_void_p(s.cuda_stream), C.byref(st)) == 0, _lib.last_error()
        tot += st.n_edges
    return tot
for mode in (False, True, False, True):
    run(mode); torch.cuda.synchronize()
    t = time.perf_counter(); e = 0
    for _ in range(5): e = run(mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("parts %d concurrent=%s: %.2f ms per 1M reads (%.1f M reads/s), edges %d" % (parts, mode, dt * 1e3, N / dt / 1e6, e), flush=True)
