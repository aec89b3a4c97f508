"""Diagnostic: which path do the extension rounds of k_filter_extract_fast take at BASELINE configs[1]?
Run with SIGAX_LIB=build/libsigax_prof.so (a build with -DSIGAX_FX_PROFILE)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import _lib, host  # noqa: E402
from siga_amd.overlap import FMIndexPair  # noqa: E402
from tests.golden.make_reads import fast_reads, rank_of_r_names, substitute  # noqa: E402

N, G, L = int(os.environ.get("FXP_N", 1000000)), int(os.environ.get("FXP_G", 5000000)), int(os.environ.get("FXP_L", 150))
reads, _ = fast_reads(G, L, N, 1)
if float(os.environ.get("FXP_ERR", 0)) > 0:
    reads = substitute(reads, float(os.environ["FXP_ERR"]), 101)
d = tempfile.mkdtemp()
prefix = os.path.join(d, "reads")
offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
host.index_build_gpu(reads.reshape(-1), offs, prefix)
pair = FMIndexPair.load(prefix)
pair.set_reads(np.full(N, L, dtype=np.uint32), rank_of_r_names(N))
lib = _lib.lib()
bt = C.c_void_p()
assert lib.sigax_batch_create(pair.handle, N, N * L, L, C.byref(bt)) == 0
buf = reads.reshape(-1).tobytes()
assert lib.sigax_batch_upload(bt, buf, offs.ctypes.data, N, None) == 0
flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
st = _lib.Stats()
assert lib.sigax_batch_run(bt, 0, 45, flags, None) == 0
assert lib.sigax_batch_finish(bt, None, C.byref(st)) == 0
out = (C.c_uint64 * 32)()
lib.sigax_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 32)]
assert lib.sigax_debug_counters(bt, C.byref(out)) == 0
names = ["rounds", "two-step ok", "one-step single-symbol", "one-step counted/ended", "generic: crosses granule", "generic: branch/$",
         "items", "items <=16 blocks", "two-step tried", "two-step: crosses line", "two rounds at once", "items with intersecting blocks",
         "rounds: blocks in 1 line", "rounds: blocks in 2 adjacent lines", "rounds: blocks in more lines"]
for w, base in (("W=32", 0), ("W=64", 16)):
    print(w, {n: int(out[base + i]) for i, n in enumerate(names[:(15 if base == 0 else 8)])})
print("W=32 by round 0,1,2,3+: single-line rounds", [int(out[24 + i]) for i in range(4)], "of", [int(out[28 + i]) for i in range(4)])
print(st.as_dict())
