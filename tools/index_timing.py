"""`siga index` phase by phase at a given read count (SIGAX_BUILD_TIMING=1, SIGA_TIMING=1), run twice on the same file.
gpurun -- python tools/index_timing.py [N]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import host  # noqa: E402
from tests.golden.make_reads import fast_reads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
d = tempfile.mkdtemp()
reads, _ = fast_reads(5 * N, 150, N, 1)
fa = os.path.join(d, "reads.fa")
with open(fa, "wb") as f:
    f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
del reads
env = dict(os.environ, SIGA_TIMING="1", SIGAX_BUILD_TIMING="1")
for k in range(2):
    t0 = time.time()
    r = subprocess.run([host.CLI_PATH, "index", "-t", "64", "reads.fa"], cwd=d, env=env, capture_output=True, text=True)
    print("run %d: rc %d, %.2f s" % (k, r.returncode, time.time() - t0))
    sys.stdout.write(r.stderr)
    sys.stdout.flush()
