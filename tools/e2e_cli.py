"""End-to-end `siga index` + `siga overlap` on a C2-sized FASTA through the CLI (host parse, GPU path, gz ASQG)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import host, build
from tests.golden.make_reads import fast_reads
build.build_all()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
wd = "/tmp/siga_e2e_%d" % N
os.makedirs(wd, exist_ok=True)
reads, _ = fast_reads(N * 5, 150, N, 1)
with open(wd + "/reads.fa", "wb") as f:
    for i in range(N):
        f.write(b">r%d\n" % i + bytes(reads[i]) + b"\n")
for cmd in (["index", "-t", "64", "reads.fa"], ["overlap", "-m", "45", "-t", "8", "reads.fa"]):
    t = time.time()
    r = subprocess.run([host.CLI_PATH] + cmd, cwd=wd)
    print(" ".join(cmd), "rc", r.returncode, "%.2f s" % (time.time() - t), flush=True)
print("asqg.gz bytes", os.path.getsize(wd + "/reads.asqg.gz"))
import gzip, hashlib
t = time.time(); data = gzip.open(wd + "/reads.asqg.gz", "rb").read(); print("gunzip ok: %d bytes, %d ED lines, md5 %s (%.1f s)" % (len(data), data.count(b"\nED\t"), hashlib.md5(data).hexdigest(), time.time() - t))
r = subprocess.run(["gzip", "-t", wd + "/reads.asqg.gz"]); print("gzip -t rc", r.returncode)
