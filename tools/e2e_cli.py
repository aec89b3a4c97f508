"""End to end through the CLI at BASELINE configs[1] (1M x 150 bp reads as FASTA): `siga index`, `siga overlap -m 45`, and
the CPU figure beside it: the oracle's `siga overlap -t <all cores>` restatement (parse, OpenMP overlap, serial VT/ED text)
on the same files.  gpurun -- python tools/e2e_cli.py [N] [gpus]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import host  # noqa: E402
from tests.golden.make_reads import fast_reads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
GPUS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
G, L = 5 * N, 150
d = tempfile.mkdtemp()
reads, _ = fast_reads(G, L, N, 1)
fa = os.path.join(d, "reads.fa")
with open(fa, "wb") as f:
    f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
env = dict(os.environ, SIGA_TIMING="1")


def run(args):
    t0 = time.time()
    r = subprocess.run([host.CLI_PATH] + args, cwd=d, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    print("siga %s: rc %d, %.2f s" % (" ".join(args), r.returncode, dt))
    sys.stdout.write(r.stderr)
    return dt


t_index = run(["index", "-t", "64", "reads.fa"])
t_overlap = run(["overlap", "-m", "45", "-t", "8", "reads.fa"])
for extra in filter(None, os.environ.get("E2E_AGAIN", "").split(";")):  # the same run with other settings: "A=1,B=2;C=3"
    os.rename(os.path.join(d, "reads.asqg.gz"), os.path.join(d, "first.asqg.gz"))
    keep = dict(env)
    env.update(kv.split("=", 1) for kv in extra.split(","))
    print("again with", extra)
    run(["overlap", "-m", "45", "-t", "8", "reads.fa"])
    env.clear()
    env.update(keep)
    a = subprocess.run(["gzip", "-dc", os.path.join(d, "reads.asqg.gz")], capture_output=True).stdout
    b = subprocess.run(["gzip", "-dc", os.path.join(d, "first.asqg.gz")], capture_output=True).stdout
    print("same ASQG text as the first run:", a == b)
    del a, b
    os.remove(os.path.join(d, "reads.asqg.gz"))
    os.rename(os.path.join(d, "first.asqg.gz"), os.path.join(d, "reads.asqg.gz"))
if GPUS > 1:
    os.rename(os.path.join(d, "reads.asqg.gz"), os.path.join(d, "one.asqg.gz"))
    env["SIGA_DEVICE_MAP"] = ",".join(["0"] * GPUS) if os.environ.get("E2E_REHEARSE") else ""
    t_multi = run(["overlap", "-m", "45", "-t", "8", "--gpus", str(GPUS), "reads.fa"])
    same = open(os.path.join(d, "reads.asqg.gz"), "rb").read() == open(os.path.join(d, "one.asqg.gz"), "rb").read()
    print("--gpus %d: %.2f s, .asqg.gz identical to the 1-GPU file: %s" % (GPUS, t_multi, same))
gz = os.path.join(d, "reads.asqg.gz")
t0 = time.time()
text = subprocess.run(["gzip", "-dc", gz], capture_output=True).stdout
print("asqg.gz %d bytes -> %d bytes of text, %d ED lines, md5 %s (gunzip %.1f s)" % (
    os.path.getsize(gz), len(text), text.count(b"\nED\t"), hashlib.md5(text).hexdigest(), time.time() - t0))
print("GPU end to end: index %.2f s + overlap %.2f s -> %.2f M reads/s through `siga overlap`" % (t_index, t_overlap, N / t_overlap / 1e6))
if os.environ.get("E2E_CPU", "1") != "0":
    from oracle import pyoracle as po
    po.build()
    t0 = time.time()
    fwd = po.Index.load(os.path.join(d, "reads.bwt"), os.path.join(d, "reads.sai"))
    rev = po.Index.load(os.path.join(d, "reads.rbwt"), os.path.join(d, "reads.rsai"))
    t_load = time.time() - t0
    secs = po.build_asqg_mt(fwd, rev, fa, 45, os.path.join(d, "cpu.asqg"))
    total = t_load + sum(secs.values())
    same = open(os.path.join(d, "cpu.asqg"), "rb").read() == text
    print("CPU (%d threads) end to end: index load %.2f s, parse %.2f s, overlap %.2f s, VT+ED text %.2f s = %.2f s -> %.3f M reads/s; "
          "ASQG text identical to the GPU's: %s" % (po.max_threads(), t_load, secs["parse"], secs["overlap"], secs["text"], total,
                                                    N / total / 1e6, same))
    print("speed-up of `siga overlap` end to end: %.1fx" % (total / t_overlap))
