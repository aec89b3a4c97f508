"""`siga correct` and `siga rmdup` end to end through the CLI (BASELINE configs[3]'s shape: 1M x 150 bp reads with 1 %
substitutions for correct; the same reads plus 5 % exact duplicates for rmdup).  gpurun -- python tools/e2e_aux.py [N]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from siga_amd import host  # noqa: E402
from tests.golden.make_reads import fast_reads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = tempfile.mkdtemp()
env = dict(os.environ, SIGA_TIMING="1")


def run(args):
    t0 = time.time()
    r = subprocess.run([host.CLI_PATH] + args, cwd=d, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    print("siga %s: rc %d, %.2f s" % (" ".join(args), r.returncode, dt))
    sys.stdout.write("".join(l + "\n" for l in r.stderr.splitlines() if "batch" not in l))
    return dt


def fasta(path, reads, names=None):
    with open(path, "wb") as f:
        f.write(b"".join(b">%s\n%s\n" % (names[i] if names else b"r%d" % i, bytes(r)) for i, r in enumerate(reads)))


reads, _ = fast_reads(5 * N, 150, N, 1)
rng = np.random.default_rng(4)
noisy = reads.copy()
flips = rng.random(noisy.shape) < 0.01
noisy[flips] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(flips.sum()))]
fasta(os.path.join(d, "noisy.fa"), noisy)
run(["index", "-t", "64", "noisy.fa"])
t = run(["correct", "-t", "8", "-k", "31", "noisy.fa"])
lines = open(os.path.join(d, "noisy.ec.fa"), "rb").read().split(b"\n")
names, seqs = lines[0:-1:2], lines[1::2]
ids = np.array([int(nm.split()[0][2:]) for nm in names])  # ">r<i>": reads that fail the k-mer check are not written
keep = np.array([len(x) == 150 for x in seqs[:len(ids)]])
fixed = np.frombuffer(b"".join(x for x, k in zip(seqs, keep) if k), dtype=np.uint8).reshape(-1, 150)
ids = ids[keep]
print("siga correct: %.2f M reads/s end to end; %d of %d reads written; mismatches against the error-free reads, over the written ones: %d -> %d" % (
    N / t / 1e6, len(names), N, int((noisy[ids] != reads[ids]).sum()), int((fixed != reads[ids]).sum())))

dup = np.concatenate([reads, reads[rng.integers(0, N, size=N // 20)]])
fasta(os.path.join(d, "dup.fa"), dup)
run(["index", "-t", "64", "dup.fa"])
t = run(["rmdup", "-t", "8", "dup.fa"])
kept = open(os.path.join(d, "dup.rmdup.fa"), "rb").read().count(b">")
dups = open(os.path.join(d, "dup.rmdup.dups.fa"), "rb").read().count(b">")
# a read and its reverse complement are one sequence to rmdup (OverlapBuilder::duplicate looks on both strands)
comp = np.zeros(256, dtype=np.uint8)
comp[list(b"ACGT")] = list(b"TGCA")
rc = comp[dup[:, ::-1]]
first_diff = np.argmax(dup != rc, axis=1)
rows = np.arange(len(dup))
use_rc = (dup[rows, first_diff] > rc[rows, first_diff])
canon = np.where(use_rc[:, None], rc, dup)
print("siga rmdup: %.2f M reads/s end to end; %d reads in, %d kept, %d duplicates set aside (distinct sequences up to reverse complement: %d)" % (
    len(dup) / t / 1e6, len(dup), kept, dups, len(np.unique(canon.view("S150")))))
