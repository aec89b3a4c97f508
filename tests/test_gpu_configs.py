"""The BASELINE.json configurations other than configs[1], on the GPU (configs[1] at full size: test_gpu_fullsize.py,
configs[3] `siga correct`: test_gpu_correct_scale.py).

C1  E. coli MiSeq shape: `siga index` -> `siga overlap -m 85` through the CLI on 100 k reads from a 4.6 Mb genome
    (real MiSeq data is not available offline), ASQG bytes == the oracle's.
C3  20 M x 150 bp from a 100 Mb genome: the 8-GPU job's index (3.02e9 symbols, u32 positions, beyond the two-step
    table's range), one rank's shard of reads on this GPU.
C5  50 M x 250 bp from a 230 Mb genome (the stand-in for chr1), seed 3: the FULL configuration -- index
    of 1.255e10 symbols per strand with 64-bit positions (WIDE kernels, superblock counters), built on the GPU, 190 GB of
    tables on the device (rank granules, two-step lines, bare 34-bit row-table entries + text: the entries with symbols do
    not fit), rank 0's shard of 8 = 6.25 M reads through sigax_overlap_batch in pieces.
For C3/C5 the oracle checks a random sample of the shard read by read (it runs from the same index files), and the
shard is checked whole through size-independent properties: idempotence, invariance under re-sharding (read_base),
self-containment blocks, dedup rule and order of the edge records, N_occ_min additivity."""
import gzip
import os
import subprocess
import time

import numpy as np
import pytest

from tests.fixtures import CACHE

pytestmark = pytest.mark.gpu


def _name_ranks(n):
    from tests.golden.make_reads import rank_of_r_names
    return rank_of_r_names(n)


def test_c1_ecoli_shape_cli_index_overlap_m85(tmp_path):
    from oracle import pyoracle as po
    from siga_amd import host
    from tests.golden.make_reads import fast_reads
    N, G, L, M = 100000, 4600000, 150, 85
    reads, _ = fast_reads(G, L, N, 11)
    cwd = str(tmp_path)
    fa = os.path.join(cwd, "ecoli.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
    assert subprocess.run([host.CLI_PATH, "index", "-t", "8", "ecoli.fa"], cwd=cwd).returncode == 0
    assert subprocess.run([host.CLI_PATH, "overlap", "-m", str(M), "-t", "8", "ecoli.fa"], cwd=cwd).returncode == 0
    got = gzip.open(os.path.join(cwd, "ecoli.asqg.gz"), "rb").read()
    fwd = po.Index.load(os.path.join(cwd, "ecoli.bwt"), os.path.join(cwd, "ecoli.sai"))
    rev = po.Index.load(os.path.join(cwd, "ecoli.rbwt"), os.path.join(cwd, "ecoli.rsai"))
    # the GPU-built index files are the host SA-IS's files
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "-t", "8", "-p", "cpu", "ecoli.fa"], cwd=cwd).returncode == 0
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        assert open(os.path.join(cwd, "ecoli" + ext), "rb").read() == open(os.path.join(cwd, "cpu" + ext), "rb").read(), ext
    po.build_asqg(fwd, rev, fa, M, os.path.join(cwd, "oracle.asqg"))
    want = open(os.path.join(cwd, "oracle.asqg"), "rb").read()
    assert got == want
    assert want.count(b"\nED\t") > 10000


def _big_case(tag, N, G, L, seed, world, sample, M=45):
    """Build (GPU) and load the index of an N-read job, run rank 0's shard of `world`, check it."""
    import siga_amd
    from oracle import pyoracle as po
    from siga_amd import host
    from tests.bigcheck import assert_same_blocks, blocks_matrix, edges_matrix, expected_edges
    from tests.golden.make_reads import fast_reads
    from siga_amd.sharding import shard_range
    t0 = time.time()
    reads, _ = fast_reads(G, L, N, seed)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    d = os.path.join(CACHE, tag)
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "reads")
    t1 = time.time()
    host.index_build_gpu(reads.reshape(-1), offs, prefix)
    t2 = time.time()
    pair = siga_amd.FMIndexPair.load(prefix)
    t3 = time.time()
    info = pair.info()
    rank = _name_ranks(N)
    pair.set_reads(np.full(N, L, dtype=np.uint32), rank)
    lo, hi = shard_range(N, 0, world)
    n = hi - lo
    b = siga_amd.OverlapBuilder(pair)

    def run(a, z):
        return b.overlap((reads[a:z].reshape(-1), np.arange(0, (z - a + 1) * L, L, dtype=np.uint64)), M, read_base=a, edges=True)

    full = run(lo, hi)
    t4 = time.time()
    again = run(lo, hi)
    for k in ("block_offs", "blocks", "substring", "edges"):
        assert full[k].tobytes() == again[k].tobytes(), k
    cuts = [lo, lo + n // 3 + 1, lo + (2 * n) // 3 - 5, hi]
    parts = [run(cuts[i], cuts[i + 1]) for i in range(3)]
    assert np.concatenate([p["blocks"] for p in parts]).tobytes() == full["blocks"].tobytes()
    assert np.concatenate([p["edges"] for p in parts]).tobytes() == full["edges"].tobytes()
    assert sum(p["stats"]["n_occ_find"] + p["stats"]["n_occ_extract"] for p in parts) == \
        full["stats"]["n_occ_find"] + full["stats"]["n_occ_extract"]
    boffs = full["block_offs"].astype(np.int64)
    blk = full["blocks"]
    assert boffs[0] == 0 and boffs[-1] == len(blk) and np.all(np.diff(boffs) >= 0)
    owner = np.repeat(np.arange(n), np.diff(boffs))
    for af in (0, 3):  # every read contains itself in both indexes
        sel = (blk["length"] == L) & (blk["af"] == af)
        assert sel.sum() == n and np.array_equal(owner[sel], np.arange(n))
    assert not full["substring"].any()
    rest = blk[blk["length"] != L]
    assert rest["length"].min() >= M and rest["length"].max() < L
    assert np.all(blk["capped0_hi"] >= blk["capped0_lo"]) and np.all(blk["capped0_hi"] < N)
    e = full["edges"]
    assert np.all(rank[e["query"]] > rank[e["target"]]) and np.all(np.diff(e["query"].astype(np.int64)) >= 0)
    assert e["query"].min() >= lo and e["query"].max() < hi
    t5 = time.time()
    # oracle: a random sample of the shard, read by read, against the same index files
    fwd = po.Index.load(prefix + ".bwt", prefix + ".sai")
    rev = po.Index.load(prefix + ".rbwt", prefix + ".rsai")
    t6 = time.time()
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(n, size=min(sample, n), replace=False))
    want = po.overlap_batch(fwd, rev, (reads[lo + pick].reshape(-1), np.arange(0, (len(pick) + 1) * L, L, dtype=np.uint64)), M)
    got10 = blocks_matrix(blk)
    for k, r in enumerate(pick):
        w = want["blocks"][int(want["block_offs"][k]):int(want["block_offs"][k + 1])]
        g = got10[boffs[r]:boffs[r + 1]]
        assert np.array_equal(g, w), (tag, int(lo + r))
    # edge records of the whole shard from the GPU's blocks by the numpy restatement of the converter
    exp = expected_edges(got10, full["block_offs"], fwd.sai(), rev.sai(), np.full(N, L, dtype=np.uint32), rank, read_base=lo)
    assert np.array_equal(edges_matrix(e), exp)
    # the index files this test and the oracle sample both read were written by the GPU suffix sorter: every pair of
    # adjacent BWT rows of both strands is in the order of record (checked on the device: sigax_index_check_order)
    for which in (0, 1):
        bad, first, und = pair.check_order(which)
        assert bad == 0 and und == 0, (tag, which, bad, first, und)
    t7 = time.time()
    s = full["stats"]
    print("%s: %d symbols/strand wide=%d device %.1f GB | reads %.0fs, index build %.1fs, load %.1fs, shard of %d reads %.1fs "
          "(%d blocks, %d edges, %d slow reads), invariants %.0fs, oracle load %.0fs, sample+edges %.0fs" % (
              tag, info["n_symbols"], info["wide"], info["device_bytes"] / 1e9, t1 - t0, t2 - t1, t3 - t2, n, t4 - t3,
              s["n_blocks"], s["n_edges"], s["n_slow_reads"], t5 - t4, t6 - t5, t7 - t6))
    pair.close()
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        os.remove(prefix + ext)
    return info


def test_c3_shape_index_3e9_symbols_one_shard():
    info = _big_case("c3", 20000000, 100000000, 150, 2, world=8, sample=100000)
    assert info["n_symbols"] == 20000000 * 151 and info["wide"] == 0


def test_c5_full_size_wide_index_one_shard_of_eight():
    N = 50000000
    info = _big_case("c5", N, 230000000, 250, 3, world=8, sample=50000)
    assert info["n_symbols"] == N * 251 >= 2**32 and info["wide"] == 1


def test_pipeline_on_reads_with_errors_correct_index_rmdup_overlap(tmp_path):
    """The reference's pipeline ORDER on reads with sequencing errors (examples/siga-ecoli-miseq.sh:64-85): `index --no-reverse`
    -> `correct -k 31` -> `index` -> `rmdup` (what the extractor's own error message prescribes before an overlap,
    src/overlap_builder.cpp:755-756) -> `index` -> `overlap -m 45`, every step through the CLI on the GPU, every file byte for
    byte the oracle's run of the same step on the same input.  C2-shaped reads (150 bp, 30x) with 1 % substitutions: after
    `correct` a few hundred wrong bases are left, so the overlap stage runs the branching extractor forms on real leftovers,
    not only on the small random sets of test_gpu_random.py."""
    from oracle import pyoracle as po
    from siga_amd import host
    from tests.golden.make_reads import fast_reads, substitute
    N, G, L = 60000, 300000, 150
    clean, _ = fast_reads(G, L, N, 31)
    reads = substitute(clean, 0.01, 131)
    cwd = str(tmp_path)
    with open(os.path.join(cwd, "reads.fa"), "wb") as f:
        f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))

    def cli(*args):
        r = subprocess.run([host.CLI_PATH] + list(args), cwd=cwd, capture_output=True, text=True)
        assert r.returncode == 0, " ".join(args) + ": " + r.stderr[-2000:]

    def same(a, b):
        x, y = open(os.path.join(cwd, a), "rb").read(), open(os.path.join(cwd, b), "rb").read()
        assert x == y, "%s differs from %s (%d vs %d bytes)" % (a, b, len(x), len(y))
        return len(x)

    def oracle_index(fa, prefix):
        seqs = [s for _, _, s in __import__("siga_amd").overlap.read_sequences(os.path.join(cwd, fa))]
        fwd, rev = po.Index.build(seqs), po.Index.build(seqs, reverse=True)
        fwd.save(os.path.join(cwd, prefix + ".bwt"), os.path.join(cwd, prefix + ".sai"))
        rev.save(os.path.join(cwd, prefix + ".rbwt"), os.path.join(cwd, prefix + ".rsai"))
        return fwd, rev

    # 1. index for the corrector (forward strand only), 2. correct
    cli("index", "-t", "8", "--no-reverse", "reads.fa")
    assert not os.path.exists(os.path.join(cwd, "reads.rbwt"))
    cli("correct", "-k", "31", "-t", "8", "reads.fa")
    st = po.correct(po.Index.load(os.path.join(cwd, "reads.bwt")), os.path.join(cwd, "reads.fa"), os.path.join(cwd, "o.ec.fa"), k=31)
    same("reads.ec.fa", "o.ec.fa")
    assert st["changed"] > N // 2 and st["written"] > N * 0.9
    # 3. index of the corrected reads == the oracle's builder
    cli("index", "-t", "8", "reads.ec.fa")
    ofwd, orev = oracle_index("reads.ec.fa", "o.ec")
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        same("reads.ec" + ext, "o.ec" + ext)
    # 4. rmdup
    cli("rmdup", "-t", "8", "reads.ec.fa")
    po.rmdup(ofwd, orev, os.path.join(cwd, "reads.ec.fa"), os.path.join(cwd, "o.rmdup.fa"), os.path.join(cwd, "o.rmdup.dups.fa"))
    same("reads.ec.rmdup.fa", "o.rmdup.fa")
    same("reads.ec.rmdup.dups.fa", "o.rmdup.dups.fa")
    # 5. index + overlap of what is left
    cli("index", "-t", "8", "reads.ec.rmdup.fa")
    cli("overlap", "-m", "45", "-t", "8", "reads.ec.rmdup.fa")
    fwd, rev = oracle_index("reads.ec.rmdup.fa", "o.fin")
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        same("reads.ec.rmdup" + ext, "o.fin" + ext)
    po.build_asqg_mt(fwd, rev, os.path.join(cwd, "reads.ec.rmdup.fa"), 45, os.path.join(cwd, "o.asqg"))
    got = gzip.open(os.path.join(cwd, "reads.ec.rmdup.asqg.gz"), "rb").read()
    want = open(os.path.join(cwd, "o.asqg"), "rb").read()
    assert got == want
    assert want.count(b"\nED\t") > N // 2
    # the leftovers made the extractor branch: the same reads through the library, counting
    import siga_amd
    pair = siga_amd.FMIndexPair.load(os.path.join(cwd, "reads.ec.rmdup"))
    seqs = [s for _, _, s in siga_amd.overlap.read_sequences(os.path.join(cwd, "reads.ec.rmdup.fa"))]
    res = siga_amd.OverlapBuilder(pair).overlap(seqs, 45)
    assert res["stats"]["n_blocks"] > N
    pair.close()
