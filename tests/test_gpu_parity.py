"""GPU parity tests proper: the HIP path (through the C-ABI, include/sigax.h) against the oracle on the same inputs.
Bit-exact: Occ values, per-read block lists IN ORDER (hits text), substring flags, ASQG text."""
import os

import numpy as np
import pytest

from tests.fixtures import GOLDEN, ed_lines, fixture, md5_prefix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import siga_amd
    return siga_amd


def _pair(sa, fx):
    return sa.FMIndexPair.load(fx.prefix)


@pytest.mark.parametrize("name", ["corner", "tiny", "dup", "toy"])
def test_occ_bit_exact(sa, name):
    fx = fixture(name)
    pair = _pair(sa, fx)
    n = len(fx.fwd)
    rng = np.random.default_rng(1)
    pos = np.concatenate([np.arange(min(n, 300), dtype=np.uint64), rng.integers(0, n, 2000, dtype=np.uint64),
                          np.array([n - 1, max(n - 2, 0), 2**64 - 1], dtype=np.uint64)])
    for which, orc in ((0, fx.fwd), (1, fx.rev)):
        got = pair.occ(pos, which)
        for k, p in enumerate(pos):
            assert list(got[k]) == list(orc.occ(int(p))), (name, which, int(p))
    assert pair.info()["pred"] == list(map(int, fx.fwd.pred()))


CASES = [
    ("corner", 10, True, True), ("corner", 10, False, True), ("corner", 30, True, True),
    ("rep", 10, True, True), ("rep", 10, False, True),
    ("dup", 8, True, True), ("dup", 8, False, True), ("dup", 8, True, False),
    ("tiny", 20, True, True), ("tiny", 20, False, True), ("tiny", 59, True, True), ("tiny", 1, True, True),
    ("ragged", 15, True, True), ("ragged", 15, False, False),
    ("toy", 45, True, True), ("toy", 45, False, True), ("toy", 45, True, False),
]


@pytest.mark.parametrize("name,m,irr,rc", CASES)
def test_hits_and_asqg_bit_exact(sa, name, m, irr, rc):
    from siga_amd.overlap import format_hits
    fx = fixture(name)
    pair = _pair(sa, fx)
    want_asqg, want_hits, st = fx.oracle_asqg(m, irreducible=irr, rc=rc, hits=True)
    builder = sa.OverlapBuilder(pair, fx.prefix, irreducible=irr, rc=rc)
    got_asqg, res = builder.build(fx.fa, m)
    got_hits = format_hits(res)
    if got_hits != want_hits:
        gl, wl = got_hits.split("\n"), want_hits.split("\n")
        bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
        raise AssertionError("hits differ for %d reads; first read %d:\n got  %s\n want %s" % (
            len(bad), bad[0], gl[bad[0]][:600], wl[bad[0]][:600]))
    assert got_asqg == want_asqg
    s = res["stats"]
    assert s["n_blocks"] == st["blocks"]
    assert s["n_occ_find"] + s["n_occ_extract"] == st["n_occ_min"]
    assert s["n_edges"] == len(ed_lines(want_asqg))


@pytest.mark.parametrize("irr", [True, False])
def test_deep_coverage_items_bit_exact(sa, irr):
    """160x coverage: a (read, side) item holds about 110 blocks, more than the 64 lanes of a wave -- the four-blocks-per-
    lane path of the 64-lane launch (irreducible mode) or the general kernel (exhaustive mode).  Same bytes as the oracle."""
    from siga_amd.overlap import format_hits
    fx = fixture("deep")
    pair = _pair(sa, fx)
    want_asqg, want_hits, st = fx.oracle_asqg(30, irreducible=irr, hits=True)
    got_asqg, res = sa.OverlapBuilder(pair, fx.prefix, irreducible=irr).build(fx.fa, 30)
    assert format_hits(res) == want_hits
    assert got_asqg == want_asqg
    s = res["stats"]
    assert s["n_occ_find"] + s["n_occ_extract"] == st["n_occ_min"]
    if irr:
        assert s["n_slow_reads"] < len(fx.reads) // 10, s  # the lane-group path took them, not the one-lane kernel


def test_non_acgt_reads_block_parity(sa):
    """Reads holding an N: per-read block lists must still equal the oracle's (no edges: see fixtures.py)."""
    from oracle import pyoracle as po
    fx = fixture("ragged_n")
    pair = _pair(sa, fx)
    res = sa.OverlapBuilder(pair).overlap(fx.seqs, 15)
    offs = res["block_offs"]
    for r, seq in enumerate(fx.seqs):
        want, sub, _, _ = po.overlap(fx.fwd, fx.rev, seq, 15)
        got = res["blocks"][int(offs[r]):int(offs[r + 1])]
        cols = ["capped0_lo", "capped0_hi", "capped1_lo", "capped1_hi", "raw0_lo", "raw0_hi", "raw1_lo", "raw1_hi",
                "length", "af"]
        gl = [[int(b[c]) for c in cols] for b in got]
        assert gl == [list(map(int, w)) for w in want], r
        assert bool(res["substring"][r]) == sub


def test_toy_asqg_matches_reference_md5(sa):
    """End of the chain: GPU ASQG md5 == the md5 prefix of the reference's own output."""
    fx = fixture("toy")
    pair = _pair(sa, fx)
    text, _ = sa.OverlapBuilder(pair, fx.prefix).build(fx.fa, GOLDEN["toy"]["min_overlap"])
    assert md5_prefix(text) == GOLDEN["toy"]["md5"]["asqg_t1"]


def test_kmer_counts(sa):
    fx = fixture("tiny")
    pair = _pair(sa, fx)
    seqs = fx.seqs
    kmers = [s[i:i + 21] for s in seqs[:40] for i in (0, 7, 30)] + ["A" * 21, "ACGT" * 5 + "N"]
    got = pair.kmer_counts(kmers)
    want = [fx.fwd.occurrences(k) for k in kmers]
    assert list(map(int, got)) == want


def test_empty_batch_and_errors(sa):
    fx = fixture("tiny")
    pair = _pair(sa, fx)
    res = sa.OverlapBuilder(pair).overlap([], 20)
    assert len(res["blocks"]) == 0 and list(res["block_offs"]) == [0]
    with pytest.raises(sa.SigaxError):
        sa.FMIndexPair.load("/nonexistent/prefix")
    with pytest.raises(sa.SigaxError):  # edges without read metadata
        sa.OverlapBuilder(_pair(sa, fx)).overlap(fx.seqs[:3], 20, edges=True)


# ---- host C++ side (siga_amd/host): OverlapBuilder::build and the CLI, drop-in file naming -----------------------
def test_cli_overlap_writes_reference_asqg_gz(sa, tmp_path):
    """`siga overlap -m 45 toy.fa` in a scratch CWD: <stem>.asqg.gz must gunzip to the oracle's ASQG bytes
    (= the reference's own output by md5)."""
    import gzip
    import shutil
    import subprocess
    from siga_amd import host
    fx = fixture("toy")
    cwd = str(tmp_path)
    for ext in (".fa", ".bwt", ".rbwt", ".sai", ".rsai"):
        shutil.copy(fx.prefix + ext, cwd)
    r = subprocess.run([host.CLI_PATH, "overlap", "-m", "45", "-t", "4", "toy.fa"], cwd=cwd, capture_output=True)
    assert r.returncode == 0, r.stderr
    got = gzip.open(cwd + "/toy.asqg.gz", "rb").read().decode()
    want, _, _ = fx.oracle_asqg(45)
    assert got == want
    assert md5_prefix(got) == GOLDEN["toy"]["md5"]["asqg_t1"]
    x = subprocess.run([host.CLI_PATH, "overlap", "-m", "45", "-x", "--no-opposite-strand", "-p", "toy", "toy.fa"], cwd=cwd)
    assert x.returncode == 0
    wantx, _, _ = fx.oracle_asqg(45, irreducible=False, rc=False)
    assert gzip.open(cwd + "/toy.asqg.gz", "rb").read().decode() == wantx
    bad = subprocess.run([host.CLI_PATH, "overlap", "-p", "nosuchprefix", "toy.fa"], cwd=cwd, capture_output=True)
    assert bad.returncode == 255


def test_host_builder_vertex_tags_and_fastq(sa, tmp_path):
    """VT tags from FASTA comments (CR/BX/EX) and FASTQ input through the C++ OverlapBuilder::build."""
    from siga_amd import host
    from oracle import pyoracle as po
    fx = fixture("tiny")
    reads = fx.reads[:60]
    fq = tmp_path / "t.fq"
    with open(fq, "w") as f:
        for i, (n, s) in enumerate(reads):
            c = " BX:Z:AC%d CR:i:%d junk EX:Z:e" % (i, i) if i % 3 == 0 else (" CR:i:oops" if i % 3 == 1 else "")
            f.write("@%s%s\n%s\n+\n%s\n" % (n, c, s, "I" * len(s)))
    prefix = str(tmp_path / "t")
    host.index_file(str(fq), prefix)
    out = str(tmp_path / "t.asqg")
    host.overlap_file(str(fq), prefix, 20, out)
    fwd = po.Index.load(prefix + ".bwt", prefix + ".sai")
    rev = po.Index.load(prefix + ".rbwt", prefix + ".rsai")
    po.build_asqg(fwd, rev, str(fq), 20, str(tmp_path / "o.asqg"))
    assert open(out).read() == open(tmp_path / "o.asqg").read()


def test_mid_fixture_through_gpu_matches_reference_md5(sa, tmp_path):
    """40 000 x 150 bp (SURVEY.md App. C `mid`): GPU ASQG md5 == the reference's recorded md5 prefix."""
    from siga_amd import host
    fx = fixture("mid")
    out = str(tmp_path / "mid.asqg")
    host.overlap_file(fx.fa, fx.prefix, GOLDEN["mid"]["min_overlap"], out)
    text = open(out).read()
    assert md5_prefix(text) == GOLDEN["mid"]["md5"]["asqg_t1"]
    assert len(ed_lines(text)) == GOLDEN["mid"]["ed_irreducible"]


# ---- `siga rmdup`: OverlapBuilder::duplicate on the GPU + Hits2FastaConverter on the host --------
@pytest.mark.parametrize("name", ["dup", "corner", "tiny", "ragged"])
def test_rmdup_matches_oracle(sa, name, tmp_path):
    import shutil
    import subprocess
    from oracle import pyoracle as po
    from siga_amd import host
    fx = fixture(name)
    po.rmdup(fx.fwd, fx.rev, fx.fa, str(tmp_path / "o.fa"), str(tmp_path / "o.dups.fa"))
    host.rmdup_file(fx.fa, fx.prefix, str(tmp_path / "g.fa"), str(tmp_path / "g.dups.fa"))
    assert open(tmp_path / "g.fa").read() == open(tmp_path / "o.fa").read()
    assert open(tmp_path / "g.dups.fa").read() == open(tmp_path / "o.dups.fa").read()
    if name == "dup":  # the CLI, with the reference's output names in the CWD (src/rmdup.cpp:40-44)
        cwd = str(tmp_path)
        for ext in (".fa", ".bwt", ".rbwt", ".sai", ".rsai"):
            shutil.copy(fx.prefix + ext, cwd)
        assert subprocess.run([host.CLI_PATH, "rmdup", "dup.fa"], cwd=cwd).returncode == 0
        assert open(cwd + "/dup.rmdup.fa").read() == open(tmp_path / "o.fa").read()
        assert open(cwd + "/dup.rmdup.dups.fa").read() == open(tmp_path / "o.dups.fa").read()
        assert open(tmp_path / "o.dups.fa").read().count(">") > 100  # the fixture is mostly duplicates / substrings


@pytest.mark.parametrize("name", ["dup", "corner", "ragged", "toy"])
def test_duplicate_blocks_match_oracle(sa, name):
    """sigax SIGAX_DUPLICATE blocks == OverlapBuilder::duplicate of the oracle (src/overlap_builder.cpp:1184-1195),
    block by block and read by read, substring flags included."""
    from oracle import pyoracle as po
    from tests.bigcheck import blocks_matrix
    fx = fixture(name)
    pair = _pair(sa, fx)
    res = sa.OverlapBuilder(pair).duplicate(fx.seqs)
    want = po.overlap_batch(fx.fwd, fx.rev, fx.seqs, 0, duplicate=True)
    assert np.array_equal(res["block_offs"], want["block_offs"])
    assert np.array_equal(blocks_matrix(res["blocks"]), want["blocks"])
    assert np.array_equal(res["substring"].astype(bool), want["substring"].astype(bool))
    assert res["stats"]["n_blocks"] == int(want["block_offs"][-1]) > 0


# ---- `siga correct` k-mer path (BASELINE configs[3]) ------------------------------------------------
def _noisy(reads, frac, seed, with_n=False):
    import random
    rnd = random.Random(seed)
    out = []
    for n, s in reads:
        if rnd.random() < frac:
            for _ in range(rnd.choice([1, 1, 2])):
                p = rnd.randrange(len(s))
                s = s[:p] + rnd.choice([c for c in "ACGT" if c != s[p]]) + s[p + 1:]
        if with_n and rnd.random() < 0.02:
            p = rnd.randrange(len(s))
            s = s[:p] + "N" + s[p + 1:]
        out.append((n, s))
    return out


@pytest.mark.parametrize("k,fmt", [(21, "fa"), (31, "fq"), (15, "fa")])
def test_correct_matches_oracle(sa, tmp_path, k, fmt):
    import random
    from oracle import pyoracle as po
    from siga_amd import host
    fx = fixture("toy")
    reads = _noisy(fx.reads, 0.3, 11 + k, with_n=(k == 15))
    path = str(tmp_path / ("reads." + fmt))
    rnd = random.Random(5)
    with open(path, "w") as f:
        for i, (n, s) in enumerate(reads):
            if fmt == "fq":
                q = "".join(chr(33 + rnd.choice([2, 12, 19, 20, 30, 40])) for _ in s)
                f.write("@%s%s\n%s\n+\n%s\n" % (n, " c%d" % i if i % 5 == 0 else "", s, q))
            else:
                f.write(">%s\n%s\n" % (n, s))
        if fmt == "fa":
            f.write(">short\nACGTACGT\n")  # shorter than k: never written
    prefix = str(tmp_path / "reads")
    host.index_file(path, prefix)
    st = po.correct(po.Index.load(prefix + ".bwt"), path, str(tmp_path / "o.ec"), k=k)
    host.correct_file(path, prefix, str(tmp_path / "g.ec"), k=k)
    assert open(tmp_path / "g.ec").read() == open(tmp_path / "o.ec").read()
    assert st["changed"] > 100 and st["written"] > 2000


def test_correct_cli_defaults(sa, tmp_path):
    """`siga correct reads.fa` -> reads.ec.fa in the CWD (src/correct.cpp:34-37), defaults k=31 x=3 i=10 O=1."""
    import subprocess
    from oracle import pyoracle as po
    from siga_amd import host
    fx = fixture("toy")
    cwd = str(tmp_path)
    with open(cwd + "/reads.fa", "w") as f:
        for n, s in _noisy(fx.reads, 0.2, 3):
            f.write(">%s\n%s\n" % (n, s))
    assert subprocess.run([host.CLI_PATH, "index", "reads.fa"], cwd=cwd).returncode == 0
    assert subprocess.run([host.CLI_PATH, "correct", "reads.fa"], cwd=cwd).returncode == 0
    po.correct(po.Index.load(cwd + "/reads.bwt"), cwd + "/reads.fa", cwd + "/o.ec")
    assert open(cwd + "/reads.ec.fa").read() == open(cwd + "/o.ec").read()


def _batch_download(sa, L, bt):
    import ctypes as C
    from siga_amd import _lib
    from siga_amd.overlap import BLOCK_DTYPE, EDGE_DTYPE
    res = _lib.Result()
    assert L.sigax_batch_download(bt, C.byref(res)) == 0, _lib.last_error()
    try:
        n = res.n_reads
        offs = np.ctypeslib.as_array(res.block_offs, shape=(n + 1,)).copy()
        blocks = np.frombuffer(C.string_at(res.blocks, int(offs[-1]) * BLOCK_DTYPE.itemsize), dtype=BLOCK_DTYPE).copy()
        sub = np.ctypeslib.as_array(res.substring, shape=(max(n, 1),))[:n].copy()
        eds = np.frombuffer(C.string_at(res.edges, int(res.n_edges) * EDGE_DTYPE.itemsize), dtype=EDGE_DTYPE).copy()
    finally:
        L.sigax_result_free(C.byref(res))
    return offs, blocks, sub, eds


def test_batches_in_flight_on_one_index(sa):
    """Three batch objects of one index run at once, each from its own stream (the library queues their launches on
    the index's shared finder / filter streams): every result equals the one-shot call's."""
    import ctypes as C
    from siga_amd import _lib
    from siga_amd.overlap import pack_reads, name_ranks
    fx = fixture("toy")
    pair = _pair(sa, fx)
    reads = sa.overlap.read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    n = len(seqs)
    cuts = [0, n // 3, n // 2, n]
    builder = sa.OverlapBuilder(pair, fx.prefix)
    want = [builder.overlap(seqs[cuts[i]:cuts[i + 1]], 45, read_base=cuts[i], edges=True) for i in range(3)]
    L = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    for rounds in range(2):  # second round: the batch objects are reused
        bts, streams = [], []
        for i in range(3):
            part = seqs[cuts[i]:cuts[i + 1]]
            buf, offs = pack_reads(part)
            bt = C.c_void_p()
            assert L.sigax_batch_create(pair.handle, len(part), len(buf), max(map(len, part)), C.byref(bt)) == 0
            sp = C.c_void_p()
            assert L.sigax_stream_create(0, C.byref(sp)) == 0, _lib.last_error()
            assert L.sigax_batch_upload(bt, buf, offs.ctypes.data, len(part), sp) == 0, _lib.last_error()
            assert L.sigax_batch_set_subbatches(bt, 1 + i) == 0
            bts.append(bt)
            streams.append(sp)
        for rep in range(2):
            for i in range(3):
                assert L.sigax_batch_run(bts[i], cuts[i], 45, flags, streams[i]) == 0, _lib.last_error()
            for i in (2, 0, 1):  # finished in another order than submitted
                stats = _lib.Stats()
                assert L.sigax_batch_finish(bts[i], streams[i], C.byref(stats)) == 0, _lib.last_error()
                offs, blocks, sub, eds = _batch_download(sa, L, bts[i])
                assert np.array_equal(offs, want[i]["block_offs"])
                assert blocks.tobytes() == want[i]["blocks"].tobytes()
                assert np.array_equal(sub, want[i]["substring"])
                assert eds.tobytes() == want[i]["edges"].tobytes()
        for bt, sp in zip(bts, streams):
            L.sigax_batch_destroy(bt)
            L.sigax_stream_destroy(0, sp)


def _run_batch(sa, L, pair, seqs, m, flags):
    """one run through the device-resident batch API -> (offs, blocks, substring, edges, stats dict, run-info dict)"""
    import ctypes as C
    from siga_amd import _lib
    from siga_amd.overlap import pack_reads
    buf, offs = pack_reads(seqs)
    bt = C.c_void_p()
    assert L.sigax_batch_create(pair.handle, len(seqs), len(buf), max(map(len, seqs)), C.byref(bt)) == 0, _lib.last_error()
    try:
        assert L.sigax_batch_upload(bt, buf, offs.ctypes.data, len(seqs), None) == 0, _lib.last_error()
        assert L.sigax_batch_run(bt, 0, m, flags, None) == 0, _lib.last_error()
        stats, ri = _lib.Stats(), _lib.RunInfo()
        assert L.sigax_batch_finish(bt, None, C.byref(stats)) == 0, _lib.last_error()
        assert L.sigax_batch_run_info(bt, C.byref(ri)) == 0, _lib.last_error()
        return _batch_download(sa, L, bt) + (stats.as_dict(), ri.as_dict())
    finally:
        L.sigax_batch_destroy(bt)


def _same_run(a, b):
    assert np.array_equal(a[0], b[0])
    assert a[1].tobytes() == b[1].tobytes()
    assert np.array_equal(a[2], b[2])
    assert a[3].tobytes() == b[3].tobytes()
    for k in ("n_blocks", "n_edges", "n_candidate_blocks", "n_occ_find", "n_occ_extract", "n_substring"):
        assert a[4][k] == b[4][k], k


@pytest.mark.parametrize("name", ["toy", "ragged"])
def test_deep_start_table_serves_the_runs_it_may(sa, name):
    """The block finder's deep start table (sigax_index_prepare_overlap, fm_layout.h): chains start K = min(min-overlap, 56)
    symbols in after one lookup.  A run whose min-overlap is at least the table's K uses it, a run below it walks, asking
    for a smaller min-overlap replaces the table -- and every run gives the blocks, edges and rank-evaluation counts of the
    index without the table (which the suite above pins to the oracle), with fewer table sectors asked for."""
    from siga_amd import _lib
    from siga_amd.overlap import name_ranks
    fx = fixture(name)
    reads = sa.overlap.read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    L = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    meta = (np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    def plain_run(m):
        # a fresh index per run: the first run on an index never has the table (it comes with reuse or with prepare_overlap)
        plain = sa.FMIndexPair.load(fx.prefix, resident=False)
        plain.set_reads(*meta)
        try:
            return _run_batch(sa, L, plain, seqs, m, flags)
        finally:
            plain.close()

    pair = sa.FMIndexPair.load(fx.prefix, resident=False)
    pair.set_reads(*meta)
    big, small = (45, 20) if name == "toy" else (30, 16)
    want = {m: plain_run(m) for m in (big, big + 5, small, small + 3, small - 4)}
    assert all(w[5]["deep_k"] == 0 for w in want.values())
    pair.prepare_overlap(big)
    got = _run_batch(sa, L, pair, seqs, big, flags)
    if not (got[5]["row_bits"] and got[5]["row_text"]):
        # this index has no row table + text to read the K-mers off (direct maps, SIGAX_ROWEND=0, SIGAX_LOOKAHEAD=0: the wide
        # suite runs this test under those too): no table, every chain walks, same bytes
        assert got[5]["deep_k"] == 0
        _same_run(got, want[big])
        return
    if os.environ.get("SIGAX_FIND_DEEP") == "0" or os.environ.get("SIGAX_FIND_DEEP_USE") == "0":
        assert got[5]["deep_k"] == 0
        _same_run(got, want[big])
        return
    if os.environ.get("SIGAX_DEEP_K"):
        return  # K forced by the environment: the K arithmetic below does not apply (the parity suite covers those tables)
    assert got[5]["deep_k"] == big
    _same_run(got, want[big])
    assert got[4]["n_sectors_find"] < want[big][4]["n_sectors_find"]
    got = _run_batch(sa, L, pair, seqs, big + 5, flags)   # a larger min-overlap starts from the same table
    assert got[5]["deep_k"] == big
    _same_run(got, want[big + 5])
    got = _run_batch(sa, L, pair, seqs, small, flags)     # a smaller one cannot
    assert got[5]["deep_k"] == 0
    _same_run(got, want[small])
    pair.prepare_overlap(small)                           # ... until the table is replaced
    for m in (small, small + 3, big):
        got = _run_batch(sa, L, pair, seqs, m, flags)
        assert got[5]["deep_k"] == small, m
        _same_run(got, want[m])
    got = _run_batch(sa, L, pair, seqs, small - 4, flags)
    assert got[5]["deep_k"] == 0
    _same_run(got, want[small - 4])
    pair.prepare_overlap(8)                               # below 16 symbols the 12-mer table is all there is: no-op
    assert _run_batch(sa, L, pair, seqs, small, flags)[5]["deep_k"] == small


def test_deep_start_table_comes_with_reuse(sa):
    """Without sigax_index_prepare_overlap the table is built in the background once the index has been asked for as many
    reads as it holds, for the min-overlap of the run at hand; runs before it is there walk -- same bytes."""
    import time
    from siga_amd import _lib
    from siga_amd.overlap import name_ranks
    fx = fixture("toy")
    reads = sa.overlap.read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    L = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    pair = sa.FMIndexPair.load(fx.prefix, resident=False)
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    first = _run_batch(sa, L, pair, seqs, 45, flags)
    assert first[5]["deep_k"] == 0
    if not (first[5]["row_bits"] and first[5]["row_text"]) or os.environ.get("SIGAX_FIND_DEEP") == "0" or \
            os.environ.get("SIGAX_FIND_DEEP_USE") == "0" or os.environ.get("SIGAX_DEEP_K"):
        pytest.skip("no row table + text (or the table is switched off / its K forced) in this environment")
    seen = 0
    for _ in range(50):
        got = _run_batch(sa, L, pair, seqs, 45, flags)
        _same_run(got, first)
        seen = got[5]["deep_k"]
        if seen:
            break
        time.sleep(0.1)
    assert seen == 45


def test_forward_only_index_serves_the_corrector_and_refuses_overlaps(sa, tmp_path):
    """`siga index --no-reverse` writes <prefix>.bwt/.sai alone and `siga correct` loads just that (src/correct.cpp:41-47,
    examples/siga-ecoli-miseq.sh:64-70): sigax_index_open with rbwt_path = NULL.  Occ, k-mer counts and corrections equal the
    two-strand index's; an overlap run and Occ on the missing strand come back as SIGAX_E_STATE, not as a crash."""
    import ctypes as C
    from oracle import pyoracle as po
    from siga_amd import _lib, host
    fx = fixture("toy")
    L = _lib.lib()
    h = C.c_void_p()
    assert L.sigax_index_open((fx.prefix + ".bwt").encode(), None, None, None, 0, C.byref(h)) == 0, _lib.last_error()
    fwd = sa.FMIndexPair(h.value)
    both = _pair(sa, fx)
    pos = np.arange(0, len(fx.fwd), 53, dtype=np.uint64)
    assert np.array_equal(fwd.occ(pos, 0), both.occ(pos, 0))
    kmers = [s[i:i + 25] for s in fx.seqs[:200] for i in (0, 17, 60)]
    assert np.array_equal(fwd.kmer_counts(kmers), both.kmer_counts(kmers))
    out = np.zeros((3, 5), dtype=np.uint64)
    p3 = np.arange(3, dtype=np.uint64)
    assert L.sigax_occ_batch(fwd.handle, 1, p3.ctypes.data, 3, out.ctypes.data) == -6
    with pytest.raises(sa.overlap.SigaxError) as e:
        sa.OverlapBuilder(fwd).overlap(fx.seqs[:10], 45)
    assert e.value.code == -6 and "reverse strand" in str(e.value)
    # the CLI on a forward-only index in the CWD
    import shutil
    import subprocess
    cwd = str(tmp_path)
    with open(cwd + "/reads.fa", "w") as f:
        for n, s in _noisy(fx.reads, 0.2, 3):
            f.write(">%s\n%s\n" % (n, s))
    assert subprocess.run([host.CLI_PATH, "index", "--no-reverse", "reads.fa"], cwd=cwd).returncode == 0
    assert not os.path.exists(cwd + "/reads.rbwt")
    assert subprocess.run([host.CLI_PATH, "correct", "-k", "41", "reads.fa"], cwd=cwd).returncode == 0
    po.correct(po.Index.load(cwd + "/reads.bwt"), cwd + "/reads.fa", cwd + "/o.ec", k=41)
    assert open(cwd + "/reads.ec.fa").read() == open(cwd + "/o.ec").read()
    r = subprocess.run([host.CLI_PATH, "overlap", "-m", "45", "reads.fa"], cwd=cwd, capture_output=True, text=True)
    assert r.returncode != 0 and "rbwt" in r.stderr
    fwd.close()
    both.close()


class _DeviceWords:
    """a u32 array in device memory, through the HIP runtime libsigax.so is bound to (dlsym on the library's handle looks
    through its dependencies: the process may hold a second runtime, PyTorch's own)"""

    def __init__(self, values):
        import ctypes as C
        from siga_amd import _lib
        self.hip = _lib.lib()
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        v = np.ascontiguousarray(values, dtype=np.uint32)
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), max(v.nbytes, 4)) == 0
        assert self.hip.hipMemcpy(self.ptr, v.ctypes.data, v.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def free(self):
        self.hip.hipFree(self.ptr)


@pytest.mark.parametrize("name", ["toy", "ragged"])
def test_reads_in_any_order_under_their_ids(sa, name):
    """sigax_batch_upload_read_ids / _set_device_read_ids: a batch's reads may be any subset of the indexed reads in any
    order (one rank's share under key-range sharding, siga_amd/sharding.py).  A shuffled half of the read set, given with its
    ids: every read's blocks and substring flag are those of the plain run over the whole set, and its edge records are the
    plain run's records of that query, in the same order.  Ids for another number of reads, and ids beyond the indexed reads,
    are refused."""
    import ctypes as C
    from siga_amd import _lib
    from siga_amd.overlap import pack_reads, name_ranks
    fx = fixture(name)
    pair = _pair(sa, fx)
    reads = sa.overlap.read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    n = len(seqs)
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    L = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    offs0, blocks0, sub0, eds0, st0, _ = _run_batch(sa, L, pair, seqs, 45, flags)
    by_query = {}
    for e in eds0:
        by_query.setdefault(int(e["query"]), []).append(e.tobytes())
    ids = np.random.default_rng(5).permutation(n)[: max(n // 2, 1)].astype(np.uint32)
    part = [seqs[i] for i in ids]
    buf, offs = pack_reads(part)
    bt = C.c_void_p()
    assert L.sigax_batch_create(pair.handle, len(part), len(buf), max(map(len, part)), C.byref(bt)) == 0, _lib.last_error()
    d_ids = d_bad = None
    try:
        assert L.sigax_batch_upload(bt, buf, offs.ctypes.data, len(part), None) == 0, _lib.last_error()
        d_ids = _DeviceWords(ids)
        for form in ("host", "device"):
            if form == "host":
                assert L.sigax_batch_upload_read_ids(bt, ids.ctypes.data, len(ids), None) == 0, _lib.last_error()
            else:
                assert L.sigax_batch_set_device_read_ids(bt, d_ids.ptr, len(ids)) == 0, _lib.last_error()
            for rep in range(2):
                assert L.sigax_batch_run(bt, 12345, 45, flags, None) == 0, _lib.last_error()  # read_base is not looked at
                stats = _lib.Stats()
                assert L.sigax_batch_finish(bt, None, C.byref(stats)) == 0, _lib.last_error()
                o, b, s, e = _batch_download(sa, L, bt)
                want_e = []
                for r, q in enumerate(ids):
                    q = int(q)
                    assert b[o[r]:o[r + 1]].tobytes() == blocks0[offs0[q]:offs0[q + 1]].tobytes(), (form, r, q)
                    assert s[r] == sub0[q]
                    want_e += by_query.get(q, [])
                assert e.tobytes() == b"".join(want_e), form
        # ids for another number of reads
        assert L.sigax_batch_set_device_read_ids(bt, d_ids.ptr, len(ids) + 1) == 0
        assert L.sigax_batch_run(bt, 0, 45, flags, None) == _lib.SIGAX_E_STATE
        # an id beyond the indexed reads: the host form refuses it at once, the device form when the run is finished
        bad = ids.copy()
        bad[len(bad) // 2] = n
        assert L.sigax_batch_upload_read_ids(bt, bad.ctypes.data, len(bad), None) == _lib.SIGAX_E_ARG
        d_bad = _DeviceWords(bad)
        assert L.sigax_batch_set_device_read_ids(bt, d_bad.ptr, len(bad)) == 0
        assert L.sigax_batch_run(bt, 0, 45, flags, None) == 0, _lib.last_error()
        assert L.sigax_batch_finish(bt, None, C.byref(_lib.Stats())) == _lib.SIGAX_E_ARG
        assert "beyond" in _lib.last_error()
        # forgotten again: consecutive reads from read_base
        assert L.sigax_batch_set_device_read_ids(bt, None, 0) == 0
        buf2, offs2 = pack_reads(seqs[: len(part)])
        assert L.sigax_batch_upload(bt, buf2, offs2.ctypes.data, len(part), None) == 0, _lib.last_error()
        assert L.sigax_batch_run(bt, 0, 45, flags, None) == 0, _lib.last_error()
        assert L.sigax_batch_finish(bt, None, C.byref(_lib.Stats())) == 0, _lib.last_error()
        o, b, s, e = _batch_download(sa, L, bt)
        assert e.tobytes() == b"".join(x for q in range(len(part)) for x in by_query.get(q, []))
    finally:
        L.sigax_batch_destroy(bt)
        for d in (d_ids, d_bad):
            if d is not None:
                d.free()


def test_finder_choice_measured_by_the_batch_object(sa):
    """Between 2^30 and 2^31 symbols a batch object times its finder both ways and keeps the faster (sigax_api.cpp:
    want_coop; run_info.coop says which one a run used): two runs per lane, two cooperative, then the choice -- the same
    bytes every time.  The fixtures are far below that range: all runs per lane, unless SIGAX_COOP_TUNE_MIN=0 puts them in
    it (tests/test_gpu_wide.py runs this test that way)."""
    import ctypes as C
    from siga_amd import _lib
    from siga_amd.overlap import pack_reads, name_ranks
    fx = fixture("rep")
    pair = _pair(sa, fx)
    reads = sa.overlap.read_sequences(fx.fa)
    seqs = [r[2] for r in reads]
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([r[0] for r in reads]))
    L = _lib.lib()
    flags = _lib.SIGAX_IRREDUCIBLE | _lib.SIGAX_RC | _lib.SIGAX_EDGES
    want = _run_batch(sa, L, pair, seqs, 30, flags)
    buf, offs = pack_reads(seqs)
    bt = C.c_void_p()
    assert L.sigax_batch_create(pair.handle, len(seqs), len(buf), max(map(len, seqs)), C.byref(bt)) == 0, _lib.last_error()
    used = []
    try:
        assert L.sigax_batch_upload(bt, buf, offs.ctypes.data, len(seqs), None) == 0, _lib.last_error()
        for rep in range(7):
            assert L.sigax_batch_run(bt, 0, 30, flags, None) == 0, _lib.last_error()
            stats, ri = _lib.Stats(), _lib.RunInfo()
            assert L.sigax_batch_finish(bt, None, C.byref(stats)) == 0, _lib.last_error()
            assert L.sigax_batch_run_info(bt, C.byref(ri)) == 0
            used.append(int(ri.coop))
            _same_run(_batch_download(sa, L, bt) + (stats.as_dict(), ri.as_dict()), want)
    finally:
        L.sigax_batch_destroy(bt)
    static = any(os.environ.get(k) for k in ("SIGAX_FIND_COOP", "SIGAX_COOP_MIN_SYMBOLS")) or os.environ.get("SIGAX_FORCE_WIDE")
    if os.environ.get("SIGAX_COOP_TUNE_MIN") == "0" and not static and os.environ.get("SIGAX_TWO_STEP") != "0":
        assert used[:4] == [0, 0, 1, 1] and used[4] == used[5] == used[6], used
    elif not static:
        assert used == [0] * 7, used
