"""`siga index` on the GPU (sigax_build_strand, siga_amd/csrc/sigax_index_build.hip) against the oracle's index builder
(model B) and against the host's own SA-IS: .bwt/.rbwt/.sai/.rsai byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from tests.fixtures import GOLDEN, fixture, md5_prefix

pytestmark = pytest.mark.gpu

EXTS = (".bwt", ".rbwt", ".sai", ".rsai")


def _same(prefix_a, prefix_b):
    for ext in EXTS:
        a, b = open(prefix_a + ext, "rb").read(), open(prefix_b + ext, "rb").read()
        assert a == b, "%s differs (%d vs %d bytes)" % (ext, len(a), len(b))


@pytest.mark.parametrize("name", ["corner", "rep", "dup", "tiny", "ragged", "ragged_n", "toy", "mid"])
def test_gpu_index_matches_oracle_builder(name, tmp_path):
    from siga_amd import host
    fx = fixture(name)
    prefix = str(tmp_path / name)
    host.index_file_gpu(fx.fa, prefix)
    _same(prefix, fx.prefix)
    if name == "toy":  # the reference's own files, as recorded by the survey
        assert md5_prefix(open(prefix + ".bwt", "rb").read()) == GOLDEN["toy"]["md5"]["bwt"]
        assert md5_prefix(open(prefix + ".sai", "rb").read()) == GOLDEN["toy"]["md5"]["sai"]


@pytest.mark.parametrize("name", ["dup", "toy"])
def test_gpu_index_in_many_groups(name, tmp_path, monkeypatch):
    """A tiny sort workspace cuts the suffixes into many prefix groups: same files."""
    from siga_amd import host
    fx = fixture(name)
    monkeypatch.setenv("SIGAX_BUILD_GROUP", "700")
    prefix = str(tmp_path / name)
    host.index_file_gpu(fx.fa, prefix)
    _same(prefix, fx.prefix)


def _pack(seqs):
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return "".join(seqs).encode(), offs


def _oracle_files(seqs, prefix):
    from oracle import pyoracle as po
    po.Index.build(seqs).save(prefix + ".bwt", prefix + ".sai")
    po.Index.build(seqs, reverse=True).save(prefix + ".rbwt", prefix + ".rsai")


def test_gpu_index_repeats_and_degenerate_sets(tmp_path):
    """Deep repeats (segments beyond one wave, many global rounds), identical reads in a row, homopolymers, reads that
    are prefixes of each other, a single read, one-base reads."""
    from siga_amd import host
    rng = np.random.default_rng(3)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 300))
    cases = {
        "same": [base[:80]] * 150,                                      # 150 identical reads: every segment has 150 rows
        "polya": ["A" * 70] * 40 + ["A" * k for k in range(1, 60)],
        "tandem": ["ACG" * 30, "CGA" * 30, "GAC" * 30] * 30 + [base[i:i + 90] for i in range(0, 200, 3)],
        "deep": [base[i:i + 100] for i in range(0, 200)] * 2,          # 200x coverage of a short genome, each read twice
        "runs": ["A" * 200] * 60 + ["C" * 150] * 40 + ["ACGT" * 10],  # BWT runs of thousands of symbols: many 31-unit runs
        "single": ["ACGTTGCA"],
        "ones": ["A", "C", "A", "T", "N", "A"],
    }
    for name, seqs in cases.items():
        want = str(tmp_path / (name + "_o"))
        got = str(tmp_path / (name + "_g"))
        _oracle_files(seqs, want)
        buf, offs = _pack(seqs)
        host.index_build_gpu(buf, offs, got)
        _same(got, want)


def test_gpu_index_empty_set(tmp_path):
    from siga_amd import host
    got = str(tmp_path / "empty")
    host.index_build_gpu(b"", np.zeros(1, dtype=np.uint64), got)
    assert os.path.getsize(got + ".bwt") == 30  # header only (src/bwt.cpp:121-178)


def test_gpu_index_matches_host_sais_at_c2_scale(tmp_path):
    """BASELINE configs[1] reads (1M x 150 bp): the device builder and the host's multi-threaded suffix sort write the
    same four files."""
    from siga_amd import host
    from tests.golden.make_reads import fast_reads
    import time
    N, G, L = 1000000, 5000000, 150
    reads, _ = fast_reads(G, L, N, 1)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    t0 = time.time()
    host.index_build_gpu(reads.reshape(-1), offs, str(tmp_path / "g"))
    t1 = time.time()
    host.index_build(reads.reshape(-1), offs, str(tmp_path / "h"), threads=max(2, min(os.cpu_count() or 2, 64)))
    t2 = time.time()
    print("index build 1M x 150: GPU %.2f s, host %.2f s" % (t1 - t0, t2 - t1))
    _same(str(tmp_path / "g"), str(tmp_path / "h"))


def test_cli_index_on_gpu(tmp_path):
    from siga_amd import host
    fx = fixture("toy")
    cwd = str(tmp_path)
    r = subprocess.run([host.CLI_PATH, "index", "-p", "t", fx.fa], cwd=cwd, capture_output=True)
    assert r.returncode == 0, r.stderr
    _same(os.path.join(cwd, "t"), fx.prefix)
    r = subprocess.run([host.CLI_PATH, "index", "--cpu", "-p", "c", fx.fa], cwd=cwd, capture_output=True)
    assert r.returncode == 0, r.stderr
    _same(os.path.join(cwd, "c"), fx.prefix)


def _rl_units(bwt_ranks):
    """RL units of src/rlstring.h:10-63 with the 31-cap of src/bwt.cpp:17 for a sequence of symbol ranks"""
    out = []
    i = 0
    while i < len(bwt_ranks):
        j = i
        while j < len(bwt_ranks) and bwt_ranks[j] == bwt_ranks[i] and j - i < 31:
            j += 1
        out.append((bwt_ranks[i] << 5) | (j - i))
        i = j
    return np.array(out, dtype=np.uint8)


def _index_with_sentinels_ordered_by_read(seqs, reverse=False):
    """A valid FM-index of the same reads in ANOTHER suffix order: every read's own '$', ordered by read index (what `-a sais`
    gives) instead of one shared '$' with comparisons running on into the next read."""
    if reverse:
        seqs = [s[::-1] for s in seqs]
    rank = {"A": 1, "C": 2, "G": 3, "T": 4}
    rows = [(("", i), i, len(s)) for i, s in enumerate(seqs)]  # the '$' suffixes first, by read index
    rows += sorted(((s[t:], i), i, t) for i, s in enumerate(seqs) for t in range(len(s)))
    bwt = [rank[seqs[i][t - 1]] if t > 0 else 0 for _, i, t in rows]
    sai = np.array([i for _, i, t in rows if t == 0], dtype=np.uint32)  # the full-read suffixes in row order (no empty reads here)
    return _rl_units(bwt), sai, len(bwt)


def test_order_check_passes_builders_and_flags_another_suffix_order():
    """sigax_index_check_order compares every pair of adjacent BWT rows on the device (row table = suffix array, stretch text =
    reads).  The oracle's and the GPU builder's indexes are in order; an index of the same reads with the sentinels ordered
    by read index -- a valid FM-index, LF walks and all, but not the order of record -- is flagged."""
    import siga_amd
    from siga_amd.overlap import name_ranks
    for name in ("toy", "dup", "tiny"):
        fx = fixture(name)
        pair = siga_amd.FMIndexPair.load(fx.prefix)
        pair.set_reads(np.array([len(s) for s in fx.seqs], dtype=np.uint32), name_ranks([n for n, _ in fx.reads]))
        for which in (0, 1):
            bad, first, und = pair.check_order(which)
            assert bad == 0, (name, which, bad, first)
        pair.close()
    fx = fixture("dup")
    seqs = fx.seqs
    runs, sai, nsym = _index_with_sentinels_ordered_by_read(seqs)
    rruns, rsai, _ = _index_with_sentinels_ordered_by_read(seqs, reverse=True)
    assert nsym == sum(len(s) + 1 for s in seqs) and len(sai) == len(seqs)
    pair = siga_amd.FMIndexPair.from_memory(runs, rruns, nsym, len(seqs), sai, rsai)
    pair.set_reads(np.array([len(s) for s in seqs], dtype=np.uint32), name_ranks([n for n, _ in fx.reads]))
    bad, first, und = pair.check_order(0)
    assert bad > 0
    pair.close()
