"""Randomised differential tests: read sets with substitution errors (branching in the irreducible extractor),
duplicates, substrings, mixed lengths and both strands, through the GPU path vs the oracle -- hits text (block lists in
order) and ASQG must be identical.  Seeds are fixed; each case is a few hundred to a few thousand reads."""
import os
import random

import pytest

from oracle import pyoracle as po
from tests.fixtures import CACHE
from tests.golden.make_reads import revcomp

pytestmark = pytest.mark.gpu


def make_case(seed):
    rnd = random.Random(seed)
    G = rnd.choice([1500, 4000, 12000])
    genome = "".join(rnd.choice("ACGT") for _ in range(G))
    if rnd.random() < 0.5:  # a repeat: the same segment twice
        seg = genome[100:100 + rnd.choice([60, 150])]
        p = rnd.randrange(G // 2, G - len(seg))
        genome = genome[:p] + seg + genome[p + len(seg):]
    fixed = rnd.random() < 0.6
    L = rnd.choice([50, 80, 120])
    cov = rnd.choice([8, 15, 30, 60])
    err = rnd.choice([0.0, 0.005, 0.02, 0.04])
    n = max(50, min(4000, G * cov // L))
    reads = []
    for i in range(n):
        l = L if fixed else rnd.randrange(max(25, L // 2), L + 1)
        p = rnd.randrange(0, G - l + 1)
        s = genome[p:p + l]
        if rnd.random() < 0.5:
            s = revcomp(s)
        s = "".join((rnd.choice([c for c in "ACGT" if c != b]) if rnd.random() < err else b) for b in s)
        reads.append(("q%d" % i, s))
    for _ in range(rnd.choice([0, 3, 20])):  # exact duplicates and substrings under new names
        nme, s = rnd.choice(reads)
        if rnd.random() < 0.5 and len(s) > 30:
            a = rnd.randrange(0, len(s) - 25)
            s = s[a:a + rnd.randrange(25, len(s) - a + 1)]
        reads.append(("d%d" % len(reads), s))
    m = rnd.choice([12, 20, 31])
    return reads, m, rnd.random() < 0.75, rnd.random() < 0.8


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_case_bit_exact(seed, tmp_path):
    import siga_amd
    from siga_amd import host
    from siga_amd.overlap import format_hits
    reads, m, irr, rc = make_case(seed)
    d = str(tmp_path)
    fa = d + "/r.fa"
    with open(fa, "w") as f:
        for n, s in reads:
            f.write(">%s\n%s\n" % (n, s))
    prefix = d + "/r"
    host.index_file(fa, prefix, threads=2)
    fwd = po.Index.load(prefix + ".bwt", prefix + ".sai")
    rev = po.Index.load(prefix + ".rbwt", prefix + ".rsai")
    st = po.build_asqg(fwd, rev, fa, m, d + "/o.asqg", d + "/o.hits", irr, rc)
    pair = siga_amd.FMIndexPair.load(prefix)
    text, res = siga_amd.OverlapBuilder(pair, prefix, irreducible=irr, rc=rc).build(fa, m)
    got_hits, want_hits = format_hits(res), open(d + "/o.hits").read()
    if got_hits != want_hits:
        gl, wl = got_hits.split("\n"), want_hits.split("\n")
        bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
        raise AssertionError("seed %d (m=%d irr=%s rc=%s, %d reads): %d reads differ; first %d:\n got  %s\n want %s" % (
            seed, m, irr, rc, len(reads), len(bad), bad[0], gl[bad[0]][:500], wl[bad[0]][:500]))
    assert text == open(d + "/o.asqg").read()
    s = res["stats"]
    assert s["n_occ_find"] + s["n_occ_extract"] == st["n_occ_min"]
    # rmdup on the same set
    po.rmdup(fwd, rev, fa, d + "/o.rm.fa", d + "/o.rm.dups.fa")
    host.rmdup_file(fa, prefix, d + "/g.rm.fa", d + "/g.rm.dups.fa")
    assert open(d + "/g.rm.fa").read() == open(d + "/o.rm.fa").read()
    assert open(d + "/g.rm.dups.fa").read() == open(d + "/o.rm.dups.fa").read()


@pytest.mark.parametrize("L", [250, 400, 700])
def test_long_reads_bit_exact(L, tmp_path):
    """Read lengths at which the finder changes form: 128 reads of a workgroup staged in LDS (one launch per strand),
    64 reads staged (both strands in one launch), reads too long to stage (one-step finder)."""
    import siga_amd
    from siga_amd import host
    from siga_amd.overlap import format_hits
    rnd = random.Random(1000 + L)
    G = 12 * L
    genome = "".join(rnd.choice("ACGT") for _ in range(G))
    reads = []
    for i in range(G * 12 // L):
        l = L if i % 5 else rnd.randrange(L // 2, L)
        p = rnd.randrange(0, G - l + 1)
        s = genome[p:p + l]
        if rnd.random() < 0.5:
            s = revcomp(s)
        s = "".join((rnd.choice([c for c in "ACGT" if c != b]) if rnd.random() < 0.003 else b) for b in s)
        reads.append(("q%d" % i, s))
    d = str(tmp_path)
    fa = d + "/r.fa"
    with open(fa, "w") as f:
        for n, s in reads:
            f.write(">%s\n%s\n" % (n, s))
    prefix = d + "/r"
    host.index_file(fa, prefix, threads=2)
    fwd = po.Index.load(prefix + ".bwt", prefix + ".sai")
    rev = po.Index.load(prefix + ".rbwt", prefix + ".rsai")
    st = po.build_asqg(fwd, rev, fa, 40, d + "/o.asqg", d + "/o.hits", True, True)
    pair = siga_amd.FMIndexPair.load(prefix)
    text, res = siga_amd.OverlapBuilder(pair, prefix).build(fa, 40)
    assert format_hits(res) == open(d + "/o.hits").read()
    assert text == open(d + "/o.asqg").read()
    s = res["stats"]
    assert s["n_occ_find"] + s["n_occ_extract"] == st["n_occ_min"]
