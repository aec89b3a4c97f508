"""Pins the oracle (CPU restatement) to every reference observation available for this path:
the adjacent KATs in the reference's own tests and the reference outputs recorded in SURVEY.md App. C
(tests/golden/survey_appc.json).  CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.fixtures import GOLDEN, ed_lines, fixture, md5_prefix


# ---- KATs restated from the reference's own tests ------------------------------------------------
def test_alphabet_ranks():  # test/index_test.cpp:13-31
    L = po.lib()
    for ch, r in zip("$ACGT", range(5)):
        assert L.orc_torank(ord(ch)) == r
        assert chr(L.orc_tochar(r)) == ch
    for other in "N\0acgtX-":  # alphabet.h:19-39: every other byte -> 0
        assert L.orc_torank(ord(other)) == 0


def test_rl_encoding():  # test/index_test.cpp:33-84
    out = (C.c_uint8 * 16)()
    n = po.lib().orc_rl_encode(b"AAACGGGTA", 9, out, 16)
    assert n == 5
    runs = [(out[i] >> 5, out[i] & 31) for i in range(n)]
    assert runs == [(1, 3), (2, 1), (3, 3), (4, 1), (1, 1)]
    big = (C.c_uint8 * 16)()
    n = po.lib().orc_rl_encode(b"A" * 70, 70, big, 16)  # 31-cap split, rlstring.h:13, bwt.cpp:17
    assert [big[i] & 31 for i in range(n)] == [31, 31, 8]


def test_tag_values():  # test/overlap_test.cpp:9-28
    buf = C.create_string_buffer(64)
    assert po.lib().orc_tag_roundtrip(b"test:i:100", ord("i"), b"CR", buf, 64) == 1
    assert buf.value == b"CR:i:100"
    assert po.lib().orc_tag_roundtrip(b"test:Z:100.0", ord("Z"), b"BX", buf, 64) == 1
    assert buf.value == b"BX:Z:100.0"
    assert po.lib().orc_tag_roundtrip(b"test:f:100.0", ord("f"), b"ER", buf, 64) == 1
    assert po.lib().orc_tag_roundtrip(b"test:Z:1", ord("i"), b"CR", buf, 64) == 0


def test_revcomp_and_stem():  # test/preprocess_test.cpp:30-43, test/utils_test.cpp:32-36
    buf = C.create_string_buffer(8)
    po.lib().orc_revcomp(b"ACGTGAC", 7, 0, buf)
    assert buf.raw[:7] == b"CAGTGCA"
    po.lib().orc_revcomp(b"CAGTGCA", 7, 1, buf)
    assert buf.raw[:7] == b"GTCACGT"
    out = C.create_string_buffer(64)
    for p in (b"a.txt", b"a.txt.gz", b"a.txt.bz2", b"/x/y/a.fa"):
        po.lib().orc_stem(p, out, 64)
        assert out.value == b"a"


# ---- reference outputs recorded at survey time -----------------------------------------------------
def test_corner_fixture():
    g = GOLDEN["corner"]
    fx = fixture("corner")
    assert list(fx.fwd.sai()) == g["sai"]
    asqg, hits, _ = fx.oracle_asqg(g["min_overlap"], hits=True)
    assert ed_lines(asqg) == g["ed"]
    assert hits.split("\n")[3] == g["hit_line_d"]
    vt = [l for l in asqg.split("\n") if l.startswith("VT")]
    assert [l.split("\t")[1] for l in vt if l.endswith("SS:i:1")] == g["substring_reads"]
    assert asqg.split("\n")[0] == "HT\tVN:i:1\tOL:i:10\tCN:i:1"
    x, _, _ = fx.oracle_asqg(g["min_overlap"], irreducible=False)
    assert ed_lines(x) == g["ed"]


def test_toy_fixture_md5_and_counts():
    g = GOLDEN["toy"]
    fx = fixture("toy")
    assert md5_prefix(open(fx.fa, "rb").read()) == g["md5"]["fa"]
    assert md5_prefix(open(fx.prefix + ".bwt", "rb").read()) == g["md5"]["bwt"]
    assert md5_prefix(open(fx.prefix + ".sai", "rb").read()) == g["md5"]["sai"]
    asqg, _, st = fx.oracle_asqg(g["min_overlap"])
    assert md5_prefix(asqg) == g["md5"]["asqg_t1"]
    assert len(ed_lines(asqg)) == g["ed_irreducible"]
    assert st["blocks"] == g["blocks"] and st["occ_calls"] == g["occ_calls"]
    x, _, _ = fx.oracle_asqg(g["min_overlap"], irreducible=False)
    assert len(ed_lines(x)) == g["ed_exhaustive"]
    n, _, _ = fx.oracle_asqg(g["min_overlap"], rc=False)
    assert len(ed_lines(n)) == g["ed_irreducible_norc"]


def test_rep_fixture_resolve_path():
    g = GOLDEN["rep"]
    fx = fixture("rep")
    asqg, _, _ = fx.oracle_asqg(g["min_overlap"])
    assert ed_lines(asqg) == g["ed_irreducible"]
    x, _, _ = fx.oracle_asqg(g["min_overlap"], irreducible=False)
    assert ed_lines(x) == [g["ed_irreducible"][i] for i in g["ed_exhaustive_order"]]


def test_dup_fixture():
    g = GOLDEN["dup"]
    fx = fixture("dup")
    asqg, _, _ = fx.oracle_asqg(g["min_overlap"])
    assert len(ed_lines(asqg)) == g["ed"]
    assert asqg.count("SS:i:1") == g["substring_reads"]


@pytest.mark.slow
def test_mid_fixture_md5_and_counts():
    g = GOLDEN["mid"]
    fx = fixture("mid")
    import os
    assert os.path.getsize(fx.prefix + ".bwt") == g["bwt_bytes"]
    assert len(fx.fwd) == g["symbols"]
    asqg, _, st = fx.oracle_asqg(g["min_overlap"])
    assert md5_prefix(asqg) == g["md5"]["asqg_t1"]
    assert len(ed_lines(asqg)) == g["ed_irreducible"]
    assert st["blocks"] == g["blocks"] and st["occ_calls"] == g["occ_calls"]


# ---- internal consistency of the restatement -------------------------------------------------------
def test_occ_matches_naive_count():
    fx = fixture("tiny")
    runs = fx.fwd.runs()
    bwt = np.repeat(runs >> 5, runs & 31)
    assert len(bwt) == len(fx.fwd)
    cum = np.zeros((len(bwt) + 1, 5), dtype=np.uint64)
    for r in range(5):
        cum[1:, r] = np.cumsum(bwt == r)
    for i in list(range(0, len(bwt), 37)) + [len(bwt) - 1]:
        assert list(fx.fwd.occ(i)) == list(cum[i + 1])
    assert list(fx.fwd.occ(2**64 - 1)) == [0] * 5  # fmindex.cpp:191: ++i wraps to 0
    assert list(fx.fwd.pred()) == [0] + list(np.cumsum([np.sum(bwt == r) for r in range(4)]))
