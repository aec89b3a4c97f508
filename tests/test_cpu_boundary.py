"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/sigax.h declares, refuses to
run without a device (no CPU fallback), and the host logic (index builder, readers, stem, formatters) matches the
oracle and the reference's KATs."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from siga_amd import _lib, host
from siga_amd import overlap as ov
from tests.fixtures import GOLDEN, ROOT, fixture, md5_prefix


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "sigax.h")).read()
    declared = set(re.findall(r"\b(sigax_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (sigax_[a-z_0-9]+)", out))
    assert declared <= exported


def test_struct_sizes_match_header():
    assert _lib.BLOCK_DTYPE.itemsize == 80 and _lib.EDGE_DTYPE.itemsize == 16
    assert C.sizeof(_lib.Stats) == 11 * 8


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fx = fixture("corner")
    with pytest.raises(ov.SigaxError) as e:
        ov.FMIndexPair.load(fx.prefix)
    assert e.value.code == -3  # SIGAX_E_DEVICE
    n = C.c_int(-1)
    assert _lib.lib().sigax_device_count(C.byref(n)) != 0 or n.value == 0


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "siga_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("the oracle", "").replace("oracle-", ""), os.path.join(dirpath, f)


@pytest.mark.parametrize("name", ["corner", "rep", "dup", "tiny", "toy", "ragged_n"])
def test_siga_index_matches_oracle_builder(name, tmp_path):
    """`siga index` (own SA-IS, siga_amd/host) is byte-identical to the oracle's naive suffix sort (model B)."""
    fx = fixture(name)
    prefix = str(tmp_path / name)
    host.index_file(fx.fa, prefix, threads=2)
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        assert open(prefix + ext, "rb").read() == open(fx.prefix + ext, "rb").read(), ext
    if name == "toy":
        assert md5_prefix(open(prefix + ".bwt", "rb").read()) == GOLDEN["toy"]["md5"]["bwt"]
        assert md5_prefix(open(prefix + ".sai", "rb").read()) == GOLDEN["toy"]["md5"]["sai"]


def test_cli_index_and_help(tmp_path):
    fx = fixture("corner")
    cwd = str(tmp_path)
    fa = os.path.join(cwd, "reads.fa.gz")
    import gzip
    with gzip.open(fa, "wb") as f:
        f.write(open(fx.fa, "rb").read())
    r = subprocess.run([host.CLI_PATH, "index", "-t", "2", fa], cwd=cwd)
    assert r.returncode == 0
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):  # Utils::stem("reads.fa.gz") == "reads"; outputs land in the CWD
        assert open(os.path.join(cwd, "reads" + ext), "rb").read() == open(fx.prefix + ext, "rb").read()
    h = subprocess.run([host.CLI_PATH, "overlap", "--help"], capture_output=True)
    assert h.returncode == 0 and b"--min-overlap" in h.stdout  # help returns 256 -> exit status 0
    bad = subprocess.run([host.CLI_PATH, "index", "/nonexistent.fa"], cwd=cwd, capture_output=True)
    assert bad.returncode == 255  # runner returned -1


def test_stem_rule():  # test/utils_test.cpp:32-36
    for p in ("a.txt", "a.txt.gz", "a.txt.bz2", "/x/y/a.fa", "a"):
        assert host.stem(p) == "a"


def test_reader_semantics(tmp_path):  # src/kseq.cpp:127-228, test/preprocess_test.cpp:30-43
    fa = tmp_path / "x.fa"
    fa.write_text(">r1 BX:Z:ACGT CR:i:7\nACGT\nAC\n\n>r2\tcomment\n  GGTT  \n>r3\nTT\n")
    recs = ov.read_sequences(str(fa))
    assert recs == [("r1", "BX:Z:ACGT CR:i:7", "ACGTAC"), ("r2", "comment", "GGTT"), ("r3", "", "TT")]
    fq = tmp_path / "x.fq"
    fq.write_text("@q1 c\nACGT\n+\nIIII\n@q2\nGG\n+q2\nII\n")
    assert ov.read_sequences(str(fq)) == [("q1", "c", "ACGT"), ("q2", "", "GG")]


def test_vertex_tags_and_edge_coords():
    assert ov._vertex_tags("BX:Z:ACGT CR:i:7 EX:Z:foo") == "\tCR:i:7\tBX:Z:ACGT\tEX:Z:foo"  # asqg.cpp:171-186 order
    assert ov._vertex_tags("CR:i:x") == "\tCR:i:0" and ov._vertex_tags("CR:Z:7") == ""
    # OverlapBlock::overlap (overlap_builder.cpp:158-175) on the corner fixture's first ED line "b a 0 20 30 9 29 30"
    assert ov.edge_coords(21, 3, 30, 30) == (0, 20, 9, 29)
    assert ov.edge_coords(21, 0, 30, 30) == (9, 29, 0, 20)


def test_name_ranks_are_string_order():
    r = ov.name_ranks(["r10", "r9", "r1", "r10"])
    assert list(r) == [1, 2, 0, 1]  # "r1" < "r10" < "r9"; equal names share a rank (overlap_builder.cpp:358,365)


@pytest.mark.slow
def test_parallel_suffix_sort_equals_sais(tmp_path):
    """threads >= 3 selects the multi-threaded bucket sort: same files as the SA-IS path (and as the oracle) on `mid`."""
    fx = fixture("mid")
    prefix = str(tmp_path / "mid")
    host.index_file(fx.fa, prefix, threads=4)
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        assert open(prefix + ext, "rb").read() == open(fx.prefix + ext, "rb").read(), ext


def test_gzip_writer_single_member_roundtrip(tmp_path):
    """The ASQG writer's gzip stream: one member, valid for zlib/gzip, across block (1 MiB) and flush (64 MiB) edges."""
    import gzip
    import zlib
    rng = np.random.default_rng(3)
    line = b"ED\tr123 r45 0 104 150 45 149 150 0 0\n"
    for n, pieces in ((0, 1), (1, 1), (len(line) * 1000, 7), ((1 << 20) + 17, 3), (70 << 20, 41)):
        data = (line * (n // len(line) + 1))[:n]
        if n > 100:
            noise = rng.integers(65, 90, size=min(n, 1 << 16), dtype=np.uint8).tobytes()
            data = noise + data[len(noise):]
        path = str(tmp_path / ("x%d.gz" % n))
        host.write_file(path, data, pieces)
        assert gzip.open(path, "rb").read() == data
        raw = open(path, "rb").read()
        d = zlib.decompressobj(16 + zlib.MAX_WBITS)  # exactly one gzip member, nothing after it
        assert d.decompress(raw) == data and d.eof and d.unused_data == b""
    plain = str(tmp_path / "x.txt")
    host.write_file(plain, line * 10, 3)
    assert open(plain, "rb").read() == line * 10


def test_parallel_loader_equals_record_reader(tmp_path):
    """The chunk-parallel FASTA loader and the serial FASTQ walk of the host library give the records of the
    record-at-a-time reader (src/kseq.cpp:140-228 semantics), quirks included: blank and padded lines, multi-line
    sequences, comments, a named record without sequence (the reader gives up there), a header without text (its
    sequence leaks into the next record), gz input.  Files whose records have their bases on one line take the loader's
    by-reference chunks (bases copied once, from the file image), files with multi-line records the copying ones, a file
    with both kinds either per chunk (SIGA_LOADER_COPY=1 forces the copying way everywhere)."""
    import gzip
    import random
    rnd = random.Random(4)

    def seq(n):
        return "".join(rnd.choice("ACGT") for _ in range(n))

    big = []
    for i in range(30000):
        s = seq(rnd.choice([30, 70, 151]))
        body = s if i % 3 else s[:20] + "\n  " + s[20:] + "  \n\n"
        big.append(">r%d%s\n%s\n" % (i, " BX:Z:AC CR:i:%d" % i if i % 11 == 0 else ("\tx" if i % 13 == 0 else ""), body))
    # records with their bases on one line (what reads files are): the chunks' bases stay in the file image until the join
    one = [">s%d%s\n%s%s\n" % (i, " c=%d" % i if i % 7 == 0 else "", "  " if i % 5 == 0 else "", seq(rnd.choice([25, 100, 150]))) + ("\n" if i % 17 == 0 else "")
           for i in range(30000)]
    cases = {
        "big.fa": "".join(big),
        "one.fa": "".join(one),
        "one_then_many.fa": "".join(one[:15000]) + "".join(big[:15000]),   # chunks of either kind in one file
        "one_stop.fa": "".join(one[:12345]) + ">empty\n>after\nACGT\n" + "".join(one[12345:]),
        "one_nameless.fa": "".join(one[:700]) + ">\nAAAA\n>next\nCCCC\n" + "".join(one[700:1500]),
        "one_tail.fa": "".join(one[:20000]) + ">last",
        "one_noeol.fa": "".join(one[:9999]) + ">z\nACGTACGT",
        "stop.fa": "".join(big[:20000]) + ">empty\n>after\nACGT\n" + "".join(big[20000:]),
        "nameless.fa": "".join(big[:500]) + ">\nAAAA\n>next\nCCCC\n" + "".join(big[500:900]),
        "spaces.fa": " >x y\n AC GT \n\n\r\n>z\nTT\n",
        "tail.fa": "".join(big[:100]) + ">last",
        "reads.fq": "".join("@q%d%s\n%s\n+%s\n%s\n" % (i, " c" if i % 5 == 0 else "", s, "q%d%s" % (i, " c" if i % 5 == 0 else "") if i % 2 else "",
                                                          "I" * len(s)) for i, s in ((i, seq(50 + i % 7)) for i in range(5000))),
        "bad.fq": "@a\nACGT\n+\nIIII\n@b\nACGT\n+\nIII\n@c\nAC\n+\nII\n",
    }
    for name, text in cases.items():
        path = str(tmp_path / name)
        open(path, "w").write(text)
        for p in (path, path + ".gz"):
            if p.endswith(".gz"):
                with gzip.open(p, "wb") as f:
                    f.write(text.encode())
            na = host.parse_file(p, str(tmp_path / "a.txt"), parallel=True, threads=7)
            nb = host.parse_file(p, str(tmp_path / "b.txt"), parallel=False)
            assert na == nb, (name, na, nb)
            assert open(tmp_path / "a.txt", "rb").read() == open(tmp_path / "b.txt", "rb").read(), name
    assert host.parse_file(str(tmp_path / "stop.fa"), str(tmp_path / "a.txt")) == 20000
    assert host.parse_file(str(tmp_path / "one.fa"), str(tmp_path / "a.txt")) == 30000
    assert host.parse_file(str(tmp_path / "one_stop.fa"), str(tmp_path / "a.txt")) == 12345
    assert host.parse_file(str(tmp_path / "one_noeol.fa"), str(tmp_path / "a.txt")) == 10000
    assert host.parse_file(str(tmp_path / "bad.fq"), str(tmp_path / "a.txt")) == 1


def test_gzip_writer_bytes_do_not_depend_on_the_call_pattern(tmp_path):
    import gzip
    import random
    rnd = random.Random(1)
    data = bytes(rnd.getrandbits(8) & 0x3F | 0x40 for _ in range(3 * (1 << 20) + 12345))
    outs = []
    for pieces in (1, 7, 1000):
        p = str(tmp_path / ("w%d.gz" % pieces))
        host.write_file(p, data, pieces=pieces)
        outs.append(open(p, "rb").read())
        assert gzip.decompress(outs[-1]) == data
    assert outs[0] == outs[1] == outs[2]


def test_index_algorithm_sais_gives_the_own_sentinel_order(tmp_path):
    """`siga index -a sais` = SAISBuilder (src/suffix_array_builder.cpp:31-172): suffixes compared as strings up to the end of
    their read (mkqs over the characters), ties by read index (SuffixIndexCmp, :194-199) -- every read's own sentinel, ordered
    by read index -- NOT the default "sais2" order.  The host sorter's files equal a Python restatement of that order (RL units
    with the 31-cap, .sai ids) on read sets with duplicates, substrings and mixed lengths; the CLI takes the option."""
    import subprocess
    import numpy as np
    from siga_amd import host
    from tests.fixtures import fixture
    from tests.test_gpu_index_build import _index_with_sentinels_ordered_by_read
    for name in ("dup", "tiny", "toy", "corner"):
        fx = fixture(name)
        prefix = str(tmp_path / name)
        host.index_file_sais(fx.fa, prefix, threads=3)
        for ext, rev in ((".bwt", False), (".rbwt", True)):
            runs, sai, nsym = _index_with_sentinels_ordered_by_read(fx.seqs, reverse=rev)
            raw = open(prefix + ext, "rb").read()
            assert raw[:2] == b"\xca\xca" and int.from_bytes(raw[10:18], "little") == nsym
            assert np.array_equal(np.frombuffer(raw[30:], dtype=np.uint8), runs), (name, ext)
            ids = [int(l.split()[0]) for l in open(prefix + (".rsai" if rev else ".sai")).read().split("\n")[3:] if l]
            assert ids == sai.tolist(), (name, ext)
        if name == "dup":  # differs from the order of record where reads repeat
            assert open(prefix + ".bwt", "rb").read() != open(fx.prefix + ".bwt", "rb").read()
    fx = fixture("tiny")
    cwd = str(tmp_path)
    assert subprocess.run([host.CLI_PATH, "index", "-a", "SAIS", "-t", "2", "-p", "cli", fx.fa], cwd=cwd).returncode == 0
    assert open(os.path.join(cwd, "cli.bwt"), "rb").read() == open(str(tmp_path / "tiny") + ".bwt", "rb").read()
    assert subprocess.run([host.CLI_PATH, "index", "-a", "ropebwt", fx.fa], cwd=cwd, capture_output=True).returncode == 255
    # reads with other bases are refused for this algorithm
    bad = os.path.join(cwd, "n.fa")
    open(bad, "w").write(">a\nACGTNACGT\n>b\nACGTT\n")
    assert subprocess.run([host.CLI_PATH, "index", "-a", "sais", bad], cwd=cwd, capture_output=True).returncode == 255


def test_bzip2_input_reads_like_plain_text(tmp_path):
    """Utils::ifstream opens ".bz2" through a bzip2 filter (src/utils.cpp:91-126): the parallel loader and the record-at-a-time
    reader give the same records from reads.fa.bz2 (one stream, and two concatenated streams) as from reads.fa; a truncated
    file fails to load instead of reading as a shorter set; `siga index` takes the file and names its outputs by the stem
    (src/utils.cpp:128-135)."""
    import bz2
    fx = fixture("tiny")
    text = open(fx.fa, "rb").read()
    cut = text.index(b">", len(text) // 2)
    variants = {"one.fa.bz2": bz2.compress(text), "two.fa.bz2": bz2.compress(text[:cut]) + bz2.compress(text[cut:])}
    plain = str(tmp_path / "plain.txt")
    n = host.parse_file(fx.fa, plain)
    assert n == len(fx.reads)
    for name, data in variants.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        for parallel in (True, False):
            out = str(tmp_path / (name + (".par" if parallel else ".ser")))
            assert host.parse_file(p, out, parallel=parallel) == n
            assert open(out, "rb").read() == open(plain, "rb").read()
    bad = str(tmp_path / "bad.fa.bz2")
    open(bad, "wb").write(variants["one.fa.bz2"][:-20])
    assert host.parse_file(bad, str(tmp_path / "bad.out")) < 0
    cwd = str(tmp_path)
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "-t", "2", "one.fa.bz2"], cwd=cwd).returncode == 0
    for ext in (".bwt", ".rbwt", ".sai", ".rsai"):
        assert open(os.path.join(cwd, "one" + ext), "rb").read() == open(fx.prefix + ext, "rb").read(), ext


def test_ini_file_sets_options_the_command_line_overrides(tmp_path):
    """-s/--ini=FILE (src/main.cpp:62-77): the file's key=value lines fill the option tree, the command line's options go over
    it.  `siga index --ini` with prefix / algorithm / threads in the file names its outputs by the file's prefix unless -p says
    otherwise; a file that cannot be read, or a line without '=', ends the run with status 1 before anything is written."""
    fx = fixture("tiny")
    cwd = str(tmp_path)
    ini = os.path.join(cwd, "siga.ini")
    open(ini, "w").write("; defaults of this project\nprefix = fromini\nalgorithm=sais\nthreads=2\nno-reverse=\n[other]\nprefix=ignored\n")
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "--ini", ini, fx.fa], cwd=cwd).returncode == 0
    assert os.path.exists(os.path.join(cwd, "fromini.bwt")) and not os.path.exists(os.path.join(cwd, "fromini.rbwt"))
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "-s", ini, "-a", "sais2", "-p", "cli", fx.fa], cwd=cwd).returncode == 0
    assert open(os.path.join(cwd, "cli.bwt"), "rb").read() == open(fx.prefix + ".bwt", "rb").read()  # -a sais2 won over the file
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "--ini=" + os.path.join(cwd, "missing.ini"), fx.fa], cwd=cwd,
                          capture_output=True).returncode == 1
    bad = os.path.join(cwd, "bad.ini")
    open(bad, "w").write("prefix x\n")
    assert subprocess.run([host.CLI_PATH, "index", "--cpu", "-s", bad, fx.fa], cwd=cwd, capture_output=True).returncode == 1


def test_sai_tables_parse_in_chunks_like_they_do_serially(tmp_path):
    """SuffixArray's text format (src/suffix_array.cpp:57-95: magic, strings, elems, then "<read> <j>" lines) is read before any
    device is touched.  Tables of 2^18 rows and more are parsed in chunks on the host's threads: a good table gets as far as the
    size check against the .bwt (which names both tables' row counts), a bad read id is reported with ITS row whichever chunk
    holds it, a cut-off body is refused, and a layout other than one pair per line is still read (serially)."""
    fx = fixture("tiny")
    n = 300_000
    rng = np.random.default_rng(5)
    ids = rng.permutation(n)

    def table(rows, elems=None, per_line=1):
        pairs = ["%d 0" % r for r in rows]
        body = "\n".join(" ".join(pairs[i:i + per_line]) for i in range(0, len(pairs), per_line))
        return "51914\n%d\n%d\n%s\n" % (n, len(rows) if elems is None else elems, body)

    def open_with(text, rtext=None):
        sai, rsai = str(tmp_path / "t.sai"), str(tmp_path / "t.rsai")
        open(sai, "w").write(text)
        open(rsai, "w").write(text if rtext is None else rtext)
        h = C.c_void_p()
        rc = _lib.lib().sigax_index_open((fx.prefix + ".bwt").encode(), (fx.prefix + ".rbwt").encode(), sai.encode(), rsai.encode(),
                                         0, C.byref(h))
        assert rc != 0 and not h.value
        return rc, _lib.last_error()

    good = table(ids)
    rc, msg = open_with(good)
    assert rc == -2 and "(%d, %d entries) do not match" % (n, n) in msg, msg  # SIGAX_E_IO from the size check: both parsed
    rc, msg = open_with(table(ids, per_line=2))
    assert rc == -2 and "(%d, %d entries) do not match" % (n, n) in msg, msg
    for row in (7, n // 2 + 3, n - 2):  # first chunk, a middle one, the last
        bad = ids.copy()
        bad[row] = n + 11
        rc, msg = open_with(table(bad))
        assert rc == -2 and "read id %d at row %d" % (n + 11, row) in msg, msg
        rc, msg = open_with(good, rtext=table(bad))  # the reverse table's error comes through from its thread
        assert rc == -2 and "read id %d at row %d" % (n + 11, row) in msg, msg
    rc, msg = open_with(table(ids[:n - 5], elems=n))
    assert rc == -2 and "truncated .sai body" in msg, msg


def test_line_coder_blocks_inflate_to_the_text(tmp_path):
    """Blocks that are mostly bases go through the writer's own deflate coder (line_deflate.hpp: matches against the previous
    line's same column, one dynamic Huffman code per block, CRC-32 by carry-less multiplication) instead of zlib.  Whatever
    the text looks like -- VT lines with names that grow a digit, FASTA, lines longer than deflate's 32 KiB window, one
    endless line, a single base repeated (one literal symbol), runs longer than the longest match, no line end at the end,
    block edges inside a line, FASTQ records (lines matched four lines up) -- gzip must give the text back, as ONE member, and the bytes must not depend on how the text
    was handed over.  SIGA_GZIP_LEVEL=6 sends everything through zlib (the reference's setting) and reads back the same."""
    import gzip
    import zlib
    rng = np.random.default_rng(11)

    def bases(n):
        return rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()

    genome = bases(200_000)

    def read(i, length=150):
        p = (i * 7919) % (len(genome) - length)
        return genome[p:p + length]

    texts = {
        "vt": b"".join(b"VT\tr%d\t%s\tSS:i:%d\n" % (i, read(i), i % 7 == 0) for i in range(9_990, 30_000)),
        "vt_ragged": b"".join(b"VT\tread/%d\t%s\tSS:i:0\n" % (i, read(i, 30 + (i * 37) % 220)) for i in range(20_000)),
        "fasta": b"".join(b">r%d some comment\n%s\n" % (i, read(i)) for i in range(20_000)),
        "fastq": b"".join(b"@r%d/1\n%s\n+\n%s\n" % (i, read(i), bytes(33 + (j * 7 + i) % 40 for j in range(150))) for i in range(12_000)),
        "long_lines": b"".join(b">c%d\n%s\n" % (i, bases(40_000 + 1000 * i)) for i in range(12)),
        "one_line": bases(1_500_000),
        "same_base": b"A" * 2_200_000,
        "same_line": (b"ACGTTGCAACGT" * 30 + b"\n") * 9_000 + b"ACGT",
        "short": b"ACGTACGTAC\n" * 9,
        "identical_reads": (b"VT\tx\t" + read(5) + b"\tSS:i:0\n") * 20_000,
    }
    for name, data in texts.items():
        outs = []
        for pieces in (1, 13):
            path = str(tmp_path / ("%s_%d.gz" % (name, pieces)))
            host.write_file(path, data, pieces)
            raw = open(path, "rb").read()
            d = zlib.decompressobj(16 + zlib.MAX_WBITS)
            assert d.decompress(raw) == data and d.eof and d.unused_data == b"", name
            outs.append(raw)
        assert outs[0] == outs[1], name
    # the coder is on: VT lines come out smaller than zlib's best makes them
    vt = texts["vt"]
    assert len(open(str(tmp_path / "vt_1.gz"), "rb").read()) < len(zlib.compress(vt, 9))
    code = ("import sys; sys.path.insert(0, %r); from siga_amd import host; host.write_file(sys.argv[1], open(sys.argv[2], 'rb').read(), 3)"
            % ROOT)
    src = str(tmp_path / "vt.txt")
    open(src, "wb").write(vt)
    ref = str(tmp_path / "vt_level6.gz")
    import sys
    assert subprocess.run([sys.executable, "-c", code, ref, src], env=dict(os.environ, SIGA_GZIP_LEVEL="6")).returncode == 0
    assert gzip.open(ref, "rb").read() == vt
    assert len(open(ref, "rb").read()) > len(open(str(tmp_path / "vt_1.gz"), "rb").read())


def test_host_read_table_ranks_names_in_string_order(tmp_path):
    """Hit2OverlapConverter's ReadInfo table (src/overlap_builder.cpp:333-343,358,365: edges are kept by comparing read NAMES
    with std::string's operator<): the host ranks all names with a sample sort on (first eight bytes, index) pairs.  Names that
    tie in their first eight bytes, names shorter than that, duplicates, bytes above 0x7f (unsigned order) and a set large
    enough for many buckets must rank exactly as Python sorts the byte strings."""
    rng = np.random.default_rng(2)
    n = 120_000
    names = []
    for i in range(n):
        kind = i % 6
        if kind == 0:
            names.append(b"r%d" % rng.integers(0, 50_000))           # duplicates, lengths 2..6
        elif kind == 1:
            names.append(b"sample_A/lane3/read%d" % i)               # long common prefix: every compare is a tie-break
        elif kind == 2:
            names.append(bytes(rng.integers(33, 127, size=rng.integers(1, 12), dtype=np.uint8)).replace(b">", b"x"))
        elif kind == 3:
            names.append(b"\xc3\xa9" + b"%d" % rng.integers(0, 1000))  # bytes >= 0x80 sort after ASCII
        elif kind == 4:
            names.append(b"read%07d" % (i // 2))                      # ties at exactly eight bytes and beyond
        else:
            names.append(b"x" * int(rng.integers(1, 20)))             # prefixes of each other
    lens = rng.integers(1, 40, size=n)
    fa = str(tmp_path / "names.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">%s some comment\n%s\n" % (nm, b"A" * int(l)) for nm, l in zip(names, lens)))
    out = str(tmp_path / "table.txt")
    for threads in (1, 5):
        assert host.read_table(fa, out, threads=threads) == n
        got = np.loadtxt(out, dtype=np.int64)
        order = {nm: k for k, nm in enumerate(sorted(set(names)))}
        assert got[:, 0].tolist() == [order[nm] for nm in names]
        assert got[:, 1].tolist() == lens.tolist()


@pytest.mark.slow
def test_sai_writer_slices_join_up(tmp_path):
    """The .sai text (src/suffix_array.cpp:17-44: magic, strings, elems, then "<read> 0" per row) is formatted in slices of
    2^20 rows on several threads and written in order: on a read set of more than 2^20 reads the file must hold every read
    once, in rows of that exact form, and the library must read it back (sizes agree with the .bwt)."""
    n, L = (1 << 20) + 12345, 12
    rng = np.random.default_rng(8)
    reads = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, L))
    fa = str(tmp_path / "many.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
    prefix = str(tmp_path / "many")
    host.index_file(fa, prefix, threads=4)
    for ext in (".sai", ".rsai"):
        lines = open(prefix + ext, "rb").read().split(b"\n")
        assert lines[:3] == [b"51914", b"%d" % n, b"%d" % n] and lines[-1] == b"" and len(lines) == n + 4
        rows = np.array([l.split(b" ") for l in lines[3:-1]])
        assert rows.shape == (n, 2) and np.all(rows[:, 1] == b"0")
        ids = rows[:, 0].astype(np.int64)
        assert np.array_equal(np.sort(ids), np.arange(n))


@pytest.mark.parametrize("gz", [False, True])
def test_asqg_text_formatters_without_a_gpu(tmp_path, gz):
    """The text side of OverlapBuilder::build (src/overlap_builder.cpp:291-329,345-375,423-509) on the CPU: parallel loader,
    VT lines with the comment tags, ED lines through the raw-pointer formatter sized for the longest names, the block-parallel
    output stream -- for random edge records and substring flags, against the Python mirror's text.  (What the sanitizer
    builds of tools/sanitize_host.sh exercise of the formatter: names of very different lengths, every flag combination.)"""
    import gzip
    import random
    from siga_amd import host
    from siga_amd.overlap import EDGE_DTYPE, format_asqg, read_sequences
    rnd = random.Random(7)
    n = 5000
    recs = []
    for i in range(n):
        name = "r%d" % i if i % 3 else "read_with_a_much_longer_name_%d_%s" % (i, "x" * rnd.randrange(0, 60))
        com = rnd.choice(["", " CR:i:%d" % rnd.randrange(100), " BX:Z:ACGT-1 EX:Z:foo", " free text here"])
        seq = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(30, 200)))
        recs.append(">%s%s\n%s\n" % (name, com, seq))
    fa = str(tmp_path / "r.fa")
    open(fa, "w").write("".join(recs))
    reads = read_sequences(fa)
    assert len(reads) == n
    k = 20000
    ed = np.zeros(k, dtype=EDGE_DTYPE)
    for j in range(k):
        q, t = rnd.randrange(n), rnd.randrange(n)
        ed[j] = (q, t, rnd.randrange(1, min(len(reads[q][2]), len(reads[t][2])) + 1), rnd.randrange(8))
    sub = np.array([rnd.random() < 0.1 for _ in range(n)], dtype=np.uint8)
    out = str(tmp_path / ("o.asqg.gz" if gz else "o.asqg"))
    assert host.format_asqg(fa, sub, ed, 45, out, threads=5) == n
    got = (gzip.open(out, "rb") if gz else open(out, "rb")).read().decode("latin-1")
    want = format_asqg(reads, {"substring": sub, "edges": ed}, 45)
    assert got == want


@pytest.mark.parametrize("threads,cap", [(1, None), (3, "1")])
def test_vt_lines_ahead_give_the_in_order_file(tmp_path, threads, cap, monkeypatch):
    """VT lines ahead of the batches (siga_host.cpp, VtAhead: text with SS:i:0 and its 1 MiB deflate blocks made before the
    substring flags are known; SIGA_VT_AHEAD=1) against the in-order path and the Python mirror: the .asqg.gz files are
    the same BYTES -- with clean stretches (blocks used as they are), stretches with substring reads (chunks formatted
    again, their blocks deflated by the writer), text taken in pieces that do not line up with the chunks, several waves with
    a carried tail, and a cap that keeps the threads one wave ahead of the writer."""
    import gzip
    from siga_amd import host
    from siga_amd.overlap import EDGE_DTYPE, format_asqg, read_sequences
    rng = np.random.default_rng(11)
    n = 90000
    codes = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 150))]
    lens = rng.integers(100, 151, size=n)
    fa = str(tmp_path / "r.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">r%d%s\n%s\n" % (i, b" CR:i:7" if i % 1000 == 0 else b"", codes[i, :lens[i]].tobytes()) for i in range(n)))
    sub = np.zeros(n, dtype=np.uint8)
    sub[[5, 4095, 4096, 30000, 30001, 61439, 89999]] = 1  # chunks 0, 1, 7, 14 and the last; the others stay as made ahead
    ed = np.zeros(3, dtype=EDGE_DTYPE)
    ed[:] = [(0, 1, 50, 0), (2, 3, 60, 1), (n - 1, 0, 45, 2)]
    monkeypatch.setenv("SIGA_BATCH_READS", "7001")
    monkeypatch.setenv("SIGA_VT_AHEAD", "1")
    if cap:
        monkeypatch.setenv("SIGA_VT_AHEAD_BYTES", cap)
    ahead = str(tmp_path / "a.asqg.gz")
    assert host.format_asqg(fa, sub, ed, 45, ahead, threads=threads) == n
    monkeypatch.setenv("SIGA_NO_VT_AHEAD", "1")
    plain = str(tmp_path / "p.asqg.gz")
    assert host.format_asqg(fa, sub, ed, 45, plain, threads=threads) == n
    a, p = open(ahead, "rb").read(), open(plain, "rb").read()
    assert a == p and len(a) > 1 << 20
    monkeypatch.setenv("SIGA_SYNC_WRITE", "1")  # the caller writes the file itself instead of the writer's thread
    sync = str(tmp_path / "s.asqg.gz")
    assert host.format_asqg(fa, sub, ed, 45, sync, threads=threads) == n
    assert open(sync, "rb").read() == p
    monkeypatch.delenv("SIGA_SYNC_WRITE")
    text = gzip.decompress(a).decode("latin-1")
    assert text.count("SS:i:1") == 7 and len(text) > 12 << 20
    assert text == format_asqg(read_sequences(fa), {"substring": sub, "edges": ed}, 45)
    # a file without .gz: the text alone is made ahead
    monkeypatch.delenv("SIGA_NO_VT_AHEAD")
    raw = str(tmp_path / "a.asqg")
    assert host.format_asqg(fa, sub, ed, 45, raw, threads=threads) == n
    assert open(raw, "rb").read().decode("latin-1") == text


@pytest.mark.parametrize("sync", [False, True])
@pytest.mark.parametrize("name", ["/dev/full", "/dev/full.gz"])
def test_a_short_write_is_an_error_whichever_thread_writes(name, sync, monkeypatch):
    """Utils::ofstream as used for the ASQG (src/utils.cpp:92-126; a failed stream makes OverlapBuilder::build return false,
    src/overlap_builder.cpp:425-427): a device that takes no bytes must surface as an error from close(), with the file written
    by the writer's own thread (default) or by the caller (SIGA_SYNC_WRITE=1), plain or gzip."""
    from siga_amd import host
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full here")
    if sync:
        monkeypatch.setenv("SIGA_SYNC_WRITE", "1")
    if name.endswith(".gz"):  # (a name the writer treats as gzip: a link to the device)
        import tempfile
        d = tempfile.mkdtemp()
        name = os.path.join(d, "full.gz")
        os.symlink("/dev/full", name)
    data = np.random.default_rng(3).integers(65, 90, size=5 << 20, dtype=np.uint8).tobytes()
    with pytest.raises(IOError):
        host.write_file(name, data, pieces=7)
