"""Shared fixture builders: read sets + oracle-built index files in a cache directory."""
import hashlib
import json
import os

from oracle import pyoracle as po
from tests.golden import make_reads as mr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_appc.json")))
CACHE = os.environ.get("SIGA_TEST_CACHE", os.path.join(ROOT, "build", "test_cache"))


def md5_prefix(data):
    if isinstance(data, str):
        data = data.encode()
    return hashlib.md5(data).hexdigest()[:8]


def named_reads(name):
    if name == "corner":
        return [tuple(x) for x in GOLDEN["corner"]["reads"]]
    if name == "rep":
        return mr.rep_reads()
    if name == "dup":
        return mr.dup_reads()
    if name in ("toy", "mid"):
        return mr.survey_reads(*GOLDEN[name]["gen"])
    if name == "deep":   # 160x coverage: about 110 blocks per (read, side), beyond one wave's 64 lanes
        return mr.survey_reads(2500, 100, 4000, 21)
    if name == "tiny":   # 400 x 60 bp from 2 kb, 12x
        return mr.survey_reads(2000, 60, 400, 99)
    if name in ("ragged", "ragged_n"):  # mixed lengths incl. reads shorter than min-overlap, duplicates
        base = mr.survey_reads(1500, 50, 200, 5)
        out = []
        for i, (n, s) in enumerate(base):
            if i % 7 == 0:
                s = s[:20 + (i % 25)]
            if name == "ragged_n" and i % 31 == 0:
                # non-ACGT bytes rank as '$' (alphabet.h:19-39).  Block lists stay well defined; turning them into
                # edges indexes the .sai table out of range in the reference, so only blocks are compared.
                s = s[:10] + "N" + s[11:]
            out.append((n, s))
        out.append(("dupA", base[3][1]))
        out.append(("x1", "A"))
        out.append(("x2", "ACGTACGTACGT"))
        return out
    raise KeyError(name)


class Fixture:
    """Reads + index files (<dir>/<name>.{fa,bwt,rbwt,sai,rsai}) built by the oracle's index builder."""

    def __init__(self, name):
        self.name = name
        self.reads = named_reads(name)
        self.dir = os.path.join(CACHE, name)
        os.makedirs(self.dir, exist_ok=True)
        self.prefix = os.path.join(self.dir, name)
        self.fa = self.prefix + ".fa"
        seqs = [s for _, s in self.reads]
        if not all(os.path.exists(self.prefix + e) for e in (".fa", ".bwt", ".rbwt", ".sai", ".rsai")):
            with open(self.fa, "w") as f:
                f.write(mr.fasta_text(self.reads))
            fwd = po.Index.build(seqs)
            rev = po.Index.build(seqs, reverse=True)
            fwd.save(self.prefix + ".bwt", self.prefix + ".sai")
            rev.save(self.prefix + ".rbwt", self.prefix + ".rsai")
        self.fwd = po.Index.load(self.prefix + ".bwt", self.prefix + ".sai")
        self.rev = po.Index.load(self.prefix + ".rbwt", self.prefix + ".rsai")

    @property
    def seqs(self):
        return [s for _, s in self.reads]

    def oracle_asqg(self, m, irreducible=True, rc=True, hits=False):
        tag = "%d%s%s" % (m, "" if irreducible else "x", "" if rc else "n")
        out = self.prefix + "." + tag + ".oracle.asqg"
        hp = self.prefix + "." + tag + ".oracle.hits" if hits else ""
        st = po.build_asqg(self.fwd, self.rev, self.fa, m, out, hp, irreducible, rc)
        return open(out).read(), (open(hp).read() if hits else None), st


_FIX = {}


def fixture(name):
    if name not in _FIX:
        _FIX[name] = Fixture(name)
    return _FIX[name]


def ed_lines(asqg_text):
    return [l[3:] for l in asqg_text.split("\n") if l.startswith("ED\t")]


def non_ed_lines(asqg_text):
    return [l for l in asqg_text.split("\n") if l and not l.startswith("ED\t")]
