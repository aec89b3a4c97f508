"""Python mirror of the reference's overlap interface on top of the C-ABI (include/sigax.h).

Names follow the reference: `FMIndexPair.load(prefix)` stands for the two FMIndex::load calls of
Overlapping::run (src/overlap.cpp:41-42); `OverlapBuilder(fmi_pair, prefix, irreducible, rc)` mirrors
src/overlap_builder.h:19-45 with `overlap(reads, min_overlap)` (batched OverlapBuilder::overlap) and
`build(input, min_overlap, output)` (HT/VT/ED text as src/overlap_builder.cpp:423-509 writes it at -t 1).
All compute is in libsigax.so; nothing here computes overlaps on the CPU.
"""
import ctypes as C
import gzip
import os

import numpy as np

from . import _lib
from ._lib import BLOCK_DTYPE, EDGE_DTYPE, SIGAX_DUPLICATE, SIGAX_EDGES, SIGAX_IRREDUCIBLE, SIGAX_RC


class SigaxError(RuntimeError):
    def __init__(self, code, where):
        super().__init__("%s failed (%d): %s" % (where, code, _lib.last_error()))
        self.code = code


def _check(code, where):
    if code != 0:
        raise SigaxError(code, where)


def pack_reads(seqs):
    if isinstance(seqs, tuple):  # (uint8 array of concatenated bases, offsets u64[n+1]): BASELINE-sized sets, no copies
        buf, offs = seqs
        return np.ascontiguousarray(buf, dtype=np.uint8), np.ascontiguousarray(offs, dtype=np.uint64)
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


def name_ranks(names):
    """rank of each name under std::string operator< (bytewise), equal names share a rank."""
    bs = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    order = {b: i for i, b in enumerate(sorted(set(bs)))}
    return np.array([order[b] for b in bs], dtype=np.uint32)


def read_sequences(path):
    """FASTA/FASTQ reader with the reference's semantics (src/kseq.cpp:127-228): returns (name, comment, seq)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read().decode("latin-1")
    out = []
    ws = " \t\n\v\f\r"
    lines = [l.strip(ws) for l in data.split("\n")]
    lines = [l for l in lines if l]
    if not lines:
        return out

    def split_name(n):
        for i, ch in enumerate(n):
            if ch in " \t":
                return n[:i], n[i + 1:]
        return n, ""

    if data[:1] == ">":
        name, seq = None, ""
        for l in lines:
            if l.startswith(">"):
                if seq and name:
                    out.append(split_name(name) + (seq,))
                    seq = ""
                elif name:
                    return out
                name = l[1:]
            else:
                seq += l
        if seq and name:
            out.append(split_name(name) + (seq,))
    elif data[:1] == "@":
        i = 0
        while i + 4 <= len(lines):
            n, s, p, q = lines[i:i + 4]
            if not n.startswith("@") or not p.startswith("+") or len(q) != len(s):
                break
            if not (len(p) == 1 or p.endswith(n[1:])):
                break
            out.append(split_name(n[1:]) + (s,))
            i += 4
    return out


def _copy_records(ptr, n, dtype):
    """n records of `dtype` at the C pointer `ptr` -> an owned numpy array (ctypes.string_at stops at 2 GiB: one rank's shard
    of BASELINE configs[4] returns 27 M blocks of 80 bytes)"""
    out = np.empty(n, dtype=dtype)
    if n:
        C.memmove(out.ctypes.data, ptr, n * dtype.itemsize)
    return out


class FMIndexPair:
    """Both FM-indexes (+ .sai tables) resident on one GPU."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def load(cls, prefix, device=0, with_sai=True, resident=True):
        """resident: the index stays open for many more reads than it holds (tests, bench.py): sigax_index_prepare builds the
        extractor's row tables at once.  False = what one pass of `siga overlap` does: tables only after a full pass."""
        h = C.c_void_p()
        sai = (prefix + ".sai").encode() if with_sai else None
        rsai = (prefix + ".rsai").encode() if with_sai else None
        _check(_lib.lib().sigax_index_open((prefix + ".bwt").encode(), (prefix + ".rbwt").encode(), sai, rsai,
                                           device, C.byref(h)), "sigax_index_open")
        pair = cls(h.value)
        pair._prepare_pending = bool(resident)  # at set_reads() (the longest read is known then) or the first batch
        pair._resident = bool(resident)
        return pair

    @classmethod
    def from_memory(cls, runs, rruns, n_symbols, n_strings, sai=None, rsai=None, device=0, resident=True):
        runs = np.ascontiguousarray(runs, dtype=np.uint8)
        rruns = np.ascontiguousarray(rruns, dtype=np.uint8)
        h = C.c_void_p()
        ps = pr = None
        if sai is not None:
            sai = np.ascontiguousarray(sai, dtype=np.uint32)
            rsai = np.ascontiguousarray(rsai, dtype=np.uint32)
            ps, pr = sai.ctypes.data, rsai.ctypes.data
        _check(_lib.lib().sigax_index_open_mem(runs.ctypes.data, len(runs), rruns.ctypes.data, len(rruns), n_symbols,
                                               n_strings, ps, pr, device, C.byref(h)), "sigax_index_open_mem")
        pair = cls(h.value)
        pair._prepare_pending = bool(resident)  # at set_reads() (the longest read is known then) or the first batch
        pair._resident = bool(resident)
        return pair

    def close(self):
        if self._h:
            _lib.lib().sigax_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self):
        inf = _lib.IndexInfo()
        _check(_lib.lib().sigax_index_info_get(self._h, C.byref(inf)), "sigax_index_info_get")
        return {"n_symbols": inf.n_symbols, "n_strings": inf.n_strings, "device_bytes": inf.device_bytes,
                "pred": list(inf.pred), "device": inf.device, "wide": inf.wide}

    def set_reads(self, lengths, ranks):
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
        _check(_lib.lib().sigax_index_set_reads(self._h, lengths.ctypes.data, ranks.ctypes.data, len(lengths)),
               "sigax_index_set_reads")
        if getattr(self, "_prepare_pending", False):
            self.prepare()

    def prepare(self):
        """sigax_index_prepare: the extractor's row tables in place now"""
        self._prepare_pending = False
        _check(_lib.lib().sigax_index_prepare(self._h), "sigax_index_prepare")

    def prepare_overlap(self, min_overlap):
        """sigax_index_prepare_overlap: row tables + the block finder's deep start table for this minimum overlap"""
        self._prepare_pending = False
        self._deep_for = min(getattr(self, "_deep_for", 1 << 30), int(min_overlap))
        _check(_lib.lib().sigax_index_prepare_overlap(self._h, int(min_overlap)), "sigax_index_prepare_overlap")

    def check_order(self, which=0):
        """sigax_index_check_order: (pairs of adjacent BWT rows out of suffix order, first such row, undecided pairs)"""
        bad, first, und = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(_lib.lib().sigax_index_check_order(self._h, which, C.byref(bad), C.byref(first), C.byref(und)), "sigax_index_check_order")
        return int(bad.value), int(first.value), int(und.value)

    def occ(self, positions, which=0):
        """FMIndex::getOcc for many positions -> [n,5] ($,A,C,G,T)."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        out = np.zeros((len(pos), 5), dtype=np.uint64)
        _check(_lib.lib().sigax_occ_batch(self._h, which, pos.ctypes.data, len(pos), out.ctypes.data), "sigax_occ_batch")
        return out

    def kmer_counts(self, kmers):
        k = len(kmers[0])
        buf = b"".join(x.encode() if isinstance(x, str) else x for x in kmers)
        out = np.zeros(len(kmers), dtype=np.uint64)
        _check(_lib.lib().sigax_kmer_count_batch(self._h, buf, k, len(kmers), out.ctypes.data), "sigax_kmer_count_batch")
        return out


class OverlapBuilder:
    """src/overlap_builder.h:19-45 over the GPU library."""

    def __init__(self, fmi, prefix="default", irreducible=True, rc=True):
        self.fmi = fmi
        self.prefix = prefix
        self.irreducible = irreducible
        self.rc = rc

    def _flags(self, edges):
        return (SIGAX_IRREDUCIBLE if self.irreducible else 0) | (SIGAX_RC if self.rc else 0) | (SIGAX_EDGES if edges else 0)

    def duplicate(self, seqs, read_base=0, edges=False):
        """Batched OverlapBuilder::duplicate (src/overlap_builder.cpp:1184-1195)."""
        return self.overlap(seqs, 0, read_base, edges, _flags=SIGAX_DUPLICATE | (SIGAX_EDGES if edges else 0))

    def overlap(self, seqs, min_overlap, read_base=0, edges=False, _flags=None):
        """Batched OverlapBuilder::overlap.  Returns dict(block_offs, blocks, substring, edges, stats)."""
        buf, offs = pack_reads(seqs)
        if getattr(self.fmi, "_resident", False) and _flags is None and min_overlap < getattr(self.fmi, "_deep_for", 1 << 30):
            self.fmi.prepare_overlap(min_overlap)  # an index that stays open: tables for this minimum overlap (and larger ones)
        elif getattr(self.fmi, "_prepare_pending", False):
            self.fmi.prepare()
        res = _lib.Result()
        if isinstance(buf, np.ndarray):
            buf = C.c_char_p(buf.ctypes.data) if buf.size else b""
        _check(_lib.lib().sigax_overlap_batch(self.fmi.handle, buf, offs.ctypes.data, len(offs) - 1, read_base, min_overlap,
                                              self._flags(edges) if _flags is None else _flags, C.byref(res)),
               "sigax_overlap_batch")
        try:
            n = res.n_reads
            block_offs = np.ctypeslib.as_array(res.block_offs, shape=(n + 1,)).copy()
            nb = int(block_offs[-1])
            blocks = _copy_records(res.blocks, nb, BLOCK_DTYPE)
            substring = np.ctypeslib.as_array(res.substring, shape=(max(n, 1),))[:n].copy()
            ne = int(res.n_edges)
            eds = _copy_records(res.edges, ne, EDGE_DTYPE)
            stats = res.stats.as_dict()
        finally:
            _lib.lib().sigax_result_free(C.byref(res))
        return {"block_offs": block_offs, "blocks": blocks, "substring": substring, "edges": eds, "stats": stats}

    def build(self, input_path, min_overlap, output_path=None):
        """HT + VT + ED text exactly as OverlapBuilder::build writes it at -t 1 (src/overlap_builder.cpp:423-483).
        Returns the text; also writes it to output_path (gz if it ends with .gz)."""
        reads = read_sequences(input_path)
        names = [r[0] for r in reads]
        seqs = [r[2] for r in reads]
        lengths = np.array([len(s) for s in seqs], dtype=np.uint32)
        self.fmi.set_reads(lengths, name_ranks(names))
        res = self.overlap(seqs, min_overlap, 0, edges=True)
        text = format_asqg(reads, res, min_overlap)
        if output_path:
            opener = gzip.open if output_path.endswith(".gz") else open
            with opener(output_path, "wb") as f:
                f.write(text.encode("latin-1"))
        return text, res


def edge_coords(length, af, qlen, tlen):
    """OverlapBlock::overlap (src/overlap_builder.cpp:158-175)."""
    s0, e0 = qlen - length, qlen - 1
    s1, e1 = 0, length - 1
    if af & 1:
        s0, e0 = qlen - e0 - 1, qlen - s0 - 1
    if af & 2:
        s1, e1 = tlen - e1 - 1, tlen - s1 - 1
    return s0, e0, s1, e1


def _vertex_tags(comment):
    """OverlapPostProcess (src/overlap_builder.cpp:304-316) + VertexRecord << (src/asqg.cpp:171-186)."""
    cov = bar = ext = None
    if comment:
        for tok in comment.split(" "):
            parts = tok.split(":")
            if tok.startswith("BX"):
                if len(parts) == 3 and parts[1] == "Z":
                    w = parts[2].split()
                    bar = w[0] if w else ""
            elif tok.startswith("CR"):
                if len(parts) == 3 and parts[1] == "i":
                    cov = _parse_int(parts[2])
            elif tok.startswith("EX"):
                if len(parts) == 3 and parts[1] == "Z":
                    w = parts[2].split()
                    ext = w[0] if w else ""
    out = ""
    if cov is not None:
        out += "\tCR:i:%d" % cov
    if bar is not None:
        out += "\tBX:Z:%s" % bar
    if ext is not None:
        out += "\tEX:Z:%s" % ext
    return out


def _parse_int(s):
    """std::istream >> int: optional leading whitespace, sign, digits; 0 on failure; clamped on overflow."""
    s = s.lstrip(" \t\n\v\f\r")
    i = 0
    if i < len(s) and s[i] in "+-":
        i += 1
    j = i
    while j < len(s) and s[j].isdigit():
        j += 1
    if j == i:
        return 0
    v = int(s[:j])
    return max(-2**31, min(2**31 - 1, v))


def format_asqg(reads, res, min_overlap):
    lines = ["HT\tVN:i:1\tOL:i:%d\tCN:i:1" % min_overlap]
    sub = res["substring"]
    for i, (name, comment, seq) in enumerate(reads):
        lines.append("VT\t%s\t%s\tSS:i:%d%s" % (name, seq, 1 if sub[i] else 0, _vertex_tags(comment)))
    lens = [len(r[2]) for r in reads]
    for e in res["edges"]:
        q, t, ln, af = int(e["query"]), int(e["target"]), int(e["length"]), int(e["af"])
        s0, e0, s1, e1 = edge_coords(ln, af, lens[q], lens[t])
        lines.append("ED\t%s %s %d %d %d %d %d %d %d 0" % (reads[q][0], reads[t][0], s0, e0, lens[q], s1, e1, lens[t],
                                                          1 if af & 4 else 0))
    return "\n".join(lines) + "\n"


def format_hits(res):
    """Hits text (src/overlap_builder.cpp:234-241), one line per read."""
    out = []
    offs, b = res["block_offs"], res["blocks"]
    for r in range(len(offs) - 1):
        lo, hi = int(offs[r]), int(offs[r + 1])
        parts = ["%d %d %d " % (r, 1 if res["substring"][r] else 0, hi - lo)]
        for k in range(lo, hi):
            x = b[k]
            af = int(x["af"])
            parts.append("%d %d %d %d %d %d %d %d %d %d%d%d " % (
                x["capped0_lo"], x["capped0_hi"], x["capped1_lo"], x["capped1_hi"], x["raw0_lo"], x["raw0_hi"],
                x["raw1_lo"], x["raw1_hi"], x["length"], (af >> 2) & 1, (af >> 1) & 1, af & 1))
        out.append("".join(parts))
    return "\n".join(out) + "\n"
