"""ctypes driver for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (siga_amd/) never does.  See oracle/siga_oracle.hpp for the parity-pin status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, u64, cp = C.c_void_p, C.c_uint64, C.c_char_p
        pu64 = C.POINTER(C.c_uint64)
        L.orc_index_build.restype = vp
        L.orc_index_build.argtypes = [cp, pu64, u64, C.c_int]
        L.orc_index_load.restype = vp
        L.orc_index_load.argtypes = [cp, cp]
        L.orc_index_save.argtypes = [vp, cp, cp]
        L.orc_index_free.argtypes = [vp]
        for f in ("orc_length", "orc_nruns", "orc_nstrings", "orc_sai_size"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [vp]
        L.orc_runs.restype = C.POINTER(C.c_uint8)
        L.orc_runs.argtypes = [vp]
        L.orc_sai.restype = C.POINTER(C.c_uint32)
        L.orc_sai.argtypes = [vp]
        L.orc_occ.argtypes = [vp, u64, pu64]
        L.orc_pred.argtypes = [vp, pu64]
        L.orc_getchar.argtypes = [vp, u64]
        L.orc_occurrences.restype = u64
        L.orc_occurrences.argtypes = [vp, cp, u64]
        L.orc_overlap.restype = C.c_int64
        L.orc_overlap.argtypes = [vp, vp, cp, u64, u64, C.c_int, C.c_int, pu64, u64, C.POINTER(C.c_int), pu64]
        L.orc_build_asqg.argtypes = [vp, vp, cp, u64, C.c_int, C.c_int, cp, cp, pu64]
        L.orc_build_asqg_mt.argtypes = [vp, vp, cp, u64, C.c_int, C.c_int, cp, C.c_int, C.POINTER(C.c_double)]
        L.orc_rmdup.argtypes = [vp, vp, cp, cp, cp]
        L.orc_correct.argtypes = [vp, cp, cp, u64, C.c_int, u64, u64, pu64]
        L.orc_overlap_batch_timed.restype = C.c_double
        L.orc_overlap_batch_timed.argtypes = [vp, vp, cp, pu64, u64, u64, C.c_int, C.c_int, C.c_int, pu64]
        L.orc_overlap_batch.restype = C.c_int64
        L.orc_overlap_batch.argtypes = [vp, vp, vp, pu64, u64, u64, C.c_int, C.c_int, C.c_int, C.c_int, pu64, pu64, u64,
                                        C.POINTER(C.c_uint8), pu64]
        L.orc_rl_encode.restype = u64
        L.orc_rl_encode.argtypes = [cp, u64, C.POINTER(C.c_uint8), u64]
        L.orc_stem.argtypes = [cp, C.c_char_p, u64]
        L.orc_revcomp.argtypes = [cp, u64, C.c_int, C.c_char_p]
        L.orc_tag_roundtrip.argtypes = [cp, C.c_int, cp, C.c_char_p, u64]
        _LIB = L
    return _LIB


def pack_reads(reads):
    """list of str/bytes -> (concatenated bytes, offsets u64[n+1])"""
    bs = [r.encode() if isinstance(r, str) else r for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Index:
    """One strand's index (RL-BWT + markers + .sai table) held by the oracle."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index handle is NULL")
        self.h = handle

    @classmethod
    def build(cls, reads, reverse=False):
        seqs, offs = pack_reads(reads)
        return cls(lib().orc_index_build(seqs, _p64(offs), len(reads), 1 if reverse else 0))

    @classmethod
    def load(cls, bwt_path, sai_path=""):
        return cls(lib().orc_index_load(bwt_path.encode(), sai_path.encode()))

    def save(self, bwt_path, sai_path):
        if lib().orc_index_save(self.h, bwt_path.encode(), sai_path.encode()) != 0:
            raise IOError("oracle index save failed")

    def __del__(self):
        try:
            if self.h:
                lib().orc_index_free(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return lib().orc_length(self.h)

    @property
    def nstrings(self):
        return lib().orc_nstrings(self.h)

    def runs(self):
        n = lib().orc_nruns(self.h)
        return np.ctypeslib.as_array(lib().orc_runs(self.h), shape=(n,)).copy() if n else np.zeros(0, np.uint8)

    def sai(self):
        n = lib().orc_sai_size(self.h)
        return np.ctypeslib.as_array(lib().orc_sai(self.h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def occ(self, i):
        out = np.zeros(5, dtype=np.uint64)
        lib().orc_occ(self.h, C.c_uint64(i & 0xFFFFFFFFFFFFFFFF), _p64(out))
        return out

    def pred(self):
        out = np.zeros(5, dtype=np.uint64)
        lib().orc_pred(self.h, _p64(out))
        return out

    def getchar(self, i):
        return chr(lib().orc_getchar(self.h, i))

    def occurrences(self, w):
        w = w.encode() if isinstance(w, str) else w
        return lib().orc_occurrences(self.h, w, len(w))


def overlap(fwd, rev, seq, min_overlap, irreducible=True, rc=True, cap=4096):
    """OverlapBuilder::overlap for one read -> (blocks[k,10] u64, substring, occ_calls, n_occ_min)."""
    seq = seq.encode() if isinstance(seq, str) else seq
    out = np.zeros((cap, 10), dtype=np.uint64)
    sub = C.c_int(0)
    st = np.zeros(2, dtype=np.uint64)
    k = lib().orc_overlap(fwd.h, rev.h, seq, len(seq), min_overlap, int(irreducible), int(rc), _p64(out), cap,
                          C.byref(sub), _p64(st))
    if k > cap:
        return overlap(fwd, rev, seq, min_overlap, irreducible, rc, cap=int(k))
    return out[:k].copy(), bool(sub.value), int(st[0]), int(st[1])


def build_asqg(fwd, rev, reads_path, min_overlap, asqg_path, hits_path="", irreducible=True, rc=True):
    st = np.zeros(3, dtype=np.uint64)
    r = lib().orc_build_asqg(fwd.h, rev.h, reads_path.encode(), min_overlap, int(irreducible), int(rc),
                             asqg_path.encode(), hits_path.encode(), _p64(st))
    if r != 0:
        raise RuntimeError("orc_build_asqg failed: %d" % r)
    return {"occ_calls": int(st[0]), "n_occ_min": int(st[1]), "blocks": int(st[2])}


def build_asqg_mt(fwd, rev, reads_path, min_overlap, asqg_path, threads=0, irreducible=True, rc=True):
    """`siga overlap -t N` end to end on the CPU: returns seconds of {parse, overlap (OpenMP), VT + ED text}."""
    secs = (C.c_double * 3)()
    r = lib().orc_build_asqg_mt(fwd.h, rev.h, reads_path.encode(), min_overlap, int(irreducible), int(rc), asqg_path.encode(),
                                threads, secs)
    if r != 0:
        raise RuntimeError("orc_build_asqg_mt failed: %d" % r)
    return {"parse": secs[0], "overlap": secs[1], "text": secs[2]}


def rmdup(fwd, rev, reads_path, fasta_path, dups_path):
    r = lib().orc_rmdup(fwd.h, rev.h, reads_path.encode(), fasta_path.encode(), dups_path.encode())
    if r != 0:
        raise RuntimeError("orc_rmdup failed: %d" % r)


def correct(fwd, reads_path, out_path, k=31, threshold=3, rounds=10, offset=1):
    st = np.zeros(2, dtype=np.uint64)
    r = lib().orc_correct(fwd.h, reads_path.encode(), out_path.encode(), k, threshold, rounds, offset, _p64(st))
    if r != 0:
        raise RuntimeError("orc_correct failed: %d" % r)
    return {"written": int(st[0]), "changed": int(st[1])}


def overlap_batch_timed(fwd, rev, reads, min_overlap, irreducible=True, rc=True, threads=0):
    seqs, offs = pack_reads(reads)
    out = np.zeros(3, dtype=np.uint64)
    sec = lib().orc_overlap_batch_timed(fwd.h, rev.h, seqs, _p64(offs), len(reads), min_overlap, int(irreducible),
                                        int(rc), threads, _p64(out))
    return sec, {"blocks": int(out[0]), "substring": int(out[1]), "n_occ_min": int(out[2])}


def overlap_batch(fwd, rev, reads, min_overlap, irreducible=True, rc=True, duplicate=False, threads=0, cap=None):
    """OverlapBuilder::overlap (or ::duplicate) for many reads at once, OpenMP over reads.
    reads: list of str/bytes, or (uint8 array of concatenated bases, offsets u64[n+1]).
    Returns dict(block_offs u64[n+1], blocks u64[k,10], substring u8[n], n_occ_min)."""
    if isinstance(reads, tuple):
        buf, offs = reads
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        ptr = buf.ctypes.data
    else:
        seqs, offs = pack_reads(reads)
        keep = C.create_string_buffer(seqs, len(seqs) + 1)
        ptr = C.addressof(keep)
    n = len(offs) - 1
    cap = cap or max(16 * n, 1024)
    block_offs = np.zeros(n + 1, dtype=np.uint64)
    blocks = np.zeros((cap, 10), dtype=np.uint64)
    sub = np.zeros(max(n, 1), dtype=np.uint8)
    nmin = np.zeros(1, dtype=np.uint64)
    k = lib().orc_overlap_batch(fwd.h, rev.h, ptr, _p64(offs), n, min_overlap, int(irreducible), int(rc), int(duplicate), threads,
                                _p64(block_offs), _p64(blocks), cap, sub.ctypes.data_as(C.POINTER(C.c_uint8)), _p64(nmin))
    if k > cap:
        return overlap_batch(fwd, rev, reads, min_overlap, irreducible, rc, duplicate, threads, cap=int(k))
    return {"block_offs": block_offs, "blocks": blocks[:k].copy(), "substring": sub[:n], "n_occ_min": int(nmin[0])}


def max_threads():
    return lib().orc_max_threads()
