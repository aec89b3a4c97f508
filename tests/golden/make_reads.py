"""Deterministic read-set generators for fixtures, tests and the bench.

`survey_reads(G, L, N, seed)` reproduces the generator the survey used for its `toy` / `mid`
fixtures (Python `random`, seeds recorded there), so the md5 prefixes
recorded in that appendix can be checked.  `corner`, `rep` and `dup` are the hand-made fixtures
described in the same appendix.  `fast_reads` is the numpy generator bench.py uses at BASELINE sizes
(uniform i.i.d. genome, uniform positions, strand flipped with p=0.5, unique
(pos,strand), names r<i>).
"""
import random

import numpy as np

_TR = str.maketrans("ACGT", "TGCA")


def revcomp(s):
    return s.translate(_TR)[::-1]


def survey_reads(G, L, N, seed):
    random.seed(seed)
    g = "".join(random.choice("ACGT") for _ in range(G))
    seen = set()
    out = []
    while len(out) < N:
        p = random.randrange(0, G - L + 1)
        s = random.random() < 0.5
        if (p, s) in seen:
            continue
        seen.add((p, s))
        r = g[p:p + L]
        if s:
            r = revcomp(r)
        out.append(("r%d" % len(out), r))
    return out


def fasta_text(named_reads):
    return "".join(">%s\n%s\n" % (n, s) for n, s in named_reads)


def corner_reads():
    """6 hand-written 30-mers: b overlaps a by 21, c == a, d is a 16-mer substring of a, e == revcomp(b), f."""
    random.seed(4242)
    g = "".join(random.choice("ACGT") for _ in range(80))
    a = g[0:30]
    b = g[9:39]
    c = a
    d = a[5:21]
    e = revcomp(b)
    f = revcomp(g[22:52])
    return [("a", a), ("b", b), ("c", c), ("d", d), ("e", e), ("f", f)]


def rep_reads():
    """SURVEY.md App. C `rep`: exercises SubMaximalBlockFilter::resolve's re-mapping path."""
    random.seed(5)

    def rnd(n):
        return "".join(random.choice("ACGT") for _ in range(n))

    P = "ACGTTGCAAG"
    s0 = rnd(40) + P * 4
    s1 = P * 3 + rnd(50)
    s2 = P + rnd(70)
    s3 = rnd(30) + s0[:50]
    s4 = P * 2 + rnd(60)
    return [("s0", s0), ("s1", s1), ("s2", s2), ("s3", s3), ("s4", s4)]


def dup_reads():
    """SURVEY.md App. C `dup`: heavy duplicates + substrings, variable length."""
    random.seed(77)
    g = "".join(random.choice("ACGT") for _ in range(60))
    out = []
    for i in range(300):
        L = random.choice([12, 16, 20])
        p = random.randrange(0, 60 - L + 1)
        out.append(("q%d" % i, g[p:p + L]))
    return out


def fast_reads(G, L, N, seed, by_position=False, subset=None):
    """numpy generator for BASELINE-sized sets -> (uint8 array [N, L] of ASCII, genome array).  by_position: the same reads
    in genome order instead of random order (a measurement aid: what locality between neighbouring reads would be worth).
    subset=(lo, hi): only reads lo..hi-1 of the same set (one rank's shard, without materialising the other ranks'); or an
    index array: those reads, in that order."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    npos = G - L + 1
    if 2 * npos < N:
        raise ValueError("not enough distinct (pos,strand) draws")
    # unique (pos, strand) keys in random order: oversample, de-duplicate, shuffle, cut
    keys = np.unique(rng.integers(0, 2 * npos, size=int(N * 1.25) + 64, dtype=np.int64))
    while len(keys) < N:
        keys = np.unique(np.concatenate([keys, rng.integers(0, 2 * npos, size=N, dtype=np.int64)]))
    rng.shuffle(keys)
    keys = keys[:N]
    if by_position:
        keys = np.sort(keys)
    if subset is not None:
        keys = keys[subset] if isinstance(subset, np.ndarray) else keys[subset[0]:subset[1]]
    pos = (keys >> 1).astype(np.int64)
    strand = (keys & 1).astype(bool)
    codes = np.lib.stride_tricks.sliding_window_view(genome, L)[pos]  # [N, L] copy
    codes[strand] = (3 - codes[strand])[:, ::-1]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return lut[codes], genome


def rank_of_r_names(n):
    """rank of the name "r<i>" under std::string operator< for i in [0, n): decimal strings compare like their digits
    left-aligned, a proper prefix first (sorting 2e7 Python strings would take a minute)."""
    i = np.arange(n, dtype=np.int64)
    ndig = np.ones(n, dtype=np.int64)
    p = 10
    while p <= n:
        ndig += i >= p
        p *= 10
    D = int(ndig.max()) if n else 1
    key = i * (10 ** (D - ndig))
    order = np.lexsort((ndig, key))
    rank = np.empty(n, dtype=np.uint32)
    rank[order] = np.arange(n, dtype=np.uint32)
    return rank


def substitute(reads, rate, seed):
    """ASCII read array [N, L] -> copy with each base replaced by a different one with probability `rate`."""
    rng = np.random.default_rng(seed)
    code = np.zeros(256, dtype=np.uint8)
    code[np.frombuffer(b"ACGT", dtype=np.uint8)] = np.arange(4, dtype=np.uint8)
    c = code[reads]
    hit = rng.random(reads.shape) < rate
    c = np.where(hit, (c + rng.integers(1, 4, size=reads.shape, dtype=np.uint8)) & 3, c)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[c]
