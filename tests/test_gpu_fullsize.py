"""BASELINE.json configs[1] at full size (1M x 150 bp from a 5 Mb genome, m = 45) on the GPU: size-independent
properties, since the oracle takes minutes there -- idempotence, shard invariance (the multi-GPU decomposition),
structural invariants of the output, and an oracle spot-check on a random sample of reads against the full index."""
import os

import numpy as np
import pytest

from tests.fixtures import CACHE

pytestmark = pytest.mark.gpu

N, G, L, M, SEED = 1000000, 5000000, 150, 45, 1


@pytest.fixture(scope="module")
def c2():
    import siga_amd
    from siga_amd import host
    from tests.golden.make_reads import fast_reads
    reads, _ = fast_reads(G, L, N, SEED)
    d = os.path.join(CACHE, "c2")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "reads")
    if not all(os.path.exists(prefix + e) for e in (".bwt", ".rbwt", ".sai", ".rsai")):
        host.index_build(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64), prefix, threads=2)
    pair = siga_amd.FMIndexPair.load(prefix)
    names = np.char.add("r", np.arange(N).astype(str))
    order = np.argsort(names, kind="stable")
    rank = np.empty(N, dtype=np.uint32)
    rank[order] = np.arange(N, dtype=np.uint32)
    pair.set_reads(np.full(N, L, dtype=np.uint32), rank)
    return {"reads": reads, "pair": pair, "prefix": prefix, "rank": rank, "sa": siga_amd}


def _run(c2, lo, hi):
    b = c2["sa"].OverlapBuilder(c2["pair"])
    return b.overlap([bytes(r) for r in c2["reads"][lo:hi]], M, read_base=lo, edges=True)


def test_fullsize_invariants_idempotence_and_shards(c2):
    full = _run(c2, 0, N)
    again = _run(c2, 0, N)
    for k in ("block_offs", "blocks", "substring", "edges"):
        assert full[k].tobytes() == again[k].tobytes(), k  # idempotent, deterministic order
    offs = full["block_offs"]
    assert offs[0] == 0 and offs[-1] == len(full["blocks"]) and np.all(np.diff(offs.astype(np.int64)) >= 0)
    b = full["blocks"]
    # every read matches itself in both indexes: two length-L blocks, flags 000 and 011
    # (plus 101 / 110 ones when the read's reverse complement is itself a read)
    first = b[offs[:-1].astype(np.int64)]
    assert np.all(first["length"] == L) and np.all(first["af"] == 0)
    owner = np.repeat(np.arange(N), np.diff(offs.astype(np.int64)))
    for af in (0, 3):
        sel = (b["length"] == L) & (b["af"] == af)
        assert sel.sum() == N and np.array_equal(owner[sel], np.arange(N))
    assert not full["substring"].any()  # unique (pos, strand) draws: no read is a substring of another
    rest = b[b["length"] != L]
    assert rest["length"].min() >= M and rest["length"].max() < L
    assert np.all(b["capped0_hi"] >= b["capped0_lo"]) and np.all(b["capped0_hi"] < N)
    e = full["edges"]
    assert np.all(c2["rank"][e["query"]] > c2["rank"][e["target"]])  # dedup rule, overlap_builder.cpp:365
    assert np.all(np.diff(e["query"].astype(np.int64)) >= 0)          # hits order = read order
    assert full["stats"]["n_slow_reads"] < N // 100
    # shard invariance: 3 uneven contiguous shards with read_base give the same blocks and edges
    cuts = [0, 333333, 600001, N]
    parts = [_run(c2, cuts[i], cuts[i + 1]) for i in range(3)]
    assert np.concatenate([p["blocks"] for p in parts]).tobytes() == full["blocks"].tobytes()
    assert np.concatenate([p["edges"] for p in parts]).tobytes() == full["edges"].tobytes()
    assert sum(p["stats"]["n_occ_find"] + p["stats"]["n_occ_extract"] for p in parts) == \
        full["stats"]["n_occ_find"] + full["stats"]["n_occ_extract"]
    c2["full"] = full


def test_fullsize_sample_against_oracle(c2):
    from oracle import pyoracle as po
    full = c2.get("full") or _run(c2, 0, N)
    fwd = po.Index.load(c2["prefix"] + ".bwt", c2["prefix"] + ".sai")
    rev = po.Index.load(c2["prefix"] + ".rbwt", c2["prefix"] + ".rsai")
    rng = np.random.default_rng(7)
    cols = ["capped0_lo", "capped0_hi", "capped1_lo", "capped1_hi", "raw0_lo", "raw0_hi", "raw1_lo", "raw1_hi", "length", "af"]
    offs = full["block_offs"]
    for r in rng.integers(0, N, 1500):
        want, sub, _, _ = po.overlap(fwd, rev, bytes(c2["reads"][r]), M)
        got = full["blocks"][int(offs[r]):int(offs[r + 1])]
        assert [[int(x[c]) for c in cols] for x in got] == [list(map(int, w)) for w in want], int(r)
        assert bool(full["substring"][r]) == sub
