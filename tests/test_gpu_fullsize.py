"""BASELINE.json configs[1] at full size (1M x 150 bp from a 5 Mb genome, m = 45) on the GPU: idempotence, shard
invariance (the multi-GPU decomposition), structural invariants of the output, and EVERY read's block list against the
oracle (OpenMP over the host cores), plus the edge records against a numpy restatement of the converter."""
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
        host.index_build_gpu(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64), prefix)
    pair = siga_amd.FMIndexPair.load(prefix)
    names = np.char.add("r", np.arange(N).astype(str))
    order = np.argsort(names, kind="stable")
    rank = np.empty(N, dtype=np.uint32)
    rank[order] = np.arange(N, dtype=np.uint32)
    pair.set_reads(np.full(N, L, dtype=np.uint32), rank)
    return {"reads": reads, "pair": pair, "prefix": prefix, "rank": rank, "sa": siga_amd}


def _run(c2, lo, hi):
    b = c2["sa"].OverlapBuilder(c2["pair"])
    return b.overlap((c2["reads"][lo:hi].reshape(-1), np.arange(0, (hi - lo + 1) * L, L, dtype=np.uint64)), M, read_base=lo, edges=True)


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


def test_fullsize_every_read_against_oracle(c2):
    from oracle import pyoracle as po
    from tests.bigcheck import assert_same_blocks, edges_matrix, expected_edges
    full = c2.get("full") or _run(c2, 0, N)
    fwd = po.Index.load(c2["prefix"] + ".bwt", c2["prefix"] + ".sai")
    rev = po.Index.load(c2["prefix"] + ".rbwt", c2["prefix"] + ".rsai")
    want = po.overlap_batch(fwd, rev, (c2["reads"].reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64)), M)
    assert_same_blocks(full, want, "C2")
    # edge records: the converter's rule restated in numpy over the ORACLE's blocks
    exp = expected_edges(want["blocks"], want["block_offs"], fwd.sai(), rev.sai(), np.full(N, L, dtype=np.uint32), c2["rank"])
    assert np.array_equal(edges_matrix(full["edges"]), exp)
    assert len(exp) > N  # about 1.1 irreducible edges per read at 30x


def test_row_tables_come_with_reuse_not_with_the_first_pass(c2):
    """An index opened through the plain C-ABI call builds the extractor's row tables only once it has been asked for as
    many reads as it holds -- one pass of `siga overlap` (src/overlap.cpp:41-47) never pays for them -- or when the caller
    says it is here to stay (sigax_index_prepare / sigax_index_prepare_overlap); the finder's deep start table follows the
    same rule.  Same blocks and edges with the extractor walking and every chain starting twelve symbols in (first pass) and
    with the tables (after prepare); the tables show in the index's device bytes."""
    sa = c2["sa"]
    pair = sa.FMIndexPair.load(c2["prefix"], resident=False)
    try:
        pair.set_reads(np.full(N, L, dtype=np.uint32), c2["rank"])
        bare = pair.info()["device_bytes"]
        assert bare < c2["pair"].info()["device_bytes"]  # the fixture's pair was prepared at load
        b = sa.OverlapBuilder(pair)
        reads = (c2["reads"].reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64))
        first = b.overlap(reads, M, edges=True)
        assert pair.info()["device_bytes"] == bare  # a whole pass, and no tables
        full = c2.get("full") or _run(c2, 0, N)
        for k in ("block_offs", "blocks", "substring", "edges"):
            assert first[k].tobytes() == full[k].tobytes(), k
        assert first["stats"]["n_occ_find"] + first["stats"]["n_occ_extract"] == full["stats"]["n_occ_find"] + full["stats"]["n_occ_extract"]
        second = b.overlap(reads, M, edges=True)  # the index is being reused: the builds start beside this run
        pair.prepare_overlap(M)                    # ... and are waited for here
        assert pair.info()["device_bytes"] == c2["pair"].info()["device_bytes"]
        third = b.overlap(reads, M, edges=True)
        for k in ("block_offs", "blocks", "substring", "edges"):
            assert second[k].tobytes() == full[k].tobytes() and third[k].tobytes() == full[k].tobytes(), k
    finally:
        pair.close()
