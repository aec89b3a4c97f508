"""Helpers shared by the BASELINE-sized GPU tests: compare a GPU result with the oracle's batch call, and a numpy
restatement of Hit2OverlapConverter::convert (src/overlap_builder.cpp:345-375) for the edge records."""
import numpy as np

COLS = ["capped0_lo", "capped0_hi", "capped1_lo", "capped1_hi", "raw0_lo", "raw0_hi", "raw1_lo", "raw1_hi", "length", "af"]


def blocks_matrix(blocks):
    """structured sigax_block array -> u64 [k, 10] in the oracle's column order"""
    out = np.empty((len(blocks), 10), dtype=np.uint64)
    for i, c in enumerate(COLS):
        out[:, i] = blocks[c]
    return out


def assert_same_blocks(res, want, what=""):
    """res: siga_amd OverlapBuilder.overlap result; want: oracle.pyoracle.overlap_batch result"""
    assert np.array_equal(res["block_offs"], want["block_offs"]), what + " block_offs differ"
    got = blocks_matrix(res["blocks"])
    if not np.array_equal(got, want["blocks"]):
        bad = np.nonzero((got != want["blocks"]).any(axis=1))[0]
        owner = np.searchsorted(want["block_offs"], bad[0], side="right") - 1
        raise AssertionError("%s %d blocks differ; first: block %d of read %d\n got  %s\n want %s" % (
            what, len(bad), bad[0], owner, got[bad[0]], want["blocks"][bad[0]]))
    assert np.array_equal(res["substring"].astype(bool), want["substring"].astype(bool)), what + " substring flags differ"
    s = res["stats"]
    assert s["n_occ_find"] + s["n_occ_extract"] == want["n_occ_min"], what + " N_occ_min differs"


def expected_edges(blocks10, block_offs, sai, rsai, read_len, name_rank, read_base=0):
    """Hit2OverlapConverter::convert over all blocks in hits order -> u32 [e, 4] (query, target, length, af)."""
    nb = len(blocks10)
    lo, hi = blocks10[:, 0].astype(np.int64), blocks10[:, 1].astype(np.int64)
    length, af = blocks10[:, 8].astype(np.int64), blocks10[:, 9].astype(np.int64)
    owner = np.repeat(np.arange(len(block_offs) - 1, dtype=np.int64), np.diff(block_offs.astype(np.int64))) + read_base
    cnt = np.maximum(hi - lo + 1, 0)
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]]) if nb else np.zeros(0, dtype=np.int64)
    total = int(cnt.sum())
    bi = np.repeat(np.arange(nb, dtype=np.int64), cnt)
    j = lo[bi] + (np.arange(total, dtype=np.int64) - start[bi])
    rev = (af[bi] & 2) != 0
    t = np.where(rev, rsai[j], sai[j]).astype(np.int64)
    q = owner[bi]
    nq, nt = name_rank[q].astype(np.int64), name_rank[t].astype(np.int64)
    ln = length[bi]
    contained = (ln == read_len[q]) | (ln == read_len[t])          # Match::isContainment (coord.h:150-152)
    keep = (nq != nt) & ~((nq < nt) | (contained & ((af[bi] & 1) != 0)))  # :358, :365
    out = np.stack([q[keep], t[keep], ln[keep], af[bi][keep]], axis=1).astype(np.uint32)
    return out


def edges_matrix(edges):
    return np.stack([edges["query"], edges["target"], edges["length"], edges["af"]], axis=1).astype(np.uint32)


def read_sai(path):
    """.sai text (src/suffix_array.cpp:17-44) -> read ids"""
    data = np.fromfile(path, dtype=np.uint8)
    txt = data.tobytes().split(b"\n", 3)
    body = np.frombuffer(txt[3], dtype=np.uint8)
    # lines "<id> 0": parse with numpy: split on newline via loadtxt is slow for 20M lines; use a vectorised digit parse
    nl = np.nonzero(body == 10)[0]
    starts = np.concatenate([[0], nl[:-1] + 1])
    sp = nl - 2  # position of the space before the trailing '0'
    lens = sp - starts
    out = np.zeros(len(nl), dtype=np.int64)
    maxlen = int(lens.max()) if len(lens) else 0
    for k in range(maxlen):
        has = lens > k
        d = body[np.minimum(starts + k, len(body) - 1)].astype(np.int64) - 48
        out = np.where(has, out * 10 + d, out)
    return out.astype(np.uint32)
