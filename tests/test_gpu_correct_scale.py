"""BASELINE configs[3] (`siga correct`, k-mer path) beyond fixture size: 60 k reads of 150 bp at 30x with 1 % substitutions
through the host CorrectProcessor (GPU kernel k_correct) against the oracle's restatement (one thread, 2.5 k reads/s: the
sizes are what keeps this test at half a minute), k = 31 (code default) and k = 41 (example script, 20 k reads): output
files byte for byte."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,N", [(31, 60000), (41, 20000)])
def test_correct_at_scale_matches_oracle(k, N, tmp_path):
    from oracle import pyoracle as po
    from siga_amd import host
    from tests.golden.make_reads import fast_reads, substitute
    G, L = 5 * N, 150
    clean, _ = fast_reads(G, L, N, 9)
    reads = substitute(clean, 0.01, 77)
    fa = str(tmp_path / "reads.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
    prefix = str(tmp_path / "reads")
    host.index_build_gpu(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64), prefix)
    st = po.correct(po.Index.load(prefix + ".bwt"), fa, str(tmp_path / "o.ec"), k=k)
    host.correct_file(fa, prefix, str(tmp_path / "g.ec"), k=k)
    assert open(tmp_path / "g.ec", "rb").read() == open(tmp_path / "o.ec", "rb").read()
    assert st["written"] > 0.9 * N and st["changed"] > 0.5 * N


def test_correct_reads_of_all_lengths_in_both_kernel_forms(tmp_path):
    """k_correct comes in two forms: reads of up to 512 bases in a 5 KB-per-wave form that runs five waves per SIMD, longer
    ones (up to 1024) in the 10 KB form.  A batch whose longest read the caller knows (sigax_correct_batch: the host path) gets
    one launch of the right form; a device-resident batch (sigax_correct_device: offsets on the device) gets the small form
    followed by the large one for the reads the small one marked.  Reads of 40..1000 bases with substitutions: the host path
    against the oracle, the device path against the host path; a read of 1100 bases is refused by both."""
    import ctypes as C
    from oracle import pyoracle as po
    from siga_amd import _lib, host
    from siga_amd.overlap import FMIndexPair
    rng = np.random.default_rng(21)
    G = 60000
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
    lens = np.concatenate([rng.integers(40, 1001, size=1500), [512, 513, 1000, 1024, 31, 30]])
    reads = []
    for L in lens:
        p = int(rng.integers(0, G - L))
        r = genome[p:p + L].copy()
        flips = rng.random(L) < 0.01
        r[flips] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(flips.sum()))
        reads.append(r)
    # coverage for the k-mer counts: every read twice more, error-free
    clean = [genome[int(p):int(p) + 300].copy() for p in rng.integers(0, G - 300, size=6000)]
    allr = reads + clean
    fa = str(tmp_path / "reads.fa")
    with open(fa, "wb") as f:
        f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(allr)))
    prefix = str(tmp_path / "reads")
    host.index_file_gpu(fa, prefix)
    st = po.correct(po.Index.load(prefix + ".bwt"), fa, str(tmp_path / "o.ec"), k=31)
    host.correct_file(fa, prefix, str(tmp_path / "g.ec"), k=31)
    want = open(tmp_path / "o.ec", "rb").read()
    assert open(tmp_path / "g.ec", "rb").read() == want and st["changed"] > 500
    # the device-resident call on the same reads: small form, then the large one for what it marked
    pair = FMIndexPair.load(prefix, with_sai=False)
    try:
        L = _lib.lib()
        seqs = np.concatenate(allr)
        offs = np.concatenate([[0], np.cumsum([len(r) for r in allr])]).astype(np.uint64)
        n = len(allr)
        out_b = np.zeros(len(seqs), dtype=np.uint8)
        val_b = np.zeros(n, dtype=np.uint8)
        assert L.sigax_correct_batch(pair.handle, seqs.tobytes(), None, offs.ctypes.data, n, 31, 3, 10, 1, out_b.ctypes.data,
                                     val_b.ctypes.data) == 0, _lib.last_error()
        # device buffers through the HIP runtime the library itself runs on (torch brings a second copy of it)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        hip.hipFree.argtypes = [C.c_void_p]

        def dbuf(nbytes, src=None):
            q = C.c_void_p()
            assert hip.hipMalloc(C.byref(q), nbytes) == 0
            if src is not None:
                assert hip.hipMemcpy(q, src.ctypes.data, nbytes, 1) == 0  # host to device
            else:
                assert hip.hipMemset(q, 0, nbytes) == 0
            return q

        d_seqs, d_offs = dbuf(len(seqs) + 16, None), dbuf(offs.nbytes, offs)
        assert hip.hipMemcpy(d_seqs, seqs.ctypes.data, len(seqs), 1) == 0
        d_out, d_val, d_stat = dbuf(len(seqs) + 16), dbuf(n + 16), dbuf(32)
        assert hip.hipDeviceSynchronize() == 0
        assert L.sigax_correct_device(pair.handle, d_seqs, None, d_offs, n, 31, 3, 10, 1, d_out, d_val, d_stat, None) == 0, _lib.last_error()
        assert hip.hipDeviceSynchronize() == 0
        out_d, val_d, stat_d = np.zeros(len(seqs), dtype=np.uint8), np.zeros(n, dtype=np.uint8), np.zeros(4, dtype=np.uint64)
        assert hip.hipMemcpy(out_d.ctypes.data, d_out, len(seqs), 2) == 0 and hip.hipMemcpy(val_d.ctypes.data, d_val, n, 2) == 0
        assert hip.hipMemcpy(stat_d.ctypes.data, d_stat, 32, 2) == 0
        for q in (d_seqs, d_offs, d_out, d_val, d_stat):
            hip.hipFree(q)
        assert out_d.tobytes() == out_b.tobytes()
        assert val_d.tobytes() == val_b.tobytes() and int(stat_d[0]) == 0
        # beyond the large form
        big = np.concatenate([seqs[:1100], seqs[:200]])
        boffs = np.array([0, 1100, 1300], dtype=np.uint64)
        o2 = np.zeros(1300, dtype=np.uint8)
        v2 = np.zeros(2, dtype=np.uint8)
        assert L.sigax_correct_batch(pair.handle, big.tobytes(), None, boffs.ctypes.data, 2, 31, 3, 10, 1, o2.ctypes.data, v2.ctypes.data) != 0
        assert "longer than" in _lib.last_error() and v2[0] == 2
    finally:
        pair.close()
