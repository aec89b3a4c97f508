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
