"""The host pipeline of `siga overlap` (OverlapBuilder::build, siga_amd/host/siga_host.cpp): many device batches, two in
flight per GPU, several GPUs (rehearsed on one: SIGA_DEVICE_MAP maps every logical GPU of the run to device 0), index
replicas copied device to device -- the .asqg.gz bytes must not depend on any of it, and the text must be the oracle's."""
import gzip
import os

import pytest

from tests.fixtures import fixture

pytestmark = pytest.mark.gpu


def _asqg(tmp_path, tag, fx, m, monkeypatch, batch_reads=None, gpus=1):
    from siga_amd import host
    if batch_reads:
        monkeypatch.setenv("SIGA_BATCH_READS", str(batch_reads))
    else:
        monkeypatch.delenv("SIGA_BATCH_READS", raising=False)
    monkeypatch.setenv("SIGA_DEVICE_MAP", ",".join(["0"] * gpus))
    out = str(tmp_path / (tag + ".asqg.gz"))
    host.overlap_file(fx.fa, fx.prefix, m, out, gpus=gpus)
    return open(out, "rb").read()


@pytest.mark.parametrize("name,m", [("mid", 45), ("dup", 8), ("ragged", 15)])
def test_batches_and_gpus_do_not_change_the_file(name, m, tmp_path, monkeypatch):
    fx = fixture(name)
    want, _, _ = fx.oracle_asqg(m)
    one = _asqg(tmp_path, "one", fx, m, monkeypatch)
    assert gzip.decompress(one).decode() == want
    n = len(fx.reads)
    for tag, br, gpus in (("b7", max(n // 7, 1), 1), ("g2", None, 2), ("g3b", max(n // 11, 1), 3)):
        assert _asqg(tmp_path, tag, fx, m, monkeypatch, batch_reads=br, gpus=gpus) == one, tag
    # the ED text is formatted behind each batch while it fits the holding budget, at the end otherwise: same file
    monkeypatch.setenv("SIGA_ED_HOLD_BYTES", "0")
    assert _asqg(tmp_path, "late", fx, m, monkeypatch, batch_reads=max(n // 5, 1)) == one
    monkeypatch.setenv("SIGA_ED_HOLD_BYTES", "2000")  # ... and when the budget runs out half way
    assert _asqg(tmp_path, "mixed", fx, m, monkeypatch, batch_reads=max(n // 5, 1)) == one
    monkeypatch.delenv("SIGA_ED_HOLD_BYTES")


def test_index_clone_answers_like_the_original():
    import ctypes as C
    import numpy as np
    import siga_amd
    from siga_amd import _lib
    from siga_amd.overlap import FMIndexPair
    fx = fixture("toy")
    pair = FMIndexPair.load(fx.prefix)
    h = C.c_void_p()
    assert _lib.lib().sigax_index_clone(pair.handle, 0, C.byref(h)) == 0, _lib.last_error()
    twin = FMIndexPair(h.value)
    pos = np.arange(0, len(fx.fwd), 97, dtype=np.uint64)
    for which in (0, 1):
        assert np.array_equal(pair.occ(pos, which), twin.occ(pos, which))
    a = siga_amd.OverlapBuilder(pair).overlap(fx.seqs, 45)
    b = siga_amd.OverlapBuilder(twin).overlap(fx.seqs, 45)
    assert a["blocks"].tobytes() == b["blocks"].tobytes() and a["block_offs"].tobytes() == b["block_offs"].tobytes()
    twin.close()
    pair.close()
