"""sigax_locality_keys (csrc/sigax_keys.hip) against its torch restatement (siga_amd/sharding.py, pinned to the definition in
tests/test_sharding.py): same 64-bit keys for uniform and ragged reads, mixed case and non-ACGT bytes."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_sharding import _key_by_definition, _locus_reads

pytestmark = pytest.mark.gpu


def test_native_keys_equal_the_restatement():
    from siga_amd.sharding import _locality_keys_torch, key_order, locality_keys
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    for G, L, N, seed in ((50000, 150, 30000, 1), (20000, 250, 5000, 2), (5000, 16, 700, 3), (5000, 33, 701, 4)):
        reads, _ = _locus_reads(G, L, N, seed)
        rng = np.random.default_rng(seed)
        hit = rng.random(reads.shape) < 0.01   # a few lower-case and non-ACGT bytes
        reads = np.where(hit, np.frombuffer(b"acgtN", dtype=np.uint8)[rng.integers(0, 5, size=reads.shape)], reads)
        got = locality_keys(reads, device=dev, chunk=7001)   # the kernel, in several chunks
        want = _locality_keys_torch(reads, torch.device("cpu"))
        assert got.dtype == np.int64 and np.array_equal(got, want), (L, N)
        assert np.array_equal(key_order(got, device=dev), key_order(want))


def test_native_keys_of_ragged_reads():
    from siga_amd import _lib
    rng = np.random.default_rng(9)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(l))) for l in rng.integers(1, 300, size=400)]
    seqs += ["", "ACGTACGTACGTACG", "ACGTACGTACGTACGT", "acgtnACGTNNNNacgtacgtTTTT"]
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_buf = torch.from_numpy(buf.copy()).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    d_keys = torch.full((len(seqs),), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L = _lib.lib()
    assert L.sigax_locality_keys(0, d_buf.data_ptr(), d_offs.data_ptr(), len(seqs), d_keys.data_ptr(), None) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert d_keys.cpu().tolist() == [_key_by_definition(s) for s in seqs]
    assert L.sigax_locality_keys(0, None, d_offs.data_ptr(), 3, d_keys.data_ptr(), None) == _lib.SIGAX_E_ARG
    assert L.sigax_locality_keys(0, None, None, 0, None, None) == 0
