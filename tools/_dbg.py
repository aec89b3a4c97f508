import sys; sys.path.insert(0,'.')
import numpy as np
import siga_amd
from siga_amd.overlap import format_hits
from tests.fixtures import fixture
fx = fixture("mid")
pair = siga_amd.FMIndexPair.load(fx.prefix)
b = siga_amd.OverlapBuilder(pair)
res = b.overlap(fx.seqs, 45)
print(res["stats"])
want_asqg, want_hits, st = fx.oracle_asqg(45, hits=True)
got = format_hits(res).split("\n"); want = want_hits.split("\n")
bad = [i for i in range(len(want)) if got[i] != want[i]]
print("bad reads", len(bad), bad[:10])
for i in bad[:2]:
    print("got ", got[i][:400]); print("want", want[i][:400])
