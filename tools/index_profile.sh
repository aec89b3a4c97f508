#!/bin/bash
# Kernel times of `siga index` (GPU suffix sorter) at N reads of 150 bp: gpurun -- bash tools/index_profile.sh [N]
set -e
N=${1:-20000000}
OUT=$PWD/gpurun_out/index_prof
mkdir -p $OUT
D=$(mktemp -d)
python - "$N" "$D" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tests.golden.make_reads import fast_reads
N = int(sys.argv[1]); d = sys.argv[2]
reads, _ = fast_reads(5 * N, 150, N, 1)
with open(os.path.join(d, "reads.fa"), "wb") as f:
    f.write(b"".join(b">r%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
PY
cd /tmp && export TMPDIR=/tmp
SIGA=$OLDPWD/siga_amd/lib/siga
( cd $D && SIGA_CLEAN_EXIT=1 SIGA_TIMING=1 SIGAX_BUILD_TIMING=1 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $SIGA index -t 64 reads.fa ) > $OUT/run.log 2>&1 || true
tail -30 $OUT/run.log
ls -R $OUT/trace | head -20
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -25 "$F" | cut -c1-200
cp "$F" $OUT/kernel_stats.csv
rm -rf $OUT/trace $D
