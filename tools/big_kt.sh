#!/bin/bash
# GPU box: kernel trace of one rank's shard at the C5 shape (or C3 with `c3`), isolated kernels
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bigkt; mkdir -p $O
if [ "$1" = c3 ]; then ARGS="--reads-per-gpu 2500000 --genome-per-gpu 12500000 --emulate-world 8 --seed 2"; else ARGS="--reads-per-gpu 537500 --genome-per-gpu 2500000 --read-len 250 --emulate-world 32 --seed 3"; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 bench.py --cpu-sample 0 --steps 6 --warmup 2 --subbatches 1 --depth 1 $ARGS > $O/b.json 2> $O/b.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kstats_${1:-c5}.csv \;
rm -rf $O/kt
grep -E "k_filter|k_find|k_order|k_edges|k_rowend" $O/kstats_${1:-c5}.csv | cut -c1-140
