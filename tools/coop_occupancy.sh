#!/bin/bash
# GPU box: does the cooperative finder at the C3 shape follow its occupancy?  Unused dynamic LDS takes workgroups per CU
# from 4 (no padding: 40 KB each) to 3 and 2; isolated finder time per 2.5 M reads.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coopocc; mkdir -p $O
A3="--cpu-sample 0 --steps 8 --warmup 2 --isolated --reads-per-gpu 2500000 --genome-per-gpu 12500000 --emulate-world 8 --seed 2"
for pad in 0 13000 40000; do
  SIGAX_FIND_COOP_PAD=$pad timeout -k 10 500 python3 bench.py $A3 > $O/p$pad.json 2> $O/p$pad.err || exit 1
  python3 - $O/p$pad.json $pad <<EOT
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('pad', sys.argv[2], 'c3: %.2f Mreads/s step %.2f ms' % (d['value'] / 1e6, d['ms_per_step']), 'iso', {k: round(v, 2) for k, v in d['roofline']['isolated']['kernel_ms_per_step'].items()})
EOT
done
