#!/bin/bash
# GPU box: grid of the 64-lane filter/extract launches at the C5 shape (most items take them): alone 5.5 -> 4.3 ms at 3 per CU,
# pipelined step unchanged (the finder is the step)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/big; mkdir -p $O
for g in 512 768; do
for f in 768 1024; do
SIGAX_FX_GRID=$f SIGAX_FX_GRID64=$g timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 20 --warmup 3 --isolated --reads-per-gpu 537500 --genome-per-gpu 2500000 --read-len 250 --emulate-world 32 --seed 3 > $O/c5_g${g}_${f}.json 2> $O/c5_g${g}_${f}.err || exit 1
python3 - $O/c5_g${g}_${f}.json $g $f <<EOT
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('grid64', sys.argv[2], 'grid32', sys.argv[3], 'c5: %.2f Mreads/s step %.2f ms' % (d['value'] / 1e6, d['ms_per_step']), {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()}, 'iso fx %.2f' % d['roofline']['isolated']['kernel_ms_per_step']['k_filter_extract_fast'])
EOT
done
done
