#!/bin/bash
# per-GPU rate at the C3 and C5 index sizes: one rank's shard of the big job, kernels only (bench.py --emulate-world)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/big; mkdir -p $O
timeout -k 10 500 python3 bench.py --cpu-sample 0 --steps 20 --warmup 2 --isolated --reads-per-gpu 2500000 --genome-per-gpu 12500000 --emulate-world 8 --seed 2 > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"
timeout -k 10 500 python3 bench.py --cpu-sample 0 --steps 20 --warmup 2 --isolated --reads-per-gpu 537500 --genome-per-gpu 2500000 --read-len 250 --emulate-world 32 --seed 3 > $O/c5.json 2> $O/c5.err; echo "c5 rc=$?"
for f in c3 c5; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1])
r=d['roofline']
print('$f: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'find frac %.3f req %.1f'%(r['frac'], r['request_rate']['achieved']), 'iso', {k:round(v,2) for k,v in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"; done
