#!/bin/bash
# what one rank of the N-GPU weak-scaling bench sees: the N-GPU job's index, rank 0's shard (bench.py --emulate-world N)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/emu; mkdir -p $O
for n in 1 2 4 8; do
  timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated --emulate-world $n > $O/w$n.json 2> $O/w$n.err
  python3 -c "
import json
d=json.loads(open('$O/w$n.json').read().strip().split('\n')[-1])
r=d['roofline']
print('world $n: %.2f Mreads/s per rank, step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()}, 'iso', {k:round(x,2) for k,x in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"
done
SIGAX_FIND_COOP=0 timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated --emulate-world 8 > $O/w8_lane.json 2> $O/w8_lane.err
python3 -c "
import json
d=json.loads(open('$O/w8_lane.json').read().strip().split('\n')[-1])
r=d['roofline']
print('world 8 per-lane finder: %.2f Mreads/s per rank, step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()}, 'iso', {k:round(x,2) for k,x in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"
