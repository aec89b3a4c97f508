#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_coop; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_wide.py -x -q -k cooperative > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python3 bench.py --cpu-sample 0 --steps 40 --warmup 3 --isolated"
SIGAX_FIND_COOP=1 timeout -k 10 200 $B > $O/c2_coop.json 2> $O/c2_coop.err
SIGAX_FIND_COOP=0 timeout -k 10 200 $B > $O/c2_lane.json 2> $O/c2_lane.err
for f in c2_coop c2_lane; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1])
r=d['roofline']
print('$f: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'find frac %.3f req %.1f'%(r['frac'], r['request_rate']['achieved']), 'iso', {k:round(v,2) for k,v in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"; done
