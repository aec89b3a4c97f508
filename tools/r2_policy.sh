#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_policy; mkdir -p $O
B="python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated"
for v in base nt sc1 sc0sc1; do
  if [ $v = base ]; then timeout -k 10 150 $B > $O/$v.json 2> $O/$v.err
  else SIGAX_LIB="$PWD/build/libsigax_$v.so" timeout -k 10 150 $B > $O/$v.json 2> $O/$v.err; fi
  python3 -c "
import json
d=json.loads(open('$O/$v.json').read().strip().split('\n')[-1])
r=d['roofline']
print('$v: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()}, 'iso', {k:round(x,2) for k,x in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"
done
