#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_sweep2; mkdir -p $O
B="python3 bench.py --cpu-sample 0 --steps 80 --warmup 3"
for lds in 60000 50000; do for g in 512 640 768 896; do
  SIGAX_FIND_LDS=$lds SIGAX_FX_GRID=$g timeout -k 10 120 $B > $O/b_${lds}_${g}.json 2> $O/b_${lds}_${g}.err
  python3 -c "
import json
d=json.loads(open('$O/b_${lds}_${g}.json').read().strip().split('\n')[-1])
print('lds $lds fxgrid $g: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})
"
done; done
for sb in 1 2 3 4; do for dp in 2 3; do
  timeout -k 10 120 $B --subbatches $sb --depth $dp > $O/s_${sb}_${dp}.json 2> $O/s_${sb}_${dp}.err
  python3 -c "
import json
d=json.loads(open('$O/s_${sb}_${dp}.json').read().strip().split('\n')[-1])
print('subbatches $sb depth $dp: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})
"
done; done
