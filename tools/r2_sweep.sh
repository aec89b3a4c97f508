#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_sweep; mkdir -p $O
SIGAX_LIB=$PWD/build/libsigax_prof.so timeout -k 10 200 python3 tools/fx_profile.py > $O/fxprof.txt 2>&1
B="python3 bench.py --cpu-sample 0 --steps 60 --warmup 3"
for lds in 60000 80500; do for g in 768 1024 1536 2048; do
  SIGAX_FIND_LDS=$lds SIGAX_FX_GRID=$g timeout -k 10 120 $B > $O/b_${lds}_${g}.json 2> $O/b_${lds}_${g}.err
  python3 -c "
import json,sys
d=json.loads(open('$O/b_${lds}_${g}.json').read().strip().split('\n')[-1])
print('lds $lds fxgrid $g: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})
" >> $O/sweep.txt 2>&1
done; done
cat $O/fxprof.txt | tail -4; cat $O/sweep.txt
