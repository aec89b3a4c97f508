#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_lean; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated"
run() { # tag grid [env]
  env $3 SIGAX_FX_GRID=$2 timeout -k 10 150 $B > $O/$1.json 2> $O/$1.err
  python3 -c "
import json
d=json.loads(open('$O/$1.json').read().strip().split('\n')[-1])
r=d['roofline']
print('$1: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()}, 'iso', {k:round(x,2) for k,x in r.get('isolated',{}).get('kernel_ms_per_step',{}).items()}, 'slow', d['config']['slow_path_reads'])
"
}
run nolean_768 768 SIGAX_FX_NO_LEAN=1
run lean_768 768 X=1
run lean_1024 1024 X=1
run lean_1280 1280 X=1
run lean_1536 1536 X=1
run lean_2048 2048 X=1
