#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_lean3; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_wide.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated"
timeout -k 10 150 $B > $O/c2.json 2> $O/c2.err
python3 -c "
import json
d=json.loads(open('$O/c2.json').read().strip().split('\n')[-1])
print('c2: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()}, 'iso', {k:round(x,2) for k,x in d['roofline'].get('isolated',{}).get('kernel_ms_per_step',{}).items()})
"
bash tools/r2_big.sh 2>&1 | tail -2
