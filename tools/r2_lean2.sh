#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_lean2; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_wide.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python3 bench.py --cpu-sample 0 --steps 60 --warmup 3"
for g in 768 1024; do
SIGAX_LIB="$PWD/build/libsigax_lw6.so" SIGAX_FX_GRID=$g timeout -k 10 150 $B > $O/lw6_$g.json 2> $O/lw6_$g.err
python3 -c "
import json
d=json.loads(open('$O/lw6_$g.json').read().strip().split('\n')[-1])
print('lw6 grid $g: %.2f Mreads/s step %.2f ms'%(d['value']/1e6,d['ms_per_step']), {k:round(x,2) for k,x in d['kernel_ms_per_step'].items()})
"
done
bash tools/r2_big.sh 2>&1 | tail -2
