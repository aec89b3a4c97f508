#!/bin/bash
# The per-lane finder's residency cap (dynamic LDS per workgroup: 160 KB / budget workgroups of four waves per CU), swept
# with today's kernels.  gpurun -- bash tools/find_lds_sweep.sh
O=gpurun_out/find_lds
mkdir -p $O
export SIGAX_TABLES_SYNC=1
for lds in 60000 53000 40000 32000 26000; do
  SIGAX_FIND_LDS=$lds python3 bench.py --cpu-sample 0 --steps 100 --warmup 10 > $O/l$lds.json 2> $O/l$lds.err
  python3 - $O/l$lds.json $lds <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("budget %6s  %7.2f M reads/s  step %.3f ms  find/launch %.3f ms" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("avg_launch_ms", 0)))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
