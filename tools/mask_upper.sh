#!/bin/bash
# A/B of the per-lane two-step finder with and without the upper position's loads masked to the lanes that need them.
# gpurun -- bash tools/mask_upper.sh
O=gpurun_out/mask_upper
mkdir -p $O
export SIGAX_TABLES_SYNC=1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
run() {
  local name=$1; shift
  env "$@" python3 bench.py --cpu-sample 0 --steps 100 --warmup 10 --isolated > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    iso = d.get("isolated", {}).get("kernel_ms_per_step", {})
    print("%-12s %7.2f M reads/s  step %.3f ms  find/launch %.3f ms  frac %.3f  isolated %s" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("avg_launch_ms", 0), d["roofline"]["frac"], {a: round(b, 2) for a, b in iso.items()}))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run off SIGAX_FIND_MASK_UPPER=0
run on SIGAX_FIND_MASK_UPPER=1
run off2 SIGAX_FIND_MASK_UPPER=0
run on2 SIGAX_FIND_MASK_UPPER=1
