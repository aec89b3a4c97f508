#!/bin/bash
# bench.py (BASELINE configs[1]) with the finder and the filter/extract + tail streams on disjoint sets of CUs
# (SIGAX_CU_SPLIT=K: K CUs for filter/extract + tail).  gpurun -- bash tools/cu_split.sh
O=gpurun_out/cu_split
mkdir -p $O
export SIGAX_TABLES_SYNC=1
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --cpu-sample 0 --steps 100 --warmup 10 > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["config"].get("kernel_ms_per_step", {})
    print("%-22s %7.2f M reads/s  step %.3f ms  find/launch %.3f ms  %s" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("avg_launch_ms", 0), {a: round(b, 2) for a, b in k.items()} if isinstance(k, dict) else ""))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run k0 SIGAX_CU_SPLIT=0
run k32 SIGAX_CU_SPLIT=32
run k64 SIGAX_CU_SPLIT=64
run k96 SIGAX_CU_SPLIT=96
run k128 SIGAX_CU_SPLIT=128
run k64_lds40k SIGAX_CU_SPLIT=64 SIGAX_FIND_LDS=40000
run k96_lds40k SIGAX_CU_SPLIT=96 SIGAX_FIND_LDS=40000
run k0_again SIGAX_CU_SPLIT=0
