#!/bin/bash
# GPU box: the bench step under a list of environment settings, one line each (same box, same index files: an A/B table).
#   gpurun -- bash tools/sweep_env.sh OUTDIR "BENCH ARGS" "NAME:VAR=1,VAR2=x" "NAME2:" ...
O=$1; shift
A=$1; shift
mkdir -p $O
export SIGAX_TABLES_SYNC=1
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  envs=$(echo "$vars" | tr ',' ' ')
  env $envs timeout -k 10 ${SWEEP_TIMEOUT:-400} python3 bench.py --cpu-sample 0 --no-e2e --upload-steps 0 $A > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; continue; }
  python3 - $O/$name.json "$name" "$vars" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
iso = d["roofline"].get("isolated", {}).get("kernel_ms_per_step", {})
print("%-14s %-40s %7.2f M reads/s  step %7.3f ms  %s  alone %s" % (sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"],
      {k[2:6]: round(v, 2) for k, v in d["kernel_ms_per_step"].items() if k != "order_reads"}, {k[2:6]: round(v, 2) for k, v in iso.items()}))
PY
done
