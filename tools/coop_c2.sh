#!/bin/bash
# GPU box: the cooperative finder at C2 with its residency capped by a persistent grid (workgroups per CU) against the
# per-lane finder, pipelined
mkdir -p gpurun_out/grid
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --cpu-sample 0 --steps 150 --warmup 5 > gpurun_out/grid/$name.json 2> gpurun_out/grid/$name.err || exit 1
  python - $name <<EOT
import json, sys
d = json.load(open("gpurun_out/grid/%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"] / 1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
EOT
}
run lane X=1
for w in ${WGS:-3 4 5}; do for g in ${GRIDS:-768 1024}; do run coop_w${w}_g$g SIGAX_FIND_COOP=1 SIGAX_FIND_COOP_WGS=$w SIGAX_FX_GRID=$g; done; done
