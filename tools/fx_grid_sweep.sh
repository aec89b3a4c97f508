#!/bin/bash
# GPU box: pipelined and isolated step at several filter/extract grids (workgroups; default 3 per CU = 768) and finder
# LDS budgets (residency cap: 60000 = 2 workgroups per CU)
mkdir -p gpurun_out/grid
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --cpu-sample 0 --steps 150 --warmup 5 > gpurun_out/grid/$name.json 2> gpurun_out/grid/$name.err || exit 1
  python - $name <<EOT
import json, sys
d = json.load(open("gpurun_out/grid/%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"] / 1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
EOT
}
for g in ${GRIDS:-384 512 640 768}; do run g$g SIGAX_FX_GRID=$g; done
for l in ${LDS:-40000 50000 80000}; do run l$l SIGAX_FIND_LDS=$l; done
for l in ${LDS2:-40000 80000}; do run g512l$l SIGAX_FX_GRID=512 SIGAX_FIND_LDS=$l; done
