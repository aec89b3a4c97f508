#!/bin/bash
# GPU box: s_setprio in the per-lane finder, pipelined step at C2.  Needs build/libsigax_prio{1,2}.so, built here first with
#   python -c "from siga_amd import build as b; [b.build_libsigax(force=True, out='build/libsigax_prio%d.so' % p, defines=('SIGAX_FIND_PRIO=%d' % p,)) for p in (1, 2)]"
mkdir -p gpurun_out/grid
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --cpu-sample 0 --steps 150 --warmup 5 > gpurun_out/grid/$name.json 2> gpurun_out/grid/$name.err || exit 1
  python - $name <<EOT
import json, sys
d = json.load(open("gpurun_out/grid/%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"] / 1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
EOT
}
run base X=1
for p in 1 2; do run prio$p SIGAX_LIB=build/libsigax_prio$p.so; run prio${p}_g1024 SIGAX_LIB=build/libsigax_prio$p.so SIGAX_FX_GRID=1024; done
run base2 X=1
