mkdir -p gpurun_out/grid
run() { name=$1; shift; extra=""
  while [ "${1#--}" != "$1" ]; do extra="$extra $1 $2"; shift 2; done
  env "$@" timeout -k 10 200 python bench.py --cpu-sample 0 --steps 200 --warmup 5 $extra > gpurun_out/grid/$name.json 2> gpurun_out/grid/$name.err || exit 1
  python - $name <<EOT
import json, sys
d = json.load(open("gpurun_out/grid/%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"] / 1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
EOT
}
run s1d2 --subbatches 1 --depth 2 X=1
run s1d3 --subbatches 1 --depth 3 X=1
run s1d4 --subbatches 1 --depth 4 X=1
run s2d3 --subbatches 2 --depth 3 X=1
run s1d3b --subbatches 1 --depth 3 X=1
run s1d3g640 --subbatches 1 --depth 3 SIGAX_FX_GRID=640
run s1d3g1024 --subbatches 1 --depth 3 SIGAX_FX_GRID=1024
