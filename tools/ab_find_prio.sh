mkdir -p gpurun_out/grid
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --cpu-sample 0 --steps 150 --warmup 5 > gpurun_out/grid/$name.json 2> gpurun_out/grid/$name.err || exit 1
  python - $name <<EOT
import json, sys
d = json.load(open("gpurun_out/grid/%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"] / 1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
EOT
}
run base SIGAX_FX_GRID=640
run prio3 SIGAX_FX_GRID=640 SIGAX_LIB=build/libsigax_prio3.so
run base768 SIGAX_FX_GRID=768
run prio3_768 SIGAX_FX_GRID=768 SIGAX_LIB=build/libsigax_prio3.so
run prio3_1024 SIGAX_FX_GRID=1024 SIGAX_LIB=build/libsigax_prio3.so
