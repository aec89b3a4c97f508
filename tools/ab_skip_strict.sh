mkdir -p gpurun_out/errp
for e in 0.0003 0.001 0.003; do
timeout -k 10 200 python bench.py --cpu-sample 0 --steps 40 --warmup 40 --error-rate $e > gpurun_out/errp/p_$e.json 2> gpurun_out/errp/p_$e.err || exit 1
SIGAX_FX_SKIP_STRICT=1 timeout -k 10 200 python bench.py --cpu-sample 0 --steps 40 --warmup 40 --error-rate $e > gpurun_out/errp/ps_$e.json 2> gpurun_out/errp/ps_$e.err || exit 1
done
python - <<EOT
import json
for e in ("0.0003","0.001","0.003"):
  for f in ("p_","ps_"):
    d=json.load(open("gpurun_out/errp/%s%s.json"%(f,e))); print(f, e, round(d["value"]/1e6,2), round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
EOT
