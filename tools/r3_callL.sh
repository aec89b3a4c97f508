bash tools/run_profiles.sh all
O=gpurun_out/prof
tail -3 $O/bench_full.err
python3 - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().split("\n")[-1])
print(round(d["value"]/1e6,2), d["ms_per_step"], d["kernel_ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
