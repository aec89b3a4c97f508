mkdir -p gpurun_out/r3v && O=gpurun_out/r3v
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --cpu-sample 0 > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"; tail -3 $O/gloo2.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("$O/gloo2.json").read().strip().split("\n")[-1])
print(round(d["value"]/1e6,2), d["n_gpus"], d["ms_per_step"], d["config"]["workload"][:160], d["config"]["edges"], d["config"]["reads_per_gpu"])
PY
python -m pytest tests/test_gpu_wide.py -x -q -k "extractor_forms" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 100 > $O/default.json 2> $O/default.err; python -c "
import json
d=json.loads(open('$O/default.json').read().strip().split('\n')[-1]); print('default', round(d['value']/1e6,2), d['config']['row_table'], d['cpu_baseline']['value'])"
