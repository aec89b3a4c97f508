mkdir -p gpurun_out/r3m && O=gpurun_out/r3m
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q -k "correct" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in 12 13 14; do
  SIGAX_KMER_PREFIX=$v timeout -k 10 300 python bench.py --workload correct --steps 10 --warmup 2 --cpu-sample 0 > $O/correct_$v.json 2> $O/correct_$v.err; echo "correct $v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/correct_$v.json").read().strip().split("\n")[-1])
print("$v", round(d["value"]/1e6,2), "M reads/s", d["ms_per_step"])
PY
done
SIGAX_KMER_PREFIX=13 timeout -k 10 300 python bench.py --workload correct --kmer 41 --steps 10 --warmup 2 --cpu-sample 0 > $O/correct41_13.json 2> $O/correct41_13.err; python -c "
import json
d=json.loads(open('$O/correct41_13.json').read().strip().split('\n')[-1]); print('k41 p13', round(d['value']/1e6,2))"
