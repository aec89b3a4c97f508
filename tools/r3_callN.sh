mkdir -p gpurun_out/r3n && O=gpurun_out/r3n
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_wide.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
export SIGAX_TABLES_SYNC=1
bash tools/err_rates.sh > $O/err16.log 2>&1; cp gpurun_out/err/rates.txt $O/rates16.txt; cat $O/rates16.txt
for e in 0.003 0.01; do
  SIGAX_FX_NO_16=1 timeout -k 10 300 python bench.py --cpu-sample 0 --steps 40 --warmup 40 --error-rate $e --isolated > $O/no16_e$e.json 2> $O/no16_e$e.err
  python - $e <<PY
import json, sys
e = sys.argv[1]
d = json.load(open("$O/no16_e%s.json" % e))
iso = d["roofline"]["isolated"]["kernel_ms_per_step"]
print("no16 rate %s: %.1f M reads/s, step %.2f ms; alone: find %.2f fx %.2f general %.2f" % (e, d["value"]/1e6, d["ms_per_step"], iso["k_find"], iso["k_filter_extract_fast"], iso["k_filter_extract"]))
PY
done
