mkdir -p gpurun_out/r3b && O=gpurun_out/r3b
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_configs.py::test_c5_full_size_wide_index_one_shard_of_eight --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
export SIGAX_TABLES_SYNC=1
for v in on off reuse syms0; do
  case $v in
    on) E="";; off) E="SIGAX_READ_ORDER=0";; reuse) E="X=1";; syms0) E="SIGAX_ROW_SYMS=0";;
  esac
  X=""; [ $v = reuse ] && X="--reuse-order"
  env $E SIGAX_VERBOSE=1 timeout -k 10 200 python bench.py --steps 100 --cpu-sample 0 $X > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1])
print("$v", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items()}, d["config"]["candidate_slots_per_chain"], d["config"]["candidate_arena_bytes"]/1e9, d["config"]["row_table"], d["config"]["reruns"])
PY
done
