mkdir -p gpurun_out/r3e && O=gpurun_out/r3e
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_configs.py::test_c5_full_size_wide_index_one_shard_of_eight > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
export SIGAX_TABLES_SYNC=1
run() { # tag, dir, env...
  tag=$1; dir=$2; shift; shift
  (cd $dir && env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/bench_$tag.json 2> $GRAFT_REPO_ROOT/$O/bench_$tag.err; echo "bench $tag rc=$?")
  python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().split("\n")[-1])
print("$tag", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items()})
PY
}
for i in 1 2; do
run new$i . X=1
run r2off$i build/r2tree SIGAX_READ_ORDER=0
done
unset SIGAX_TABLES_SYNC
timeout -k 10 800 python -m pytest tests/test_gpu_configs.py::test_c5_full_size_wide_index_one_shard_of_eight -x -q -s > $O/c5.log 2>&1; echo "c5 rc=$?"; tail -15 $O/c5.log | cut -c1-600
