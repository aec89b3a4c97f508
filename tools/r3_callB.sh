mkdir -p gpurun_out/r3c && O=gpurun_out/r3c
python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_gpu_random.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
export SIGAX_TABLES_SYNC=1
run() { # tag, env..., -- args
  tag=$1; shift
  env "$@" SIGAX_VERBOSE=1 timeout -k 10 200 python bench.py --steps 100 --cpu-sample 0 > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().split("\n")[-1])
print("$tag", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items()})
PY
}
run new X=1
run new_on SIGAX_READ_ORDER=1
run new_worstcap SIGAX_CAND_CAP=worst
(cd build/r2tree && env SIGAX_TABLES_SYNC=1 SIGAX_READ_ORDER=0 timeout -k 10 200 python bench.py --steps 100 --cpu-sample 0 > ../../$O/bench_r2_off.json 2> ../../$O/bench_r2_off.err; echo "r2 off rc=$?")
(cd build/r2tree && env SIGAX_TABLES_SYNC=1 timeout -k 10 200 python bench.py --steps 100 --cpu-sample 0 > ../../$O/bench_r2_on.json 2> ../../$O/bench_r2_on.err; echo "r2 on rc=$?")
run new2 X=1
for t in r2_off r2_on new2; do python - <<PY
import json
d=json.loads(open("$O/bench_$t.json").read().strip().split("\n")[-1])
print("$t", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items()})
PY
done
# what the ordering costs alone: kernel trace of 5 steps with the order forced on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SIGAX_READ_ORDER=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_on --output-format csv -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --depth 1 > $O/kt_on.json 2> $O/kt_on.err
f=$(find $O/kt_on -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -40
find $O -name "*kernel_trace.csv" -delete
