#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/run_profiles.sh): rocprofv3 kernel-trace summaries and PMC passes of
# bench.py, left under gpurun_out/ for tools/collect_profiles.py.  Counters are collected in their own passes.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --cpu-sample 0 --steps 8 --warmup 2"
P="python3 bench.py --cpu-sample 0 --steps 3 --warmup 1"
run() { # name, rocprof args..., -- cmd
  name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err
  echo "$name rc=$?" >> gpurun_out/profiles.log
}
rm -f gpurun_out/profiles.log
run q_kt   --kernel-trace --stats -d gpurun_out/q_kt   --output-format csv -- $B &&
run q_kt1  --kernel-trace --stats -d gpurun_out/q_kt1  --output-format csv -- $B --subbatches 1 --depth 1 &&
run q_fetch  --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d gpurun_out/q_fetch  --output-format csv -- $P &&
run q_fetch1 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d gpurun_out/q_fetch1 --output-format csv -- $P --subbatches 1 --depth 1 &&
run q_tcc  --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/q_tcc  --output-format csv -- $P &&
run q_tcc1 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/q_tcc1 --output-format csv -- $P --subbatches 1 --depth 1 &&
python3 bench.py --isolated > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench_full rc=$?" >> gpurun_out/profiles.log
cat gpurun_out/profiles.log
