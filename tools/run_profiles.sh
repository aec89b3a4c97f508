#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/run_profiles.sh [kt|pmc|all]): rocprofv3 kernel-trace summaries and PMC
# passes of bench.py, left under gpurun_out/ for tools/collect_profiles.py.  Counters are collected in their own
# passes (one TCC counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit one pass together).
what=${1:-all}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --cpu-sample 0 --steps 8 --warmup 2"
P="python3 bench.py --cpu-sample 0 --steps 3 --warmup 1"
ISO="--subbatches 1 --depth 1"
run() { # name, rocprof args..., -- cmd
  name=$1; shift
  timeout -k 10 200 rocprofv3 "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err
  rc=$?
  echo "$name rc=$rc" >> gpurun_out/profiles.log
  return $rc
}
rm -f gpurun_out/profiles.log
if [ "$what" = kt ] || [ "$what" = all ]; then
  run q_kt   --kernel-trace --stats -d gpurun_out/q_kt   --output-format csv -- $B &&
  run q_kt1  --kernel-trace --stats -d gpurun_out/q_kt1  --output-format csv -- $B $ISO || exit 1
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  run q_fetch  --kernel-trace --pmc FETCH_SIZE -d gpurun_out/q_fetch  --output-format csv -- $P &&
  run q_write  --kernel-trace --pmc WRITE_SIZE -d gpurun_out/q_write  --output-format csv -- $P &&
  run q_tcc    --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/q_tcc --output-format csv -- $P &&
  run q_fetch1 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/q_fetch1 --output-format csv -- $P $ISO &&
  run q_write1 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/q_write1 --output-format csv -- $P $ISO &&
  run q_tcc1   --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/q_tcc1 --output-format csv -- $P $ISO || exit 1
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  python3 bench.py --isolated > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
  echo "bench_full rc=$?" >> gpurun_out/profiles.log
fi
cat gpurun_out/profiles.log
