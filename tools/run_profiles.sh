#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/run_profiles.sh [kt|pmc|calib|correct|all]): rocprofv3 kernel-trace summaries
# and PMC passes of bench.py, left under gpurun_out/prof/ for tools/collect_profiles.py.  Counters are collected in their
# own passes with --kernel-trace only (one TCC counter group per pass).
what=${1:-all}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
# the row tables at open rather than beside the second run: every profiled launch is then a steady-state one
export SIGAX_TABLES_SYNC=1
B="python3 bench.py --cpu-sample 0 --no-e2e --upload-steps 0 --steps 20 --warmup 3"
P="python3 bench.py --cpu-sample 0 --no-e2e --upload-steps 0 --steps 3 --warmup 1"
ISO="--subbatches 1 --depth 1"
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
HM="TCC_HIT_sum TCC_MISS_sum"
run() { # name, rocprof args..., -- cmd
  name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" > $O/$name.json 2> $O/$name.err
  rc=$?
  echo "$name rc=$rc" >> $O/profiles.log
  return $rc
}
rm -f $O/profiles.log
if [ "$what" = kt ] || [ "$what" = all ]; then
  run q_kt   --kernel-trace --stats -d $O/q_kt   --output-format csv -- $B &&
  run q_kt1  --kernel-trace --stats -d $O/q_kt1  --output-format csv -- $B $ISO || exit 1
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  run q_rd   --kernel-trace --pmc $RD -d $O/q_rd   --output-format csv -- $P &&
  run q_wr   --kernel-trace --pmc $WR -d $O/q_wr   --output-format csv -- $P &&
  run q_hm   --kernel-trace --pmc $HM -d $O/q_hm   --output-format csv -- $P &&
  run q_fetch --kernel-trace --pmc FETCH_SIZE -d $O/q_fetch --output-format csv -- $P &&
  run q_rd1  --kernel-trace --pmc $RD -d $O/q_rd1  --output-format csv -- $P $ISO &&
  run q_wr1  --kernel-trace --pmc $WR -d $O/q_wr1  --output-format csv -- $P $ISO &&
  run q_hm1  --kernel-trace --pmc $HM -d $O/q_hm1  --output-format csv -- $P $ISO || exit 1
fi
if [ "$what" = calib ] || [ "$what" = all ]; then
  run c_rd    --kernel-trace --pmc $RD -d $O/c_rd --output-format csv -- build/fetch_calib &&
  run c_fetch --kernel-trace --pmc FETCH_SIZE -d $O/c_fetch --output-format csv -- build/fetch_calib &&
  run c_hm    --kernel-trace --pmc $HM -d $O/c_hm --output-format csv -- build/fetch_calib || exit 1
fi
if [ "$what" = correct ] || [ "$what" = all ]; then
  C="python3 bench.py --workload correct --cpu-sample 0 --steps 5 --warmup 1"
  run k_kt --kernel-trace --stats -d $O/k_kt --output-format csv -- $C &&
  run k_rd --kernel-trace --pmc $RD -d $O/k_rd --output-format csv -- $C &&
  run k_hm --kernel-trace --pmc $HM -d $O/k_hm --output-format csv -- $C || exit 1
fi
if [ "$what" = full ] || [ "$what" = all ]; then
  # the plain default run (300 timed steps, CPU baseline leg, end_to_end and upload_inclusive legs) as the driver runs it, plus --isolated
  timeout -k 10 600 python3 bench.py --isolated > $O/bench_full.json 2> $O/bench_full.err; echo "bench_full rc=$?" >> $O/profiles.log
  # BASELINE configs[3]: the corrector at k = 31 (code default) and k = 41 (the example script's), CPU baseline leg included
  for k in 31 41; do
    timeout -k 10 400 python3 bench.py --workload correct --kmer $k --steps 20 --warmup 3 > $O/bench_correct_k$k.json 2> $O/bench_correct_k$k.err; echo "bench_correct_k$k rc=$?" >> $O/profiles.log
  done
fi
find $O -name "*kernel_trace.csv" -size +4M -delete
cat $O/profiles.log
