#!/bin/bash
# round-2 smoke of the new bench legs on the GPU box (gpurun -- bash tools/r2_check.sh)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_chk; mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --isolated > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/log
timeout -k 10 300 python3 bench.py --workload correct --steps 20 --warmup 2 > $O/correct31.json 2> $O/correct31.err; echo "correct31 rc=$?" >> $O/log
timeout -k 10 300 python3 bench.py --workload correct --kmer 41 --steps 20 --warmup 2 --cpu-sample 20000 > $O/correct41.json 2> $O/correct41.err; echo "correct41 rc=$?" >> $O/log
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --reads-per-gpu 200000 --genome-per-gpu 1000000 > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?" >> $O/log
SIGAX_TWO_STEP=0 timeout -k 10 300 python3 bench.py --steps 50 --warmup 3 --cpu-sample 0 > $O/bench_onestep.json 2> $O/bench_onestep.err; echo "onestep rc=$?" >> $O/log
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TCC_EA0_RD[A-Za-z0-9_]*\|TCC_EA0_WR[A-Za-z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_MISS[A-Za-z_]*\|TCC_HIT[A-Za-z_]*\|TCC_REQ[A-Za-z_]*" $O/counters.txt | sort -u > $O/counters_tcc.txt
for c in FETCH_SIZE "TCC_MISS_sum TCC_HIT_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c -d $O/calib_$n --output-format csv -- build/fetch_calib > $O/calib_$n.txt 2>&1; echo "calib $n rc=$?" >> $O/log
done
cat $O/log
