#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/shapes.sh c3|c5 [bench|kt|pmc ...]): one rank's shard of the 8-GPU jobs of
# BASELINE configs[2] (c3) and configs[4] at full size (c5) on this GPU (bench.py --emulate-world 8): the bench line, the
# rocprofv3 kernel-trace summary of the same command and the TCC size-class PMC passes, left under gpurun_out/shapes/ for
# tools/collect_shapes.py.  Counters are collected in their own passes with --kernel-trace only.
shape=${1:-c3}; shift
what=${@:-bench kt pmc}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=${SHAPES_OUT:-gpurun_out/shapes}; mkdir -p $O
export SIGAX_TABLES_SYNC=1   # row tables at open: every profiled launch is a steady-state one
if [ "$shape" = c3 ]; then
  A="--emulate-world 8 --cpu-sample 0 --upload-steps 0"    # defaults = configs[2]: 2.5 M reads per rank from 20 M x 150 bp, 100 Mb, seed 2
  STEPS="--steps 20 --warmup 2"; PSTEPS="--steps 3 --warmup 1"
else
  A="--emulate-world 8 --reads-per-gpu 6250000 --genome-per-gpu 28750000 --read-len 250 --seed 3 --max-local-reads 1000000 --cpu-sample 0 --upload-steps 0"
  STEPS="--steps 10 --warmup 2"; PSTEPS="--steps 2 --warmup 1"
fi
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
for w in $what; do
  case $w in
    bench) timeout -k 10 900 python3 bench.py $A $STEPS --isolated > $O/${shape}_bench.json 2> $O/${shape}_bench.err; echo "$shape bench rc=$?" ;;
    kt)    timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/${shape}_kt --output-format csv -- python3 bench.py $A $STEPS > $O/${shape}_kt.json 2> $O/${shape}_kt.err; echo "$shape kt rc=$?" ;;
    pmc)   timeout -k 10 600 rocprofv3 --kernel-trace --pmc $RD -d $O/${shape}_rd --output-format csv -- python3 bench.py $A $PSTEPS > $O/${shape}_rd.json 2> $O/${shape}_rd.err; echo "$shape rd rc=$?"
           timeout -k 10 600 rocprofv3 --kernel-trace --pmc $WR -d $O/${shape}_wr --output-format csv -- python3 bench.py $A $PSTEPS > $O/${shape}_wr.json 2> $O/${shape}_wr.err; echo "$shape wr rc=$?" ;;
  esac
  tail -2 $O/${shape}_*.err | cut -c1-300
done
# keep what travels back small: the per-dispatch traces are large, the stats and counter tables are what is read
find $O -name "*kernel_trace.csv" -size +8M -delete
ls -la $O
