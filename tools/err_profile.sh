#!/bin/bash
# GPU box: where do reads with substitutions spend filter/extract time?  Path counters (SIGAX_FX_PROFILE build) and a kernel
# trace of the isolated kernels at a given error rate.  gpurun -- bash tools/err_profile.sh 0.001
e=${1:-0.001}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/errp; mkdir -p $O
SIGAX_LIB=build/libsigax_prof.so FXP_ERR=$e timeout -k 10 200 python3 tools/fx_profile.py > $O/paths_$e.txt 2> $O/paths_$e.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$e --output-format csv -- python3 bench.py --cpu-sample 0 --steps 10 --warmup 2 --subbatches 1 --depth 1 --error-rate $e > $O/bench_$e.json 2> $O/bench_$e.err
find $O/kt_$e -name "*kernel_stats.csv" -exec cp {} $O/kstats_$e.csv \;
rm -rf $O/kt_$e
