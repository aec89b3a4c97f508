mkdir -p gpurun_out/r3g && O=gpurun_out/r3g
python -m pytest tests/test_gpu_wide.py tests/test_gpu_two_ranks.py tests/test_gpu_correct_scale.py -x -q -k "correct or locality or two_ranks or self_launch or candidate" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"].get("finder"), d["roofline"].get("avg_launch_ms"))
PY
}
for v in on off; do
  E="X=1"; [ $v = off ] && E="SIGAX_KMER_PREFIX=0"
  env $E timeout -k 10 300 python bench.py --workload correct --steps 10 --warmup 2 --cpu-sample 0 > $O/correct_$v.json 2> $O/correct_$v.err; echo "correct $v rc=$?"; show correct_$v
done
export SIGAX_TABLES_SYNC=1
for v in auto off; do
  E="X=1"; [ $v = off ] && E="SIGAX_READ_ORDER=0"
  env $E timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 2 > $O/c3_$v.json 2> $O/c3_$v.err; echo "c3 $v rc=$?"; show c3_$v
done
C5="--emulate-world 8 --reads-per-gpu 6250000 --genome-per-gpu 28750000 --read-len 250 --seed 3 --max-local-reads 1000000 --cpu-sample 0 --steps 10 --warmup 2"
for v in auto off; do
  E="X=1"; [ $v = off ] && E="SIGAX_READ_ORDER=0"
  env $E timeout -k 10 500 python bench.py $C5 > $O/c5_$v.json 2> $O/c5_$v.err; echo "c5 $v rc=$?"; show c5_$v
done
