mkdir -p gpurun_out/r3i && O=gpurun_out/r3i
python -m pytest tests/test_gpu_wide.py tests/test_gpu_index_build.py -x -q -k "locality or order_check or regrow or general or arena" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"].get("finder"))
PY
}
export SIGAX_TABLES_SYNC=1
for i in 1 2; do
for v in auto off; do
  E="X=1"; [ $v = off ] && E="SIGAX_READ_ORDER=0"
  env $E timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 > $O/c3_$v$i.json 2> $O/c3_$v$i.err; echo "c3 $v$i rc=$?"; show c3_$v$i
done
done
for i in 1 2; do
for v in on off; do
  E="SIGAX_READ_ORDER=1"; [ $v = off ] && E="X=1"
  env $E timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $O/c2_$v$i.json 2> $O/c2_$v$i.err; echo "c2 $v$i rc=$?"; show c2_$v$i
done
done
C5="--emulate-world 8 --reads-per-gpu 6250000 --genome-per-gpu 28750000 --read-len 250 --seed 3 --max-local-reads 1000000 --cpu-sample 0 --steps 10 --warmup 3"
timeout -k 10 500 python bench.py $C5 > $O/c5_auto.json 2> $O/c5_auto.err; echo "c5 rc=$?"; show c5_auto
