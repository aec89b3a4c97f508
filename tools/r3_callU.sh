mkdir -p gpurun_out/r3u && O=gpurun_out/r3u
python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -10 $O/pytest.log
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
iso=d["roofline"].get("isolated",{}).get("kernel_ms_per_step",{})
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,2) for k,x in d.get("kernel_ms_per_step",{}).items()}, "alone", {k:round(x,2) for k,x in iso.items()}, d["config"]["row_table"], round(d["config"]["index_device_bytes"]/1e9,1))
PY
}
export SIGAX_TABLES_SYNC=1
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 --isolated > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 c2_direct X=1
run2 c2_rowtab SIGAX_XMAP=0
run2 c2_direct2 X=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 --isolated > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_direct X=1
run3 c3_rowtab SIGAX_XMAP=0
C5="--emulate-world 8 --reads-per-gpu 6250000 --genome-per-gpu 28750000 --read-len 250 --seed 3 --max-local-reads 1000000 --cpu-sample 0 --steps 10 --warmup 3 --isolated"
run5() { tag=$1; shift; env "$@" timeout -k 10 500 python bench.py $C5 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run5 c5_direct X=1
run5 c5_rowtab SIGAX_XMAP=0
