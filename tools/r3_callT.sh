mkdir -p gpurun_out/r3t && O=gpurun_out/r3t
python -m pytest tests/test_gpu_wide.py -x -q -k "locality" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
iso=d["roofline"].get("isolated",{}).get("kernel_ms_per_step",{})
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,2) for k,x in d.get("kernel_ms_per_step",{}).items()}, "alone", {k:round(x,2) for k,x in iso.items()})
PY
}
export SIGAX_TABLES_SYNC=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 --isolated > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_off SIGAX_READ_ORDER=0
run3 c3_on SIGAX_READ_ORDER=1
run3 c3_on_norefine SIGAX_READ_ORDER=1 SIGAX_ORDER_REFINE=0
run3 c3_off2 SIGAX_READ_ORDER=0
run3 c3_on2 SIGAX_READ_ORDER=1
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 --isolated > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 c2_off X=1
run2 c2_on SIGAX_READ_ORDER=1
