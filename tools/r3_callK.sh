mkdir -p gpurun_out/r3l && O=gpurun_out/r3l
python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest.log
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"].get("finder"))
PY
}
export SIGAX_TABLES_SYNC=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_on1 SIGAX_READ_ORDER=1
run3 c3_off1 SIGAX_READ_ORDER=0
run3 c3_on2 SIGAX_READ_ORDER=1
run3 c3_off2 SIGAX_READ_ORDER=0
