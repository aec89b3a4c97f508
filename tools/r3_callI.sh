mkdir -p gpurun_out/r3j && O=gpurun_out/r3j
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"].get("finder"))
PY
}
export SIGAX_TABLES_SYNC=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_auto_q4 X=1
run3 c3_auto_q8 GPU_MAX_HW_QUEUES=8
run3 c3_auto_q16 GPU_MAX_HW_QUEUES=16
run3 c3_off_q8 GPU_MAX_HW_QUEUES=8 SIGAX_READ_ORDER=0
run3 c3_off_q16 GPU_MAX_HW_QUEUES=16 SIGAX_READ_ORDER=0
run3 c3_auto_q8_b16 GPU_MAX_HW_QUEUES=8 SIGAX_ORDER_BITS=16
run3 c3_auto_q8_b23 GPU_MAX_HW_QUEUES=8 SIGAX_ORDER_BITS=23
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 c2_q4 X=1
run2 c2_q8 GPU_MAX_HW_QUEUES=8
run2 c2_q16 GPU_MAX_HW_QUEUES=16
run2 c2_on_q8 GPU_MAX_HW_QUEUES=8 SIGAX_READ_ORDER=1
