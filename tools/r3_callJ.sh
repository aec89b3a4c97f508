mkdir -p gpurun_out/r3k && O=gpurun_out/r3k
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"].get("finder"))
PY
}
export SIGAX_TABLES_SYNC=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_auto X=1
run3 c3_off SIGAX_READ_ORDER=0
run3 c3_auto_p2k SIGAX_FIND_COOP_PAD=2048
run3 c3_off_p2k SIGAX_FIND_COOP_PAD=2048 SIGAX_READ_ORDER=0
run3 c3_auto_p8k SIGAX_FIND_COOP_PAD=8192
run3 c3_off_p8k SIGAX_FIND_COOP_PAD=8192 SIGAX_READ_ORDER=0
run3 c3_auto_q8 GPU_MAX_HW_QUEUES=8
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 c2_q4 X=1
run2 c2_q8 GPU_MAX_HW_QUEUES=8
run2 c2_q4b X=1
run2 c2_q8b GPU_MAX_HW_QUEUES=8
run2 c2_on SIGAX_READ_ORDER=1
C5="--emulate-world 8 --reads-per-gpu 6250000 --genome-per-gpu 28750000 --read-len 250 --seed 3 --max-local-reads 1000000 --cpu-sample 0 --steps 10 --warmup 3"
run5() { tag=$1; shift; env "$@" timeout -k 10 500 python bench.py $C5 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run5 c5_p0 X=1
run5 c5_p2k SIGAX_FIND_COOP_PAD=2048
