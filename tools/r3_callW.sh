mkdir -p gpurun_out/r3w && O=gpurun_out/r3w
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,2) for k,x in d.get("kernel_ms_per_step",{}).items()})
PY
}
export SIGAX_TABLES_SYNC=1
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 base X=1
for v in 1 2 4 7; do run2 nt$v SIGAX_LIB=$PWD/build/libsigax_nt$v.so; done
run2 base2 X=1
run3() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --emulate-world 8 --cpu-sample 0 --steps 20 --warmup 3 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run3 c3_base X=1
run3 c3_nt7 SIGAX_LIB=$PWD/build/libsigax_nt7.so
run3 c3_nt1 SIGAX_LIB=$PWD/build/libsigax_nt1.so
