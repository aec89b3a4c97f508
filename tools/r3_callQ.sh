mkdir -p gpurun_out/r3q gpurun_out/err gpurun_out/e2e && O=gpurun_out/r3q
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()})
PY
}
export SIGAX_TABLES_SYNC=1
run2() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $tag; }
run2 base X=1
run2 prio1 SIGAX_LIB=$PWD/build/libsigax_prio1.so
run2 prio2 SIGAX_LIB=$PWD/build/libsigax_prio2.so
run2 prio3 SIGAX_LIB=$PWD/build/libsigax_prio3.so
run2 base2 X=1
for d in 2 4; do env X=1 timeout -k 10 200 python bench.py --steps 200 --cpu-sample 0 --depth $d > $O/depth$d.json 2> $O/depth$d.err; show depth$d; done
run2 fxgrid2 SIGAX_FX_GRID=512
run2 fxgrid4 SIGAX_FX_GRID=1024
unset SIGAX_TABLES_SYNC
bash tools/err_rates.sh > gpurun_out/err/log.txt 2>&1; cat gpurun_out/err/rates.txt
E2E_REHEARSE=1 timeout -k 10 400 python tools/e2e_cli.py 1000000 2 > gpurun_out/e2e/e2e_1m.txt 2>&1; tail -5 gpurun_out/e2e/e2e_1m.txt | cut -c1-400
