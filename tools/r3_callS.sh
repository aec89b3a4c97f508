mkdir -p gpurun_out/r3s && O=gpurun_out/r3s
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"]["finder"])
PY
}
export SIGAX_TABLES_SYNC=1
for w in 4 2; do
  for v in lane coop; do
    E="SIGAX_FIND_COOP=0"; [ $v = coop ] && E="SIGAX_FIND_COOP=1"
    env $E timeout -k 10 400 python bench.py --emulate-world $w --cpu-sample 0 --steps 20 --warmup 3 > $O/emu${w}_$v.json 2> $O/emu${w}_$v.err; echo "emu${w}_$v rc=$?"; show emu${w}_$v
  done
done
env SIGAX_FIND_COOP=1 timeout -k 10 200 python bench.py --cpu-sample 0 --steps 100 > $O/c2_coop.json 2> $O/c2_coop.err; show c2_coop
