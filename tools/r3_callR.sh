mkdir -p gpurun_out/r3r && O=gpurun_out/r3r
python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
show() { python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]/1e6,2), "M reads/s step", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d.get("kernel_ms_per_step",{}).items()}, d["config"]["workload"][:90], d["config"]["finder"])
PY
}
for w in 2 4; do timeout -k 10 400 python bench.py --emulate-world $w --cpu-sample 0 --steps 20 --warmup 3 > $O/emu$w.json 2> $O/emu$w.err; echo "emu$w rc=$?"; show emu$w; done
timeout -k 10 300 python bench.py > $O/default.json 2> $O/default.err; echo "default rc=$?"; show default
