#!/bin/bash
# GPU box: does the finder gain when its per-launch table fits the 256 MiB Infinity Cache?  BASELINE configs[1] at the same
# coverage with fewer reads: the table of one strand is 2 bytes per symbol (302 MB at 1 M reads).  Finder time per read, alone.
mkdir -p gpurun_out/icfit && O=gpurun_out/icfit
export SIGAX_TABLES_SYNC=1
for n in 1000000 800000 667000 500000 250000; do
  g=$((n * 5))
  timeout -k 10 200 python bench.py --steps 100 --cpu-sample 0 --isolated --reads-per-gpu $n --genome-per-gpu $g > $O/n$n.json 2> $O/n$n.err
  python - <<PY
import json
d=json.loads(open("$O/n$n.json").read().strip().split("\n")[-1])
iso=d["roofline"]["isolated"]["kernel_ms_per_step"]; k=d["kernel_ms_per_step"]; n=$n
print("reads %8d table %4.0f MB: %.1f M reads/s; per 1 M reads: find %.2f ms in the pipeline, %.2f alone; fx %.2f / %.2f" % (n, d["roofline"]["table_bytes_per_launch"]/1e6, d["value"]/1e6, k["k_find"]*1e6/n, iso["k_find"]*1e6/n, k["k_filter_extract_fast"]*1e6/n, iso["k_filter_extract_fast"]*1e6/n))
PY
done
