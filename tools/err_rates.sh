#!/bin/bash
# GPU box: `python bench.py --error-rate e` over a range of substitution rates (40 warm-up runs so that the batch objects'
# choice of first launch has settled), one line per rate into gpurun_out/err/rates.txt
mkdir -p gpurun_out/err
: > gpurun_out/err/rates.txt
for e in 0 0.0003 0.001 0.003 0.01; do
  timeout -k 10 300 python bench.py --cpu-sample 0 --no-e2e --upload-steps 0 --steps 40 --warmup 40 --error-rate $e --isolated > gpurun_out/err/e$e.json 2> gpurun_out/err/e$e.err || exit 1
  python - $e >> gpurun_out/err/rates.txt <<EOT
import json, sys
e = sys.argv[1]
d = json.load(open("gpurun_out/err/e%s.json" % e))
iso = d["roofline"]["isolated"]["kernel_ms_per_step"]
print("rate %s: %.1f M reads/s, step %.2f ms; alone per 1 M reads: find %.2f ms, filter/extract %.2f ms (+ general %.2f); edges %d, rounds-equivalent n_occ/read %.0f" % (
    e, d["value"] / 1e6, d["ms_per_step"], iso["k_find"], iso["k_filter_extract_fast"], iso["k_filter_extract"], d["config"]["edges"], d["config"]["n_occ_min_per_read"]))
EOT
done
cat gpurun_out/err/rates.txt
