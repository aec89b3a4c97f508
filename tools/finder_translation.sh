#!/bin/bash
# GPU box: where does the finder wait?  Address-translation and memory-latency counters of the finder launches at C2
# (k_find_n2, 0.3 GB table per launch) and at the C3 shape (k_find_c2, 6 GB), kernels back to back.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/xlat; mkdir -p $O
export SIGAX_TABLES_SYNC=1
C2="--cpu-sample 0 --no-e2e --upload-steps 0 --steps 3 --warmup 1 --subbatches 1 --depth 1"
C3="$C2 --reads-per-gpu 2500000 --genome-per-gpu 12500000 --emulate-world 8 --seed 2"
# at most a few counters of one block per pass ("Request exceeds the capabilities of the hardware to collect" otherwise)
P1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum"
P2="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
P3="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
for cfg in c2 c3; do
  if [ $cfg = c2 ]; then ARGS=$C2; T=120; else ARGS=$C3; T=240; fi
  timeout -k 10 $T rocprofv3 --kernel-trace --pmc $P1 -d $O/${cfg}_a --output-format csv -- python3 bench.py $ARGS > $O/${cfg}_a.json 2> $O/${cfg}_a.err || exit 1
  timeout -k 10 $T rocprofv3 --kernel-trace --pmc $P2 -d $O/${cfg}_b --output-format csv -- python3 bench.py $ARGS > $O/${cfg}_b.json 2> $O/${cfg}_b.err || exit 1
  timeout -k 10 $T rocprofv3 --kernel-trace --pmc $P3 -d $O/${cfg}_c --output-format csv -- python3 bench.py $ARGS > $O/${cfg}_c.json 2> $O/${cfg}_c.err || exit 1
done
python3 - <<EOT
import csv, glob, collections
for cfg in ("c2", "c3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("a", "b", "c"):
        for f in glob.glob("$O/%s_%s/*/*counter_collection.csv" % (cfg, p)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("k_find"):
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        g = lambda n: m.get(n, 0.0)
        print(cfg, k, "per launch:",
              "UTCL1 requests %.1f M, misses %.1f M (%.1f %%)" % (g("TCP_UTCL1_REQUEST_sum") / 1e6, g("TCP_UTCL1_TRANSLATION_MISS_sum") / 1e6, 100 * g("TCP_UTCL1_TRANSLATION_MISS_sum") / max(g("TCP_UTCL1_REQUEST_sum"), 1)),
              "| L1->L2 read latency %.0f cycles" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1)),
              "| memory read latency %.0f cycles (EA level / requests), requests %.1f M" % (g("TCC_EA0_RDREQ_LEVEL_sum") / max(g("TCC_EA0_RDREQ_sum"), 1), g("TCC_EA0_RDREQ_sum") / 1e6),
              "| UTCL2 busy %.2f of GUI active" % (g("GRBM_UTCL2_BUSY") / max(g("GRBM_GUI_ACTIVE"), 1)))
EOT
