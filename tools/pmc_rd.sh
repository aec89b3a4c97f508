cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $RD -d $O/t_rd1 --output-format csv -- python3 bench.py --cpu-sample 0 --steps 3 --warmup 1 --subbatches 1 --depth 1 > $O/t_rd1.json 2> $O/t_rd1.err
f=$(find $O/t_rd1 -name "*counter_collection.csv" | head -1)
python3 - $f <<EOT
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "")[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if "k_f" in k or "k_order" in k or "k_edges" in k:
        print(k, cnt[k], {c: round(v / cnt[k] / 1e6, 2) for c, v in acc[k].items()})
EOT
