#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq; mkdir -p $O
P="python3 bench.py --cpu-sample 0 --no-e2e --upload-steps 0 --steps 3 --warmup 1 ${SQ_ARGS:-}"
export SIGAX_TABLES_SYNC=1
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq --output-format csv -- $P > $O/sq.json 2> $O/sq.err; echo "sq rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq1 --output-format csv -- $P --subbatches 1 --depth 1 > $O/sq1.json 2> $O/sq1.err; echo "sq1 rc=$?"
python3 - <<'PY'
import csv,glob,collections
for d in ("sq","sq1"):
    f=sorted(glob.glob("gpurun_out/sq/%s/*/*_counter_collection.csv"%d))
    if not f: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        agg[r['Kernel_Name'].split('(')[0].replace('void ','')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if k.startswith('k_find') or k.startswith('k_filter_extract_fast'):
            m={c:sum(x)/len(x) for c,x in v.items()}
            wc=m.get('SQ_WAVE_CYCLES',1)
            print(d,k,'launches',len(v['SQ_WAVE_CYCLES']),'wave_cycles %.3g'%wc,'wait_any %.2f wait_inst %.2f active %.2f'%(m.get('SQ_WAIT_ANY',0)/wc,m.get('SQ_WAIT_INST_ANY',0)/wc,m.get('SQ_ACTIVE_INST_ANY',0)/wc),'valu %.3g salu %.3g lds %.3g vmem_rd %.3g'%(m.get('SQ_INSTS_VALU',0),m.get('SQ_INSTS_SALU',0),m.get('SQ_INSTS_LDS',0),m.get('SQ_INSTS_VMEM_RD',0)))
PY
