"""Turn the rocprofv3 outputs tools/run_profiles.sh left under gpurun_out/ into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else 'r01'
FIND = 'k_find_n2'
def latest(pat):
    fs = sorted(glob.glob(pat), key=os.path.getmtime)
    return fs[-1] if fs else None
def last_json(path):
    return json.loads(open(path).read().strip().split('\n')[-1])
for tag, d in (('default', 'q_kt'), ('sub1', 'q_kt1')):
    ks = latest('gpurun_out/%s/*/*_kernel_stats.csv' % d)
    if not ks:
        continue
    shutil.copy(ks, 'profiles/%s_kernel_stats_%s.csv' % (R, tag))
    b = last_json('gpurun_out/%s.json' % d)
    json.dump(b, open('profiles/%s_bench_under_rocprof_%s.json' % (R, tag), 'w'))
    rows = {r['Name'].split('(')[0]: r for r in csv.DictReader(open(ks))}
    kf = rows[FIND]
    print(tag, 'rocprof k_find calls', kf['Calls'], 'avg ms', float(kf['AverageNs']) / 1e6, '| bench avg_launch_ms',
          b['roofline']['avg_launch_ms'], 'value', b['value'], 'frac', b['roofline']['frac'])
res = {}
launches = {}
for tag, names in (('default', ['q_fetch', 'q_write', 'q_tcc']), ('subbatches_1', ['q_fetch1', 'q_write1', 'q_tcc1'])):
    try:
        launches[tag] = last_json('gpurun_out/%s.json' % names[0])['launches_per_step']
    except Exception:
        launches[tag] = None
    o = {}
    for name in names:
        f = latest('gpurun_out/%s/*/*_counter_collection.csv' % name)
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if 'k_' in k:
                o.setdefault(k, {}).update({c: {'mean_per_launch': sum(x) / len(x), 'launches': len(x)} for c, x in v.items()})
    res[tag] = o
    print(tag, o.get(FIND))
json.dump(res, open('profiles/%s_pmc_per_launch.json' % R, 'w'), indent=1)
def hb(o):
    k = o[FIND]
    return (k['FETCH_SIZE']['mean_per_launch'] + k['WRITE_SIZE']['mean_per_launch']) * 1024
src = "profiles/" + R + "_pmc_per_launch.json %s: (FETCH_SIZE + WRITE_SIZE) KB x 1024; 64-byte sector requests, no x2 correction (DESIGN.md 4)"
tr = {}
for tag in ('default', 'subbatches_1'):
    key = "k_find/1000000/5000000/150/%s" % launches.get(tag)
    try:
        tr[key] = {"hbm_bytes_per_launch": hb(res[tag]), "source": src % tag}
    except KeyError:
        pass
if tr:
    json.dump(tr, open('profiles/traffic.json', 'w'), indent=1)
print(tr)
if os.path.exists('gpurun_out/bench_full.json'):
    b = last_json('gpurun_out/bench_full.json')
    json.dump(b, open('profiles/%s_bench_default.json' % R, 'w'))
    print(json.dumps(b)[:700])
