"""Turn the rocprofv3 outputs a gpurun call left under gpurun_out/ into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil
def latest(pat):
    fs = sorted(glob.glob(pat), key=os.path.getmtime)
    return fs[-1]
for tag, d in (('default', 'q_kt'), ('sub1', 'q_kt1')):
    ks = latest('gpurun_out/%s/*/*_kernel_stats.csv' % d)
    shutil.copy(ks, 'profiles/r01_kernel_stats_%s.csv' % tag)
    b = json.load(open('gpurun_out/%s.json' % d))
    json.dump(b, open('profiles/r01_bench_under_rocprof_%s.json' % tag, 'w'))
    rows = {r['Name'].split('(')[0]: r for r in csv.DictReader(open(ks))}
    kf = rows['void k_find<false>']
    print(tag, 'rocprof k_find calls', kf['Calls'], 'avg ms', float(kf['AverageNs']) / 1e6, '| bench avg_launch_ms',
          b['roofline']['avg_launch_ms'], 'value', b['value'], 'frac', b['roofline']['frac'])
res = {}
for tag, names in (('default_4_launches_per_step', ['q_fetch', 'q_tcc']), ('subbatches_1', ['q_fetch1', 'q_tcc1'])):
    o = {}
    for name in names:
        f = latest('gpurun_out/%s/*/*_counter_collection.csv' % name)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if 'k_' in k:
                o.setdefault(k, {}).update({c: {'mean_per_launch': sum(x) / len(x), 'launches': len(x)} for c, x in v.items()})
    res[tag] = o
    print(tag, o['void k_find<false>'])
json.dump(res, open('profiles/r01_pmc_per_launch.json', 'w'), indent=1)
def hb(o):
    k = o['void k_find<false>']
    return (k['FETCH_SIZE']['mean_per_launch'] + k['WRITE_SIZE']['mean_per_launch']) * 1024
src = "profiles/r01_pmc_per_launch.json %s: (FETCH_SIZE + WRITE_SIZE) KB x 1024; 64-byte sector requests, no x2 correction (DESIGN.md 4)"
tr = {"k_find/1000000/5000000/150/4": {"hbm_bytes_per_launch": hb(res['default_4_launches_per_step']), "source": src % "default_4_launches_per_step"},
      "k_find/1000000/5000000/150/1": {"hbm_bytes_per_launch": hb(res['subbatches_1']), "source": src % "subbatches_1"}}
json.dump(tr, open('profiles/traffic.json', 'w'), indent=1)
print(tr)
if os.path.exists('gpurun_out/bench_full.json'):
    shutil.copy('gpurun_out/bench_full.json', 'profiles/r01_bench_default.json')
    print(open('gpurun_out/bench_full.json').read()[:600])
