"""Turn the rocprofv3 outputs tools/run_profiles.sh left under gpurun_out/prof/ into the committed summaries under
profiles/: <R>_kernel_stats_*.csv, <R>_bench_under_rocprof_*.json, <R>_pmc_per_launch.json, <R>_fetch_calibration.json and
traffic.json (what bench.py prints as roofline.traffic).

HBM-side bytes per launch come from the L2's memory-side request counters by size class,
    read  = 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x TCC_EA0_RDREQ_32B
    write = 64 x TCC_EA0_WRREQ_64B + 32 x (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)
because FETCH_SIZE = 64 B x TCC_EA0_RDREQ whatever the request size (tools/fetch_calib.hip: it reads exactly half of a
coalesced stream's bytes, and 64 B per line for random 16-byte pieces of 128-byte lines)."""
import collections, csv, glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else 'r04'
O = 'gpurun_out/prof'
def latest(pat):
    fs = sorted(glob.glob(pat), key=os.path.getmtime)
    return fs[-1] if fs else None
def last_json(path):
    return json.loads(open(path).read().strip().split('\n')[-1])
def counters(name):
    f = latest('%s/%s/*/*_counter_collection.csv' % (O, name))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            out[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: {'mean_per_launch': sum(x) / len(x), 'launches': len(x)} for c, x in v.items()} for k, v in out.items()}
def rd_bytes(c):
    g = lambda n: c.get(n, {}).get('mean_per_launch', 0.0)
    return 128 * g('TCC_EA0_RDREQ_128B_sum') + 64 * g('TCC_EA0_RDREQ_64B_sum') + 32 * g('TCC_EA0_RDREQ_32B_sum')
def wr_bytes(c):
    g = lambda n: c.get(n, {}).get('mean_per_launch', 0.0)
    return 64 * g('TCC_EA0_WRREQ_64B_sum') + 32 * (g('TCC_EA0_WRREQ_sum') - g('TCC_EA0_WRREQ_64B_sum'))

for tag, d in (('default', 'q_kt'), ('sub1', 'q_kt1'), ('correct', 'k_kt')):
    ks = latest('%s/%s/*/*_kernel_stats.csv' % (O, d))
    if not ks:
        continue
    shutil.copy(ks, 'profiles/%s_kernel_stats_%s.csv' % (R, tag))
    b = last_json('%s/%s.json' % (O, d))
    json.dump(b, open('profiles/%s_bench_under_rocprof_%s.json' % (R, tag), 'w'))
    rows = {r['Name'].split('(')[0].replace('void ', ''): r for r in csv.DictReader(open(ks))}
    for k, r in rows.items():
        if k.startswith('k_find') or k.startswith('k_filter_extract_fast') or k.startswith('k_correct'):
            print(tag, 'rocprof', k, 'calls', r['Calls'], 'avg ms', float(r['AverageNs']) / 1e6)
    print(tag, 'bench avg_launch_ms', b['roofline']['avg_launch_ms'], 'value', b['value'], 'frac', b['roofline']['frac'])

res = {}
tr = json.load(open('profiles/traffic.json')) if os.path.exists('profiles/traffic.json') else {}
for tag, suffix in (('default', ''), ('subbatches_1', '1')):
    merged = collections.defaultdict(dict)
    for name in ('q_rd', 'q_wr', 'q_hm', 'q_fetch'):
        for k, v in counters(name + suffix).items():
            if k.startswith('k_'):
                merged[k].update(v)
    res[tag] = merged
    try:
        bj = last_json('%s/q_rd%s.json' % (O, suffix))
        launches = bj['launches_per_step']
        sha = bj['config'].get('kernels_sha')
    except Exception:
        continue
    fx_rd = fx_wr = 0.0
    for k, c in merged.items():
        if k.startswith('k_find'):
            tr['k_find/1000000/5000000/150/%d' % launches] = {
                'hbm_bytes_per_launch': rd_bytes(c) + wr_bytes(c), 'read': rd_bytes(c), 'write': wr_bytes(c), 'kernel': k, 'kernels_sha': sha,
                'source': 'profiles/%s_pmc_per_launch.json %s: TCC_EA0_RDREQ/_WRREQ by size class (tools/collect_profiles.py)' % (R, tag)}
        if k.startswith('k_filter_extract_fast') or k.startswith('k_fx_route'):
            fx_rd += rd_bytes(c); fx_wr += wr_bytes(c)
    if fx_rd:
        tr['k_filter_extract_fast/1000000/5000000/150/%d' % launches] = {
            'hbm_bytes_per_launch': fx_rd + fx_wr, 'read': fx_rd, 'write': fx_wr, 'kernels_sha': sha,
            'source': 'profiles/%s_pmc_per_launch.json %s: 32-lane + 64-lane launch of one sub-batch' % (R, tag)}
kc = collections.defaultdict(dict)
for name in ('k_rd', 'k_hm'):
    for k, v in counters(name).items():
        if k.startswith('k_correct'):
            kc[k].update(v)
if kc:
    res['correct'] = kc
    # one correction call = the launch of the small form + the launch of the 1024 form behind it: both counted
    tot = sum(rd_bytes(c) for c in kc.values())
    try:
        csha = last_json('%s/k_rd.json' % O)['config'].get('kernels_sha')
    except Exception:
        csha = None
    tr['k_correct/1000000/5000000/150/31'] = {'hbm_bytes_per_launch': tot, 'read': tot, 'kernel': ' + '.join(sorted(kc)), 'kernels_sha': csha,
                                              'source': 'profiles/%s_pmc_per_launch.json correct (reads only)' % R}
json.dump(res, open('profiles/%s_pmc_per_launch.json' % R, 'w'), indent=1)
if tr:
    json.dump(tr, open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(tr, indent=1))
cal = collections.defaultdict(dict)
for name in ('c_rd', 'c_fetch', 'c_hm'):
    for k, v in counters(name).items():
        if 'calib' in k:
            cal[k].update({c: x['mean_per_launch'] for c, x in v.items()})
if cal:
    known = {'k_calib_stream': 4294967296, 'k_calib_line1': 33554432 * 64, 'k_calib_line2': 33554432 * 128, 'k_calib_gran': 33554432 * 64}
    for k, c in cal.items():
        c['bytes_by_size_class'] = 128 * c.get('TCC_EA0_RDREQ_128B_sum', 0) + 64 * c.get('TCC_EA0_RDREQ_64B_sum', 0) + 32 * c.get('TCC_EA0_RDREQ_32B_sum', 0)
        c['bytes_needed_at_sector_granularity'] = known.get(k)
        c['FETCH_SIZE_bytes'] = c.get('FETCH_SIZE', 0) * 1024
    json.dump(cal, open('profiles/%s_fetch_calibration.json' % R, 'w'), indent=1)
    print(json.dumps(cal, indent=1))

if os.path.exists('%s/bench_full.json' % O):
    b = last_json('%s/bench_full.json' % O)
    json.dump(b, open('profiles/%s_bench_default.json' % R, 'w'))
    print('bench_full', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline'].get('isolated', {}).get('frac'))

for k in (31, 41):
    f = '%s/bench_correct_k%d.json' % (O, k)
    if os.path.exists(f):
        try:
            b = last_json(f)
            json.dump(b, open('profiles/%s_bench_correct_k%d.json' % (R, k), 'w'))
            print('correct k', k, b['value'], b['ms_per_step'])
        except Exception as e:
            print('correct k', k, 'no bench line', e)
