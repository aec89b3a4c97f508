"""Turn what tools/shapes.sh left under gpurun_out/shapes/ (one rank's shard of the 8-GPU jobs of BASELINE configs[2] = c3
and configs[4] = c5, `bench.py --emulate-world 8`) into the committed summaries under profiles/:
<R>_kernel_stats_<shape>.csv (rocprofv3 --kernel-trace --stats), <R>_bench_under_rocprof_<shape>.json (the bench line of that
same run), <R>_bench_<shape>.json (the plain run with --isolated), <R>_pmc_<shape>.json (TCC size-class counters per kernel
and launch) and the shape's keys of traffic.json (what bench.py prints as roofline.traffic).  Memory-side bytes per launch as
in tools/collect_profiles.py: 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B (+ writes likewise)."""
import collections, csv, glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else 'r04'
O = os.environ.get('SHAPES_OUT', 'gpurun_out/shapes')
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
tpath = 'profiles/traffic.json'
tr = json.load(open(tpath)) if os.path.exists(tpath) else {}
for shape in ('c3', 'c5'):
    ks = latest('%s/%s_kt/*/*_kernel_stats.csv' % (O, shape))
    if ks:
        shutil.copy(ks, 'profiles/%s_kernel_stats_%s.csv' % (R, shape))
        b = last_json('%s/%s_kt.json' % (O, shape))
        json.dump(b, open('profiles/%s_bench_under_rocprof_%s.json' % (R, shape), 'w'))
        for r in csv.DictReader(open(ks)):
            k = r['Name'].split('(')[0].replace('void ', '')
            if k.startswith('k_find') or k.startswith('k_filter_extract_fast') or k.startswith('k_read_keys') or 'read_keys' in k:
                print(shape, 'rocprof', k, 'calls', r['Calls'], 'avg ms', float(r['AverageNs']) / 1e6)
        print(shape, 'bench avg_launch_ms', b['roofline']['avg_launch_ms'], 'value', b['value'], 'frac', b['roofline']['frac'])
    if os.path.exists('%s/%s_bench.json' % (O, shape)):
        try:
            json.dump(last_json('%s/%s_bench.json' % (O, shape)), open('profiles/%s_bench_%s.json' % (R, shape), 'w'))
        except Exception as e:
            print(shape, 'no bench line', e)
    merged = collections.defaultdict(dict)
    for name in ('rd', 'wr'):
        for k, v in counters('%s_%s' % (shape, name)).items():
            if k.startswith('k_'):
                merged[k].update(v)
    if not merged:
        continue
    json.dump(merged, open('profiles/%s_pmc_%s.json' % (R, shape), 'w'), indent=1)
    try:
        b = last_json('%s/%s_rd.json' % (O, shape))
    except Exception:
        continue
    key = '%d/%d/%d/%d' % (b['config']['reads_per_gpu'], b['config']['genome_bp'], b['config']['read_len'], b['launches_per_step'])
    if (b['config'].get('sharding') or {}).get('by', '').startswith('locality'):
        key += '/key'  # as bench.py names the entry of a key-range shard
    sha = b['config'].get('kernels_sha')
    fx_rd = fx_wr = 0.0
    for k, c in merged.items():
        if k.startswith('k_find'):
            tr['k_find/' + key] = {'hbm_bytes_per_launch': rd_bytes(c) + wr_bytes(c), 'read': rd_bytes(c), 'write': wr_bytes(c), 'kernel': k, 'kernels_sha': sha,
                                   'source': 'profiles/%s_pmc_%s.json: TCC_EA0_RDREQ/_WRREQ by size class (tools/collect_shapes.py)' % (R, shape)}
        if k.startswith('k_filter_extract_fast') or k.startswith('k_fx_route'):
            fx_rd += rd_bytes(c); fx_wr += wr_bytes(c)
    if fx_rd:
        tr['k_filter_extract_fast/' + key] = {'hbm_bytes_per_launch': fx_rd + fx_wr, 'read': fx_rd, 'write': fx_wr, 'kernels_sha': sha,
                                              'source': 'profiles/%s_pmc_%s.json: the launch chain of one sub-batch' % (R, shape)}
    print(shape, key, {k: v['hbm_bytes_per_launch'] for k, v in tr.items() if key in k})
json.dump(tr, open(tpath, 'w'), indent=1)
