"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
gfx950 corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE (KB) under-reports wide coalesced streaming reads by
exactly 2x -> doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane stores.
One launch = one Dispatch_Id: rows are first summed per dispatch (rocprofv3 may emit several rows per dispatch -- per XCD / dimension),
`launches` counts distinct dispatches.  Exits non-zero when a pass left no counter rows (a timed-out / failed pass must not become a
half-empty profile that bench.py later attaches to a line)."""
import csv, glob, json, sys, collections
out = sys.argv[1]
per_dispatch = {'fetch': collections.defaultdict(float), 'write': collections.defaultdict(float)}
name_of = {}
for kind, key in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    files = glob.glob('%s/%s/*/*counter_collection.csv' % (out, kind))
    rows = 0
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != key:
                continue
            did = (f, r.get('Dispatch_Id', rows))
            per_dispatch[kind][did] += float(r['Counter_Value'])
            name_of[did] = r['Kernel_Name']
            rows += 1
    if not rows:
        sys.stderr.write('pmc_traffic: the %s pass left no %s rows under %s/%s\n' % (kind, key, out, kind))
        sys.exit(3)
acc = collections.defaultdict(lambda: {'fetch_kb': 0.0, 'write_kb': 0.0, 'n_fetch': 0, 'n_write': 0})
for kind in ('fetch', 'write'):
    for did, kb in per_dispatch[kind].items():
        a = acc[name_of[did]]
        a[kind + '_kb'] += kb
        a['n_' + kind] += 1
res = {}
for k, v in acc.items():
    n = max(v['n_fetch'], v['n_write'], 1)
    res[k[:120]] = {'launches': n,
                    'fetch_bytes_per_launch': 2.0 * 1024 * v['fetch_kb'] / max(v['n_fetch'], 1),
                    'write_bytes_per_launch': 1024 * v['write_kb'] / max(v['n_write'], 1)}
    res[k[:120]]['hbm_bytes_per_launch'] = res[k[:120]]['fetch_bytes_per_launch'] + res[k[:120]]['write_bytes_per_launch']
top = dict(sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:24])
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
top['_meta'] = {'csrc_sha256': bench.csrc_fingerprint(),
                'command': 'bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-dp-overhead (PYLC_NO_SIDE_STREAM=1), two rocprofv3 --pmc passes',
                'corrections': 'FETCH_SIZE x2 (gfx950), WRITE_SIZE exact; KB -> bytes; rows summed per Dispatch_Id, launches = distinct dispatches'}
print(json.dumps(top, indent=1))
