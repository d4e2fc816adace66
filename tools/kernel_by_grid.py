"""Per (kernel, grid size) launch count and average duration from a rocprofv3 kernel trace CSV, restricted to kernels whose name contains
one of the given substrings (default: the BatchNorm family); also the gap between each such kernel's end and the start of the NEXT kernel on
its queue.    python tools/kernel_by_grid.py <kernel_trace.csv> [substr ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:] or ['bn_']
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
acc = collections.defaultdict(lambda: [0, 0, 0])
for q, rs in byq.items():
    rs.sort(key=lambda r: r['s'])
    for i, r in enumerate(rs):
        if not any(s in r['Kernel_Name'] for s in subs):
            continue
        key = (r['Kernel_Name'].split('(')[0][-70:], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))
        a = acc[key]
        a[0] += 1; a[1] += r['e'] - r['s']
        if i + 1 < len(rs): a[2] += max(rs[i + 1]['s'] - r['e'], 0)
for (name, grid), (n, t, gap) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-72s blocks %6d  n %5d  avg %8.1f us  gap-after %6.1f us  total %8.2f ms' % (name, grid, n, t / n / 1e3, gap / n / 1e3, t / 1e6))
