"""Per-queue busy time and top kernels from a rocprofv3 kernel trace CSV (last `steps` steps of the run)."""
import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
# one step = from one adamw kernel to the next
ad = [r for r in rows if 'adamw' in r['Kernel_Name']]
t0, t1 = ad[-3]['e'], ad[-1]['e']
sel = [r for r in rows if r['s'] >= t0 and r['e'] <= t1]
print('window: 2 steps, %.2f ms per step' % ((t1 - t0) / 2e6))
byq = collections.defaultdict(list)
for r in sel:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    k = collections.Counter()
    for r in rs:
        k[r['Kernel_Name'].split('(')[0][-60:]] += r['e'] - r['s']
    print('queue', q, 'kernels', len(rs), 'busy %.2f ms per step' % (busy / 2e6))
    for name, t in k.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
        print('    %-62s %.2f ms/step' % (name, t / 2e6))
# union busy time
ev = sorted([(r['s'], 1) for r in sel] + [(r['e'], -1) for r in sel])
cur = 0; last = None; tot = 0
for t, d in ev:
    if cur > 0: tot += t - last
    cur += d; last = t
print('union busy %.2f ms per step' % (tot / 2e6))
