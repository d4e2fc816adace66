import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'gather' not in k and 'wgrad' not in k and 'gg_pl' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%d  mean %.4g' % (c, len(v), sum(v) / len(v)))
