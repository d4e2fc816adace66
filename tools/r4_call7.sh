#!/bin/bash
set -o pipefail
out=gpurun_out/r4g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1; do
  PYLC_EVAL_ONLY=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr$m -- python3 tools/eval_ab.py r101_512_bs32 > $out/run$m.log 2>&1
  f=$(find $out/tr$m -name "*kernel_stats.csv" | head -1)
  echo "== eval_planes=$m"; tail -2 $out/run$m.log | cut -c1-200
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%7.2f ms %6d calls %8.1f us avg  %s' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
print('total %.1f ms' % (tot / 1e6))
PY
  rm -rf $out/tr$m
done
