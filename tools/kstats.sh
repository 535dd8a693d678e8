#!/bin/bash
# usage (GPU box): tools/kstats.sh <cmd...>  -> per-kernel rocprofv3 --stats summary
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o r -- "$@" > /tmp/kst.out 2> /tmp/kst.err || { echo "rocprof failed"; tail -5 /tmp/kst.err; }
tail -3 /tmp/kst.out
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kst/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('rf::', '').replace('void ', '').split('(')[0]
    if float(r['Percentage']) < 0.05: continue
    print(f"{n:36s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={float(r['Percentage']):6.2f}")
PY
