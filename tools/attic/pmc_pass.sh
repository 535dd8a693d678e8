#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh "<cmd>" "CTR1 CTR2" ["CTR3 ..."] ...   one rocprofv3 --pmc pass per group
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CMD="$1"; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1)); D=/tmp/pmcpass$i; rm -rf $D
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $D -o r -- $CMD > /dev/null 2> /tmp/pmcpass$i.log || echo "pass $i ($ctrs) timed out or failed"
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -z "$f" ] && { echo "no output for $ctrs"; tail -3 /tmp/pmcpass$i.log; continue; }
  python3 - "$f" "$ctrs" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("rf::", "").replace("void ", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
print("## counters:", sys.argv[2])
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    if not k.startswith("k"): continue
    print(f"{k:34s} n={len(n[k]):4d} " + " ".join(f"{c}={v/len(n[k]):.5g}" for c, v in sorted(agg[k].items())))
PY
done
