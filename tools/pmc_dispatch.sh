#!/bin/bash
# usage (GPU box): tools/pmc_dispatch.sh "<cmd>" <kernel-substring> "CTR1 CTR2 ..." [...]  -> per-dispatch counter rows, in dispatch order
CMD="$1"; KSUB="$2"; shift; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1)); D=/tmp/pmcd$i; rm -rf $D
  timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $D -o r -- $CMD > /dev/null 2> /tmp/pmcd$i.log || echo "pass $i failed"
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -z "$f" ] && { echo "no output for $ctrs"; tail -3 /tmp/pmcd$i.log; continue; }
  python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if sys.argv[2] not in r.get("Kernel_Name", ""): continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in d.values() for c in v})
print("dispatch " + " ".join(f"{n:>28s}" for n in names))
for k in sorted(d)[-16:]:
    print(f"{k:8d} " + " ".join(f"{d[k].get(n, 0):28.5g}" for n in names))
PY
done
