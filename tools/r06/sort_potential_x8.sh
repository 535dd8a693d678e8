#!/bin/bash
# round 6, task 3: what a global ray order is worth on the OUT-OF-CACHE atrium (--scene-scale 8), where the closest-hit launches are bound by L1->L2 requests.
# ms per order (HIP events around the launch alone) + per-dispatch L1->L2 read requests and L2 hit / miss counts of the same launches (separate rocprofv3 --pmc passes).
# usage (GPU box, from the repo root): bash tools/r06/sort_potential_x8.sh [tiles = 128] [spp = 64]
TILES=${1:-128}; SPP=${2:-64}
OUT=$PWD/gpurun_out/r06_sort_x8; mkdir -p $OUT
REPO=$PWD
export PYTHONPATH=$REPO
python tools/gpu_sort_potential.py $TILES $SPP plain 8 > $OUT/potential_x8.log 2>&1
tail -40 $OUT/potential_x8.log
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1)); D=/tmp/sortpmc$i; rm -rf $D
  RF_SORT_POTENTIAL_CHILD=1 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $D -o r -- python $REPO/tools/gpu_sort_potential.py $TILES $SPP plain 8 > $OUT/pmc$i.stdout 2> $OUT/pmc$i.stderr || echo "pass $i failed"
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -z "$f" ] && { echo "no output for $ctrs"; tail -3 $OUT/pmc$i.stderr; continue; }
  python3 - "$f" "$OUT/pmc$i.stdout" > $OUT/pmc$i.table <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
labels = [l[6:].strip() for l in open(sys.argv[2]) if l.startswith("LABEL ")]
d = collections.OrderedDict()
for r in rows:
    if "kTraceWide" not in r.get("Kernel_Name", ""): continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
names = sorted({c for v in d.values() for c in v})
print(f"# {len(ids)} kTraceWide dispatches, {len(labels)} labels")
print("label".ljust(34) + " ".join(f"{n:>30s}" for n in names) + "   per ray: " + " ".join(names))
if len(ids) == len(labels):
    seen = set()
    for k, lab in zip(ids, labels):
        if lab in seen: continue   # (every order is launched twice: the first one is kept)
        seen.add(lab)
        rays = int(lab.split("|")[1])
        print(lab.ljust(34) + " ".join(f"{d[k].get(n, 0):30.6g}" for n in names) + "   " + " ".join(f"{d[k].get(n, 0) / rays:10.3f}" for n in names))
else:
    for k in ids: print(str(k).ljust(34) + " ".join(f"{d[k].get(n, 0):30.6g}" for n in names))
PY
  cat $OUT/pmc$i.table | head -60
done
