#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the same
# bench command.  Summaries land in gpurun_out/prof/ ; copy the ones to be judged into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps ${STEPS:-32} --warmup 4 --no-cpu-baseline"

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.log
for f in $(find $OUT/trace -name '*kernel_stats.csv' -o -name '*_stats.csv' | head -5); do cp $f $OUT/; done

i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc$i -o r -- $CMD > /dev/null 2> $OUT/pmc$i.log
  f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$ctrs" > $OUT/pmc${i}_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("rf::", "").replace("void ", "").split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
print("counters:", sys.argv[2])
for k in sorted(agg, key=lambda k: -len(n[k])):
    print(f"{k:60s} dispatches={len(n[k]):5d} " + " ".join(f"{c}={v:.6g} (per dispatch {v/len(n[k]):.6g})" for c, v in agg[k].items()))
PY
  fi
  rm -rf $OUT/pmc$i
done
rm -rf $OUT/trace/*/*.db 2>/dev/null
ls -la $OUT
