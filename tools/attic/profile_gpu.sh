#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the same
# bench command.  Summaries land in gpurun_out/prof/ ; copy the ones to be judged into profiles/.
#   STEPS (default 128) / WARMUP (default 8): must equal bench.py's defaults for pmc_traffic.json to apply.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-128}; WARMUP=${WARMUP:-32}
CMD="python $REPO/bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-counting"

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.log
for f in $(find $OUT/trace -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats.csv; done

i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc$i -o r -- $CMD > /dev/null 2> $OUT/pmc$i.log
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
rm -rf $OUT/trace
# HBM-side traffic of the closest-hit traversal kernel per launch, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE (KB) x 2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane reads) + WRITE_SIZE (KB)
python3 - $OUT $STEPS <<'PY'
import json, re, sys
out, steps = sys.argv[1], int(sys.argv[2])
def per_dispatch(path, ctr):
    for line in open(path):
        if line.startswith("kTraceWide<false, false, false>"):
            n = int(re.search(r"dispatches=\s*(\d+)", line).group(1))
            v = float(re.search(ctr + r"=\S+ \(per dispatch ([0-9.e+]+)\)", line).group(1))
            return n, v
    raise SystemExit("kernel not found in " + path)
n, fetch_kb = per_dispatch(out + "/pmc1_summary.txt", "FETCH_SIZE")
_, write_kb = per_dispatch(out + "/pmc2_summary.txt", "WRITE_SIZE")
bench = json.loads(open(out + "/bench_under_trace.json").read().strip().splitlines()[-1])
json.dump({"kernel": "kTraceWide<closest>", "workload": "1920x1080x8", "launches_per_128_steps": round(bench["roofline"]["launches"] * 128 / steps),
           "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024), "fetch_size_kb_per_dispatch": fetch_kb, "write_size_kb_per_dispatch": write_kb,
           "dispatches_averaged": n,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-cpu-baseline --no-counting` (tools/profile_gpu.sh): FETCH_SIZE x 2 "
                     "(gfx950 correction for 16 B/lane reads) + WRITE_SIZE, averaged over all launches of the kernel (warm-up batch = same size); "
                     "fabric-side requests of the 8 L2s, Infinity-Cache hits included; scene = synthetic atrium"},
          open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/pmc_traffic.json").read())
PY
ls -la $OUT
