#!/bin/bash
# Runs on the GPU box (via gpurun):  bash tools/roofline_pmc.sh [outdir]
#   1. tools/microbench/fetch_calib: known byte counts per access pattern, plain (times) and under rocprofv3 --pmc
#      (FETCH_SIZE, WRITE_SIZE, L1 accesses, L2 hit/miss, EA request sizes) -> calibration of the counters
#   2. the bench command: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group
#   3. tools/roofline_post.py: calibration factors + per-ray counter bytes of the closest-hit kernel -> pmc_per_ray.json
# Counters are collected in their own runs (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/roofline}
case $OUT in /*) ;; *) OUT=$PWD/$OUT ;; esac     # the passes run from /tmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-4}; WARMUP=${WARMUP:-2}
# BENCH_ARGS: extra bench.py arguments (e.g. "--scene-scale 8 --spp-per-step 8"); CALIB_FROM: a committed profile directory whose
# calibration files are reused instead of re-running the microbenchmark (same chip, same rocprofv3)
BENCH="python $REPO/bench.py --steps $STEPS --warmup $WARMUP --repeat 1 --no-cpu-baseline --no-counting --no-live-counters --no-regimes ${BENCH_ARGS:-}"   # (--repeat 1: the counters are divided by the rays of ONE timed region + warm-up)
CALIB=$REPO/tools/microbench/fetch_calib
[ -x $CALIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $CALIB $REPO/tools/microbench/fetch_calib.hip
VCALIB=$REPO/tools/microbench/valu_calib
[ -x $VCALIB ] || /opt/rocm/bin/hipcc -std=c++20 -O3 --offload-arch=gfx950 -o $VCALIB $REPO/tools/microbench/valu_calib.hip

summarize() {   # csv, counters -> per-kernel per-dispatch averages
python3 - "$1" "$2" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("rf::", "").replace("void ", "").split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
print("counters:", sys.argv[2])
for k in sorted(agg, key=lambda k: -len(n[k])):
    print(f"{k:60s} dispatches={len(n[k]):5d} " + " ".join(f"{c}={v:.6g} (per dispatch {v/len(n[k]):.6g})" for c, v in sorted(agg[k].items())))
PY
}
pmc() {   # tag, command, counters
  local tag=$1 cmd=$2 ctrs=$3
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/raw_$tag -o r -- $cmd > /dev/null 2> $OUT/$tag.log
  local f=$(find $OUT/raw_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then summarize "$f" "$ctrs" > $OUT/${tag}_summary.txt; else echo "no output for $tag ($ctrs)"; tail -3 $OUT/$tag.log; fi
  rm -rf $OUT/raw_$tag $OUT/$tag.log
}

# ---- 1. calibration microbenchmark
if [ -n "${CALIB_FROM:-}" ]; then
  cp $REPO/$CALIB_FROM/calib_plain.jsonl $REPO/$CALIB_FROM/calib_pmc*_summary.txt $OUT/
  cp $REPO/$CALIB_FROM/valu_calib_w*.jsonl $OUT/ 2> /dev/null
  echo "$CALIB_FROM" > $OUT/calibration_reused_from.txt
else
$CALIB > $OUT/calib_plain.jsonl 2> $OUT/calib_plain.err
# (round 4) VALU issue rates per instruction class, at the traversal kernel's occupancy (6 waves per SIMD) and at 8; EXEC masks full .. one lane
for w in 6 8; do $VCALIB $w 1500 > $OUT/valu_calib_w$w.jsonl 2>> $OUT/calib_plain.err; done
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); pmc calib_pmc$i "$CALIB" "$ctrs"
done
fi

# ---- 2. the bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
for f in $(find $OUT/trace -name '*kernel_stats.csv' | head -1); do cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/trace
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); pmc pmc$i "$BENCH" "$ctrs"
done

# ---- 2b. per bounce: the same counters per DISPATCH of the two traversal kernels (the launches of one batch are bounces 1..B)
perdispatch() {   # tag, counters
  local tag=$1 ctrs=$2
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/raw_$tag -o r -- $BENCH > /dev/null 2> $OUT/$tag.log
  local f=$(find $OUT/raw_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
python3 - "$f" > $OUT/${tag}.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    k = r.get("Kernel_Name", "")
    kind = "closest" if "kTraceWide<false" in k else "shadow" if "kTraceWide<true" in k else "shadow_look" if "kShadowFirstLook" in k else None   # (shadow_look: the dense first pass of a bounce's any-hit launch)
    if kind is None: continue
    d.setdefault((kind, int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in d.values() for c in v})
print("kernel,dispatch," + ",".join(names))
for (kind, disp) in sorted(d, key=lambda t: (t[0], t[1])):
    print(f"{kind},{disp}," + ",".join(repr(d[(kind, disp)].get(n, 0.0)) for n in names))
PY
  else echo "no output for $tag ($ctrs)"; tail -3 $OUT/$tag.log; fi
  rm -rf $OUT/raw_$tag $OUT/$tag.log
}
perdispatch per_bounce_pmc1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
perdispatch per_bounce_pmc2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"
perdispatch per_bounce_pmc3 "TCC_HIT_sum TCC_MISS_sum"
perdispatch per_bounce_pmc4 "FETCH_SIZE"
perdispatch per_bounce_pmc5 "WRITE_SIZE"

# ---- 3. post-process
python3 $REPO/tools/roofline_post.py $OUT
ls -la $OUT
