#!/bin/bash
# usage (GPU box): tools/pmc_bounces.sh <out.csv> <spp> "<option set>" ["<option set>" ...]
# Per-DISPATCH counters of the traversal launches (closest-hit and shadow, bounce 1..B of ONE batch of the atrium at 1080p) for each
# option set ('-' = defaults), one process per option set and counter group:  variant,kernel,bounce,rays,<counters...>
# Ray counts per bounce come from the renderer's own bounce statistics (printed by the same process).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; SPP=$2; shift; shift
case $OUT in /*) ;; *) OUT=$PWD/$OUT ;; esac
mkdir -p $(dirname $OUT); cd /tmp && export TMPDIR=/tmp
GROUPS_=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "FETCH_SIZE")
rm -f $OUT.parts; : > $OUT.parts
for v in "$@"; do
  g=0
  for ctrs in "${GROUPS_[@]}"; do
    g=$((g+1)); D=/tmp/pmcb_$g; rm -rf $D
    timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $D -o r -- python $REPO/tools/gpu_variant_bounces.py $SPP "$v" > /tmp/pmcb_$g.out 2> /tmp/pmcb_$g.log || echo "pass $g of '$v' failed"
    f=$(find $D -name '*counter_collection.csv' | head -1)
    [ -z "$f" ] && { echo "no output for $v / $ctrs"; tail -3 /tmp/pmcb_$g.log; continue; }
    python3 - "$f" "$v" /tmp/pmcb_$g.out >> $OUT.parts <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
rays = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
d = collections.OrderedDict()
for r in rows:
    k = r.get("Kernel_Name", "")
    kind = "closest" if "kTraceWide<false" in k else "shadow" if "kTraceWide<true" in k else None
    if kind is None: continue
    d.setdefault((kind, int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for kind in ("closest", "shadow"):
    ids = sorted(i for (k, i) in d if k == kind)[-len(rays[kind]):]      # the last frame's launches (the first frame warms up)
    for b, i in enumerate(ids):
        for c, val in d[(kind, i)].items():
            print(json.dumps(dict(variant=sys.argv[2], kernel=kind, bounce=b + 1, rays=rays[kind][b], ms=rays["ms_" + kind][b], counter=c, value=val)))
PY
    rm -rf $D
  done
done
python3 - $OUT.parts $OUT <<'PY'
import json, sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]):
    r = json.loads(l)
    e = t.setdefault((r["variant"], r["kernel"], r["bounce"]), dict(rays=r["rays"], ms=r["ms"]))
    e[r["counter"]] = r["value"]
names = sorted({c for e in t.values() for c in e if c not in ("rays", "ms")})
with open(sys.argv[2], "w") as f:
    f.write("variant,kernel,bounce,rays,ms_unprofiled_pass," + ",".join(names) + "\n")
    for (v, k, b), e in t.items():
        f.write(f'"{v}",{k},{b},{e["rays"]},{e["ms"]:.3f},' + ",".join(repr(e.get(n, 0.0)) for n in names) + "\n")
# derived per-ray table
print(f"{'variant':46s} {'kernel':8s} b {'Mrays':>7s} {'L1acc/ray':>9s} {'L1/clk/CU':>9s} {'L2req/ray':>9s} {'Greq/s':>7s} {'VALU/ray':>8s} {'VALUshare':>9s} {'lanes':>5s} {'SALU/ray':>8s} {'VMEM/ray':>8s} {'LDS/ray':>7s} {'L2hit':>5s} {'ms(gui)':>7s} {'fabricB/ray':>11s} {'fabric TB/s':>11s}")
for (v, k, b), e in t.items():
    rays = max(e["rays"], 1); cyc = (e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) or 1.0      # the counter is summed over the 8 XCDs
    ms = cyc / 2.4e6
    valu = e.get("SQ_INSTS_VALU", 0.0)
    print(f"{v[:46]:46s} {k:8s} {b} {rays/1e6:7.1f} {e.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/rays:9.1f} {e.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/cyc/256:9.3f} "
          f"{e.get('TCP_TCC_READ_REQ_sum',0)/rays:9.2f} {e.get('TCP_TCC_READ_REQ_sum',0)/(ms*1e-3)/1e9:7.1f} {valu/rays:8.1f} "
          f"{valu*4.0/1024/cyc:9.3f} {e.get('SQ_THREAD_CYCLES_VALU',0)/max(e.get('SQ_ACTIVE_INST_VALU',0)*64,1):5.2f} {e.get('SQ_INSTS_SALU',0)/rays:8.1f} {e.get('SQ_INSTS_VMEM_RD',0)/rays:8.1f} {e.get('SQ_INSTS_LDS',0)/rays:7.1f} "
          f"{e.get('TCC_HIT_sum',0)/max(e.get('TCC_HIT_sum',0)+e.get('TCC_MISS_sum',0),1):5.2f} {ms:7.2f} {e.get('FETCH_SIZE',0)*1024*0.93/rays:11.1f} {e.get('FETCH_SIZE',0)*1024*0.93/(ms*1e-3)/1e12:11.3f}")
PY
rm -f $OUT.parts
