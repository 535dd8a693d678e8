#!/bin/bash
# round 6, task 8: texels in 8 x 8 tiles -- kShade per bounce (ms from the kernel trace, L1->L2 read requests per hit from one --pmc pass) with the option off / on, plain atrium 1080p 64 spp
OUT=$PWD/gpurun_out/r06_texel; mkdir -p $OUT
REPO=$PWD; export PYTHONPATH=$REPO
python tools/r06/ab_variants.py 64 "texel_tiles=0" "texel_tiles=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_texel_tiles.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  D=/tmp/texpmc$v; rm -rf $D
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $D -o r -- python $REPO/tools/gpu_variant_bounces.py 64 "texel_tiles=$v" > $OUT/pmc_$v.out 2> $OUT/pmc_$v.err || echo "pass $v failed"
  python3 - $D $OUT/pmc_$v.out $v <<'PY' | tee -a $OUT/kshade_requests.txt
import csv, sys, glob, json, collections
d, outp, v = sys.argv[1], sys.argv[2], sys.argv[3]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rays = json.loads([l for l in open(outp) if l.startswith("{")][-1])
ctr = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if "kShade<" not in r["Kernel_Name"]: continue
    ctr.setdefault(int(r["Dispatch_Id"]), {"name": "sorted" if "kShade<true>" in r["Kernel_Name"] else "plain"})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "kShade<" in r["Kernel_Name"]: dur[int(r["Dispatch_Id"])] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
ids = sorted(ctr)[-len(rays["closest"]):]          # the measured frame's eight kShade launches
print(f"texel_tiles={v}: bounce  hits(M)  L1->L2 read req/hit  L1 accesses/hit  kShade ms (under the counter pass)")
for b, i in enumerate(ids):
    hits = rays["shadow"][b]                       # hits of bounce b = its shadow rays
    print(f"   {b + 1}  {hits / 1e6:8.1f}  {ctr[i].get('TCP_TCC_READ_REQ_sum', 0) / max(hits, 1):8.2f}  {ctr[i].get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / max(hits, 1):8.2f}  {dur.get(i, float('nan')):8.3f}")
PY
done
cd $REPO
for v in 0 1; do
  D=/tmp/textr$v; rm -rf $D
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $REPO/tools/gpu_variant_bounces.py 64 "texel_tiles=$v" > /dev/null 2>&1 )
  f=$(find $D -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $v <<'PY' | tee -a $OUT/kshade_ms.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "kShade<" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
ms = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6 for r in rows][-8:]
print(f"texel_tiles={sys.argv[2]}: kShade ms per bounce (kernel trace, no counters): " + " ".join(f"{x:6.3f}" for x in ms) + f" | sum {sum(ms):7.3f}")
PY
done
