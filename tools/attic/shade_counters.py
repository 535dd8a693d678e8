#!/usr/bin/env python3
"""kShade / kSky counters per queue entry and per hit from a profile directory's pmc*_summary.txt (tools/roofline_pmc.sh) + its bench_under_trace.json:
   tools/shade_counters.py profiles/r04_final > profiles/r04_shade/shade_counters.json      (pure text processing)"""
import json, os, re, sys
d = sys.argv[1]
bench = json.loads(open(os.path.join(d, "bench_under_trace.json")).read().strip().splitlines()[-1])
scale = (bench["steps"] + bench["warmup"]) / bench["steps"]                  # counters cover warm-up + timed batch; the same frame is traced throughout
pb = bench["per_bounce_rank0"]
entries = {"kShade<false>": pb[0]["closest_rays"] * scale, "kShade<true>": sum(b["closest_rays"] for b in pb[1:]) * scale}
hits = {"kShade<false>": pb[0]["shadow_rays"] * scale, "kShade<true>": sum(b["shadow_rays"] for b in pb[1:]) * scale}
misses = sum(b["closest_rays"] - b["shadow_rays"] for b in pb) * scale
cnt = {}
for f in sorted(os.listdir(d)):
    if not re.fullmatch(r"pmc\d+_summary\.txt", f):
        continue
    for line in open(os.path.join(d, f)):
        m = re.match(r"(\S.*?)\s+dispatches=\s*(\d+)\s+(.*)", line)
        if not m or not m.group(1).startswith(("kShade", "kSky")):
            continue
        k = cnt.setdefault(m.group(1).strip(), {"dispatches": int(m.group(2))})
        for name, val in re.findall(r"(\w+)=([0-9.e+]+) \(per dispatch", m.group(3)):
            k[name] = float(val)
stats = {r["Name"] if False else None: None for r in []}
import csv
ms = {}
for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    n = r["Name"].replace("rf::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith(("kShade", "kSky")):
        ms[n] = float(r["TotalDurationNs"]) * 1e-6
out = {"profile": os.path.basename(os.path.abspath(d)), "note": "counters summed over every dispatch of the profiled process (warm-up + timed batch, the same frame); "
       "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, uncorrected (calibration: coalesced streams count 0.5 x their bytes, random 64-byte gathers 1.07 x); "
       "per entry = per entry of the bounce's queue (closest-hit ray), per hit = per shaded hit; kSky: per path that left the scene"}
for k, c in cnt.items():
    units = entries.get(k, misses)
    per = {"ms_total": round(ms.get(k, 0.0), 2), "dispatches": c["dispatches"], "units": round(units), "unit": "queue entry" if k in entries else "miss"}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        if name in c: per[name + "_bytes_per_unit"] = round(c[name] * 1024.0 / units, 2)
    for name in ("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_EA0_RDREQ_sum", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
        if name in c: per[name + "_per_unit"] = round(c[name] / units, 3)
    if "TCC_HIT_sum" in c: per["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "TCP_TCC_READ_REQ_LATENCY_sum" in c: per["l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
    if "SQ_WAIT_ANY" in c: per["wave_cycles_waiting"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if k in hits:
        per["hits"] = round(hits[k])
        per["l1_to_l2_read_requests_per_hit"] = round(c.get("TCP_TCC_READ_REQ_sum", 0.0) / hits[k], 3)
        per["fabric_read_requests_per_hit"] = round(c.get("TCC_EA0_RDREQ_sum", 0.0) / hits[k], 3)
        per["ps_per_hit"] = round(ms.get(k, 0.0) * 1e9 / hits[k], 1)
    if ms.get(k):
        per["G_l1_to_l2_read_requests_per_s"] = round(c.get("TCP_TCC_READ_REQ_sum", 0.0) / ms[k] * 1e-6, 1)
        per["fabric_read_GBps_at_64B"] = round(c.get("TCC_EA0_RDREQ_sum", 0.0) * 64.0 / ms[k] * 1e-6, 1)
    out[k] = per
print(json.dumps(out, indent=1))
