#!/usr/bin/env python3
"""tools/roofline_post.py <dir>  -- turns the summaries tools/roofline_pmc.sh wrote into
     <dir>/calibration.json   counter / known-bytes ratios of the microbenchmark, per access pattern
     <dir>/pmc_per_ray.json   fabric-side bytes and vector-L1 accesses per ray of the closest-hit traversal kernel
     <dir>/per_bounce.json    per bounce of the profiled batch: vector-L1 accesses per ray and per clock per CU, L1 -> L2 requests,
                              L2 hit rate, VALU issue share (which unit binds where)
(pmc_per_ray.json is what bench.py reads from profiles/pmc_per_ray.json).  Pure text processing; runs anywhere."""
import json
import os
import re
import sys

out = sys.argv[1]


def summary(path):
    """-> {kernel: {counter: per-dispatch average, "_dispatches": n}}"""
    res = {}
    if not os.path.exists(path):
        return res
    for line in open(path):
        m = re.match(r"(\S.*?)\s+dispatches=\s*(\d+)\s+(.*)", line)
        if not m:
            continue
        k = m.group(1).strip()
        d = {"_dispatches": int(m.group(2))}
        for c, tot, per in re.findall(r"(\w+)=(\S+) \(per dispatch ([0-9.eE+-]+)\)", m.group(3)):
            d[c] = float(per)
        res[k] = d
    return res


def merged(prefix, count):
    res = {}
    for i in range(1, count + 1):
        for k, d in summary(os.path.join(out, f"{prefix}{i}_summary.txt")).items():
            res.setdefault(k, {}).update(d)
    return res


# ---------------------------------------------------------------- calibration
plain = {}
for line in open(os.path.join(out, "calib_plain.jsonl")):
    line = line.strip()
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    key = j["kernel"]
    if j.get("rep", 1) == 0:
        continue          # first pass of the read kernels = cold; the counters below average both, so keep rep 1 as the time
    plain[key] = j
cal = merged("calib_pmc", 6)
calibration = {}
for k, j in plain.items():
    c = cal.get(k, {})
    known = j.get("bytes", j.get("bytes_lines", j.get("bytes_requested")))
    row = dict(known_bytes=known, ms=j["ms"], rate=j.get("GBps", j.get("Grecords_per_s")))
    if "FETCH_SIZE" in c:
        row["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024.0
        row["fetch_ratio"] = row["FETCH_SIZE_bytes"] / known
    if "WRITE_SIZE" in c:
        row["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024.0
        row["write_ratio"] = row["WRITE_SIZE_bytes"] / known
    if "records" in j and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        row["l1_accesses_per_record"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / j["records"]
        row["l1_to_l2_read_requests_per_record"] = c.get("TCP_TCC_READ_REQ_sum", 0.0) / j["records"]
    if "TCC_HIT_sum" in c:
        row["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    for name in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
        if name in c:
            row[name] = c[name]
    calibration[k] = row


def ratio(kernel, field):
    return calibration.get(kernel, {}).get(field)


# The factors applied to the traversal kernel's counters:
#   reads : the record gather over a table far larger than L2 + Infinity Cache -- every record is one distinct 64-B line
#           that must come through the fabric, so known bytes = records x 64 and factor = known / FETCH_SIZE
#   writes: coalesced full-line stores of a table far larger than the caches: factor = bytes / WRITE_SIZE
fr = ratio("gather56<33>", "fetch_ratio")
wr = ratio("fill16<33>", "write_ratio")
factors = dict(fetch_calibration=(1.0 / fr) if fr else None, write_calibration=(1.0 / wr) if wr else None,
               fetch_pattern="gather56<33>: 56 of 64 B of a random 64-B record per lane, 8 GiB table (all from HBM): known = records x 64 B",
               write_pattern="fill16<33>: coalesced 16 B/lane stores of 8 GiB: known = bytes written",
               streaming_read_ratio=ratio("stream16<33>", "fetch_ratio"),
               note="ratio = counter bytes / known bytes; MI355X_MICROARCH.md predicts 0.5 for wide coalesced streaming reads on gfx950")
# ---------------------------------------------------------------- VALU issue calibration (tools/microbench/valu_calib.hip, round 4)
# cycles per wave-instruction per SIMD = wall time of a launch that keeps `waves_per_simd` waves on every SIMD x the shader clock the launch ran at
# (s_memtime / s_memrealtime inside the kernel) / instructions issued per SIMD.  `valu_issue_peak` is what the roofline's VALU ceiling uses: the FASTEST
# MIXED instruction stream measured (streams of a single class reach 2.3 cycles for plain VOP1 / VOP2 arithmetic on VGPRs, but the traversal kernel
# did not get faster when 15 of its selects per step were moved into that class: profiles/r04_lanes) -- so a launch at fraction 1 issues as fast as
# any mixed stream this chip was seen to issue.
valu = {}
for w in (6, 8):
    path = os.path.join(out, f"valu_calib_w{w}.jsonl")
    if not os.path.exists(path):
        continue
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        j = json.loads(line)
        if "class" not in j:
            continue
        cyc = j["ns_per_simd_inst"] * j["memtime_mhz"] * 1e-3
        valu.setdefault(j["class"], {}).setdefault(j["exec"], {})[f"w{w}"] = round(cyc, 3)
valu_issue = None
if valu:
    def cyc_of(cls, mask="full", w="w6"):
        return valu.get(cls, {}).get(mask, {}).get(w)
    mixed = {c: cyc_of(c) for c in valu if (c.startswith("mix_") or c in ("half_quad_step_mix", "v_cmp+v_cndmask")) and cyc_of(c)}
    fastest = min(mixed, key=mixed.get) if mixed else None
    step = cyc_of("half_quad_step_mix")
    valu_issue = dict(
        cycles_per_instruction_at_6_waves_per_simd={c: cyc_of(c) for c in sorted(valu) if cyc_of(c)},
        half_quad_step_mix_by_exec_mask={m: v.get("w6") for m, v in valu.get("half_quad_step_mix", {}).items()},
        v_fma_mix_f32_by_exec_mask={m: v.get("w6") for m, v in valu.get("v_fma_mix_f32", {}).items()},
        step_mix_cycles_per_instruction=step,
        fastest_mixed_stream=fastest, valu_issue_peak_cycles_per_instruction=mixed.get(fastest) if fastest else None,
        valu_issue_peak_instructions_per_simd_clk=round(1.0 / mixed[fastest], 4) if fastest else None,
        half_empty_exec_speedup=round(step / valu["half_quad_step_mix"]["low32"]["w6"], 3) if step and valu.get("half_quad_step_mix", {}).get("low32", {}).get("w6") else None,
        note="cycles per wave64 instruction per SIMD, wall-clock based, every SIMD of the chip loaded with 6 waves; half_empty_exec_speedup ~ 1: a VALU instruction "
             "whose EXEC mask has half (or three quarters) of its lanes off issues no faster on gfx950 -- idle lanes are lost issue slots")
json.dump(dict(factors=factors, kernels=calibration, valu_issue=valu_issue), open(os.path.join(out, "calibration.json"), "w"), indent=1)
print(json.dumps(factors, indent=1))
VALU_CPI = (valu_issue or {}).get("valu_issue_peak_cycles_per_instruction") or 4.0      # cycles per instruction at the calibrated issue peak (4: the uncalibrated round-3 model)

# ---------------------------------------------------------------- the bench command
bench = json.loads(open(os.path.join(out, "bench_under_trace.json")).read().strip().splitlines()[-1])
pm = merged("pmc", 7)
# The closest-hit traversal is one kernel template in two instantiations: kTraceWide<false, false, false, false> (bounces 1-2, plain
# 64-byte records) and kTraceWide<false, false, false, true> (bounces >= 3, compact-capable records).  Their dispatches are summed.
KS = [n for n in pm if n.startswith("kTraceWide<false, false, false")]
if not KS:
    raise SystemExit(f"no closest-hit kTraceWide instantiation in the PMC summaries: {list(pm)}")
k = {"_dispatches": sum(pm[n]["_dispatches"] for n in KS)}
for n in KS:
    for c, v in pm[n].items():
        if c != "_dispatches":
            k[c] = k.get(c, 0.0) + v * pm[n]["_dispatches"] / k["_dispatches"]        # average over ALL closest-hit dispatches
# rays behind the counters: every dispatch of the kernel in the profiled process = warm-up + timed steps (the counting pass
# is off); the same frame is traced throughout, so rays scale with the samples.  `rays` = rays per AVERAGE dispatch.
timed_rays = bench["roofline"]["rays_per_launch"] * bench["roofline"]["launches"]
all_rays = timed_rays * (bench["steps"] + bench["warmup"]) / bench["steps"]
rays = all_rays / k["_dispatches"]
W, H = re.search(r"(\d+)x(\d+)", bench["config"]["workload"]).groups()
B = re.search(r"(\d+) bounces", bench["config"]["workload"]).group(1)
name = bench["config"]["workload"].split(" -- ")[0].split(", ")[0]
fetch = k["FETCH_SIZE"] * 1024.0 / rays
write = k["WRITE_SIZE"] * 1024.0 / rays
fc = factors["fetch_calibration"] or 1.0
wc = factors["write_calibration"] or 1.0
per_ray = dict(
    profile=os.path.basename(os.path.abspath(out)), kernel="kTraceWide<closest>", instantiations={n: pm[n]["_dispatches"] for n in KS}, workload=f"{name} {W}x{H}x{B}",
    rays_per_average_dispatch=round(rays), dispatches_averaged=k["_dispatches"], steps=bench["steps"], warmup=bench["warmup"],
    fetch_size_bytes_per_ray=round(fetch, 2), write_size_bytes_per_ray=round(write, 2),
    fetch_calibration=round(fc, 4), write_calibration=round(wc, 4),
    hbm_side_bytes_per_ray=round(fc * fetch + wc * write, 2),
    l1_accesses_per_ray=round(k.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / rays, 2),
    l1_to_l2_read_requests_per_ray=round(k.get("TCP_TCC_READ_REQ_sum", 0.0) / rays, 2),
    l2_hit_rate=round(k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1.0), 4) if "TCC_HIT_sum" in k else None,
    valu_lane_utilisation=round(k["SQ_THREAD_CYCLES_VALU"] / max(k.get("SQ_ACTIVE_INST_VALU", 0.0) * 64, 1.0), 4) if "SQ_THREAD_CYCLES_VALU" in k and "SQ_ACTIVE_INST_VALU" in k else None,
    gpu_cycles_per_launch=k.get("GRBM_GUI_ACTIVE"),
    avg_launch_ms_under_trace=bench["roofline"]["avg_launch_ms"],
    source="rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-counting` (tools/roofline_pmc.sh), one counter group per run, averaged over every launch of "
           "the kernel and divided by the rays one launch traces; FETCH_SIZE / WRITE_SIZE (KB) x 1024 x the calibration factors of calibration.json; fabric-side "
           "requests of the 8 L2s (Infinity-Cache hits included)")
def kernel_block(prefix, rays_timed, label):
    """Fabric-side bytes and L1 / L2 figures per unit of work for every instantiation of a kernel whose name starts with `prefix` (or one of several)."""
    prefixes = (prefix,) if isinstance(prefix, str) else tuple(prefix)
    names = [n for n in pm if n.startswith(prefixes)]
    if not names or not rays_timed:
        return None
    kk = {"_dispatches": sum(pm[n]["_dispatches"] for n in names)}
    for n in names:
        for c, v in pm[n].items():
            if c != "_dispatches":
                kk[c] = kk.get(c, 0.0) + v * pm[n]["_dispatches"] / kk["_dispatches"]
    units = rays_timed * (bench["steps"] + bench["warmup"]) / bench["steps"] / kk["_dispatches"]
    f, w = kk.get("FETCH_SIZE", 0.0) * 1024.0 / units, kk.get("WRITE_SIZE", 0.0) * 1024.0 / units
    return dict(kernel=label, instantiations={n: pm[n]["_dispatches"] for n in names}, units_per_average_dispatch=round(units),
                fetch_size_bytes_per_unit=round(f, 2), write_size_bytes_per_unit=round(w, 2),
                l1_accesses_per_unit=round(kk.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / units, 2),
                l1_to_l2_read_requests_per_unit=round(kk.get("TCP_TCC_READ_REQ_sum", 0.0) / units, 2),
                l2_hit_rate=round(kk["TCC_HIT_sum"] / max(kk["TCC_HIT_sum"] + kk["TCC_MISS_sum"], 1.0), 4) if "TCC_HIT_sum" in kk else None)


# the other two kernels the bench line prices: the shadow traversal (per shadow ray; random record gathers like the closest-hit
# kernel: same read calibration) and kShade (per queue entry = per closest-hit ray; its reads are a mix of coalesced streams --
# counted at 0.5 -- and 128-byte record gathers -- counted at ~1 --, so its bytes are reported with BOTH factors as a bracket)
# (round 4: from bounce 2 on an any-hit launch is kShadowFirstLook -- the dense pass over the queue that answers the rays their cell's leaves stop -- followed by
# kTraceWide<true, ...> over the rest: both are "the shadow traversal", per shadow ray of the queue)
sh = kernel_block(("kTraceWide<true", "kShadowFirstLook"), bench["rays"]["shadow"], "kTraceWide<shadow> + kShadowFirstLook")
if sh:
    sh["hbm_side_bytes_per_unit"] = round(fc * sh["fetch_size_bytes_per_unit"] + wc * sh["write_size_bytes_per_unit"], 2)
    per_ray["shadow"] = sh
kd = kernel_block("kShade<", bench["rays"]["closest"], "kShade")
if kd:
    stream = factors.get("streaming_read_ratio") or 0.5
    kd["hbm_side_bytes_per_unit_low"] = round(fc * kd["fetch_size_bytes_per_unit"] + wc * kd["write_size_bytes_per_unit"], 2)
    kd["hbm_side_bytes_per_unit_high"] = round(kd["fetch_size_bytes_per_unit"] / stream + wc * kd["write_size_bytes_per_unit"], 2)
    kd["note"] = "low: every read counted as a random gather (FETCH_SIZE x %.3f); high: every read counted as a coalesced stream (FETCH_SIZE / %.3f)" % (fc, stream)
    per_ray["shade"] = kd
# ceilings measured by the calibration microbenchmark on this chip (tools/microbench/fetch_calib.hip): L1 -> L2 request rate of a
# random 64-byte record stream served by L2 / Infinity Cache (32 MiB table) and by HBM (8 GiB table), and the streaming rates
req_rates = {k2: v["rate"] * v["l1_to_l2_read_requests_per_record"] for k2, v in calibration.items()
             if k2.startswith("gather") and k2.endswith("<25>") and "l1_to_l2_read_requests_per_record" in v and not k2.startswith("gather12")}
hbm_gather = {k2: v["rate"] * 64.0 for k2, v in calibration.items() if k2.startswith("gather") and k2.endswith("<33>") and not k2.startswith("gather12")}
per_ray["ceilings"] = dict(
    l1_to_l2_requests_G_per_s=round(max(req_rates.values()), 1) if req_rates else None,
    l1_to_l2_requests_source={k2: round(v, 1) for k2, v in req_rates.items()},
    hbm_random_64B_gather_GBps=round(max(hbm_gather.values()), 1) if hbm_gather else None,
    hbm_random_64B_gather_source={k2: round(v, 1) for k2, v in hbm_gather.items()},
    hbm_stream_read_GBps=calibration.get("stream16<33>", {}).get("rate"), hbm_stream_write_GBps=calibration.get("fill16<33>", {}).get("rate"),
    valu_issue_peak_cycles_per_instruction=VALU_CPI, valu_issue_peak_source=(valu_issue or {}).get("fastest_mixed_stream") or "uncalibrated: 4 cycles per wave64 instruction",
    valu_step_mix_cycles_per_instruction=(valu_issue or {}).get("step_mix_cycles_per_instruction"), valu_half_empty_exec_speedup=(valu_issue or {}).get("half_empty_exec_speedup"),
    note="measured on this chip by tools/microbench/fetch_calib.hip: G L1->L2 read requests/s of a random 64-byte record gather over a 32 MiB table "
         "(cache resident) = records/s x requests per record; bytes/s of the same gather over an 8 GiB table (every record from HBM)")
json.dump(per_ray, open(os.path.join(out, "pmc_per_ray.json"), "w"), indent=1)
print(json.dumps(per_ray, indent=1))


# ---------------------------------------------------------------- per bounce (per dispatch of the traversal kernels)
def per_dispatch(path):
    res = {}
    if not os.path.exists(path):
        return res
    lines = open(path).read().strip().splitlines()
    if not lines:
        return res
    names = lines[0].split(",")[2:]
    for ln in lines[1:]:
        f = ln.split(",")
        res.setdefault(f[0], []).append(dict(zip(names, (float(x) for x in f[2:]))))
    return res


pb = {}
for i in (1, 2, 3, 4, 5):
    for kind, rows in per_dispatch(os.path.join(out, f"per_bounce_pmc{i}.csv")).items():
        for j, row in enumerate(rows):
            pb.setdefault(kind, {}).setdefault(j, {}).update(row)
CUS, XCDS, SIMDS = 256, 8, 1024
bounces = bench.get("per_bounce_rank0", [])
nb = len(bounces)
table = []
for kind in ("closest", "shadow"):
    rows = pb.get(kind, {})
    if not rows or not nb:
        continue
    last = sorted(rows)[-nb:]          # the timed batch's launches are the last nb dispatches of the kernel
    if kind == "shadow" and pb.get("shadow_look"):
        # the first looks of the timed batch are the LAST dispatches of that kernel and belong to its last bounces (a cold first batch has none; bounce 1 has none
        # by default): their counters -- cycles included -- are added to the traversal launch of the same bounce
        looks = pb["shadow_look"]
        lk = sorted(looks)[-min(nb - 1, len(looks)):] if nb > 1 else []
        merged_rows = {j: dict(rows[j]) for j in last}
        for j, jl in zip(last[len(last) - len(lk):], lk):
            for c, v in looks[jl].items():
                merged_rows[j][c] = merged_rows[j].get(c, 0.0) + v
        rows = merged_rows
    for b, j in zip(bounces, last):
        c = rows[j]
        rays = b[f"{kind}_rays"]
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS          # the counter is summed over the 8 XCDs
        e = dict(kernel=kind, bounce=b["bounce"], rays=rays, ms=b[f"ms_{kind}"], gpu_cycles=round(cyc))
        if rays and cyc:
            e["l1_accesses_per_ray"] = round(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / rays, 1)
            e["l1_accesses_per_clk_per_cu"] = round(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / cyc / CUS, 3)
            e["l1_to_l2_requests_per_ray"] = round(c.get("TCP_TCC_READ_REQ_sum", 0.0) / rays, 2)
            if "SQ_INSTS_VALU" in c:
                e["valu_instructions_per_ray"] = round(c["SQ_INSTS_VALU"] / rays, 1)
                e["valu_issue_share"] = round(c["SQ_INSTS_VALU"] * VALU_CPI / SIMDS / cyc, 3)     # against the calibrated issue peak (valu_calib: fastest mixed stream)
                e["valu_issue_share_4_cycle_model"] = round(c["SQ_INSTS_VALU"] * 4.0 / SIMDS / cyc, 3)     # rounds 2-3: 4 cycles per wave64 instruction per SIMD (uncalibrated: > 1 at bounce 1)
                if "SQ_THREAD_CYCLES_VALU" in c:
                    e["valu_active_lanes_per_instruction"] = round(c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_INSTS_VALU"] * 64.0, 1.0), 3)
                    # informative, NOT the ceiling: valu_calib's step mix issues 1 / half_empty_exec_speedup x slower under a half-empty EXEC mask; scaled linearly
                    # with the idle-lane share of this launch's instructions, that is how busy the VALU is AT THAT OCCUPANCY (may exceed 1 by the noise of the model)
                    hs = (valu_issue or {}).get("half_empty_exec_speedup")
                    if hs:
                        penalty = 1.0 + min(max(1.0 - e["valu_active_lanes_per_instruction"], 0.0), 0.75) / 0.5 * (1.0 / hs - 1.0)
                        e["valu_issue_share_exec_adjusted"] = round(e["valu_issue_share"] * penalty, 3)
                e["salu_instructions_per_ray"] = round(c.get("SQ_INSTS_SALU", 0.0) / rays, 1)
            if "TCC_HIT_sum" in c:
                e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3)
            if "FETCH_SIZE" in c:
                # fabric-side bytes of THIS launch (random record gathers: the gather calibration), and the rate they moved at
                e["hbm_side_bytes_per_ray"] = round((fc * c["FETCH_SIZE"] + wc * c.get("WRITE_SIZE", 0.0)) * 1024.0 / rays, 1)
                if b[f"ms_{kind}"] > 0:
                    e["hbm_side_GBps"] = round(e["hbm_side_bytes_per_ray"] * rays / (b[f"ms_{kind}"] * 1e-3) / 1e9, 1)
        table.append(e)
if table and per_ray.get("shadow"):
    # the kernel block above averages over EVERY dispatch of the process, i.e. also over the first batch, whose any-hit launches fill a cold occluder grid (full
    # walks, no kShadowFirstLook); what the bench line prices is the steady state: the timed batch's launches, weighted by their rays
    rows_s = [e for e in table if e["kernel"] == "shadow" and e.get("rays") and "hbm_side_bytes_per_ray" in e]
    tot = sum(e["rays"] for e in rows_s)
    if tot:
        wavg = lambda key: round(sum(e.get(key, 0.0) * e["rays"] for e in rows_s) / tot, 2)
        per_ray["shadow"].update(hbm_side_bytes_per_unit=wavg("hbm_side_bytes_per_ray"), l1_to_l2_read_requests_per_unit=wavg("l1_to_l2_requests_per_ray"),
                                 l1_accesses_per_unit=wavg("l1_accesses_per_ray"), valu_instructions_per_unit=wavg("valu_instructions_per_ray"),
                                 averaged_over="the launches of the timed batch (warm occluder grid), weighted by their rays; fetch_size / write_size / l2_hit_rate above: every dispatch of the process")
if table:
    per_ray["per_bounce"] = table
    json.dump(per_ray, open(os.path.join(out, "pmc_per_ray.json"), "w"), indent=1)
    json.dump(dict(profile=os.path.basename(os.path.abspath(out)), spp_of_the_batch=bench["config"]["spp"],
                   valu_issue_peak_cycles_per_instruction=VALU_CPI,
                   note="one row per launch of the timed batch; l1_accesses_per_clk_per_cu is against the ceiling of 1 (one vector-L1 tag access per clock "
                        "per CU); valu_issue_share = wave instructions x the calibrated cycles per instruction (calibration.json: valu_issue, the fastest mixed "
                        "stream of tools/microbench/valu_calib) / SIMD-cycles available", rows=table),
              open(os.path.join(out, "per_bounce.json"), "w"), indent=1)
    for e in table:
        print(e)
