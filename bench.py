#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without a launcher's WORLD_SIZE in the
                                                            environment this process re-launches itself as N ranks under
                                                            torch.distributed.run --standalone on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the driver's form)

The timed region is run --repeat R times (default 3; each is EXACTLY K steps between barriers); `value` and `ms_per_step`
are the MEDIAN repeat's, the others are listed under `repeats`.

Workload (BASELINE.json configs[2]/[3]): Sponza stand-in ("synthetic atrium": the real assets/Sponza.glb is absent
from the reference mount), 1920x1080, 8 bounces, default camera and sky.

A STEP is SPP_PER_STEP = 16 samples per pixel of the whole frame through the full wavefront pipeline
(raygen -> [closest-hit traversal -> shade/NEE -> shadow traversal] x 8 -> sky -> accumulate); K timed steps = 16 K spp
(the default K = 16 is config 3's 256 spp).  Samples are traced in equal batches of up to the tuned depth (1 Gi paths = 514 spp
of a 1080p frame per batch; a rank that owns 1/N of the tiles traces N times as many samples per batch), whatever K
is.  Inputs (scene, textures, tables) are resident in HBM before the timed region.  `value` = rays traced (closest-hit
+ shadow traversals, counted on the device) by all ranks per second.  With N > 1 the frame is tile-sharded (strong
scaling: total work fixed) and the per-rank tile buffers are brought to rank 0 once at frame end by the C++ RCCL
exchange (rf_renderer_gather_frame: grouped ncclSend/ncclRecv + device un-tile), inside the timed region.

The JSON line also carries
  roofline     the closest-hit traversal kernel against the HBM roofline, in MEASURED bytes: `traffic` = fabric-side
               bytes per launch from rocprofv3 FETCH_SIZE / WRITE_SIZE -- two child passes of this workload under
               `rocprofv3 --pmc <one counter> --kernel-trace` after the timed region (`traffic_live`; the committed
               per-ray profile profiles/pmc_per_ray.json serves when rocprofv3 is missing or --no-live-counters is
               given), calibrated on known byte counts in the same access pattern --, `achieved` = traffic / average
               launch duration (HIP events on the renderer's stream inside the timed region), frac = achieved / 8 TB/s.  SURVEY.md 8(d)'s
               algorithmic figure (48 B per reference node visit ...) is reported beside it under `algorithmic`
               and is NOT divided by the HBM peak: the 34 MB of BVH + triangles live in L2 / Infinity Cache, so it is
               a cache rate.  `l1` is the ceiling that binds: vector-L1 line accesses per second against
               256 CUs x 1 access/clk x 2.4 GHz.
  cpu_baseline the CPU oracle (a port of the reference's algorithm) on a bounded crop of the same frame and the
               reference's bvh-visualizer primary-ray loop, on this box's host cores (1 thread and all of them).
  parity_crop  the GPU frame of the timed region against the oracle image of that crop.
  occluder_cache  one more, UNTIMED repeat of the same frames with the any-hit launches' occluder cache off (a shadow
               ray first visits the leaves that stopped the last rays from its cell of the scene): the rate and the shadow
               launches' time without it, the whole image compared bit for bit, and how many shadow rays the first look
               settled.  Every shadow ray is traced and counted in both; one unsharded GPU only (--no-occluder-ablation skips it).
  shortcuts_off  (round 6) one more untimed repeat with BOTH result-invisible shortcuts off -- the occluder cache and kShade's own-triangle test -- i.e. the figure for a
               scene in which neither finds anything to do; image compared bit for bit.  `value_traced_only` beside `value` counts only the rays that entered a traversal launch.
  f32_transcendentals  (round 6) one more untimed repeat in the opt-in `transcendentals` = 1 mode (device f32 math library instead of the specified f64 evaluation): rate,
               kernel times and the image graded by SURVEY 8(d)'s tolerance against the default mode's.
  batch_depth_curve  (round 6) the same frames traced 16 / 64 / all spp per batch, and one rank's shard at world 8 (1/8 of the tiles) in one batch: Mrays/s against paths in flight.
  self_shadow  the same for kShade's own-triangle test of the shadow rays (the reference offsets a hit point along the geometric normal whatever
               the side, so a ray towards a sun behind that normal is stopped by the triangle it starts on): how many shadow rays kShade
               settles, the rate and the shadow / shade times with the test off, image compared bit for bit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L1_PEAK_GACC = 256 * 2.4        # 256 CUs x one vector-L1 (TCP) tag access per clock x 2.4 GHz max clock = 614.4 G accesses/s
SPP_PER_STEP = 16


def log(*a):
    print(*a, file=sys.stderr, flush=True)


REAL_ASSET_CANDIDATES = ("assets/Sponza.pt", "assets/Sponza.glb", "assets/Sponza.gltf", "assets/Sponza/Sponza.gltf", "assets/Sponza/glTF/Sponza.gltf")


def find_real_asset():
    """The asset the metric names, if a user put it next to the repo (the reference keeps it at assets/Sponza.glb and its
    pt-format-tool writes Sponza.pt beside it: src/pt-format-tool/main.cpp:31-34).  It is absent from the reference mount
    (.MISSING_LARGE_BLOBS), so normally nothing is found and the synthetic stand-in is used."""
    for rel in REAL_ASSET_CANDIDATES:
        for base in (ROOT, os.getcwd()):
            p = os.path.join(base, rel)
            if os.path.isfile(p):
                return p
    return ""


def load_scene(path, scale=1, gpu_builder_device=None, detail="plain"):
    """-> (PtFormat, info).  path: a .pt / .glb / .gltf (data = "real"); else the synthetic atrium, tessellated `scale` x finer
    (detail = "clutter": its harder variant with cloth, displaced spheres, chains, cables and plants)."""
    import rayfinder_amd as rf
    from rayfinder_amd import scenes
    if path:
        if not path.endswith(".pt") and gpu_builder_device is not None:
            rf.set_bake_bvh_builder(gpu_builder_device)
        pt = rf.PtFormat.load(path) if path.endswith(".pt") else rf.PtFormat.from_gltf(path)
        rf.set_bake_bvh_builder(None)
        v = pt.view()
        return pt, dict(name=os.path.basename(path), triangles=int(v.num_triangle_position_attributes), textures=int(v.num_textures), real=True)
    if scale > 1 and gpu_builder_device is not None:
        rf.set_bake_bvh_builder(gpu_builder_device)      # same node bytes as the host builder (bvh_build below), 40x faster at this size
    try:
        return scenes.atrium(scale, detail)
    finally:
        rf.set_bake_bvh_builder(None)


def oracle_scene(pt):
    from oracle import orc
    a = pt.arrays()
    descs, off = [], 0
    for (px, w, h) in a["baseColorTextures"]:
        descs.append((w, h, off)); off += px.size
    texels = np.concatenate([px for (px, _, _) in a["baseColorTextures"]])
    return orc.OracleScene(a["bvhNodes"], a["trianglePositionAttributes"], a["triangleVertexAttributes"], np.array(descs, np.uint32), texels), a


def host_cpus():
    """-> (hardware threads this process may run on, CPU quota of the container in CPUs or None).  The GPU boxes report 256
    hardware threads but run the job under a cgroup CPU quota (cpu.max = 16 CPUs): threads beyond ~2x the quota only add
    scheduling overhead (tools/cpu_scaling.py: 16 threads 42.6, 32 threads 48.9, 256 threads 27.9 Mrays/s)."""
    try:
        hw = len(os.sched_getaffinity(0))
    except AttributeError:
        hw = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    return hw, quota


def cpu_baseline(pt, width, height, bounces, first_frame, spp, seconds_budget, gpu_image):
    """Oracle (kind "port") on a centred crop of EXACTLY the frames the timed region traced (same sample indices, same
    order), all host cores; the crop is also the parity check of the timed frame.  Plus the reference's own CPU
    loop, bvh-visualizer's primary rays (src/bvh-visualizer/main.cpp:60-78), at 1 thread and at all cores."""
    import rayfinder_amd as rf
    from oracle import orc
    sc, a = oracle_scene(pt)
    hw_threads, quota = host_cpus()
    cores = hw_threads if quota is None else max(1, min(hw_threads, int(round(2 * quota))))     # threads used by the all-cores legs
    cam = rf.camera_to_array(rf.fly_camera(width, height))
    sky = rf.aligned_sky_state(rf.make_sky())
    rp = orc.make_render_params(width, height, cam, spp, bounces, 0.25, sky)
    # calibration: 1 thread PINNED to one of the CPUs this process may use (an unpinned thread migrates between the 256 hardware threads of the box under a
    # 16-CPU quota: 0.378 Mrays/s on the driver's box against 0.71 in the profile run), three 32x8 patches of 2 frames each, the MEDIAN rate
    cw, ch = 32, 8
    x0, y0 = (width - cw) // 2, (height - ch) // 2
    affinity = None
    try:
        affinity = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(affinity)[len(affinity) // 2]})
    except (AttributeError, OSError):
        affinity = None
    singles, st = [], None
    try:
        # (one untimed pass first: the first touch of the oracle's scene arrays -- 150 MB of nodes, triangles and texels -- is page faults and cold caches,
        # the 0.35-against-0.73 spread of the earlier single-thread figures)
        orc.render(sc, rp, first_frame, 2, x0, y0, x0 + cw, y0 + ch, accumulated_start=0)
        for _ in range(3):
            t0 = time.perf_counter()
            _, st = orc.render(sc, rp, first_frame, 2, x0, y0, x0 + cw, y0 + ch, accumulated_start=0)
            dt1 = max(time.perf_counter() - t0, 1e-4)
            singles.append((st.closestRays + st.shadowRays) / dt1)
    finally:
        if affinity is not None:
            os.sched_setaffinity(0, affinity)
    single = sorted(singles)[1]
    rays_per_pixel = (st.closestRays + st.shadowRays) / (cw * ch * 2) * spp        # at the full sample count
    # crop sized for the budget (assuming ~50 % parallel efficiency up to 64 threads: the loop is memory-latency bound)
    want = seconds_budget * (cores * 0.5 if quota is None else min(cores, quota) * 0.8) * single
    area = max(want / max(rays_per_pixel, 1.0), 64.0 * 64.0)
    # at least two rows per thread (one-row blocks, dealt statically), at least 64x64 (the parity crop), at most the frame
    ch2 = int(min(height, max(64, 2 * cores, int(area ** 0.5) // 4 * 4)))
    cw2 = int(min(width, max(64, int(area / ch2) // 8 * 8)))
    x0, y0 = (width - cw2) // 2, (height - ch2) // 2
    image = np.zeros((height, width, 4), np.float32)
    # all cores: C-side threads inside the oracle (pthreads; rows dealt in static one-row blocks, block b to thread b % cores --
    # "row-parallel threads, static scanline blocks"); no Python in the timed loop
    t0 = time.time()
    _, st_all, started = orc.render_threads(sc, rp, first_frame, spp, x0, y0, x0 + cw2, y0 + ch2, cores, 1, image=image, accumulated_start=0)
    dt = max(time.time() - t0, 1e-6)
    rays = st_all.closestRays + st_all.shadowRays
    # ... and once more at cores = floor(CPU quota) (one thread per CPU the container is actually given), on a quarter of the crop
    quota_leg = None
    if quota is not None and int(quota) >= 1 and int(quota) != cores:
        qc = int(quota)
        qh = max(min(ch2 // 4, ch2), min(ch2, 2 * qc))
        qy = y0 + (ch2 - qh) // 2
        t0 = time.time()
        _, st_q, started_q = orc.render_threads(sc, rp, first_frame, spp, x0, qy, x0 + cw2, qy + qh, qc, 1, accumulated_start=0)
        dq = max(time.time() - t0, 1e-6)
        quota_leg = dict(value=round((st_q.closestRays + st_q.shadowRays) / dq * 1e-6, 3), cores=int(started_q), crop=[x0, qy, cw2, qh])
    # parity of the timed GPU frame on that crop
    g, c = gpu_image[y0:y0 + ch2, x0:x0 + cw2, :3], image[y0:y0 + ch2, x0:x0 + cw2, :3]
    same_nan = bool(np.array_equal(np.isnan(g), np.isnan(c)))
    identical = int(((g.view(np.uint32) == c.view(np.uint32)).all(axis=-1) | np.isnan(c).any(axis=-1)).sum())
    parity = dict(crop=[x0, y0, cw2, ch2], spp=spp, pixels=cw2 * ch2, bit_identical_pixels=identical, nan_pixels_match=same_nan,
                  verdict="bit-identical" if identical == cw2 * ch2 and same_nan else "DIFFERENT")
    # the reference's CPU path: buildBvh is timed in `bvh_build`; here its primary-ray loop at bvh-visualizer's own size
    vw, vh = 1280, 720
    vcam = rf.camera_to_array(rf.bvh_visualizer_camera(a["bvhNodes"], np.float32(np.float32(vw) / np.float32(vh))))
    tris36 = a["bvhPositionAttributes"]
    t0 = time.time()
    one = orc.bvh_visualize(a["bvhNodes"], tris36, vcam, vw, vh, vh * 3 // 8, vh * 5 // 8)   # the middle quarter of the rows on one thread
    viz_single = vw * (vh * 5 // 8 - vh * 3 // 8) / max(time.time() - t0, 1e-6)
    t0 = time.time()
    orc.bvh_visualize_threads(a["bvhNodes"], tris36, vcam, vw, vh, cores, 1)
    vdt = max(time.time() - t0, 1e-6)
    del one
    base = dict(value=round(rays / dt * 1e-6, 3), unit="Mrays/s", cores=cores, kind="port",
                sample=f"oracle/rf_oracle.c full path tracer, centred {cw2}x{ch2} crop of the {width}x{height} frame, {spp} spp (the timed frames), {bounces} bounces, "
                       f"{rays} rays in {dt:.1f} s on {started} C-side threads (pthreads in the oracle, static one-row blocks; the host has {hw_threads} hardware threads"
                       + (f", the container a CPU quota of {quota:g} CPUs" if quota is not None else "") + f"); 1 thread: {single * 1e-6:.3f} Mrays/s",
                host_hardware_threads=hw_threads, cpu_quota=quota,
                single_thread_value=round(single * 1e-6, 3),
                single_thread=dict(value=round(single * 1e-6, 3), samples=[round(v * 1e-6, 3) for v in singles], pinned=affinity is not None,
                                   spread=dict(min=round(min(singles) * 1e-6, 3), max=round(max(singles) * 1e-6, 3), across_runs_of_round_5="0.7 - 2.8 Mrays/s on the same build: the box's cores are shared "
                                               "with other tenants, so this figure is a floor of what one reference thread does, not a constant; no ratio against it is claimed"),
                                   note="one thread pinned to one allowed CPU (sched_setaffinity), the centred 32x8 patch x 2 frames three times after one untimed pass, the median"),
                at_quota_cores=quota_leg,
                bvh_visualizer_primary_rays=dict(unit="Mrays/s", image=f"{vw}x{vh}", one_thread=round(viz_single * 1e-6, 3),
                                                 all_cores=round(vw * vh / vdt * 1e-6, 3), cores=cores,
                                                 note="the reference's only CPU traversal loop (src/bvh-visualizer/main.cpp:60-78, single-threaded there), restated in oracle/rf_oracle.c"))
    return base, parity


def parity_of_crop(pt, width, height, bounces, first_frame, spp, gpu_image, crop_w=48, crop_h=32):
    """The oracle on a small centred crop of exactly the frames `gpu_image` holds -> dict(crop, spp, pixels, bit_identical_pixels, verdict)."""
    import rayfinder_amd as rf
    from oracle import orc
    sc, _ = oracle_scene(pt)
    hw_threads, quota = host_cpus()
    cores = hw_threads if quota is None else max(1, min(hw_threads, int(round(2 * quota))))
    rp = orc.make_render_params(width, height, rf.camera_to_array(rf.fly_camera(width, height)), spp, bounces, 0.25, rf.aligned_sky_state(rf.make_sky()))
    x0, y0 = (width - crop_w) // 2, (height - crop_h) // 2
    image = np.zeros((height, width, 4), np.float32)
    orc.render_threads(sc, rp, first_frame, spp, x0, y0, x0 + crop_w, y0 + crop_h, cores, 1, image=image, accumulated_start=0)
    g, c = gpu_image[y0:y0 + crop_h, x0:x0 + crop_w, :3], image[y0:y0 + crop_h, x0:x0 + crop_w, :3]
    same_nan = bool(np.array_equal(np.isnan(g), np.isnan(c)))
    identical = int(((g.view(np.uint32) == c.view(np.uint32)).all(axis=-1) | np.isnan(c).any(axis=-1)).sum())
    return dict(crop=[x0, y0, crop_w, crop_h], spp=spp, pixels=crop_w * crop_h, bit_identical_pixels=identical, nan_pixels_match=same_nan,
                verdict="bit-identical" if identical == crop_w * crop_h and same_nan else "DIFFERENT")


def run_regime(name, detail, scale, steps, sps, width, height, bounces, device, with_parity=True, repeats=3):
    """One of the OTHER stand-ins next to the headline: the same timed-region recipe on a renderer of its own -- warm-up of 2 steps,
    `repeats` timed regions of `steps` steps (median), one more with the occluder cache off, the oracle on a small crop of the timed frames."""
    import rayfinder_amd as rf
    t_all = time.time()
    pt, info = load_scene("", scale, device, detail)
    spp, warm = sps * steps, sps * 2
    cam, sky = rf.fly_camera(width, height), rf.make_sky()
    r = rf.ReferencePathTracer(rf.make_render_parameters(width, height, cam, spp, bounces, sky, 0.25), pt.scene(), device_ordinal=device)
    try:
        r.set_option("reserve_samples", spp)
        r.render(warm); r.synchronize()
        r.set_timing(True)
        runs = []
        for rep in range(repeats):
            r.set_render_parameters(rf.make_render_parameters(width, height, cam, spp, bounces, sky, 0.5 + 0.125 * (rep % 2)))
            r.reset_stats()
            r.synchronize()
            t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
            runs.append((dt, r.stats()))
        dt, st = sorted(runs, key=lambda x: x[0])[(repeats - 1) // 2]
        image = r.read_accumulation()[0]
        first_frame = warm + (repeats - 1) * spp
        layouts = r.layout_info(bounces)          # what the renderer picked by itself for this scene (rf_renderer_layout_info)
        r.set_option("occluder_cache_bounces", 0)
        r.set_render_parameters(rf.make_render_parameters(width, height, cam, spp, bounces, sky, 0.375))
        r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt_off = time.perf_counter() - t0
        st_off = r.stats()
        img_off = r.read_accumulation()[0]
    finally:
        r.close()
    rays = st["closest_rays"] + st["shadow_rays"]
    out = dict(workload=f"{info['name']}, {width}x{height}, {bounces} bounces, {sps} spp per step x {steps} steps = {spp} spp", scene_triangles=info.get("triangles"),
               steps=steps, value=round(rays / dt * 1e-6, 1), unit="Mrays/s", repeats=[round((s_["closest_rays"] + s_["shadow_rays"]) / t_ * 1e-6, 1) for (t_, s_) in runs],
               ms_per_step=round(dt / steps * 1e3, 3),
               kernel_ms={k: round(st[k], 3) for k in ("ms_raygen", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")},
               value_with_cache_off=round((st_off["closest_rays"] + st_off["shadow_rays"]) / dt_off * 1e-6, 1),
               shadow_rays_settled_by_kshade_fraction=round(st.get("shadow_rays_self_answered", 0) / max(st["shadow_rays"], 1), 4),
               cache_off_image_bit_identical=bool(np.array_equal(np.asarray(img_off).view(np.uint32), np.asarray(image).view(np.uint32))),
               record_layouts=layouts)
    if with_parity:
        out["parity_crop"] = parity_of_crop(pt, width, height, bounces, first_frame, spp, image)
    out["wall_s"] = round(time.time() - t_all, 1)
    log(f"[bench] regime {name}: {out['value']} Mrays/s (cache off {out['value_with_cache_off']}), parity {out.get('parity_crop', {}).get('verdict')}, {out['wall_s']} s")
    return out


def cold_start(pt, width, height, bounces, device, spp=256):
    """BASELINE.json config 3 as written, on a FRESH renderer: 256 spp in one call, no warm-up steps -- the occluder grid empty, kShadowFirstLook not yet running
    (it starts with the second batch), caches cold.  Path state is allocated before the clock (hipMalloc is not the hot path)."""
    import rayfinder_amd as rf
    cam, sky = rf.fly_camera(width, height), rf.make_sky()
    r = rf.ReferencePathTracer(rf.make_render_parameters(width, height, cam, spp, bounces, sky, 0.25), pt.scene(), device_ordinal=device)
    try:
        r.set_option("reserve_samples", spp)
        r.set_timing(True); r.reset_stats(); r.synchronize()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); dt = time.perf_counter() - t0
        st = r.stats()
    finally:
        r.close()
    return dict(value=round((st["closest_rays"] + st["shadow_rays"]) / dt * 1e-6, 1), spp=spp, timed_region_s=round(dt, 4), ms_shadow=round(st["ms_shadow"], 3),
                ms_closest=round(st["ms_closest"], 3), note="fresh renderer, no warm-up steps, one render call of 256 spp (config 3 as written); allocation outside the clock")


def find_pmc_profile(workload):
    """The committed per-ray counter figures for this workload: profiles/pmc_per_ray*.json whose "workload" matches
    (scene name, frame, bounces -- per ray, so they hold for any --steps / N)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_per_ray*.json"))):
        try:
            pmc = json.load(open(path))
        except Exception as e:  # a malformed file must not take the bench line down
            log(f"[bench] {path} ignored: {e}")
            continue
        if pmc.get("workload") == workload:
            return pmc, os.path.relpath(path, ROOT)
    return None, None


def live_traffic(argv_scene, steps, sps, width, height, bounces, timeout_s=90):
    """`traffic` measured in THIS run: two more passes of the same workload, each as a child process under
    `rocprofv3 --pmc <one counter> --kernel-trace` (FETCH_SIZE, then WRITE_SIZE: separate passes, counters with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes), summed over the closest-hit launches of the child's one timed batch.
    -> dict(fetch_kb, write_kb, launches, closest_rays) or None (rocprofv3 missing, a pass failed or timed out: the committed per-ray profile serves then)."""
    import csv, glob, shutil, subprocess, tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        log("[bench] live counters: rocprofv3 not found")
        return None
    out = {}
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "0", "--repeat", "1", "--spp-per-step", str(sps), "--width", str(width),
             "--height", str(height), "--bounces", str(bounces), "--no-cpu-baseline", "--no-counting", "--no-live-counters", "--no-regimes"] + argv_scene
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rf_pmc_", dir="/tmp")
        try:
            p = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if p.returncode != 0 or not files or not lines:
                log(f"[bench] live counters: the {counter} pass failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-300:]}")
                return None
            line = json.loads(lines[-1])
            total, dispatches = 0.0, set()
            for r in csv.DictReader(open(files[0])):
                if "kTraceWide<false" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                    total += float(r["Counter_Value"]); dispatches.add(r.get("Dispatch_Id"))
            out[counter] = total
            out["launches"] = len(dispatches)
            out["closest_rays"] = line["rays"]["closest"]
        except Exception as e:  # noqa: BLE001  (a profiler that is absent, hangs or changes its output must not take the bench line down)
            log(f"[bench] live counters: the {counter} pass failed: {e}")
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return dict(fetch_kb=out["FETCH_SIZE"], write_kb=out["WRITE_SIZE"], launches=out["launches"], closest_rays=out["closest_rays"])


def build_roofline(s, cs, per_bounce, workload, live=None):
    """The roofline block of the JSON line.

    Top level (the driver's contract): the closest-hit traversal kernel against HBM in MEASURED fabric-side bytes --
    traffic = bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes of this command, calibrated),
    achieved = traffic / the launch duration measured live with HIP events, frac = achieved / 8 TB/s.
    `bound` names the ceiling that actually BINDS this kernel on this workload (the one with the largest fraction among
    HBM bytes, L1->L2 requests, vector-L1 tag accesses and VALU issue), `ceilings` holds all four, `per_bounce` the same per
    launch of the timed batch (live times x profiled per-ray counters), `other_kernels` the shadow traversal and kShade."""
    launches = max(s["launches_closest"], 1)
    avg_ms = s["ms_closest"] / launches
    rays_per_launch = s["closest_rays"] / launches
    roofline = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=None, kernel="kTraceWide<closest>",
                    avg_launch_ms=round(avg_ms, 4), launches=launches, rays_per_launch=int(rays_per_launch),
                    compulsory_hbm_bytes_per_ray=40)   # 12 B origin + 12 B direction in (packed, at the ray's queue position: no queue read), 16 B hit record out
    pmc, pmc_file = find_pmc_profile(workload)
    if pmc is None and live and live.get("closest_rays") and avg_ms > 0:
        # no committed counter profile for this workload (another scene / frame size): the HBM figures come from this run's own counter passes, with the
        # calibration factors of the access pattern (they belong to the pattern, not to the scene)
        fc, wc = 0.9304, 1.0
        try:
            cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_per_ray.json")))
            fc, wc = float(cal.get("fetch_calibration") or fc), float(cal.get("write_calibration") or wc)
        except Exception:  # noqa: BLE001
            pass
        per_ray = (live["fetch_kb"] * 1024.0 * fc + live["write_kb"] * 1024.0 * wc) / live["closest_rays"]
        traffic = per_ray * rays_per_launch
        achieved = traffic / (avg_ms * 1e-3) / 1e9
        roofline.update(traffic=int(traffic), achieved=round(achieved, 1), frac=round(achieved / HBM_PEAK_GBPS, 4),
                        traffic_live=dict(live=True, fetch_size_kb=live["fetch_kb"], write_size_kb=live["write_kb"], launches_profiled=live["launches"],
                                          closest_rays_profiled=live["closest_rays"], fetch_calibration=fc, write_calibration=wc, hbm_side_bytes_per_ray=round(per_ray, 2),
                                          note="two child passes of this workload under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`; no committed per-ray profile for this "
                                               "workload, so the other ceilings (L1 tag, L1->L2 requests, VALU) are not priced"))
    if pmc is not None and avg_ms > 0:
        sec = avg_ms * 1e-3
        per_ray = float(pmc["hbm_side_bytes_per_ray"])
        live_info = dict(live=False, note="committed per-ray counter profile x the live ray count (no live counter pass in this run)")
        if live and live.get("closest_rays"):
            # FETCH_SIZE / WRITE_SIZE of the closest-hit launches measured in THIS run (two child passes under rocprofv3 --pmc), corrected with the committed
            # calibration factors of the access pattern (tools/microbench/fetch_calib: random 64-byte gathers 1 / 1.07, coalesced stores 1.0)
            fc, wc = float(pmc.get("fetch_calibration") or 1.0), float(pmc.get("write_calibration") or 1.0)
            committed = per_ray
            per_ray = (live["fetch_kb"] * 1024.0 * fc + live["write_kb"] * 1024.0 * wc) / live["closest_rays"]
            live_info = dict(live=True, fetch_size_kb=live["fetch_kb"], write_size_kb=live["write_kb"], launches_profiled=live["launches"], closest_rays_profiled=live["closest_rays"],
                             fetch_calibration=fc, write_calibration=wc, hbm_side_bytes_per_ray_committed_profile=committed,
                             note="two child passes of this workload under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (one counter per pass), summed over the closest-hit "
                                  "launches of the child's timed batch, x the calibration factors, / the rays those launches traced")
        traffic = per_ray * rays_per_launch
        achieved = traffic / sec / 1e9
        roofline.update(traffic=int(traffic), achieved=round(achieved, 1), frac=round(achieved / HBM_PEAK_GBPS, 4), traffic_live=live_info,
                        traffic_source=dict(file=pmc_file, profile=pmc.get("profile"), hbm_side_bytes_per_ray=per_ray,
                                            fetch_size_bytes_per_ray=pmc.get("fetch_size_bytes_per_ray"), fetch_calibration=pmc.get("fetch_calibration"),
                                            write_size_bytes_per_ray=pmc.get("write_size_bytes_per_ray"), write_calibration=pmc.get("write_calibration"),
                                            l2_hit_rate=pmc.get("l2_hit_rate")))
        ceil = pmc.get("ceilings", {})
        ceilings = {"hbm": dict(bytes_per_ray=per_ray, GBps=round(achieved, 1), peak_GBps=HBM_PEAK_GBPS, frac=round(achieved / HBM_PEAK_GBPS, 4),
                                random_64B_gather_GBps_measured=ceil.get("hbm_random_64B_gather_GBps"),
                                frac_of_random_gather_rate=round(achieved / ceil["hbm_random_64B_gather_GBps"], 4) if ceil.get("hbm_random_64B_gather_GBps") else None)}
        if pmc.get("l1_to_l2_read_requests_per_ray") and ceil.get("l1_to_l2_requests_G_per_s"):
            g = float(pmc["l1_to_l2_read_requests_per_ray"]) * rays_per_launch / sec / 1e9
            ceilings["l1_l2_requests"] = dict(requests_per_ray=pmc["l1_to_l2_read_requests_per_ray"], G_per_s=round(g, 1), peak_G_per_s=ceil["l1_to_l2_requests_G_per_s"],
                                              frac=round(g / ceil["l1_to_l2_requests_G_per_s"], 4), counter="TCP_TCC_READ_REQ_sum",
                                              peak_source="calibration.json: random 64-byte record gather, cache-resident table (tools/microbench/fetch_calib.hip)")
        if pmc.get("l1_accesses_per_ray"):
            gacc = float(pmc["l1_accesses_per_ray"]) * rays_per_launch / sec / 1e9
            ceilings["l1_tag"] = dict(accesses_per_ray=pmc["l1_accesses_per_ray"], G_per_s=round(gacc, 1), peak_G_per_s=L1_PEAK_GACC, frac=round(gacc / L1_PEAK_GACC, 4),
                                      counter="TCP_TOTAL_CACHE_ACCESSES_sum", peak_source="256 CUs x 1 tag access/clk x 2.4 GHz")
            roofline["l1"] = dict(accesses_per_ray=pmc["l1_accesses_per_ray"], G_accesses_per_s=round(gacc, 1), peak_G_accesses_per_s=L1_PEAK_GACC,
                                  frac=round(gacc / L1_PEAK_GACC, 4), counter="TCP_TOTAL_CACHE_ACCESSES_sum")
        # per launch of the timed batch: live HIP-event time of THIS run x the profiled per-ray counters of that bounce
        rows = []
        prof = {(r["kernel"], r["bounce"]): r for r in pmc.get("per_bounce", [])}
        for b in per_bounce:
            for kind in ("closest", "shadow"):
                r = prof.get((kind, b["bounce"]))
                ms, rays = b[f"ms_{kind}"], b[f"{kind}_rays"]
                if r is None or ms <= 0 or not rays:
                    continue
                row = dict(kernel=kind, bounce=b["bounce"], rays=rays, ms=ms, G_rays_per_s=round(rays / ms * 1e-6, 2))
                fr = {}
                if "l1_to_l2_requests_per_ray" in r and ceil.get("l1_to_l2_requests_G_per_s"):
                    g = r["l1_to_l2_requests_per_ray"] * rays / ms * 1e-6
                    row.update(l1_to_l2_requests_per_ray=r["l1_to_l2_requests_per_ray"], G_requests_per_s=round(g, 1))
                    fr["l1_l2_requests"] = g / ceil["l1_to_l2_requests_G_per_s"]
                if "l1_accesses_per_ray" in r:
                    g = r["l1_accesses_per_ray"] * rays / ms * 1e-6
                    row.update(l1_accesses_per_ray=r["l1_accesses_per_ray"])
                    fr["l1_tag"] = g / L1_PEAK_GACC
                if "valu_issue_share" in r:
                    # instructions per ray are a property of the rays; the share scales with this run's time against the profiled run's
                    share = r["valu_issue_share"] * (r["ms"] / ms) * (rays / max(r["rays"], 1)) if r.get("ms") else r["valu_issue_share"]
                    fr["valu"] = share
                    if r.get("valu_issue_share_exec_adjusted") and r["valu_issue_share"]:
                        # informative: the same share with valu_calib's penalty for partly empty EXEC masks at this launch's active-lane fraction (not a ceiling)
                        row["valu_exec_adjusted"] = round(share * r["valu_issue_share_exec_adjusted"] / r["valu_issue_share"], 3)
                        row["valu_active_lanes_per_instruction"] = r.get("valu_active_lanes_per_instruction")
                if "l2_hit_rate" in r:
                    row["l2_hit_rate"] = r["l2_hit_rate"]
                if "hbm_side_bytes_per_ray" in r:
                    gb = r["hbm_side_bytes_per_ray"] * rays / ms * 1e-6
                    row.update(hbm_side_bytes_per_ray=r["hbm_side_bytes_per_ray"], hbm_side_GBps=round(gb, 1))
                    fr["hbm"] = gb / HBM_PEAK_GBPS
                row["fractions"] = {k: round(v, 3) for k, v in fr.items()}
                # What BINDS (round 6).  The vector-L1 tag rate is a figure these launches run ALONGSIDE, not into: taking a fifth of the tag accesses away (the
                # top three quad levels served from LDS) left the closest-hit launches of bounces 3-8 where they were (+0.8 %, profiles/r05_top), while every cut of the
                # instruction count moved them one for one (profiles/r05_leafrep, r05_pins, r06_lanes).  So `l1_tag` stays in `fractions` but is not a candidate for
                # `binds`; the VALU candidate is the issue share AT THIS LAUNCH'S LANE OCCUPANCY (valu_exec_adjusted: the calibrated issue peak with the measured penalty
                # for partly empty EXEC masks) -- the product the frame pays for is issue slots x the share of lanes that carry a ray.
                cand = {k: v for k, v in fr.items() if k != "l1_tag"}
                if "valu" in cand and row.get("valu_exec_adjusted"):
                    cand["valu"] = row["valu_exec_adjusted"]
                if cand:
                    top = max(cand, key=cand.get)
                    row["binds"] = "valu_issue_x_lanes" if top == "valu" else top
                    row["frac_of_binding_ceiling"] = round(cand[top], 3)
                if "l1_tag" in fr:
                    row["runs_alongside"] = {"l1_tag": round(fr["l1_tag"], 3), "note": "not binding: profiles/r05_top (a fifth of the tag accesses removed: +0.8 %)"}
                rows.append(row)
        if rows:
            roofline["per_bounce"] = rows
            # VALU issue share of the whole kernel: ray-weighted over its launches
            cl = [r for r in rows if r["kernel"] == "closest" and "valu" in r["fractions"]]
            if cl:
                share = sum(r["fractions"]["valu"] * r["ms"] for r in cl) / sum(r["ms"] for r in cl)
                cpi = ceil.get("valu_issue_peak_cycles_per_instruction")
                ceilings["valu"] = dict(issue_share=round(share, 4), peak=1.0, frac=round(share, 4),
                                        counter=f"SQ_INSTS_VALU x {cpi if cpi else 4} cycles / (1024 SIMDs x cycles)",
                                        peak_cycles_per_instruction=cpi if cpi else 4.0,
                                        peak_source=("tools/microbench/valu_calib: the fastest mixed instruction stream measured at 6 waves per SIMD (" + str(ceil.get("valu_issue_peak_source")) + "); "
                                                     "the traversal step's own mix issues at " + str(ceil.get("valu_step_mix_cycles_per_instruction")) + " cycles per instruction; EXEC masks with half of the "
                                                     "lanes off issue " + str(ceil.get("valu_half_empty_exec_speedup")) + " x as fast (no faster)") if cpi else "uncalibrated 4-cycle model")
        roofline["ceilings"] = ceilings
        # (the kernel-level `bound`: the same rule as the per-launch `binds` -- the L1 tag rate is listed, not a candidate; the VALU figure at the kernel's lane occupancy)
        cl_adj = [r for r in rows if r["kernel"] == "closest" and r.get("valu_exec_adjusted")]
        if cl_adj and "valu" in ceilings:
            adj = sum(r["valu_exec_adjusted"] * r["ms"] for r in cl_adj) / sum(r["ms"] for r in cl_adj)
            lanes = [r for r in cl_adj if r.get("valu_active_lanes_per_instruction")]
            ceilings["valu"].update(frac_at_lane_occupancy=round(adj, 4),
                                    active_lanes_per_instruction=round(sum(r["valu_active_lanes_per_instruction"] * r["ms"] for r in lanes) / sum(r["ms"] for r in lanes), 3) if lanes else None,
                                    note="frac: SQ_INSTS_VALU x calibrated cycles / SIMD cycles (full-EXEC issue peak); frac_at_lane_occupancy: the same against what the VALU issues "
                                         "at this kernel's measured share of active lanes (a half-empty EXEC mask issues 13 % slower on gfx950: tools/microbench/valu_calib)")
        binding = max((k for k in ceilings if k != "l1_tag"), key=lambda k: ceilings[k].get("frac_at_lane_occupancy", ceilings[k]["frac"]))
        roofline["bound"] = "valu_issue_x_lanes" if binding == "valu" else binding
        roofline["frac_of_binding_ceiling"] = ceilings[binding].get("frac_at_lane_occupancy", ceilings[binding]["frac"])
        roofline["binding_ceiling_note"] = ("`frac` / achieved / peak / traffic above are the HBM figures the contract asks for (measured fabric bytes / 8 TB/s): this kernel does not run into "
                                            "HBM on a cache-resident scene.  frac_of_binding_ceiling is the roof it does run into: VALU issue at its lane occupancy (38 - 46 % of the issue "
                                            "slots it pays for carry no ray: lanes parked at a leaf while others descend, and the reverse)")
        deep = [r for r in rows if r["kernel"] == "closest" and r["bounce"] >= 3 and "binds" in r]
        first = [r for r in rows if r["kernel"] == "closest" and r["bounce"] == 1 and "binds" in r]
        roofline["bound_by_phase"] = dict(bounce_1=first[0]["binds"] if first else None,
                                          bounces_3_up=max(set(r["binds"] for r in deep), key=[r["binds"] for r in deep].count) if deep else None)
        roofline["bound_note"] = ("`bound` = the ceiling this kernel runs into on this workload (candidates: HBM bytes, L1->L2 requests, VALU issue at the launch's lane occupancy; the "
                                  "vector-L1 tag rate is reported under `runs_alongside` -- the A/B of profiles/r05_top falsified it as a bound); achieved / peak / frac / traffic above stay "
                                  "the HBM-side figures (measured fabric bytes) whatever binds")
        # one-line entries for the two other kernels the frame spends its time in
        other = {}
        if pmc.get("shadow") and s["ms_shadow"] > 0:
            k = pmc["shadow"]
            t = k["hbm_side_bytes_per_unit"] * s["shadow_rays"]
            gb = t / (s["ms_shadow"] * 1e-3) / 1e9
            other["kTraceWide<shadow>"] = dict(bound="l1_l2_requests" if roofline["bound"] != "hbm" else "hbm", unit="GB/s", peak=HBM_PEAK_GBPS, achieved=round(gb, 1), frac=round(gb / HBM_PEAK_GBPS, 4),
                                               traffic=int(t / max(s["launches_shadow"], 1)), hbm_side_bytes_per_ray=k["hbm_side_bytes_per_unit"],
                                               l1_to_l2_requests_per_ray=k.get("l1_to_l2_read_requests_per_unit"),
                                               G_requests_per_s=round(k.get("l1_to_l2_read_requests_per_unit", 0.0) * s["shadow_rays"] / (s["ms_shadow"] * 1e-3) / 1e9, 1),
                                               avg_launch_ms=round(s["ms_shadow"] / max(s["launches_shadow"], 1), 4))
        if pmc.get("shade") and s["ms_shade"] > 0:
            k = pmc["shade"]
            lo, hi = (k[f"hbm_side_bytes_per_unit_{x}"] * s["closest_rays"] / (s["ms_shade"] * 1e-3) / 1e9 for x in ("low", "high"))
            other["kShade+kSky"] = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBPS, achieved=round(lo, 1), frac=round(lo / HBM_PEAK_GBPS, 4),
                                        achieved_if_all_reads_were_streams=round(hi, 1), frac_if_all_reads_were_streams=round(hi / HBM_PEAK_GBPS, 4),
                                        traffic=int(k["hbm_side_bytes_per_unit_low"] * s["closest_rays"] / max(s["launches_shade"], 1)),
                                        hbm_side_bytes_per_queue_entry=[k["hbm_side_bytes_per_unit_low"], k["hbm_side_bytes_per_unit_high"]],
                                        avg_launch_ms=round(s["ms_shade"] / max(s["launches_shade"], 1), 4),
                                        note="the timed span holds kShade and the bounce's kSky launch; FETCH_SIZE counts random gathers at ~1.07 x and coalesced streams at 0.5 x "
                                             "their bytes, kShade mixes both: `achieved` applies the gather factor to every read (lower bracket), the other figure the stream factor")
        if other:
            roofline["other_kernels"] = other
    if cs is not None:
        bytes_closest = 28 * cs["closest_rays"] + 16 * cs["closest_rays"] + 48 * (cs["closest_node_visits"] + cs["closest_triangle_tests"])
        bytes_shadow = 28 * cs["shadow_rays"] + 4 * cs["shadow_rays"] + 48 * (cs["shadow_node_visits"] + cs["shadow_triangle_tests"])
        roofline["algorithmic"] = dict(
            bytes_per_launch=int(bytes_closest / launches), bytes_per_ray=round(bytes_closest / max(cs["closest_rays"], 1), 1),
            GBps=round(bytes_closest / launches / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
            node_visits_per_ray=round(cs["closest_node_visits"] / max(cs["closest_rays"], 1), 2),
            triangle_tests_per_ray=round(cs["closest_triangle_tests"] / max(cs["closest_rays"], 1), 2),
            shadow_kernel_GBps=round(bytes_shadow / max(s["ms_shadow"], 1e-9) / 1e6, 1),
            # what the kernel itself requests: 56 of the 64 B of a wide record (one record = both children of a reference node,
            # leaves are never fetched) + 36 B per triangle + ray I/O (24 B in, 16 B out)
            requested_GBps=round((40 * cs["closest_rays"] + 56 * cs["closest_record_fetches"] + 36 * cs["closest_triangle_tests"]) / max(s["ms_closest"], 1e-9) / 1e6, 1),
            record_fetches_per_ray=round(cs["closest_record_fetches"] / max(cs["closest_rays"], 1), 2),
            note="SURVEY.md 8(d): 28 B ray in + 16 B hit out + 48 B per reference node visit + 48 B per triangle test; a cache rate when the BVH is "
                 "resident in L2 / Infinity Cache, reported for reference and not divided by the HBM peak")
    return roofline


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher.  One rank per GPU under
    torch.distributed.run (standalone rendezvous on 127.0.0.1: the container's hostname may not resolve); rank 0 prints
    the one JSON line, this process passes the ranks' stdout / stderr through and returns their exit status."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    env["RF_BENCH_SELF_LAUNCHED"] = "1"
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed steps of 16 spp each (16 = the 256 spp of BASELINE.json config 3)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed steps of 16 spp each")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scene", default=os.environ.get("RF_SCENE", ""), help="Sponza.pt / Sponza.glb; default: assets/Sponza.{pt,glb} if present, else the synthetic atrium")
    ap.add_argument("--scene-scale", type=int, default=1, help="synthetic atrium tessellated N x finer in both grid directions (N^2 x the triangles): "
                    "8 = 17 M triangles, 2.2 GB of BVH records + triangles -- the out-of-cache regime for the HBM roofline")
    ap.add_argument("--scene-detail", choices=("plain", "clutter"), default="plain", help="clutter: the harder stand-in (draped cloth, displaced spheres, chains, diagonal cables, "
                    "plants of overlapping leaves: 358 k triangles, 78 node visits and 7.5 triangle tests per closest-hit ray against 62 / 3.0); the default and the BENCH series stay on the plain atrium")
    ap.add_argument("--spp-per-step", type=int, default=SPP_PER_STEP, help="samples per pixel in one step (default 16; profiling runs of the big scene use fewer)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (and with them the parity crop)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-counting", action="store_true", help="skip the untimed counting pass (profiling runs): the algorithmic figures are null then")
    ap.add_argument("--launch", action="store_true", help="start the rank(s) under torch.distributed.run even for --gpus 1 (with --gpus N > 1 and no "
                    "launcher environment that happens by itself)")
    ap.add_argument("--exchange-at-world-1", action="store_true", help="one rank, but through everything N > 1 ranks go through: torch.distributed (RCCL) "
                    "process group, the product's RCCL communicator, the frame-end exchange (the rank sends its shard to itself) and the device un-tile")
    ap.add_argument("--no-live-counters", action="store_true", help="skip the two extra passes under `rocprofv3 --pmc` that measure the roofline's `traffic` in this run "
                    "(the committed per-ray profile x the live ray count is used then)")
    ap.add_argument("--no-occluder-ablation", action="store_true", help="skip the untimed repeat with the occluder cache off")
    ap.add_argument("--no-regimes", action="store_true", help="skip the `regimes` block (the clutter and out-of-cache stand-ins, each on a renderer of its own after the headline) "
                    "and the cold-start figure")
    ap.add_argument("--repeat", type=int, default=3, help="timed regions of K steps each; the median one is reported (min / max beside it)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if not ("WORLD_SIZE" in os.environ and "RANK" in os.environ) and (args.gpus > 1 or args.launch):     # no launcher around this process
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import rayfinder_amd as rf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher's world is what exists; a line that says n_gpus = N while M ranks ran would be a wrong scaling point
        raise SystemExit(f"bench.py: rank {rank}: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} "
                         f"(or no launcher: `python bench.py --gpus N` starts its own ranks)")
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < max(world, local_rank + 1):
        log(f"bench.py: rank {rank} of {world}: needs {world} MI355X GPU(s), one per rank, but " + ("0 are visible (no GPU visible: the product has no CPU fallback)" if visible == 0
            else f"only {visible} are visible"))
        if world > 1:
            time.sleep(1.0)     # the launcher ends the other ranks as soon as one exits: give each the time to say which rank it was
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    dist = None
    comm = None
    rccl_ranks = 0
    multi = world > 1 or args.exchange_at_world_1
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                   # (only without a launcher: --exchange-at-world-1 run directly)
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # torch.distributed is the launcher's plumbing (rendezvous, barrier, max over ranks); the data path's one
        # exchange goes through the product's own RCCL communicator, whose id travels over the rendezvous store
        ids = [rf.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            comm = rf.TileComm(ids[0], rank, world, local_rank)      # times out with an error (RF_COMM_TIMEOUT_S) instead of hanging
            rccl_ranks = comm.info()["rccl_ranks"]                   # what RCCL itself says the communicator spans (ncclCommCount)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] rank {rank}: RCCL communicator creation FAILED: {e}")
            comm = None
    exchange = "none" if not multi else "C++ RCCL exchange (rf_renderer_gather_frame: grouped ncclSend/ncclRecv + device un-tile)"

    def all_ranks_ok(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    W, H, K, WU, B = args.width, args.height, max(args.steps, 1), max(args.warmup, 0), args.bounces
    SPS = max(args.spp_per_step, 1)
    spp, warm_spp = SPS * K, SPS * WU
    t0 = time.time()
    scene_path = args.scene or find_real_asset()
    pt, info = load_scene(scene_path, max(args.scene_scale, 1), local_rank, args.scene_detail)
    log(f"[bench] rank {rank}: scene {info} ready in {time.time() - t0:.1f} s")

    cam = rf.fly_camera(W, H)
    sky = rf.make_sky()
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.25), pt.scene(), device_ordinal=local_rank)
    r.set_tile_shard(rank, world)
    accum = None
    if multi:
        # self-test of the exchange before anything is timed.  The C++ path has only ever run at world size 1 on the
        # builder's single-GPU boxes; should it throw here on any rank, every rank falls back -- loudly, and named in the JSON
        # line -- to the round-1 plumbing (torch.distributed.gather of the compact buffers + host un-tile) so that a
        # scaling curve still exists.  (A hang cannot be caught; the layout arithmetic both sides share is tested under gloo.)
        err = "" if comm is not None else "no communicator"
        try:
            if comm is not None:
                r.gather_frame(comm, 0, loopback=(world == 1))      # the first exchange is watched: an error after RF_COMM_TIMEOUT_S instead of a hang
                r.synchronize()
        except Exception as e:  # noqa: BLE001
            err = str(e)
        if not all_ranks_ok(err == ""):
            log(f"[bench] rank {rank}: C++ RCCL exchange FAILED ({err or 'on another rank'}); falling back to torch.distributed.gather")
            exchange = f"FALLBACK torch.distributed.gather + host un-tile (the C++ RCCL exchange failed: {err or 'on another rank'})"
            if comm is not None:
                comm.close()
            comm = None
            rccl_ranks = 0
            from rayfinder_amd.sharding import shard_layout
            _, max_tiles = shard_layout(W, H, rank, world)
            accum = torch.zeros((max_tiles * 1024, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
            r.bind_accumulation_buffer(accum.data_ptr(), accum.numel() * 4)

    def exchange_frame():
        if comm is not None:
            r.gather_frame(comm, 0, loopback=(world == 1))
            return None
        if accum is not None:
            from rayfinder_amd.sharding import gather_device
            r.synchronize()
            return gather_device(accum, rank, world)
        return None

    def barrier():
        r.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # warm-up: W untimed steps (+ one exchange so that RCCL's connections exist); path state for the timed batches is
    # allocated here, whatever W is, so that the timed region never calls hipMalloc
    r.set_option("reserve_samples", spp)
    if warm_spp:
        r.render(warm_spp)
    exchange_frame()
    r.synchronize()
    # Timed region, R times over: EXACTLY K steps, barrier + device synchronize on both sides.  The clock stops when THIS
    # rank's stream is idle (after the exchange, which on rank 0 ends with the un-tile of every rank's shard); the barrier
    # that follows only lines the ranks up again and would add its own latency to every rank's figure, so it sits
    # after the clock.  The reported time is the max over ranks, and of the R repeats the MEDIAN one is reported.
    # Every repeat restarts the accumulation (frameCount keeps counting: repeat i traces frames warm_spp + i spp ..
    # warm_spp + (i + 1) spp - 1, i.e. the sample indices 0..spp-1 rotated by warm_spp -- the same frames every time, so
    # every repeat produces the same image).
    R = max(args.repeat, 1)
    r.set_timing(True)
    runs = []
    exchange_ms = []
    parts = None
    for rep in range(R):
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.5 + 0.125 * (rep % 2)))
        r.reset_stats()
        barrier()
        t0 = time.perf_counter()
        r.render(spp)
        parts = exchange_frame()
        r.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ex_ms = comm.last_exchange_ms() if comm is not None else None      # (outside the clock: HIP events around this rank's sends / receives + un-tile)
        if ex_ms is not None and dist is not None:
            t = torch.tensor([ex_ms], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ex_ms = float(t.item())
        exchange_ms.append(None if ex_ms is None else round(ex_ms, 4))
        runs.append((elapsed, r.stats(), r.bounce_stats()))
    order = sorted(range(R), key=lambda i: runs[i][0])
    median_run = order[(R - 1) // 2]          # the lower median for even R: a repeat that was actually measured
    elapsed, s, bs = runs[median_run]         # bs: queue occupancy and traversal time per bounce of that repeat (this rank)
    last_first_frame = warm_spp + (R - 1) * spp
    per_bounce = [dict(bounce=i + 1, closest_rays=int(bs["closest_rays"][i]), ms_closest=round(float(bs["ms_closest"][i]), 3),
                       shadow_rays=int(bs["shadow_rays"][i]), ms_shadow=round(float(bs["ms_shadow"][i]), 3)) for i in range(len(bs["closest_rays"]))]
    r.set_timing(False)
    tiles_rank0 = len(r.shard_tiles())

    rays_local = s["closest_rays"] + s["shadow_rays"]
    if dist is not None:
        c = torch.tensor([rays_local, s["closest_rays"], s["shadow_rays"], s["primary_rays"], s["abandoned_rays"]], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total, closest_total, shadow_total, paths_total, abandoned_total = (float(x) for x in c.tolist())
    else:
        rays_total, closest_total, shadow_total, paths_total, abandoned_total = (float(rays_local), float(s["closest_rays"]), float(s["shadow_rays"]),
                                                                                 float(s["primary_rays"]), float(s["abandoned_rays"]))

    # the frame of the timed region (read-back outside the timed region for every N)
    image = None
    if rank == 0:
        if comm is not None:
            image = comm.read_frame(r, W, H)
        elif parts is not None:
            from rayfinder_amd.sharding import assemble
            image = assemble(parts, W, H, world)
        else:
            image = r.read_accumulation()[0]

    # counting pass (untimed): node visits / triangle tests of exactly the timed frames on this rank
    cs = None
    if not args.no_counting:
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.25))
        # frameCount is now warm_spp + R spp: this pass traces the next spp frames, the same sample indices
        # (n = frameCount % spp) as every timed repeat
        r.set_counting(True)
        r.reset_stats()
        r.render(spp)
        r.synchronize()
        cs = r.stats()
        r.set_counting(False)
        assert cs["closest_rays"] == s["closest_rays"] and cs["shadow_rays"] == s["shadow_rays"], "counting pass traced different rays"

    # ---- the same K steps once more with the any-hit launches' occluder cache off (untimed; one GPU): what the figure above owes to it, and that
    # the image does not (kTraceWide, kFlagOccluderCache: a shadow ray first visits the leaves that stopped the last rays from its cell of the scene)
    occluder = None
    if rank == 0 and world == 1 and not multi and not args.no_occluder_ablation and not args.no_counting:   # (--no-counting: profiling runs trace the timed launches only)
        r.set_option("occluder_cache_bounces", 0)
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.375))
        # frameCount is warm_spp + R spp (+ spp of the counting pass): the same sample indices again
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); t_off = time.perf_counter() - t0
        s_off = r.stats()
        r.set_timing(False)
        img_off = r.read_accumulation()[0]
        r.set_option("occluder_cache_bounces", 64)
        occluder = {"enabled": True,
                    "value_with_cache_off": round((s_off["closest_rays"] + s_off["shadow_rays"]) / t_off * 1e-6, 1),
                    "ms_shadow": round(s["ms_shadow"], 3), "ms_shadow_with_cache_off": round(s_off["ms_shadow"], 3),
                    "shadow_rays_answered_by_first_look": int(s.get("shadow_rays_hint_answered", 0)),
                    "same_rays": bool(s_off["closest_rays"] == s["closest_rays"] and s_off["shadow_rays"] == s["shadow_rays"]),
                    "image_bit_identical": bool(np.array_equal(np.asarray(img_off).view(np.uint32), np.asarray(image).view(np.uint32))),
                    "note": "one untimed repeat of the same frames with occluder_cache_bounces = 0; every shadow ray is traced and counted in both"}
        log(f"[bench] occluder cache off: {occluder['value_with_cache_off']} Mrays/s, shadow launches {occluder['ms_shadow_with_cache_off']} ms against {occluder['ms_shadow']}, "
            f"image bit-identical: {occluder['image_bit_identical']}")

    # ---- and once more with kShade's own-triangle test of the shadow rays off (untimed; one GPU): the reference offsets every hit point along the GEOMETRIC normal, so a
    # shadow ray towards a sun that stands behind that normal is stopped by the triangle it starts on; kShade settles those itself (exact leaf box + triangle)
    self_shadow = None
    if occluder is not None:
        r.set_option("shadow_self_test", 0)
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.4375))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); t_ns = time.perf_counter() - t0
        s_ns = r.stats()
        r.set_timing(False)
        img_ns = r.read_accumulation()[0]
        r.set_option("shadow_self_test", 1)
        self_shadow = {"enabled": True, "shadow_rays_settled_by_kshade": int(s.get("shadow_rays_self_answered", 0)),
                       "fraction_of_shadow_rays": round(s.get("shadow_rays_self_answered", 0) / max(s["shadow_rays"], 1), 4),
                       "value_with_it_off": round((s_ns["closest_rays"] + s_ns["shadow_rays"]) / t_ns * 1e-6, 1),
                       "ms_shadow": round(s["ms_shadow"], 3), "ms_shadow_with_it_off": round(s_ns["ms_shadow"], 3),
                       "ms_shade": round(s["ms_shade"], 3), "ms_shade_with_it_off": round(s_ns["ms_shade"], 3),
                       "same_rays": bool(s_ns["closest_rays"] == s["closest_rays"] and s_ns["shadow_rays"] == s["shadow_rays"]),
                       "image_bit_identical": bool(np.array_equal(np.asarray(img_ns).view(np.uint32), np.asarray(image).view(np.uint32))),
                       "note": "one untimed repeat of the same frames with shadow_self_test = 0 (occluder cache on); every shadow ray is counted in both"}
        log(f"[bench] own-triangle test off: {self_shadow['value_with_it_off']} Mrays/s, shadow launches {self_shadow['ms_shadow_with_it_off']} ms against {self_shadow['ms_shadow']}, "
            f"kShade settles {self_shadow['fraction_of_shadow_rays']} of the shadow rays, image bit-identical: {self_shadow['image_bit_identical']}")

    # ---- both shortcuts off at once: the occluder cache AND kShade's own-triangle test -- what the line owes to two properties of the scene (98 % of the
    # stand-in's shadow rays are occluded, 62 % by the triangle they start on); the figure for a scene where neither fires
    shortcuts_off = None
    if self_shadow is not None:
        r.set_option("shadow_self_test", 0); r.set_option("occluder_cache_bounces", 0)
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.46875))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); t_b = time.perf_counter() - t0
        s_b = r.stats()
        r.set_timing(False)
        img_b = r.read_accumulation()[0]
        r.set_option("shadow_self_test", 1); r.set_option("occluder_cache_bounces", 64)
        shortcuts_off = {"value_with_cache_and_self_test_off": round((s_b["closest_rays"] + s_b["shadow_rays"]) / t_b * 1e-6, 1),
                         "ms_shadow": round(s_b["ms_shadow"], 3), "ms_shade": round(s_b["ms_shade"], 3), "ms_closest": round(s_b["ms_closest"], 3),
                         "same_rays": bool(s_b["closest_rays"] == s["closest_rays"] and s_b["shadow_rays"] == s["shadow_rays"]),
                         "image_bit_identical": bool(np.array_equal(np.asarray(img_b).view(np.uint32), np.asarray(image).view(np.uint32))),
                         "note": "one untimed repeat of the same frames with occluder_cache_bounces = 0 AND shadow_self_test = 0: every shadow ray walks the tree from the root"}
        log(f"[bench] both shortcuts off: {shortcuts_off['value_with_cache_and_self_test_off']} Mrays/s, image bit-identical: {shortcuts_off['image_bit_identical']}")

    # ---- the opt-in f32 transcendentals mode: same frames, kRaygen / kSky on the device's f32 math library; graded against the default mode's image
    f32_mode = None
    if self_shadow is not None:
        r.set_option("transcendentals", 1)
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.484375))
        r.set_timing(True); r.reset_stats()
        t0 = time.perf_counter(); r.render(spp); r.synchronize(); t_f = time.perf_counter() - t0
        s_f = r.stats()
        r.set_timing(False)
        img_f = r.read_accumulation()[0]
        r.set_option("transcendentals", 0)
        g, c = np.asarray(img_f)[..., :3].astype(np.float64), np.asarray(image)[..., :3].astype(np.float64)
        ok = ~(np.isnan(g).any(-1) | np.isnan(c).any(-1))
        within = (np.abs(g - c) <= 1e-3 * np.abs(c) + 1e-4 * spp).all(-1) & ok
        f32_mode = {"value_with_f32_transcendentals": round((s_f["closest_rays"] + s_f["shadow_rays"]) / t_f * 1e-6, 1),
                    "ms_raygen": [round(s["ms_raygen"], 3), round(s_f["ms_raygen"], 3)], "ms_shade_incl_sky": [round(s["ms_shade"], 3), round(s_f["ms_shade"], 3)],
                    "pixels_within_tolerance": round(float(within.mean()), 6), "image_mean_relative_error": float(abs(g[ok].mean() - c[ok].mean()) / max(abs(c[ok].mean()), 1e-30)),
                    "nan_pixels": [int(np.isnan(c).any(-1).sum()), int(np.isnan(g).any(-1).sum())], "bit_identical_pixels": round(float(((img_f[..., :3] == image[..., :3]).all(-1)).mean()), 4),
                    "default": "off", "note": "one untimed repeat with `transcendentals` = 1; [default mode, f32 mode]; tolerance of SURVEY 8(d): |d| <= 1e-3 |ref| + 1e-4 spp per channel "
                                              "against the DEFAULT mode's image (which the parity crop ties to the oracle bit for bit); the gain is within noise, so the mode stays off"}
        log(f"[bench] f32 transcendentals: {f32_mode['value_with_f32_transcendentals']} Mrays/s, within tolerance {f32_mode['pixels_within_tolerance']}")

    # ---- throughput against batch depth: the same frames in batches of 16 / 64 / all spp, and ONE rank's shard at world 8 (an eighth of the tiles) in one
    # batch -- what the 0.93 efficiency of the shard emulation is made of.  Untimed extras; `value` above is the configured depth.
    depth_curve = None
    if self_shadow is not None:
        depth_curve = []
        expo = 0.4921875

        def timed_frames(per_call, label):
            nonlocal expo
            expo += 1.0 / 1024.0
            r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, expo))
            r.reset_stats(); r.synchronize()
            t0 = time.perf_counter()
            left = spp
            while left > 0:
                n = min(per_call, left); r.render(n); left -= n
            r.synchronize(); dt = time.perf_counter() - t0
            st = r.stats()
            depth_curve.append({"what": label, "samples_per_batch": int(st["batch_samples_used"]), "batches": int(st["batches_traced"]),
                                "paths_per_batch": int(st["batch_samples_used"]) * len(r.shard_tiles()) * 1024, "value": round((st["closest_rays"] + st["shadow_rays"]) / dt * 1e-6, 1),
                                "ms": round(dt * 1e3, 2)})
        for per_call in sorted({min(16, spp), min(64, spp), spp}):
            timed_frames(per_call, f"whole frame, {per_call} spp per batch")
        r.set_tile_shard(0, 8)
        timed_frames(spp, f"rank 0's shard at world 8 ({len(r.shard_tiles())} of the frame's tiles), {spp} spp in one batch")
        timed_frames(spp, f"rank 0's shard at world 8, {spp} spp in one batch (second pass: allocation and grid warm)")
        r.set_tile_shard(0, 1)
        log(f"[bench] batch depth curve: {[(d['paths_per_batch'], d['value']) for d in depth_curve]}")

    # ---- roofline of the dominant kernel (closest-hit traversal) + one-line entries for the shadow traversal and kShade
    live = None
    if rank == 0 and world == 1 and not multi and not args.no_live_counters and "ROCPROFILER_REGISTER_ROOT" not in os.environ and "ROCP_TOOL_LIBRARIES" not in os.environ:
        r.synchronize()
        scene_args = (["--scene", scene_path] if scene_path else []) + ["--scene-scale", str(max(args.scene_scale, 1)), "--scene-detail", args.scene_detail]
        t_live = time.time()
        live = live_traffic(scene_args, K, SPS, W, H, B)
        log(f"[bench] live counter passes: {'ok' if live else 'unavailable'} in {time.time() - t_live:.1f} s")
    roofline = build_roofline(s, cs, per_bounce, f"{info['name']} {W}x{H}x{B}", live)

    cfg_label = {(1920, 1080, 8): "BASELINE.json config 3" if world == 1 else "BASELINE.json config 4", (3840, 2160, 16): "BASELINE.json config 5",
                 (800, 600, 4): "BASELINE.json config 2"}.get((W, H, B), "custom configuration")
    if rank == 0:
        nan_pixels = int(np.isnan(image[..., :3]).any(axis=-1).sum())
        out = {
            "metric": "Mrays/sec at 1920x1080 Sponza, 8 bounces; achieved HBM GB/s on traversal",
            "value": round(rays_total / elapsed * 1e-6, 1),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": WU,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real" if info.get("real") else "synthetic",
            "config": {"workload": f"{info['name']}{'' if scene_path else ' -- the real Sponza.glb is not in the reference mount'}, {W}x{H}, {B} bounces, "
                                   f"{SPS} spp per step x {K} steps = {spp} spp, default rayfinder camera + sky ({cfg_label}; tiled over {world} GPU(s))",
                       "spp_per_step": SPS, "spp": spp,
                       # (the timed region's batches, from rf_stats of its median repeat on rank 0: a render call that does not get the configured depth -- device memory short --
                       # traces shallower batches, same image, shorter launches; 148 B of path state per path in flight)
                       "samples_per_batch": int(s["batch_samples_used"]), "batches": int(s["batches_traced"]), "paths_per_batch": int(s["batch_samples_used"]) * tiles_rank0 * 1024,
                       "path_state_gb": round(int(s["batch_samples_used"]) * tiles_rank0 * 1024 * 148 / 1e9, 1),
                       "scene_triangles": info.get("triangles"), "scene_textures": info.get("textures"), "scene_digest": info.get("digest"),
                       "sharding": f"32x32 tiles along a Z-order curve dealt round-robin (rotated per block) over {world} rank(s), one exchange at frame end: {exchange}" if multi else "none"},
            "timed_region_s": round(elapsed, 4),
            "repeats": {"count": R, "reported": "median", "median_index": median_run,
                        "value": [round(rays_total / t_ * 1e-6, 1) for (t_, _, _) in runs],
                        "min": round(rays_total / max(t_ for (t_, _, _) in runs) * 1e-6, 1),
                        "max": round(rays_total / min(t_ for (t_, _, _) in runs) * 1e-6, 1),
                        "timed_region_s": [round(t_, 4) for (t_, _, _) in runs],
                        "note": "each repeat is the same K steps (same frames, same image) between barriers; value / ms_per_step / kernel times are the median repeat's"},
            "paths_per_s": round(paths_total / elapsed, 1),
            "rays": {"closest": int(closest_total), "shadow": int(shadow_total), "abandoned": int(abandoned_total)},
            "kernel_ms_rank0": {k: round(s[k], 3) for k in ("ms_raygen", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")},
            "exchange": exchange,
            "exchange_ms": ({"per_repeat_max_over_ranks": exchange_ms, "median": sorted(x for x in exchange_ms if x is not None)[(len([x for x in exchange_ms if x is not None]) - 1) // 2]
                             if any(x is not None for x in exchange_ms) else None,
                             "what": "HIP events on each rank's stream around its part of the one frame-end exchange (ncclSend / ncclRecv group + the root's un-tile), "
                                     "max over ranks; inside the timed region"} if multi else None),
            "rccl_ranks": rccl_ranks,      # ncclCommCount of the product's communicator (0: no RCCL exchange in this run -- one GPU, or the fallback)
            "device_memory": dict(r.memory_info(), batch_samples_used=int(s["batch_samples_used"]), batches_traced=int(s["batches_traced"]),
                                  batch_depth_used_paths=int(s["batch_samples_used"]) * tiles_rank0 * 1024,
                                  note="batch_*: the timed region's median repeat on rank 0 (rf_stats); max_paths_per_batch: the depth the handle would use now"),
            "value_traced_only": round((rays_total - float(s.get("shadow_rays_self_answered", 0)) - float(s.get("shadow_rays_hint_answered", 0))) / elapsed * 1e-6, 1) if world == 1 else None,
            "value_traced_only_note": "closest + shadow rays that entered a traversal launch (kTraceWide), i.e. `value` without the shadow rays kShade's own-triangle test and "
                                      "kShadowFirstLook settled with one box + triangle test each (they are shadow rays of the reference all the same: `value` counts them)",
            "nan_pixels": nan_pixels,
            "per_bounce_rank0": per_bounce,
            "roofline": roofline,
        }
        out["self_shadow"] = self_shadow if self_shadow is not None else {"enabled": True, "shadow_rays_settled_by_kshade": int(s.get("shadow_rays_self_answered", 0)),
                                                                          "fraction_of_shadow_rays": round(s.get("shadow_rays_self_answered", 0) / max(s["shadow_rays"], 1), 4),
                                                                          "value_with_it_off": None, "note": "the repeat with the test off was not run (sharded / multi-rank / --no-occluder-ablation / --no-counting)"}
        not_run = "not run: these untimed extras run on one unsharded GPU only (and not with --no-occluder-ablation / --no-counting)"
        out["shortcuts_off"] = shortcuts_off if shortcuts_off is not None else {"value_with_cache_and_self_test_off": None, "note": not_run}
        out["f32_transcendentals"] = f32_mode if f32_mode is not None else {"value_with_f32_transcendentals": None, "default": "off", "note": not_run}
        out["batch_depth_curve"] = depth_curve if depth_curve is not None else [{"what": not_run, "value": None}]
        out["occluder_cache"] = occluder if occluder is not None else {"enabled": True, "value_with_cache_off": None,
                                                                         "note": "the untimed repeat with the cache off runs on one unsharded GPU only (and not with --no-counting)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_crop"] = cpu_baseline(pt, W, H, B, last_first_frame, spp, args.cpu_seconds, image)
            # scene bake beside it: the reference's recursive builder as restated on the host (one
            # thread, what pt-format-tool does) and the GPU builder that emits the same bytes
            tris = pt.arrays()["bvhPositionAttributes"]
            t0 = time.perf_counter(); host_nodes, host_idx, _ = rf.build_bvh(tris); host_ms = (time.perf_counter() - t0) * 1e3
            rf.build_bvh_gpu(tris[:4096])                                       # module load / first-launch cost outside the figure
            gpu_nodes, gpu_idx, _, gpu_ms = rf.build_bvh_gpu(tris)
            out["bvh_build"] = {"triangles": int(len(tris)), "nodes": int(len(host_nodes)), "host_ms_1_thread": round(host_ms, 2),
                                "gpu_ms": round(float(gpu_ms), 3), "node_bytes_identical": bool(host_nodes.tobytes() == gpu_nodes.tobytes()),
                                "triangle_order_identical": bool(np.array_equal(host_idx, gpu_idx))}
        if (world == 1 and not multi and not args.no_regimes and not scene_path and max(args.scene_scale, 1) == 1 and args.scene_detail == "plain"
                and (W, H, B) == (1920, 1080, 8)):      # (the headline workload only: config 3's frame on the plain atrium)
            # (the headline's renderer is done: its 90 GB of path state go back before the others allocate theirs)
            r.close()
            try:
                out["occluder_cache"]["value_cold_start"] = cold_start(pt, W, H, B, local_rank)
                log(f"[bench] cold start: {out['occluder_cache']['value_cold_start']}")
            except Exception as e:  # noqa: BLE001  (an extra: it must not take the headline down)
                out["occluder_cache"]["value_cold_start"] = {"error": str(e)[:300]}
            regimes = {}
            for (name, detail, scale, steps) in (("clutter", "clutter", 1, 20), ("out_of_cache_x8", "plain", 8, 4)):   # (the step counts of their own full lines in profiles/: the whole driver command takes about a minute)
                try:
                    regimes[name] = run_regime(name, detail, scale, steps, SPS, W, H, B, local_rank, with_parity=not args.no_cpu_baseline)
                except Exception as e:  # noqa: BLE001
                    regimes[name] = {"error": str(e)[:300]}
                    log(f"[bench] regime {name} FAILED: {e}")
            regimes["note"] = ("the other stand-ins (DESIGN.md 8.1), each on a renderer of its own after the headline: same recipe (2 warm-up steps, 3 timed regions, the median), "
                               "one more repeat with the occluder cache off, the oracle on a 48x32 crop of the timed frames; `value` above stays on the plain atrium")
            out["regimes"] = regimes
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
