#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]/[3]): Sponza stand-in ("synthetic atrium", the real
assets/Sponza.glb is absent from the reference mount), 1920x1080, 8 bounces, default camera and
sky.  A STEP is one sample per pixel of the whole frame through the full wavefront pipeline
(raygen -> [closest-hit traversal -> shade/NEE -> shadow traversal] x 8 -> accumulate); K timed
steps = K spp.  Inputs (scene, textures, tables) are resident in HBM before the timed region.
`value` = rays traced (closest-hit + shadow traversals, counted on the device) by all ranks per
second.  With N > 1 the frame is tile-sharded (strong scaling: total work fixed) and the per-rank
accumulation buffers are gathered once at frame end with RCCL (inside the timed region).

The JSON line also carries
  roofline     traversal-closest kernel: algorithmic bytes per launch (28 B ray in
               + 16 B hit out + 48 B per node visit + 48 B per triangle test; visits/tests counted
               by the counting build of the same kernel on the same frames) / its average launch
               duration, measured with HIP events on the renderer's stream inside the timed region.
  cpu_baseline the CPU oracle (a port of the reference's algorithm) on a bounded crop of the same
               frame, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_scene(path):
    import rayfinder_amd as rf
    from rayfinder_amd import scenes
    if path:
        pt = rf.PtFormat.load(path) if path.endswith(".pt") else rf.PtFormat.from_gltf(path)
        v = pt.view()
        return pt, dict(name=os.path.basename(path), triangles=int(v.num_triangle_position_attributes), textures=int(v.num_textures))
    return scenes.atrium()


def cpu_baseline(pt, width, height, bounces, seconds_budget=20.0):
    """Oracle (kind "port") on a centred crop of the same frame; all host cores, row strips."""
    import threading

    import rayfinder_amd as rf
    from oracle import orc
    a = pt.arrays()
    descs, off = [], 0
    for (px, w, h) in a["baseColorTextures"]:
        descs.append((w, h, off)); off += px.size
    texels = np.concatenate([px for (px, _, _) in a["baseColorTextures"]])
    sc = orc.OracleScene(a["bvhNodes"], a["trianglePositionAttributes"], a["triangleVertexAttributes"], np.array(descs, np.uint32), texels)
    spp = 2
    rp = orc.make_render_params(width, height, rf.camera_to_array(rf.fly_camera(width, height)), spp, bounces, 0.25,
                                rf.aligned_sky_state(rf.make_sky()))
    cores = os.cpu_count() or 1
    # calibrate on a small crop (1 thread), then size the crop for the budget
    cw, ch = 64, 36
    x0, y0 = (width - cw) // 2, (height - ch) // 2
    t0 = time.time()
    _, st = orc.render(sc, rp, 0, spp, x0, y0, x0 + cw, y0 + ch)       # warm-up / calibration
    dt1 = time.time() - t0
    rays1 = st.closestRays + st.shadowRays
    # single-thread figure (what the reference's bvh-visualizer / CPU code does): ~3 s sample
    reps = int(min(max(1, 3.0 / max(dt1, 1e-3)), 64))
    rp1 = orc.make_render_params(width, height, rf.camera_to_array(rf.fly_camera(width, height)), spp * reps, bounces, 0.25,
                                 rf.aligned_sky_state(rf.make_sky()))
    t0 = time.time()
    _, st = orc.render(sc, rp1, 0, spp * reps, x0, y0, x0 + cw, y0 + ch)
    single = (st.closestRays + st.shadowRays) / (time.time() - t0)
    # rays wanted for the budget (parallel efficiency on this many-core host measured at ~10-15 % of
    # linear: the traversal is memory-latency bound on the CPU too), as crop area first, then spp
    want = seconds_budget * min(cores, 24) * 0.5 * single
    f = min(max(1.0, want / max(rays1, 1)) ** 0.5, min(width / cw, height / ch))
    cw2, ch2 = int(cw * f) // 8 * 8, max(int(ch * f) // 4 * 4, 4)
    per_spp = rays1 / spp * (cw2 * ch2) / (cw * ch)
    spp = int(min(max(2, want / max(per_spp, 1)), 64))
    rp = orc.make_render_params(width, height, rf.camera_to_array(rf.fly_camera(width, height)), spp, bounces, 0.25,
                                rf.aligned_sky_state(rf.make_sky()))
    x0, y0 = (width - cw2) // 2, (height - ch2) // 2
    image = np.zeros((height, width, 4), np.float32)
    # dynamic scheduling: 4-row strips pulled from a shared counter (rows differ a lot in cost)
    strips = list(range(y0, y0 + ch2, 4))
    lock = threading.Lock()
    nxt = [0]
    stats = []

    def work():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(strips):
                return
            _, st = orc.render(sc, rp, 0, spp, x0, strips[i], x0 + cw2, min(strips[i] + 4, y0 + ch2), image=image)
            with lock:
                stats.append(st)

    threads = [threading.Thread(target=work) for _ in range(cores)]
    t0 = time.time()
    [t.start() for t in threads]
    [t.join() for t in threads]
    dt = time.time() - t0
    rays = sum(s.closestRays + s.shadowRays for s in stats)
    return dict(value=round(rays / dt * 1e-6, 3), unit="Mrays/s", cores=cores, kind="port",
                sample=f"oracle/rf_oracle.c full path tracer, centred {cw2}x{ch2} crop of the {width}x{height} frame, {spp} spp, {bounces} bounces, "
                       f"{rays} rays in {dt:.1f} s on {cores} threads (4-row strips, dynamic); 1 thread: {single * 1e-6:.3f} Mrays/s",
                single_thread_value=round(single * 1e-6, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)   # one full batch (32 samples of 1080p = 64 Mi paths)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scene", default=os.environ.get("RF_SCENE", ""), help="Sponza.pt / Sponza.glb; default: synthetic atrium")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-counting", action="store_true", help="skip the untimed counting pass (profiling runs): roofline.achieved is null then")
    args = ap.parse_args()

    import torch
    import rayfinder_amd as rf
    from rayfinder_amd.sharding import assemble, gather_device, shard_layout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H, K, WU, B = args.width, args.height, args.steps, args.warmup, args.bounces
    t0 = time.time()
    pt, info = load_scene(args.scene)
    log(f"[bench] rank {rank}: scene {info} ready in {time.time() - t0:.1f} s")

    cam = rf.fly_camera(W, H)
    sky = rf.make_sky()
    spp = max(K, 1)
    params = rf.make_render_parameters(W, H, cam, spp, B, sky, 0.25)
    r = rf.ReferencePathTracer(params, pt.scene(), device_ordinal=local_rank)
    r.set_tile_shard(rank, world)
    tiles, max_tiles = shard_layout(W, H, rank, world)
    accum = torch.zeros((max_tiles * 1024, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    r.bind_accumulation_buffer(accum.data_ptr(), accum.numel() * 4)

    def barrier():
        r.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # warm-up: W untimed steps (+ one gather so RCCL is connected); path state for the timed batches is
    # allocated here, whatever W is, so that the timed region never calls hipMalloc
    r.set_option("reserve_samples", K)
    r.render(WU)
    r.synchronize()
    if dist is not None:
        gather_device(accum, rank, world)
    # restart the accumulation (frameCount keeps counting: sample indices are a rotation of 0..K-1)
    r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.5))
    r.set_timing(True)
    r.reset_stats()

    barrier()
    t0 = time.perf_counter()
    r.render(K)                       # EXACTLY K steps
    r.synchronize()
    parts = gather_device(accum, rank, world)   # frame-end RCCL gather (device to device; no-op at N=1)
    barrier()
    elapsed = time.perf_counter() - t0
    s = r.stats()
    bs = r.bounce_stats()   # queue occupancy and traversal time per bounce of the timed region (this rank)
    per_bounce = [dict(bounce=i + 1, closest_rays=int(bs["closest_rays"][i]), ms_closest=round(float(bs["ms_closest"][i]), 3),
                       shadow_rays=int(bs["shadow_rays"][i]), ms_shadow=round(float(bs["ms_shadow"][i]), 3)) for i in range(len(bs["closest_rays"]))]
    r.set_timing(False)

    rays_local = s["closest_rays"] + s["shadow_rays"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rays_local, s["closest_rays"], s["shadow_rays"], s["primary_rays"]], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total, closest_total, shadow_total, paths_total = (float(x) for x in c.tolist())
    else:
        rays_total, closest_total, shadow_total, paths_total = float(rays_local), float(s["closest_rays"]), float(s["shadow_rays"]), float(s["primary_rays"])

    # counting pass (untimed): node visits / triangle tests of exactly the timed frames on this rank
    cs = None
    if not args.no_counting:
        r.set_render_parameters(rf.make_render_parameters(W, H, cam, spp, B, sky, 0.25))
        # frameCount is now WU + K; the counting pass must see the same sample indices as the timed one
        # (n = frameCount % spp): the timed pass used frames WU..WU+K-1, this one WU+K..WU+2K-1 == same set mod K
        r.set_counting(True)
        r.reset_stats()
        r.render(K)
        r.synchronize()
        cs = r.stats()
        r.set_counting(False)
        assert cs["closest_rays"] == s["closest_rays"] and cs["shadow_rays"] == s["shadow_rays"], "counting pass traced different rays"

    # roofline of the dominant kernel (closest-hit traversal), SURVEY.md 8(d) bytes
    launches = max(s["launches_closest"], 1)
    avg_ms = s["ms_closest"] / launches
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            # HBM-side bytes per launch of this kernel from the committed rocprofv3 PMC passes of the SAME command
            # (same frame, bounces, steps per launch); launches differ in size, the figure is their average
            if pmc.get("kernel") == "kTraceWide<closest>" and pmc.get("workload") == f"{W}x{H}x{B}" and pmc.get("launches_per_128_steps") == round(launches * 128 / K):
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if cs is not None:
        bytes_closest = 28 * cs["closest_rays"] + 16 * cs["closest_rays"] + 48 * (cs["closest_node_visits"] + cs["closest_triangle_tests"])
        bytes_shadow = 28 * cs["shadow_rays"] + 4 * cs["shadow_rays"] + 48 * (cs["shadow_node_visits"] + cs["shadow_triangle_tests"])
        achieved = bytes_closest / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBPS, 4),
                        traffic=traffic, kernel="kTraceWide<closest>", avg_launch_ms=round(avg_ms, 4), launches=launches,
                        algorithmic_bytes_per_launch=int(bytes_closest / launches),
                        rays_per_launch=int(cs["closest_rays"] / launches),
                        node_visits_per_ray=round(cs["closest_node_visits"] / max(cs["closest_rays"], 1), 2),
                        triangle_tests_per_ray=round(cs["closest_triangle_tests"] / max(cs["closest_rays"], 1), 2),
                        shadow_kernel_GBps=round(bytes_shadow / max(s["ms_shadow"], 1e-9) / 1e6, 1),
                        # what the kernel itself requests: 64 B per wide record + 48 B per triangle + ray I/O (the wide layout
                        # needs one record per two reference node visits, so this is below the algorithmic figure)
                        requested_GBps=round((44 * cs["closest_rays"] + 64 * cs["closest_record_fetches"] + 48 * cs["closest_triangle_tests"])
                                             / max(s["ms_closest"], 1e-9) / 1e6, 1),
                        record_fetches_per_ray=round(cs["closest_record_fetches"] / max(cs["closest_rays"], 1), 2),
                        # the schema's "bound" is hbm|mfma; what actually limits this kernel (profiles/r01_final):
                        limiter="L1->VGPR return path (TD busy 86-97 %) and VALU issue (62 %); the BVH is resident in L2 / Infinity Cache, "
                                "so HBM-side traffic is ~1/8 of the algorithmic bytes and frac > 1")
    else:
        roofline = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=traffic, kernel="kTraceWide<closest>",
                        avg_launch_ms=round(avg_ms, 4), launches=launches)
    cfg_label = {(1920, 1080, 8): "BASELINE.json config 3" if world == 1 else "BASELINE.json config 4", (3840, 2160, 16): "BASELINE.json config 5",
                 (800, 600, 4): "BASELINE.json config 2"}.get((W, H, B), "custom configuration")
    if rank == 0:
        image = assemble(parts, W, H, world)      # read-back + un-tile, outside the timed region for every N
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
            "data": "synthetic",
            "config": {"workload": f"{info['name']}, {W}x{H}, {K} spp, {B} bounces, default rayfinder camera + sky ({cfg_label}; tiled over {world} GPU(s))",
                       "scene_triangles": info.get("triangles"), "scene_textures": info.get("textures"), "scene_digest": info.get("digest"),
                       "sharding": f"32x32 tiles, scrambled round-robin over {world} rank(s), one RCCL gather at frame end" if world > 1 else "none"},
            "paths_per_s": round(paths_total / elapsed, 1),
            "rays": {"closest": int(closest_total), "shadow": int(shadow_total)},
            "kernel_ms_rank0": {k: round(s[k], 3) for k in ("ms_raygen", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")},
            "nan_pixels": nan_pixels,
            "per_bounce_rank0": per_bounce,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pt, W, H, B, args.cpu_seconds)
            # scene bake beside it: the reference's recursive builder as restated on the host (one
            # thread, what pt-format-tool does) and the GPU builder that emits the same node bytes
            tris = pt.arrays()["bvhPositionAttributes"]
            t0 = time.perf_counter(); host_nodes, _, _ = rf.build_bvh(tris); host_ms = (time.perf_counter() - t0) * 1e3
            rf.build_bvh_gpu(tris[:4096])                                       # module load / first-launch cost outside the figure
            gpu_nodes, _, _, gpu_ms = rf.build_bvh_gpu(tris)
            out["bvh_build"] = {"triangles": int(len(tris)), "nodes": int(len(host_nodes)), "host_ms_1_thread": round(host_ms, 2),
                                "gpu_ms": round(float(gpu_ms), 3), "node_bytes_identical": bool(host_nodes.tobytes() == gpu_nodes.tobytes())}
        print(json.dumps(out), flush=True)
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
