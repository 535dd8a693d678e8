"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties (shard-union == whole, batching invariance, idempotence,
ray-count conservation).

Bars:  traversal outputs (hit, t, triangle, u, v, p, nodesVisited, triangle tests) BIT-EXACT.
       Radiance sums: per channel |gpu - oracle| <= 1e-3*|oracle| + 1e-4*spp on >= 99.5 % of
       pixels; in practice the f64-evaluated transcendental policy makes the
       images bit-identical, which the tests also report/assert at >= 99.9 %.
"""
import os

import numpy as np
import pytest

import rayfinder_amd as rf
from conftest import DUCK, GOLDEN, DuckOracle, bits, oracle_scene_from_pt
from oracle import orc

pytestmark = pytest.mark.gpu


def _renderer(pt, w, h, spp, bounces, cam=None, sky=None, exposure=0.25, **kw):
    cam = cam if cam is not None else rf.fly_camera(w, h)
    params = rf.make_render_parameters(w, h, cam, spp, bounces, sky if sky is not None else rf.make_sky(), exposure)
    return rf.ReferencePathTracer(params, pt.scene(), **kw), params


def _compare(gpu_sum, ref_sum, spp, min_ok=0.995, min_exact=0.999):
    g, c = gpu_sum[..., :3], ref_sum[..., :3]
    nan = np.isnan(g) | np.isnan(c)
    assert np.array_equal(np.isnan(g), np.isnan(c)), "NaN pixels differ"
    diff = np.abs(np.where(nan, 0, g - c))
    tol = 1e-3 * np.abs(np.where(nan, 0, c)) + 1e-4 * spp
    ok = float((diff <= tol).mean())
    exact = float(((g == c) | nan).mean())
    assert ok >= min_ok, (ok, exact, float(diff.max()))
    assert exact >= min_exact, (ok, exact, float(diff.max()))
    return ok, exact


# ------------------------------------------------------------------ config 1: node-visit parity
def test_primary_node_visits_bit_exact_vs_golden_256(duck_pt):
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    r, _ = _renderer(duck_pt, 256, 256, 1, 1)
    nodes = duck_pt.arrays()["bvhNodes"]
    out = r.trace_primary_stats(rf.bvh_visualizer_camera(nodes, 1.0), 256, 256)
    assert np.array_equal(out["nodesVisited"], g["viz256_nodes_visited"].astype(np.uint32))
    assert np.array_equal(np.packbits(out["hit"]), g["viz256_hit"])
    assert int(out["nodesVisited"].sum()) == 1209382 and int(out["hit"].sum()) == 19462
    assert int(out["triTests"].sum()) == int(g["viz256_tri_tests"]) == 69097


def test_primary_node_visits_bit_exact_1280x720(duck_pt, duck_oracle):
    r, _ = _renderer(duck_pt, 64, 64, 1, 1)
    aspect = np.float32(np.float32(1280) / np.float32(720))
    cam = rf.bvh_visualizer_camera(duck_oracle.nodes, aspect)
    out = r.trace_primary_stats(cam, 1280, 720)
    cpu = orc.bvh_visualize(duck_oracle.nodes, duck_oracle.tris36, rf.camera_to_array(cam), 1280, 720)
    assert np.array_equal(out["nodesVisited"], cpu["nodesVisited"])
    assert np.array_equal(out["hit"], cpu["hit"]) and np.array_equal(bits(out["t"]), bits(cpu["t"]))
    assert np.array_equal(out["triTests"], cpu["triTests"])
    assert int(out["nodesVisited"].sum()) == 9979946


def test_reference_bvh_test_grid_on_gpu(duck_pt):
    """src/tests/bvh.cpp:76-101 rays: GPU hit/t/triangle == oracle == brute force."""
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    r, _ = _renderer(duck_pt, 64, 64, 1, 1)
    out = r.intersect_rays(g["grid_rays"], 1000.0)
    assert np.array_equal(out["hit"], g["grid_hit"]) and int(out["hit"].sum()) == 1216
    assert np.array_equal(bits(out["t"]), bits(g["grid_t"]))
    assert np.array_equal(out["tri"][out["hit"] == 1], g["grid_tri"][g["grid_hit"] == 1])


def _random_rays(rng, n, lo, hi):
    o = rng.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), (n, 3))
    target = rng.uniform(lo, hi, (n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    # edge cases the reference tests exercise: axis-parallel directions (inf invDir, 0*inf NaN
    # slabs), non-unit directions, rays starting inside the model, zero components
    k = n // 10
    rays[:k, 3:] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1)).astype(np.float32)
    rays[k:2 * k, 3:] *= rng.uniform(0.1, 5.0, (k, 1)).astype(np.float32)
    rays[2 * k:3 * k, :3] = rng.uniform(lo, hi, (k, 3)).astype(np.float32)
    rays[3 * k:4 * k, 4] = 0.0
    return rays


@pytest.mark.parametrize("tmax", [10000.0, 1.5, float(np.finfo(np.float32).max)])
def test_closest_hit_random_rays_bit_exact(duck_pt, duck_oracle, tmax):
    rng = np.random.default_rng(42)
    lo, hi = duck_oracle.nodes[0]["min"].astype(np.float64), duck_oracle.nodes[0]["max"].astype(np.float64)
    rays = _random_rays(rng, 20000, lo, hi)
    r, _ = _renderer(duck_pt, 64, 64, 1, 1)
    with np.errstate(all="ignore"):
        cpu = orc.intersect_bvh_batch(duck_oracle.nodes, duck_oracle.pos48, rays, tmax)
    gpu = r.intersect_rays(rays, tmax)
    assert np.array_equal(gpu["hit"], cpu["hit"])
    assert np.array_equal(gpu["tri"], cpu["tri"])
    for k in ("t", "uv", "p"):
        assert np.array_equal(bits(gpu[k]), bits(cpu[k])), k
    assert np.array_equal(gpu["nodesVisited"], cpu["nodesVisited"])
    assert np.array_equal(gpu["triTests"], cpu["triTests"])
    assert 0.05 < cpu["hit"].mean() < 0.95


def test_any_hit_random_rays_bit_exact(duck_pt, duck_oracle):
    rng = np.random.default_rng(43)
    lo, hi = duck_oracle.nodes[0]["min"].astype(np.float64), duck_oracle.nodes[0]["max"].astype(np.float64)
    rays = _random_rays(rng, 20000, lo, hi)
    r, _ = _renderer(duck_pt, 64, 64, 1, 1)
    with np.errstate(all="ignore"):
        cpu = orc.shadow_batch(duck_oracle.nodes, duck_oracle.pos48, rays, 10000.0)
        closest = orc.intersect_bvh_batch(duck_oracle.nodes, duck_oracle.pos48, rays, 10000.0)
    gpu = r.occluded_rays(rays, 10000.0)
    assert np.array_equal(gpu, cpu)
    assert np.array_equal(gpu == 0.0, closest["hit"] == 1)   # any-hit agrees with closest-hit on occlusion
    assert r.intersect_rays(np.zeros((0, 6), np.float32), 1.0)["tri"].size == 0   # empty input


@pytest.mark.parametrize("tmax", [10000.0, 1.5])
def test_render_path_traversal_kernel_on_arbitrary_rays(duck_pt, duck_oracle, tmax):
    """The persistent wide-layout kernel the render path uses (packed slab pairs for regular rays,
    reference-ordered scalar traversal for axis-parallel / zero-component ones), driven with
    arbitrary rays through the query API: hit, t, barycentrics and offset hit point bit-exact."""
    rng = np.random.default_rng(44)
    lo, hi = duck_oracle.nodes[0]["min"].astype(np.float64), duck_oracle.nodes[0]["max"].astype(np.float64)
    rays = _random_rays(rng, 30000, lo, hi)
    rays[-50:, 3:] = 0.0                                  # degenerate directions
    rays[-100:-50, 3] = np.float32(1e-42)                 # denormal component: 1/d = inf
    rays[-150:-100, 0] = np.float32(np.nan)               # NaN origin
    rays[-200:-150, 3:] *= np.float32(1e30)               # huge directions
    # 0 * inf slabs: a zero direction component with the origin EXACTLY on a plane of a node's box
    nodes = duck_oracle.nodes
    for i in range(2000):
        nd = nodes[rng.integers(0, len(nodes))]
        ax = int(rng.integers(0, 3))
        rays[1000 + i, ax] = nd["min" if i % 2 else "max"][ax]
        rays[1000 + i, 3 + ax] = np.float32(0.0) if i % 4 < 2 else np.float32(-0.0)
    # grazing rays: the origin ON a face of a node's box (exactly, or one ulp to either side), the direction almost inside that face --
    # where a conservative interior test and the exact leaf test (half-precision quad records) have to agree with the reference
    for i in range(4000):
        nd = nodes[rng.integers(0, len(nodes))]
        ax = int(rng.integers(0, 3))
        o = rng.uniform(nd["min"].astype(np.float64), nd["max"].astype(np.float64)).astype(np.float32)
        o[ax] = nd["min" if i % 2 else "max"][ax]
        if i % 3:
            o[ax] = np.nextafter(o[ax], np.float32(np.inf if i % 3 == 1 else -np.inf))
        d = rng.normal(size=3).astype(np.float32)
        d[ax] = np.float32(rng.choice([1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3, 1e-12, -1e-12]))
        rays[4000 + i, :3] = o
        rays[4000 + i, 3:] = d
    # far DIAGONAL origins: |o| from 1e3 to 1e20 on all three axes, aimed at the scene -- beyond the conservative records' origin bound, where fma(+-65504, 1/d, -o/d)
    # of a half-precision record's EMPTY slot would collapse to near == far if such a ray ever reached a step (rf_wide.hpp, kHalfEmptyPlanes): the refill's gate must send
    # every one of them to the scalar traversal, whatever layout the launch reads
    centre = 0.5 * (lo + hi)
    for i in range(600):
        sgn = np.array([1.0 if (i >> k) & 1 else -1.0 for k in range(3)])
        far = sgn * (10.0 ** (3 + (i % 18))) * rng.uniform(0.8, 1.25, 3)
        target = rng.uniform(lo, hi)
        rays[9000 + i, :3] = far.astype(np.float32)
        d = target - far
        rays[9000 + i, 3:] = (d / np.linalg.norm(d) if i % 2 else d * 1e-12).astype(np.float32)   # (unit, or tiny: 1/d around 1e-8 ... 1e9)
    r, _ = _renderer(duck_pt, 64, 64, 1, 1)
    r.set_option("query_variant", 2)
    with np.errstate(all="ignore"):
        cpu = orc.intersect_bvh_batch(duck_oracle.nodes, duck_oracle.pos48, rays, tmax)
        cpu_vis = orc.shadow_batch(duck_oracle.nodes, duck_oracle.pos48, rays, tmax)
    gpu = r.intersect_rays(rays, tmax)
    assert np.array_equal(gpu["hit"], cpu["hit"])
    assert np.array_equal(gpu["tri"], cpu["tri"])
    for k in ("t", "uv", "p"):
        assert np.array_equal(bits(gpu[k]), bits(cpu[k])), k
    for nearest_first in (1, 0):
        r.set_option("shadow_nearest_first", nearest_first)
        assert np.array_equal(r.occluded_rays(rays, tmax), cpu_vis), nearest_first
    # the compact-capable records (1: three vector loads per descending step, the node's own x planes carried from the parent's
    # step; 2: 32-byte records, two loads, all six planes carried): the same planes and products, so the same bits on the same
    # hostile rays
    r.set_option("shadow_nearest_first", 1)
    r.set_option("dense_leaf_min", 2)                     # Duck's leaves hold up to 4 triangles: with the threshold at 2 the leaf phases of modes 3 / 4 / 5 run over dense pairs (round 5)
    for mode in (1, 2, 3, 4, 5, 6):                       # (6: the 128-byte oct records, THREE levels per fetch, closest-hit only; 5: the local-grid quad records; 3: the 128-byte quad records, two levels of the tree per fetch; 4: their
                                                          # half-precision form -- conservative interior tests, exact boxes at the leaves)
        r.set_option("query_compact", mode)
        gpu_c = r.intersect_rays(rays, tmax)
        assert np.array_equal(gpu_c["hit"], cpu["hit"]) and np.array_equal(gpu_c["tri"], cpu["tri"]), mode
        for k in ("t", "uv", "p"):
            assert np.array_equal(bits(gpu_c[k]), bits(cpu[k])), ("compact", mode, k)
        assert np.array_equal(r.occluded_rays(rays, tmax), cpu_vis), ("compact", mode)
    r.set_option("query_compact", 0)
    r.set_option("query_variant", 0)
    assert np.array_equal(r.intersect_rays(rays, tmax)["tri"], cpu["tri"])


# ------------------------------------------------------------------ config 2: radiance parity
def test_duck_render_matches_golden_crops(duck_pt):
    g = np.load(os.path.join(GOLDEN, "duck_render_golden.npz"))
    W, H, spp, bounces = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["bounces"])
    r, _ = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp and r.render_progress_percentage() == 100.0
    for (x0, y0), want in zip(g["crops"], g["sums"]):
        _compare(img[y0:y0 + 16, x0:x0 + 16], want, spp)
    assert (img[..., 3] == 0).all()       # 16-byte stride, w unused (array<vec3f>)


def test_duck_render_vs_oracle_full_frame(duck_pt, duck_oracle):
    W, H, spp, bounces = 200, 150, 16, 4
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    ref, st = orc.render(duck_oracle.scene, rp, 0, spp)
    ok, exact = _compare(img, ref, spp)
    s = r.stats()
    assert s["closest_rays"] == st.closestRays and s["shadow_rays"] == st.shadowRays
    assert s["abandoned_rays"] == 0
    # tonemapped swap-chain image (wgsl:59-63): both sides evaluate pow(aces, 1/2.2) in f64, round once to f32 and
    # quantise with floor(x * 255 + 0.5) in f32 from bit-identical sums, so the 8-bit texels are EQUAL
    bgra = r.read_tonemapped()
    srgb = orc.tonemap(ref, spp, 0.25).reshape(H, W, 3)
    want = np.floor(srgb * np.float32(255.0) + np.float32(0.5)).astype(np.int64)
    got = np.stack([(bgra >> 16) & 255, (bgra >> 8) & 255, bgra & 255], axis=-1).astype(np.int64)
    assert exact == 1.0
    assert int((got != want).sum()) == 0 and ((bgra >> 24) == 255).all()


@pytest.mark.parametrize("sky_kw", [dict(), dict(turbidity=9.5, albedo=(0.9, 0.1, 0.4), sun_zenith_degrees=82.0, sun_azimuth_degrees=200.0),
                                    dict(turbidity=1.2, albedo=(0.0, 0.0, 0.0), sun_zenith_degrees=3.0, sun_azimuth_degrees=10.0)])
def test_sky_dome_bit_exact_over_many_directions(sky_kw):
    """Nearly every path of this frame ends in the sky at bounce 1 or 2: half a million evaluations of the sky model per
    channel (acos / cos / exp in f64 rounded once; the Mie term's pow(x, 1.5) through its x * sqrt(x) route with the
    rounding-boundary guard, rf_device.hpp wPow15) against the oracle's libm calls, bit for bit."""
    tris = np.array([[[-0.3, -1.0, -0.3], [0.3, -1.0, -0.3], [0.0, -1.0, 0.3]], [[-3.0, -1.5, -3.0], [3.0, -1.5, -3.0], [0.0, -1.5, 3.0]]], np.float32)
    n = len(tris)
    normals = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (n, 3, 1))
    uvs = np.zeros((n, 3, 2), np.float32)
    pt = rf.PtFormat.from_triangles(tris.reshape(n, 9), normals.reshape(n, 9), uvs.reshape(n, 6), np.zeros(n, np.uint32), [(np.array([0xFFC0C0C0], np.uint32), 1, 1)])
    W, H, spp, bounces = 512, 384, 3, 2
    cam = rf.create_camera((0.0, 0.2, 0.0), (0.3, 1.0, 0.2), 0.0, 1.0, float(orc.degrees_to_radians(110.0)), W / H)
    sky = rf.make_sky(**sky_kw)
    r, _ = _renderer(pt, W, H, spp, bounces, cam=cam, sky=sky, exposure=0.5)
    r.render(spp)
    img, acc = r.read_accumulation()
    s = r.stats()
    r.close()
    assert acc == spp
    sc, _ = oracle_scene_from_pt(pt)
    ref, st = orc.render(sc, orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.5, rf.aligned_sky_state(sky)), 0, spp)
    assert s["closest_rays"] == st.closestRays
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
    assert float((ref[..., :3] > 0).mean()) > 0.9                      # the frame really is sky


def test_lens_sampling_and_other_sky_parameters(duck_pt, duck_oracle):
    """Thin-lens camera (aperture > 0), low sun, turbid sky, odd frame size (edge tiles)."""
    W, H, spp, bounces = 150, 90, 8, 3
    cam = rf.fly_camera(W, H, aperture=0.2, focus_distance=2.0, vfov_degrees=50.0)
    sky = rf.make_sky(4.5, (0.3, 0.5, 0.2), 75.0, 200.0)
    r, params = _renderer(duck_pt, W, H, spp, bounces, cam=cam, sky=sky)
    r.render(spp)
    img, _ = r.read_accumulation()
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(sky))
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    _compare(img, ref, spp)


def test_counting_build_matches_oracle_counts(duck_pt, duck_oracle):
    W, H, spp, bounces = 128, 96, 4, 4
    rp = None
    images = {}
    for variant, nearest_first in ((2, 0), (0, 0), (2, 1)):
        r, params = _renderer(duck_pt, W, H, spp, bounces)
        r.set_option("traversal_variant", variant)          # 2: 64-B wide nodes (production), 0: 32-B nodes
        r.set_option("shadow_nearest_first", nearest_first)  # 0: the reference's child order for shadow rays
        r.set_counting(True)
        r.render(spp)
        s = r.stats()
        images[(variant, nearest_first)] = r.read_accumulation()[0]
        if rp is None:
            rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
            _, st = orc.render(duck_oracle.scene, rp, 0, spp)
        # closest-hit traversal always follows the reference's visit order: counts are exact
        assert s["closest_node_visits"] == st.closestNodeVisits and s["closest_triangle_tests"] == st.closestTriTests
        assert s["stack_high_water"] == st.stackHigh
        assert s["primary_rays"] == W * H * spp and s["shadow_rays"] == st.shadowRays
        if not nearest_first:
            assert s["shadow_node_visits"] == st.shadowNodeVisits and s["shadow_triangle_tests"] == st.shadowTriTests
        else:
            # production default: shadow rays visit the nearer child first.  The visibility bit is order
            # independent (identical image below); this build counts box tests, and needs fewer of them.
            assert 0 < s["shadow_node_visits"] < 1.2 * st.shadowNodeVisits
        r.close()
    assert np.array_equal(bits(images[(2, 0)]), bits(images[(0, 0)]))
    assert np.array_equal(bits(images[(2, 0)]), bits(images[(2, 1)]))


# ------------------------------------------------------------------ renderer state machine
def test_frame_and_sample_bookkeeping(duck_pt, duck_oracle):
    """reference_path_tracer.cpp:556-595: one sample per render() until spp, frameCount never
    resets, setRenderParameters resets accumulation only on a change."""
    W, H, spp, bounces = 96, 64, 6, 2
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.render(2); r.render(1)
    img3, acc = r.read_accumulation()
    assert acc == 3 and r.render_progress_percentage() == pytest.approx(50.0)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    ref3, _ = orc.render(duck_oracle.scene, rp, 0, 3)
    _compare(img3, ref3, 3)
    r.set_render_parameters(params)            # identical -> no reset
    assert r.read_accumulation()[1] == 3
    r.render(10)                               # clamps at spp; frameCount advances to 13
    img6, acc = r.read_accumulation()
    assert acc == spp
    ref6, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    _compare(img6, ref6, spp)
    # change exposure -> accumulation restarts, sample indices continue from frameCount = 13
    p2 = rf.make_render_parameters(W, H, params.camera, spp, bounces, params.sky, 0.5)
    r.set_render_parameters(p2)
    assert r.read_accumulation()[1] == 0 and (r.read_accumulation()[0] == 0).all()
    r.render(2)
    img, acc = r.read_accumulation()
    assert acc == 2
    ref = np.zeros((H, W, 4), np.float32)
    for f in (13, 14):                         # oracle: two single-frame calls with acc = 0, 1
        for y in range(H):
            for x in range(0, W, 7):           # subsample columns to keep the CPU work small
                rgb, _ = orc.pixel_sample(duck_oracle.scene, rp, x, y, f)
                ref[y, x, :3] += rgb
    _compare(img[:, ::7], ref[:, ::7], 2)
    assert r.average_renderpass_duration_ms() > 0.0


def test_batching_is_invisible(duck_pt):
    """Samples traced 1 / 3 / all per batch give the same image bit for bit (f32 accumulation in
    sample order, wgsl:55)."""
    W, H, spp, bounces = 160, 96, 9, 4
    tiles = ((W + 31) // 32) * ((H + 31) // 32)
    imgs = []
    for per_batch in (1, 3, 9):
        r, _ = _renderer(duck_pt, W, H, spp, bounces, max_paths_in_flight=per_batch * tiles * 1024)
        r.render(spp)
        imgs.append(r.read_accumulation()[0])
        r.close()
    assert np.array_equal(bits(imgs[0]), bits(imgs[1])) and np.array_equal(bits(imgs[0]), bits(imgs[2]))
    r, _ = _renderer(duck_pt, W, H, spp, bounces)
    for _ in range(spp):
        r.render(1)
    assert np.array_equal(bits(r.read_accumulation()[0]), bits(imgs[0]))


def test_tile_shards_union_equals_whole(duck_pt):
    W, H, spp, bounces = 200, 150, 4, 4
    whole, _ = _renderer(duck_pt, W, H, spp, bounces)
    whole.render(spp)
    want = whole.read_accumulation()[0]
    union = np.zeros_like(want)
    seen = np.zeros((H, W), bool)
    for world in (3,):
        for rank in range(world):
            r, _ = _renderer(duck_pt, W, H, spp, bounces)
            r.set_tile_shard(rank, world)
            assert np.array_equal(r.shard_tiles(), rf.tiles_for_rank(W, H, rank, world))
            r.render(spp)
            part = r.read_accumulation()[0]
            mask = np.zeros((H, W), bool)
            tx = (W + 31) // 32
            for t in r.shard_tiles():
                mask[(t // tx) * 32:(t // tx) * 32 + 32, (t % tx) * 32:(t % tx) * 32 + 32] = True
            assert (part[~mask] == 0).all() and not (seen & mask).any()
            seen |= mask
            union[mask] = part[mask]
            r.close()
    assert seen.all() and np.array_equal(bits(union), bits(want))


def test_render_into_torch_tensor(duck_pt):
    torch = pytest.importorskip("torch")
    W, H, spp, bounces = 96, 64, 3, 2
    r, _ = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    want = r.read_accumulation()[0]
    ptr, nbytes = r.accumulation_device_buffer()
    assert nbytes == len(r.shard_tiles()) * 1024 * 16
    buf = torch.full((nbytes // 4,), 7.0, dtype=torch.float32, device="cuda:0")
    r.bind_accumulation_buffer(buf.data_ptr(), buf.numel() * 4)
    assert r.read_accumulation()[1] == 0
    r.render(spp)
    r.synchronize()
    img = rf.untile(buf.cpu().numpy(), r.shard_tiles(), W, H)
    assert np.array_equal(bits(img), bits(want))
    with pytest.raises(rf.RayfinderError):
        r.bind_accumulation_buffer(buf.data_ptr(), 16)


def test_set_render_parameters_resize_and_limits(duck_pt):
    r, params = _renderer(duck_pt, 64, 64, 2, 2, max_width=128, max_height=128)
    p2 = rf.make_render_parameters(128, 96, rf.fly_camera(128, 96), 2, 2, params.sky, 0.25)
    r.set_render_parameters(p2)
    r.render(2)
    img, acc = r.read_accumulation()
    assert img.shape == (96, 128, 4) and acc == 2 and np.isfinite(img).all() and img[..., :3].max() > 0
    with pytest.raises(rf.RayfinderError):
        r.set_render_parameters(rf.make_render_parameters(256, 256, rf.fly_camera(256, 256), 2, 2, params.sky, 0.25))
    with pytest.raises(rf.RayfinderError):
        r.set_render_parameters(rf.make_render_parameters(64, 64, rf.fly_camera(64, 64), 2, 2, rf.make_sky(turbidity=0.5), 0.25))


# ------------------------------------------------------------------ configs 3-5: the atrium
@pytest.fixture(scope="module")
def atrium():
    from rayfinder_amd import scenes
    pt, info = scenes.atrium()
    assert 255000 < info["triangles"] < 270000 and info["textures"] == 25
    return pt


def test_atrium_crops_vs_oracle_1080p_8_bounces(atrium):
    W, H, spp, bounces = 1920, 1080, 4, 8
    r, params = _renderer(atrium, W, H, spp, bounces)
    r.set_counting(True)
    r.render(spp)
    img, _ = r.read_accumulation()
    s = r.stats()
    sc, _ = oracle_scene_from_pt(atrium)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    for (x0, y0) in [(928, 508), (64, 64), (1500, 300), (1888, 1048), (700, 900)]:
        ref, st = orc.render(sc, rp, 0, spp, x0, y0, x0 + 32, y0 + 32)
        _compare(img[y0:y0 + 32, x0:x0 + 32], ref[y0:y0 + 32, x0:x0 + 32], spp)
        assert st.stackHigh < 32          # the reference's fixed stack would also suffice here
    # full-size properties
    assert not np.isnan(img).any()
    assert s["primary_rays"] == W * H * spp
    assert s["shadow_rays"] <= s["closest_rays"] <= W * H * spp * bounces
    assert s["stack_high_water"] < 32
    assert s["closest_node_visits"] > 30 * s["closest_rays"]


def test_occluder_cache_is_invisible(atrium, duck_pt):
    """The any-hit launches' occluder cache (kTraceWide, kFlagOccluderCache: a shadow ray first visits the leaf that stopped the last ray from its
    cell of the scene) only changes the ORDER in which an any-hit ray looks at leaves.  Same image bit for bit with the cache off, cold, warm,
    stale (the sun has moved since the grid was filled: every entry points at the wrong occluder), with cells so coarse that whole rooms share one, and
    with a table so small that every entry is fought over."""
    for pt, (W, H, spp, bounces) in ((atrium, (480, 270, 4, 6)), (duck_pt, (200, 150, 4, 4))):
        skies = [rf.make_sky(), rf.make_sky(3.0, (0.3, 0.3, 0.3), 60.0, 140.0), rf.make_sky()]
        off, _ = _renderer(pt, W, H, spp, bounces, sky=skies[0])
        off.set_option("occluder_cache_bounces", 0)
        want = []
        for sky in skies:
            off.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, sky, 0.25))
            off.render(spp)
            want.append(off.read_accumulation()[0])
        off.close()
        assert not np.array_equal(bits(want[0]), bits(want[1]))          # the sun really moved
        assert np.array_equal(bits(want[0]), bits(want[2]))
        for cells, log2_table in ((1024, 22), (8, 22), (4096, 6), (512, 20)):  # default; whole rooms in one cell; a 64-cell table (every entry fought over); ...
            # ... and the cache remembering a record two quad levels above the leaf (what a finely tessellated scene gets by itself: the entry is an interior record)
            os.environ.pop("RF_OCCLUDER_HINT_LEVELS", None)
            if cells == 512:
                os.environ["RF_OCCLUDER_HINT_LEVELS"] = "2"
            r, _ = _renderer(pt, W, H, spp, bounces, sky=skies[0])
            os.environ.pop("RF_OCCLUDER_HINT_LEVELS", None)
            r.set_option("occluder_grid_cells", cells)
            r.set_option("occluder_grid_log2_cells", log2_table)
            r.set_option("shadow_first_look_from_bounce", {1024: 2, 8: 1, 4096: 0, 512: 2}[cells])   # the default; kShadowFirstLook at every bounce; in-kernel first look only
            for sky, ref in zip(skies, want):                             # cold, then stale, then stale the other way round
                r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, sky, 0.25))
                r.render(spp)
                assert np.array_equal(bits(r.read_accumulation()[0]), bits(ref)), (cells, log2_table)
            r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, skies[2], 0.5))   # warm
            r.render(spp)
            assert np.array_equal(bits(r.read_accumulation()[0]), bits(want[2])), (cells, log2_table, "warm")
            r.close()


def test_shadow_rays_stopped_by_the_triangle_they_start_on_are_settled_by_kshade(atrium, duck_pt):
    """kShadeSelfShadow: the reference pushes a hit point off the surface along the GEOMETRIC normal whatever side the path came from (wgsl:511-519), so wherever the sun stands
    behind that normal shadowRay (wgsl:321-368) is stopped by the surface itself.  kShade tests exactly that -- the exact box of the triangle's leaf, then the triangle, both in
    the shading record it holds -- and only the other hits are queued for the any-hit launches.  Same image bit for bit with the option off (cache on and off, first look from
    bounce 1 / 2 / never), every shadow ray still counted, and the test really engages (more than a third of the atrium's shadow rays)."""
    for pt, (W, H, spp, bounces) in ((atrium, (640, 360, 4, 6)), (duck_pt, (200, 150, 4, 4))):
        imgs, stats = {}, {}
        for name, opts in (("off", dict(shadow_self_test=0)), ("on", {}), ("on, cache off", dict(occluder_cache_bounces=0)), ("on, first look from 1", dict(shadow_first_look_from_bounce=1)),
                           ("on, no first look", dict(shadow_first_look_from_bounce=0))):
            r, _ = _renderer(pt, W, H, spp, bounces)
            for k, v in opts.items():
                r.set_option(k, v)
            r.render(spp); r.synchronize()                               # cold grid
            r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, rf.make_sky(), 0.5))
            r.reset_stats()
            r.render(spp)                                                # warm grid: kShadowFirstLook works through kShade's list
            imgs[name], stats[name] = r.read_accumulation()[0], (r.stats(), r.bounce_stats())
            r.close()
        off = stats["off"][0]
        assert off["shadow_rays_self_answered"] == 0
        for name in imgs:
            s, b = stats[name]
            assert np.array_equal(bits(imgs[name]), bits(imgs["off"])), name
            assert s["shadow_rays"] == off["shadow_rays"] and s["closest_rays"] == off["closest_rays"], name
            assert np.array_equal(np.asarray(b["shadow_rays"]), np.asarray(stats["off"][1]["shadow_rays"])), name
            if name != "off":
                assert s["shadow_rays_self_answered"] == stats["on"][0]["shadow_rays_self_answered"], name     # (a property of the rays, not of the cache)
                assert s["shadow_rays_self_answered"] + s["shadow_rays_hint_answered"] <= s["shadow_rays"], name
        if pt is atrium:
            assert stats["on"][0]["shadow_rays_self_answered"] > off["shadow_rays"] // 3


def test_pinhole_primary_origin_as_a_kernel_argument_is_invisible(duck_pt, duck_oracle):
    """A pinhole camera's primary rays all start at camera.origin: kRaygen then writes no origins and the bounce-1 closest-hit launch takes the point as a kernel
    argument (kFlagConstOrigin).  Same image bit for bit with the option off; a camera with a lens, and a pinhole whose origin has a component that is exactly zero
    (a +-0 the lens term could flip), keep the stream -- and all of them equal the oracle."""
    W, H, spp, bounces = 160, 120, 4, 3
    sky = rf.make_sky()
    cams = {"pinhole": rf.fly_camera(W, H),
            "pinhole, origin.x == 0": rf.fly_camera(W, H, position=(0.0, 1.25, -1.25)),
            "lens": rf.fly_camera(W, H, aperture=0.05, focus_distance=2.0)}
    for name, cam in cams.items():
        imgs = []
        for opt in (1, 0):
            r, params = _renderer(duck_pt, W, H, spp, bounces, cam=cam)
            r.set_option("const_primary_origin", opt)
            r.render(spp)
            imgs.append(r.read_accumulation()[0])
            r.close()
        assert np.array_equal(bits(imgs[0]), bits(imgs[1])), name
        want, _ = orc.render(duck_oracle.scene, orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(sky)), 0, spp)
        assert np.array_equal(bits(imgs[0][..., :3]), bits(want[..., :3])), name


def test_occluder_cache_engages_on_the_atrium(atrium):
    """The cache is not silently off: on the second batch of a handle (warm grid) kShadowFirstLook settles most of the shadow rays of bounces >= 2, the shadow
    launches take clearly less time than with the cache off, and the per-bounce ray counts do not change (every shadow ray is still counted)."""
    W, H, spp, bounces = 1920, 1080, 16, 8                            # (33 M paths per batch: launches long enough for their time to mean something)
    r, _ = _renderer(atrium, W, H, spp, bounces)
    r.set_option("shadow_self_test", 0)                              # (kShade's own-triangle test would settle six shadow rays in ten before the cache sees them: its own test below)
    r.render(spp); r.synchronize()                                   # first batch: fills the grid
    res = {}
    for name, n in (("on", 64), ("off", 0)):
        r.set_option("occluder_cache_bounces", n)
        r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, rf.make_sky(), 0.3 + 0.1 * (n == 0)))
        r.set_timing(True); r.reset_stats()
        r.render(spp); r.synchronize()
        res[name] = (r.stats(), r.bounce_stats(), r.read_accumulation()[0])
    r.close()
    on, off = res["on"][0], res["off"][0]
    deep = float(np.sum(res["on"][1]["shadow_rays"][1:]))
    assert on["shadow_rays"] == off["shadow_rays"] and on["closest_rays"] == off["closest_rays"]
    assert np.array_equal(np.asarray(res["on"][1]["shadow_rays"]), np.asarray(res["off"][1]["shadow_rays"]))
    assert off["shadow_rays_hint_answered"] == 0
    assert on["shadow_rays_hint_answered"] > 0.6 * deep, (on["shadow_rays_hint_answered"], deep)
    # (the TIME the cache saves is bench.py's business -- `occluder_cache.ms_shadow` against `ms_shadow_with_cache_off` in the line: a timing assertion in the parity suite
    # turns a perf wobble of a shared or throttled box into a red run that hides every later test under `pytest -x`, VERDICT r4 -- here it is a warning)
    if not on["ms_shadow"] < 0.85 * off["ms_shadow"]:
        import warnings
        warnings.warn(f"occluder cache: shadow launches {on['ms_shadow']:.2f} ms with the cache against {off['ms_shadow']:.2f} ms without (expected < 85 %)")
    assert np.array_equal(bits(res["on"][2]), bits(res["off"][2]))


def test_atrium_config5_4k_16_bounces_crops_and_queue_occupancy(atrium):
    """BASELINE.json config 5 geometry (3840x2160, 16 bounces; NEE is always on): oracle parity on crops and
    the per-bounce queue statistics (queue occupancy per bounce)."""
    W, H, spp, bounces = 3840, 2160, 2, 16
    r, params = _renderer(atrium, W, H, spp, bounces)
    r.reset_stats()
    r.render(spp)
    img, _ = r.read_accumulation()
    s, bs = r.stats(), r.bounce_stats()
    sc, _ = oracle_scene_from_pt(atrium)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    for (x0, y0) in [(1900, 1060), (3800, 2120), (16, 1200)]:
        ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x0 + 24, y0 + 24)
        _compare(img[y0:y0 + 24, x0:x0 + 24], ref[y0:y0 + 24, x0:x0 + 24], spp)
    assert not np.isnan(img).any()
    # occupancy: bounce 1 traces every path, a path enters bounce b+1 iff it hit something at bounce b, totals add up
    cr, sr = bs["closest_rays"].astype(np.int64), bs["shadow_rays"].astype(np.int64)
    assert len(cr) == bounces and cr[0] == W * H * spp
    assert np.array_equal(cr[1:], sr[:-1]) and (sr <= cr).all() and (np.diff(cr) <= 0).all()
    assert cr.sum() == s["closest_rays"] and sr.sum() == s["shadow_rays"]


# ------------------------------------------------------------------ round 3: the BASELINE.json configurations at their FULL sample counts
def test_config2_duck_800x600_at_its_full_64_spp_matches_golden_crops(duck_pt, duck_oracle):
    """BASELINE.json config 2 as written: Duck.pt, 800x600, 64 spp, 4 bounces -- against the committed oracle crops
    (tests/golden/make_golden.py duck_render64) and one more crop rendered live by the oracle."""
    g = np.load(os.path.join(GOLDEN, "duck_render_golden_64spp.npz"))
    W, H, spp, bounces = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["bounces"])
    assert (W, H, spp, bounces) == (800, 600, 64, 4)
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp
    for (x0, y0), want in zip(g["crops"], g["sums"]):
        assert np.array_equal(bits(img[y0:y0 + 16, x0:x0 + 16, :3]), bits(want)), (x0, y0)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    x0, y0 = 430, 310
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp, x0, y0, x0 + 8, y0 + 8)
    assert np.array_equal(bits(img[y0:y0 + 8, x0:x0 + 8, :3]), bits(ref[y0:y0 + 8, x0:x0 + 8, :3]))


def test_config3_atrium_1080p_at_its_full_256_spp_crops_vs_oracle(atrium):
    """BASELINE.json config 3 as written (on the Sponza stand-in): 1920x1080, 256 spp, 8 bounces -- 8x8 oracle crops, f32 sums of
    256 samples per pixel bit for bit."""
    W, H, spp, bounces = 1920, 1080, 256, 8
    r, params = _renderer(atrium, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp and not np.isnan(img).any()
    sc, _ = oracle_scene_from_pt(atrium)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    for (x0, y0) in [(956, 536), (300, 700), (1700, 180)]:
        ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x0 + 8, y0 + 8)
        assert np.array_equal(bits(img[y0:y0 + 8, x0:x0 + 8, :3]), bits(ref[y0:y0 + 8, x0:x0 + 8, :3])), (x0, y0)
    r.close()


def test_config5_atrium_4k_16_bounces_at_its_full_1024_spp_crops_vs_oracle(atrium):
    """BASELINE.json config 5 as written (one GPU's worth: the whole frame on this device): 3840x2160, 1024 spp, 16 bounces -- ~83 G
    rays in several batches (the path state of 8.5 G paths does not fit at once); 8x8 oracle crops bit for bit."""
    W, H, spp, bounces = 3840, 2160, 1024, 16
    r, params = _renderer(atrium, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp and not np.isnan(img).any()
    assert r.stats()["abandoned_rays"] == 0
    sc, _ = oracle_scene_from_pt(atrium)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    for (x0, y0) in [(1916, 1076), (500, 1500)]:
        ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x0 + 8, y0 + 8)
        assert np.array_equal(bits(img[y0:y0 + 8, x0:x0 + 8, :3]), bits(ref[y0:y0 + 8, x0:x0 + 8, :3])), (x0, y0)
    r.close()


@pytest.mark.parametrize("depth", [70, 96, 97, 130])
def test_deep_chain_gpu_and_oracle_share_the_96_entry_stack_bound(depth):
    """The reference's 32-entry traversal stack is overrun (undefined) by a 70-deep chain; product and oracle both define a
    96-entry stack and ABANDON a ray that needs a 97th pending node with what it has found so far (rf_device.hpp,
    oracle/rf_oracle.c ORC_STACK): queries and rendered images agree bit for bit on either side of the bound."""
    nodes, tris, attrs = _chain_scene(depth)
    sc = rf.scene_from_arrays(nodes, tris, attrs, [(np.array([0xFFFFFFFF], np.uint32), 1, 1)])
    W, H, spp, bounces = 64, 64, 2, 2
    cam = rf.create_camera((0.2, 0.2, -5.0), (0.2, 0.2, 0.0), 0.0, 1.0, np.radians(20.0), 1.0)
    params = rf.make_render_parameters(W, H, cam, spp, bounces, rf.make_sky(), 1.0)
    r = rf.ReferencePathTracer(params, sc)
    rng = np.random.default_rng(depth)
    rays = np.tile(np.array([[0.25, 0.25, -5.0, 1e-3, 1e-3, 1.0]], np.float32), (256, 1))
    rays[:, :2] += rng.uniform(-0.2, 0.2, (256, 2)).astype(np.float32)
    rays[128:, 3] *= -1.0                                         # dir.x < 0: the second child (a leaf) is the near one, nothing piles up
    pos48 = np.ascontiguousarray(tris, np.float32)
    cpu = orc.intersect_bvh_batch(nodes, pos48, rays, 10000.0)
    for variant in (0, 2):                                        # scalar kernel over the 32-byte nodes / the render path's persistent kernel
        r.set_option("query_variant", variant)
        gpu = r.intersect_rays(rays, 10000.0)
        assert np.array_equal(gpu["hit"], cpu["hit"]) and np.array_equal(gpu["tri"], cpu["tri"]), (depth, variant)
        assert np.array_equal(bits(gpu["t"]), bits(cpu["t"])), (depth, variant)
        assert np.array_equal(r.occluded_rays(rays, 10000.0), orc.shadow_batch(nodes, pos48, rays, 10000.0)), (depth, variant)
    assert int(cpu["stackHigh"][:128].max()) == min(depth, 97) and int(cpu["stackHigh"][128:].max()) <= 1
    r.reset_stats()
    r.render(spp)
    img, _ = r.read_accumulation()
    s = r.stats()
    assert (s["abandoned_rays"] > 0) == (depth > 96)
    osc = orc.OracleScene(nodes, pos48, attrs, np.array([(1, 1, 0)], np.uint32), np.array([0xFFFFFFFF], np.uint32))
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 1.0, rf.aligned_sky_state(rf.make_sky()))
    ref, st = orc.render(osc, rp, 0, spp)
    if depth <= 96:
        assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3])), depth
    else:
        # past the bound the two sides may differ, and only in the product's favour: the render path's kernels push a far child
        # only if its box can still be hit, so many rays the reference-ordered stack (every far child pending) must abandon are
        # traced to the end; a ray is abandoned only when that filtered stack AND the 96-entry scalar redo both overflow
        assert st.stackHigh == 97
        same = (bits(img[..., :3]) == bits(ref[..., :3])).all(axis=-1)
        assert same.mean() > 0.5
    r.close()


def test_atrium_idempotent_and_shard_union_at_full_size(atrium):
    W, H, spp, bounces = 1920, 1080, 2, 8
    a, params = _renderer(atrium, W, H, spp, bounces)
    a.render(spp)
    img_a = a.read_accumulation()[0]
    a.set_render_parameters(rf.make_render_parameters(W, H, params.camera, spp, bounces, params.sky, 0.125))  # reset
    a.render(spp)
    img_b = a.read_accumulation()[0]
    # frameCount moved on by 2, and 2 % 2 == 0: same sample indices -> same image, bit for bit
    assert np.array_equal(bits(img_a), bits(img_b))
    union = np.zeros_like(img_a)
    tx = (W + 31) // 32
    for rank in range(2):
        a.set_tile_shard(rank, 2)
        a.render(spp)                      # frameCount 4, 5 -> sample indices 0, 1 again
        part = a.read_accumulation()[0]
        for t in a.shard_tiles():
            ys, xs = (t // tx) * 32, (t % tx) * 32
            union[ys:ys + 32, xs:xs + 32] = part[ys:ys + 32, xs:xs + 32]
    assert np.array_equal(bits(union), bits(img_a))


def test_known_answer_quad_scene_on_gpu():
    """The analytic NEE case of tests/test_oracle_pins.py, on the GPU."""
    import math
    from rayfinder_amd import scenes
    pt = scenes.quad_scene()
    cam = rf.create_camera([0, 3, 0.001], [0, 0, 0], 0.0, 1.0, orc.degrees_to_radians(40.0), 1.0)
    r, params = _renderer(pt, 16, 16, 4, 1, cam=cam, exposure=1.0)
    r.render(1)
    img, _ = r.read_accumulation()
    sky = rf.aligned_sky_state(params.sky)
    expected = sky[30:33] * (1.0 / math.pi) * math.cos(math.radians(30.0)) * 6.216817e-05
    assert np.allclose(img[8, 8, :3], expected, rtol=5e-3)
    sc, _ = oracle_scene_from_pt(pt)
    rp = orc.make_render_params(16, 16, rf.camera_to_array(cam), 4, 1, 1.0, sky)
    ref, _ = orc.render(sc, rp, 0, 1)
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))


# ------------------------------------------------------------------ two ranks sharing one GPU
def _rank_worker(rank, world, port, w, h, spp, bounces, outdir):
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch
    import torch.distributed as dist
    import rayfinder_amd as rf2
    from rayfinder_amd.sharding import gather_image, shard_layout
    dist.init_process_group("gloo", rank=rank, world_size=world)   # NCCL refuses two ranks on one GPU
    pt = rf2.PtFormat.from_gltf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Duck.glb"))
    params = rf2.make_render_parameters(w, h, rf2.fly_camera(w, h), spp, bounces, rf2.make_sky(), 0.25)
    r = rf2.ReferencePathTracer(params, pt.scene())
    r.set_tile_shard(rank, world)
    tiles, max_tiles = shard_layout(w, h, rank, world)
    accum = torch.zeros((max_tiles * 1024, 4), dtype=torch.float32, device="cuda:0")
    r.bind_accumulation_buffer(accum.data_ptr(), accum.numel() * 4)
    r.render(spp)
    r.synchronize()
    image = gather_image(accum.cpu(), w, h, rank, world)          # same code path as bench.py, CPU tensors for gloo
    if rank == 0:
        np.save(os.path.join(outdir, "sharded.npy"), image)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reassemble_the_single_rank_image(duck_pt, tmp_path):
    """End-to-end multi-rank path (tile shard -> render into a torch tensor -> gather -> un-tile)
    with two processes; the result equals the single-rank image bit for bit."""
    import socket
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    W, H, spp, bounces = 300, 200, 4, 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_worker, args=(2, port, W, H, spp, bounces, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npy")
    r, _ = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    want = r.read_accumulation()[0]
    assert np.array_equal(bits(got), bits(want))


# ------------------------------------------------------------------ layout edge cases of the wide BVH
def _soup_pt(P, tex_rgb=(200, 180, 160)):
    n = P.shape[0]
    N = np.tile(np.array([0, 1, 0], np.float32), (n, 3))
    UV = np.tile(np.array([0, 0, 1, 0, 0, 1], np.float32), (n, 1))
    r, g, b = tex_rgb
    return rf.PtFormat.from_triangles(P, N, UV, np.zeros(n, np.uint32), [(np.array([b | (g << 8) | (r << 16) | (255 << 24)], np.uint32), 1, 1)])


def _check_scene_against_oracle(pt, rays, render_wh=(64, 48), cam=None):
    sc, a = oracle_scene_from_pt(pt)
    W, H = render_wh
    cam = cam if cam is not None else rf.create_camera([0.3, 0.4, 3.0], [0, 0, 0], 0.0, 1.0, orc.degrees_to_radians(60.0), W / H)
    r, params = _renderer(pt, W, H, 4, 5, cam=cam)
    with np.errstate(all="ignore"):
        cpu = orc.intersect_bvh_batch(a["bvhNodes"], a["trianglePositionAttributes"], rays, 10000.0)
        cpu_vis = orc.shadow_batch(a["bvhNodes"], a["trianglePositionAttributes"], rays, 10000.0)
    gpu = r.intersect_rays(rays, 10000.0)
    assert np.array_equal(gpu["tri"], cpu["tri"]) and np.array_equal(bits(gpu["t"]), bits(cpu["t"]))
    assert np.array_equal(gpu["nodesVisited"], cpu["nodesVisited"])
    assert np.array_equal(r.occluded_rays(rays, 10000.0), cpu_vis)
    # the render path (wide records, persistent kernel) in counting mode, reference order
    r.set_option("shadow_nearest_first", 0)
    r.set_counting(True)
    r.render(4)
    img = r.read_accumulation()[0]
    s = r.stats()
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), 4, 5, 0.25, rf.aligned_sky_state(params.sky))
    ref, st = orc.render(sc, rp, 0, 4)
    _compare(img, ref, 4)
    assert s["closest_node_visits"] == st.closestNodeVisits and s["shadow_node_visits"] == st.shadowNodeVisits
    assert s["closest_triangle_tests"] == st.closestTriTests and s["stack_high_water"] == st.stackHigh
    return st, cpu


def test_single_leaf_root_and_big_leaves():
    rng = np.random.default_rng(3)
    rays = _random_rays(rng, 4000, np.array([-1.0, -1, -1]), np.array([1.0, 1, 1]))
    # (a) one triangle: the whole tree is one leaf (root leaf word)
    P = np.array([[-1, -1, 0, 1, -1, 0, 0, 1, 0]], np.float32)
    _check_scene_against_oracle(_soup_pt(P), rays)
    # (b) 40 identical-centroid triangles: ONE leaf of 40 (> 7 -> big-leaf table), and a mix with normal leaves
    fan = np.zeros((40, 9), np.float32)
    ang = np.linspace(0, np.pi, 40, endpoint=False)
    fan[:, 0] = np.cos(ang); fan[:, 1] = np.sin(ang); fan[:, 3] = -np.cos(ang); fan[:, 4] = -np.sin(ang)
    fan[:, 8] = 0.3; fan[:, 2] = fan[:, 5] = -0.15      # all centroids at the origin
    pt = _soup_pt(fan)
    nodes = pt.arrays()["bvhNodes"]
    assert len(nodes) == 1 and nodes[0]["triangleCount"] == 40
    _check_scene_against_oracle(pt, rays)
    others = (rng.uniform(-1, 1, (300, 1, 3)) + rng.normal(0, 0.08, (300, 3, 3))).astype(np.float32).reshape(300, 9)
    pt = _soup_pt(np.concatenate([fan + np.float32(2.0), others]))
    assert (pt.arrays()["bvhNodes"]["triangleCount"] >= 8).any()
    _check_scene_against_oracle(pt, rays)


def test_deep_tree_uses_the_scratch_part_of_the_stack():
    """Geometrically shrinking triangles along +x make the SAH builder peel a few primitives per
    level: a ~30-level chain.  Rays along the chain keep one pending far child per level, so the
    traversal stack goes past its LDS-resident part (12 entries on the render path, 24 on the query
    path) into scratch -- but stays inside the reference's 32 and the oracle's 64."""
    rng = np.random.default_rng(9)
    n = 160
    x = 4.0 * 1.3 ** (-np.arange(n, dtype=np.float64))
    sz = 0.2 * x
    P = np.zeros((n, 9))
    P[:, 0] = x; P[:, 1] = -sz; P[:, 2] = -0.3 * sz
    P[:, 3] = x; P[:, 4] = sz; P[:, 5] = -0.3 * sz
    P[:, 6] = x + 0.1 * sz; P[:, 8] = 0.6 * sz
    P = P.astype(np.float32)
    pt = _soup_pt(P)
    # rays aimed at the apex of the chain (the origin) from x = -1 pass through every box of it
    o = np.stack([np.full(3000, -1.0), rng.uniform(-0.1, 0.1, 3000), rng.uniform(-0.05, 0.05, 3000)], axis=1)
    d = -o / np.linalg.norm(o, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    rays = np.concatenate([rays, _random_rays(rng, 3000, np.array([0.0, -1, -1]), np.array([4.0, 1, 1]))])
    cam = rf.create_camera([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], 0.0, 1.0, orc.degrees_to_radians(12.0), 1.5)  # eye at the apex
    st, cpu = _check_scene_against_oracle(pt, rays, render_wh=(48, 32), cam=cam)
    assert 24 < cpu["stackHigh"].max() < 32, int(cpu["stackHigh"].max())
    assert 12 < st.stackHigh < 32, st.stackHigh


# ------------------------------------------------------------------ randomized differential test
def _random_scene(rng, kind):
    """Triangle soups that stress different corners: axis-aligned boxes (zero direction components after
    bounces, rays starting exactly on box planes), slivers and degenerate triangles, clustered duplicates
    (multi-triangle leaves), a large ground plane with small clutter."""
    if kind == "boxes":
        tris = []
        for _ in range(int(rng.integers(3, 12))):
            c = rng.uniform(-2, 2, 3); e = rng.uniform(0.1, 0.8, 3)
            lo, hi = c - e, c + e
            v = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
            for a, b, c2, d in ((0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)):
                tris += [[v[a], v[b], v[c2]], [v[a], v[c2], v[d]]]
        tris += [[[-6, -2.5, -6], [6, -2.5, -6], [6, -2.5, 6]], [[-6, -2.5, -6], [6, -2.5, 6], [-6, -2.5, 6]]]
        tris = np.array(tris, np.float32)
    elif kind == "slivers":
        c = rng.uniform(-2, 2, (400, 1, 3)); tris = (c + rng.normal(0, 0.4, (400, 3, 3)) * np.array([1.0, 0.02, 1.0])).astype(np.float32)
        tris[:20, 2] = tris[:20, 1]                               # degenerate (zero area)
    elif kind == "duplicates":
        base = (rng.uniform(-2, 2, (30, 1, 3)) + rng.normal(0, 0.5, (30, 3, 3))).astype(np.float32)
        tris = base[rng.integers(0, 30, 900)]
    else:
        c = rng.uniform(-3, 3, (1500, 1, 3)) * np.array([1, 0.3, 1]); tris = (c + rng.normal(0, 0.15, (1500, 3, 3))).astype(np.float32)
        tris = np.concatenate([tris, np.array([[[-8, -1.2, -8], [8, -1.2, -8], [8, -1.2, 8]], [[-8, -1.2, -8], [8, -1.2, 8], [-8, -1.2, 8]]], np.float32)])
    n = len(tris)
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    gn = np.cross(e1, e2); ln = np.linalg.norm(gn, axis=1, keepdims=True)
    gn = np.where(ln > 0, gn / np.maximum(ln, 1e-30), np.array([0.0, 1.0, 0.0]))
    normals = (np.repeat(gn[:, None, :], 3, 1) + rng.normal(0, 0.1, (n, 3, 3))).astype(np.float32)   # not unit: the reference does not care
    uvs = rng.uniform(-2, 3, (n, 3, 2)).astype(np.float32)
    ntex = int(rng.integers(1, 4))
    textures = []
    for _ in range(ntex):
        w, h = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        rgb = rng.integers(0, 256, (h * w, 3), dtype=np.uint32)
        textures.append(((rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(255 << 24)).astype(np.uint32), w, h))
    return rf.PtFormat.from_triangles(tris.reshape(n, 9), normals.reshape(n, 9), uvs.reshape(n, 6), rng.integers(0, ntex, n).astype(np.uint32), textures)


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_cameras_and_skies_bit_identical_to_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = ["boxes", "slivers", "duplicates", "clutter"][seed % 4]
    pt = _random_scene(rng, kind)
    W, H = int(rng.integers(5, 13)) * 8 - int(rng.integers(0, 7)), int(rng.integers(4, 10)) * 8 - int(rng.integers(0, 7))
    spp, bounces = int(rng.integers(1, 6)), int(rng.integers(1, 8))
    pos = rng.uniform(-3, 3, 3); pos[1] = abs(pos[1]) + 0.3
    cam = rf.create_camera(pos, rng.uniform(-1, 1, 3), float(rng.choice([0.0, 0.05, 0.3])), float(rng.uniform(1, 6)),
                           orc.degrees_to_radians(float(rng.uniform(30, 100))), W / H)
    sky = rf.make_sky(turbidity=float(rng.uniform(1, 10)), albedo=tuple(rng.uniform(0, 1, 3)), sun_zenith_degrees=float(rng.uniform(0, 89)),
                      sun_azimuth_degrees=float(rng.uniform(0, 360)))
    if seed % 7 in (3, 6):                                      # the occluder cache remembers a record 1 / 2 / 3 quad levels above the leaf it found the occluder in
        os.environ["RF_OCCLUDER_HINT_LEVELS"] = str(1 + seed % 3)
    try:
        r, params = _renderer(pt, W, H, spp, bounces, cam=cam, sky=sky, exposure=0.5)
    finally:
        os.environ.pop("RF_OCCLUDER_HINT_LEVELS", None)
    if seed % 4 != 2:                                           # (default since round 3: the quad records at every bounce)
        r.set_option("quad_from_bounce", 0 if seed % 4 != 1 else 3)
        r.set_option("quad_shadow_from_bounce", 0 if seed % 4 != 3 else 2)
    if seed % 3 == 1:                                           # every launch on the compact-capable records (without quad records: the deeper bounces only)
        r.set_option("compact_from_bounce", 1)
        r.set_option("compact_shadow_from_bounce", 1)
    if seed % 4 == 3:                                           # kShade appends each tile's hits in triangle order
        r.set_option("shade_sort_from_bounce", seed % 3)
    if seed % 5 in (1, 3):                                      # the quad launches (if any) on the exact quad records / the half-precision ones
        r.set_option("quad_half_from_bounce", 0 if seed % 5 == 1 else 1 + seed % 3)   # (default since the end of round 3: half precision, where the scene suits it)
        r.set_option("quad_half_shadow_from_bounce", 0 if seed % 5 == 1 else 1 + seed % 2)
    if seed % 7 in (2, 5):                                      # the quad launches on the local-grid quad records (they take over where the half-precision ones are off)
        r.set_option("quad_half_from_bounce", 0 if seed % 7 == 2 else 3)
        r.set_option("quad_half_shadow_from_bounce", 0)
        r.set_option("quad_local_from_bounce", 1)
        r.set_option("quad_local_shadow_from_bounce", 1 + seed % 2)
    if seed % 3 == 2:                                           # ... on the 32-byte records
        r.set_option("hot_from_bounce", 1)
        r.set_option("hot_shadow_from_bounce", 1)
    if seed % 6 in (0, 5):                                      # occluder cache: cells as large as the scene / a few per object (entries that rarely fit), or off
        r.set_option("occluder_grid_cells", (3, 40)[seed % 2])
    if seed % 11 == 4:
        r.set_option("occluder_cache_bounces", 0)
    if seed % 5 in (2, 4):                                      # kShadowFirstLook from bounce 1 / never (default: from bounce 2, once the grid is warm)
        r.set_option("shadow_first_look_from_bounce", 1 if seed % 5 == 2 else 0)
    if seed % 5 in (0, 3) or os.environ.get("RF_FUZZ_DENSE"):   # leaf phases over dense (lane, triangle) pairs from 2 / 3 triangles per leaf on (default: 5; "duplicates" scenes have leaves of dozens)
        r.set_option("dense_leaf_min", 2 + seed % 2)
    if seed % 8 in (2, 6) or os.environ.get("RF_FUZZ_OCT"):     # closest-hit launches on the 128-byte oct records (three levels per fetch) from bounce 1 / 2
        if os.environ.get("RF_FUZZ_OCT"):
            r.set_option("quad_from_bounce", 1)
        r.set_option("oct_from_bounce", 1 if seed % 8 == 2 else 2 if seed % 8 == 6 else 1 + seed % 3)
    if seed % 9 in (1, 5):                                      # kShade's own-triangle test of the shadow rays off (default: on)
        r.set_option("shadow_self_test", 0)
    if seed % 3 == 0:                                           # round 6: the closest-hit launches' deep refill threshold, from "a refill for every finished lane" up, ...
        r.set_option("refill_min_deep", 1 + seed % 9)
        r.set_option("leaf_vote", 1 + (seed % 7) * 9)           # ... and the number of lanes that must still descend for the descend loop to go on (1 ... 55)
    if seed % 2:                                                # one batch per sample: every batch after the first starts on a warm occluder grid
        for _ in range(spp):
            r.render(1)
    else:
        r.render(spp)
    img, acc = r.read_accumulation()
    assert acc == spp
    sc, _ = oracle_scene_from_pt(pt)
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.5, rf.aligned_sky_state(sky))
    with np.errstate(all="ignore"):
        ref, _ = orc.render(sc, rp, 0, spp)
    g, c = img[..., :3], ref[..., :3]
    assert np.array_equal(np.isnan(g), np.isnan(c)), (kind, "NaN pixels differ")
    same = (bits(g) == bits(c)) | np.isnan(g)
    assert same.all(), (kind, W, H, spp, bounces, int((~same).sum()), float(np.nanmax(np.abs(g - c))))


# ------------------------------------------------------------------ round 5: the renderer's OWN layout choice on a finely tessellated scene, under the oracle's eye
def test_fine_scene_renderer_picks_its_layouts_itself_and_matches_the_oracle():
    """VERDICT r4 item 3.  Until now the driver-run suite forced the record layouts (query_compact on Duck, option sets in the fuzz): the selector -- exact quad
    records for the primary launch, the local-grid records from bounce 2, exact quad records for the any-hit launches, occluder-cache entries that name a record
    above the leaf -- only ever ran in builder-run bench commands.  Here the atrium at 4 x tessellation (4.2 M triangles of 4.5 cm: binary16 area ratio 1.117, beyond both
    thresholds of rf_wide.hpp) is rendered with NO option set; rf_renderer_layout_info says what the renderer took, and 1080p crops at 8 spp / 8 bounces equal the
    oracle bit for bit -- with the occluder cache warm (second batch) and with it off."""
    from rayfinder_amd import scenes
    rf.set_bake_bvh_builder(0)                                        # the GPU builder: same node bytes as the host's (tests/test_gpu_bvh_build.py), seconds at this size
    try:
        pt, info = scenes.atrium(4)
    finally:
        rf.set_bake_bvh_builder(None)
    assert info["triangles"] > 4_000_000
    W, H, spp, bounces = 1920, 1080, 8, 8
    r, params = _renderer(pt, W, H, spp, bounces)
    li = r.layout_info(bounces)
    assert li["quad_half_area_ratio"] > 1.10 and not li["legacy_layouts_compiled"], li
    assert li["closest"][0] == "quad" and all(x == "quad_local" for x in li["closest"][1:]), li        # binary16 too coarse: exact records for the coherent primary launch, the local grid from bounce 2
    assert all(x == "quad" for x in li["shadow"]) and all(li["shadow_cached"]), li                     # a shadow ray crosses the whole scene: exact boxes, started at the occluder cache
    assert li["occluder_hint_levels"] >= 1 and li["dense_leaf_min"] >= 1, li                           # 4.5-cm leaves against the sun disc's footprint: the cache remembers records, not leaves
    for _ in range(2):                                                 # two batches of 4 spp: the second starts on a warm occluder grid
        r.render(spp // 2)
    img, acc = r.read_accumulation()
    assert acc == spp
    sc, _ = oracle_scene_from_pt(pt)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    for (x0, y0) in [(944, 524), (200, 860), (1500, 300)]:            # centre, floor at the lower left, upper storey
        ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x0 + 32, y0 + 32)
        g, c = img[y0:y0 + 32, x0:x0 + 32, :3], ref[y0:y0 + 32, x0:x0 + 32, :3]
        assert np.array_equal(np.isnan(g), np.isnan(c))
        assert ((bits(g) == bits(c)) | np.isnan(g)).all(), (x0, y0, int((bits(g) != bits(c)).sum()))
    s = r.stats()
    assert s["abandoned_rays"] == 0
    # the same frames without the cache: the selector's other choices stay, the image does not move
    r.set_option("occluder_cache_bounces", 0)
    assert not any(r.layout_info(bounces)["shadow_cached"])
    r.set_render_parameters(rf.make_render_parameters(W, H, params.camera, spp, bounces, params.sky, 0.3))
    r.render(spp)
    off, _ = r.read_accumulation()
    r.close()
    # (frameCount went on counting: sample indices n = frameCount % spp are the same eight, so the sums are the same sums in another order of batches -- per pixel the
    # k-ordered accumulation makes them bit-identical, kAccumulateRuns)
    assert np.array_equal(bits(off[..., :3]), bits(img[..., :3]))


# ------------------------------------------------------------------ round 2: stack overflow is counted, hostile scenes are refused
def _chain_scene(depth):
    """A hand-made, valid, maximally unbalanced tree: interior k = {interior k+1 | leaf k}, all boxes nested around the
    z axis so that a ray along +z enters every node and has to remember `depth` far children."""
    n_int = depth
    nodes = np.zeros(2 * n_int + 1, dtype=rf.NODE_DTYPE)
    tris = np.zeros((n_int + 1, 12), np.float32)
    def tri(z):
        return np.array([-1, -1, z, 0, 3, -1, z, 0, -1, 3, z, 0], np.float32)
    # leaf order: the deepest left leaf first, then the right leaves from the bottom up (preorder)
    for k in range(n_int + 1):
        tris[k] = tri(10.0 + k)
    for k in range(n_int):
        nodes[k]["min"] = (-1, -1, 0); nodes[k]["max"] = (3, 3, 1000)
        nodes[k]["splitAxis"] = 0                          # ray dir.x > 0 -> first child is the near one
        nodes[k]["secondChildOffset"] = 2 * n_int - k     # its right leaf
    for j in range(n_int + 1):                             # leaves n_int .. 2 n_int
        i = n_int + j
        nodes[i]["min"] = (-1, -1, 10.0 + j); nodes[i]["max"] = (3, 3, 10.0 + j)
        nodes[i]["trianglesOffset"] = j; nodes[i]["triangleCount"] = 1; nodes[i]["splitAxis"] = 0xFFFFFFFF
    attrs = np.zeros((n_int + 1, 20), np.float32)
    attrs[:, [2, 6, 10]] = -1.0
    return nodes, tris, attrs


def test_rays_that_outgrow_the_traversal_stack_are_counted():
    for depth, expect_abandoned in ((60, False), (130, True)):
        nodes, tris, attrs = _chain_scene(depth)
        sc = rf.scene_from_arrays(nodes, tris, attrs, [(np.array([0xFFFFFFFF], np.uint32), 1, 1)])
        W, H = 64, 64
        params = rf.make_render_parameters(W, H, rf.create_camera((0.2, 0.2, -5.0), (0.2, 0.2, 0.0), 0.0, 1.0, np.radians(20.0), 1.0), 1, 1, rf.make_sky(), 1.0)
        r = rf.ReferencePathTracer(params, sc)
        rays = np.tile(np.array([[0.25, 0.25, -5.0, 1e-3, 1e-3, 1.0]], np.float32), (256, 1))
        out = r.intersect_rays(rays, 10000.0)                  # scalar kernel over the 32-B nodes
        r.render(1)                                            # packed kernel: 12 pending far children -> scalar redo -> same cap
        s = r.stats()
        if expect_abandoned:
            assert s["abandoned_rays"] >= 256
        else:
            assert s["abandoned_rays"] == 0
            assert (out["tri"] == 0).all()                # nearest plane = the deepest-left leaf's triangle (z = 10)
            assert s["scalar_redo_rays"] > 0                   # 60 pending far children do not fit the 12-entry LDS stack
        r.close()


def test_renderer_create_refuses_malformed_scenes(duck_pt):
    a = duck_pt.arrays()
    tex = [(px, w, h) for (px, w, h) in a["baseColorTextures"]]
    params = rf.make_render_parameters(32, 32, rf.fly_camera(32, 32), 1, 1, rf.make_sky(), 1.0)
    interior = int(np.nonzero(a["bvhNodes"]["triangleCount"] == 0)[0][3])
    for mutate in ("cycle", "leaf", "texture"):
        nodes = a["bvhNodes"].copy(); attrs = a["triangleVertexAttributes"].copy()
        if mutate == "cycle":
            nodes[interior]["secondChildOffset"] = interior
        elif mutate == "leaf":
            leaf = int(np.nonzero(nodes["triangleCount"] > 0)[0][0]); nodes[leaf]["trianglesOffset"] = len(attrs)
        else:
            attrs.view(np.uint32).reshape(-1, 20)[7, 18] = 9
        with pytest.raises(rf.RayfinderError):
            rf.ReferencePathTracer(params, rf.scene_from_arrays(nodes, a["trianglePositionAttributes"], attrs, tex))


def test_rccl_frame_exchange_world_size_one(duck_pt):
    """The C++ RCCL path (rf_comm.hip) on the one GPU there is: a world-size-1 communicator, the root's shard sent to
    itself through ncclSend / ncclRecv (loopback) and un-tiled on the device == the host un-tile of the same
    buffer, bit for bit; the in-place path (no loopback) and the max-reduction too.  Two ranks cannot share a
    GPU under RCCL, so world > 1 is covered by the gloo tests of the same tile arithmetic (test_distributed_cpu)."""
    W, H, spp, bounces = 200, 150, 4, 3          # ragged right / bottom tiles
    r, _ = _renderer(duck_pt, W, H, spp, bounces)
    comm = rf.TileComm(rf.comm_unique_id(), 0, 1, 0)
    r.render(spp)
    want, acc = r.read_accumulation()
    for loopback in (True, False):
        ptr = r.gather_frame(comm, root=0, loopback=loopback)
        assert ptr
        got = comm.read_frame(r, W, H)
        assert np.array_equal(bits(got), bits(want)), loopback
    assert comm.all_reduce_max(3.5, r) == 3.5
    # display transform of the gathered device image == the renderer's own
    bgra = r.tonemap_device_image(ptr, W, H, spp)
    assert np.array_equal(bgra, r.read_tonemapped())
    comm.close()
    r.close()


def _grade(gpu_sum, ref_sum, spp):
    """SURVEY 8(d)'s stated radiance tolerance, reported in full: share of pixels within 1e-3*|ref| + 1e-4*spp per channel, image-mean relative error, NaN pixels on
    either side (counted, not hidden), share of bit-identical pixels, worst absolute difference."""
    g, c = gpu_sum[..., :3].astype(np.float64), ref_sum[..., :3].astype(np.float64)
    nan_g, nan_c = np.isnan(g).any(-1), np.isnan(c).any(-1)
    both = ~(nan_g | nan_c)
    diff = np.abs(g - c)
    tol = 1e-3 * np.abs(c) + 1e-4 * spp
    within = (diff <= tol).all(-1) & both
    return dict(within=float(within.mean()), mean_rel=float(abs(g[both].mean() - c[both].mean()) / max(abs(c[both].mean()), 1e-30)), nan_gpu=int(nan_g.sum()), nan_ref=int(nan_c.sum()),
                exact=float(((gpu_sum[..., :3] == ref_sum[..., :3]).all(-1) & both).mean()), max_abs=float(diff[both].max()) if both.any() else 0.0)


def test_f32_transcendentals_mode_within_the_stated_tolerance(duck_pt, duck_oracle, atrium):
    """the opt-in `transcendentals` = 1 mode (kRaygen's sin / cos, kSky's acos / cos / exp / pow through the device math library's f32 functions
    instead of the specified f64 evaluation; wgsl:247-275,568-616 -- WGSL's own builtins are f32 with implementation-defined ulps) graded by SURVEY 8(d)'s tolerance
    against the ORACLE: >= 99.5 % of the pixels within 1e-3*|ref| + 1e-4*spp per channel, image-mean relative error <= 1e-4, NaN pixels reported and equal.  The default
    mode's bit-identity is asserted by the rest of this file; here the two modes are also compared with each other on whole frames."""
    report = {}
    # Duck, whole frame against the oracle
    W, H, spp, bounces = 200, 150, 16, 4
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.set_option("transcendentals", 1)
    r.render(spp)
    img = r.read_accumulation()[0]
    r.close()
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    report["duck_200x150_16spp_vs_oracle"] = _grade(img, ref, spp)
    # config 2 at its full size: the committed oracle crops + f32 mode against the default mode on the whole frame
    g = np.load(os.path.join(GOLDEN, "duck_render_golden_64spp.npz"))
    W, H, spp, bounces = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["bounces"])
    imgs = []
    for mode in (0, 1):
        r, _ = _renderer(duck_pt, W, H, spp, bounces)
        r.set_option("transcendentals", mode)
        r.render(spp)
        imgs.append(r.read_accumulation()[0])
        r.close()
    crops_gpu = np.concatenate([imgs[1][y0:y0 + 16, x0:x0 + 16, :3] for (x0, y0) in g["crops"]])
    crops_ref = np.concatenate([np.asarray(w)[..., :3] for w in g["sums"]])
    assert np.array_equal(bits(np.concatenate([imgs[0][y0:y0 + 16, x0:x0 + 16, :3] for (x0, y0) in g["crops"]])), bits(crops_ref))   # (the default mode: bit-identical, as ever)
    report["duck_800x600_64spp_crops_vs_oracle"] = _grade(crops_gpu, crops_ref, spp)
    report["duck_800x600_64spp_vs_default_mode"] = _grade(imgs[1], imgs[0], spp)
    # the atrium at 1080p: oracle crops + the whole frame against the default mode
    W, H, spp, bounces = 1920, 1080, 8, 8
    imgs = []
    for mode in (0, 1):
        r, params = _renderer(atrium, W, H, spp, bounces)
        r.set_option("transcendentals", mode)
        r.render(spp)
        imgs.append(r.read_accumulation()[0])
        r.close()
    sc, _ = oracle_scene_from_pt(atrium)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    cg, cr = [], []
    for (x0, y0) in [(928, 508), (64, 64), (1500, 300)]:
        ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x0 + 32, y0 + 32)
        assert np.array_equal(bits(imgs[0][y0:y0 + 32, x0:x0 + 32, :3]), bits(ref[y0:y0 + 32, x0:x0 + 32, :3]))
        cg.append(imgs[1][y0:y0 + 32, x0:x0 + 32, :3]), cr.append(ref[y0:y0 + 32, x0:x0 + 32, :3])
    report["atrium_1080p_8spp_crops_vs_oracle"] = _grade(np.concatenate(cg), np.concatenate(cr), spp)
    report["atrium_1080p_8spp_vs_default_mode"] = _grade(imgs[1], imgs[0], spp)
    print("f32 transcendentals:", report)
    out = os.environ.get("RF_F32_REPORT")
    if out:
        import json
        with open(out, "w") as f: json.dump(report, f, indent=1)
    for name, gr in report.items():
        assert gr["within"] >= 0.995, (name, gr)
        assert gr["mean_rel"] <= 1e-4, (name, gr)
        assert gr["nan_gpu"] == gr["nan_ref"], (name, gr)
    # ... and it IS another evaluation: the images are not bit-identical to the default mode's
    assert report["atrium_1080p_8spp_vs_default_mode"]["exact"] < 1.0


@pytest.mark.parametrize("shape", [(333, 217, 5, 0), (200, 150, 7, -1), (65, 33, 3, 0), (96, 64, 2, 3)])
def test_raygen_queue_positions_without_the_atomic_are_invisible(duck_pt, shape):
    """Round 6: kRaygen computes the first queue's positions in closed form (the valid pixels of the shard's tiles in local-pixel order: `validRankInTile`) instead of appending with
    one atomic per 1 024 slots.  Same rays, same image, same ray counts on ragged frames (right / bottom tiles partly outside), sample-major slots (-1) and -- where the slot order
    keeps pixel groups together (3) and the atomic append stays -- with the option on or off; shards included."""
    W, H, spp, g = shape
    out = []
    for dense in (0, 1):
        for (rank, world) in ((0, 1), (1, 3)):
            r, _ = _renderer(duck_pt, W, H, spp, 4)
            r.set_option("dense_raygen", dense)
            r.set_option("slot_group_shift", g)
            r.set_tile_shard(rank, world)
            r.render(spp)
            s = r.stats()
            out.append((dense, rank, world, r.read_accumulation()[0], s["primary_rays"], s["closest_rays"], s["shadow_rays"]))
            r.close()
    for a, b in ((out[0], out[2]), (out[1], out[3])):
        assert np.array_equal(bits(a[3]), bits(b[3])) and a[4:] == b[4:], (shape, a[1], a[2])
    assert out[0][4] == W * H * spp


@pytest.mark.parametrize("spp", [100, 300, 700])
def test_accumulation_of_deep_batches_in_sample_order(duck_pt, duck_oracle, spp):
    """kAccumulateRuns stages the runs of 4 / 2 / 1 pixels per workgroup in LDS depending on the samples per batch (round 6: <= 160 / <= 640 / more): the per-channel sums are the
    reference's sequential f32 additions in sample order (wgsl:55-57) whichever instantiation runs -- one batch of 100 / 300 / 700 samples against the oracle, bit for bit."""
    W, H, bounces = 40, 24, 2
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    s = r.stats()
    r.close()
    assert acc == spp and s["batch_samples_used"] == spp and s["batches_traced"] == 1
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.25, rf.aligned_sky_state(params.sky))
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))


def test_refill_threshold_is_invisible(atrium, duck_pt):
    """The refill threshold of the closest-hit launches is a scheduling choice: same image bit for bit from "a refill for every finished lane" (1) to "almost never" (63)."""
    for pt, (W, H, spp, bounces) in ((atrium, (480, 270, 4, 8)), (duck_pt, (200, 150, 4, 4))):
        r, _ = _renderer(pt, W, H, spp, bounces)
        r.render(spp)
        want = r.read_accumulation()[0]
        for refill in (22, 1, 8, 63, 40):
            r.set_option("refill_min_deep", refill)
            r.set_option("leaf_vote", {22: 20, 1: 64, 8: 1, 63: 14, 40: 33}[refill])     # (and the descend loop's exit vote: scheduling only)
            r.set_render_parameters(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, rf.make_sky(), 0.25 + 0.001 * refill))   # (exposure: restarts the accumulation)
            r.render(spp)
            assert np.array_equal(bits(r.read_accumulation()[0]), bits(want)), refill
        r.close()


def _local_world_gather(pt, W, H, spp, bounces, world, root, loopback=False):
    """N renderers + N communicators of the LOCAL test transport in this one process, one host thread per rank (ctypes releases the GIL): every rank renders
    its shard and runs the product's own rf_renderer_gather_frame; returns the root's gathered image and every rank's exchange time."""
    import threading
    uid = rf.comm_unique_id()
    out, errors = {}, []

    def rank_main(rank):
        try:
            r, _ = _renderer(pt, W, H, spp, bounces)
            r.set_tile_shard(rank, world)
            comm = rf.TileComm(uid, rank, world, 0)
            assert comm.local_transport() and comm.info()["rccl_ranks"] == world
            r.render(spp)
            ptr = r.gather_frame(comm, root=root, loopback=loopback)
            assert bool(ptr) == (rank == root)
            if rank == root:
                out["image"] = comm.read_frame(r, W, H)
            out[("ms", rank)] = comm.last_exchange_ms()
            out[("max", rank)] = comm.all_reduce_max(float(rank), r)   # (also a barrier: nobody tears its buffers down while a peer still copies from them)
            comm.close()
            r.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hangs in the exchange"
    assert all(out[("max", k)] == float(world - 1) for k in range(world))
    return out


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_local_transport_gather_with_many_owners(duck_pt, monkeypatch, world):
    """TileComm::gatherFrame + kUntile had only ever run with ONE owner (world 1).  RCCL refuses two ranks per device, so the transport -- and
    only the transport -- is swapped (RF_COMM_TRANSPORT=local: each ncclSend / ncclRecv pair of the plan becomes a device-to-device copy between the ranks'
    buffers, rf_comm.hip): N renderers render N shards, every rank runs the product's gather, and the image kUntile assembles on the root from its OWN shard
    (read in place) and the N - 1 staged ones must equal the single-rank image bit for bit -- for roots 0 and N - 1, a ragged frame, and loop-back on top."""
    monkeypatch.setenv("RF_COMM_TRANSPORT", "local")
    monkeypatch.setenv("RF_COMM_TIMEOUT_S", "120")
    for (W, H, spp, bounces) in ((200, 150, 3, 3), (333, 217, 2, 2)):   # both ragged at the right / bottom edge; 35 and 77 tiles
        whole, _ = _renderer(duck_pt, W, H, spp, bounces)
        whole.render(spp)
        want = whole.read_accumulation()[0]
        whole.close()
        for root in sorted({0, world - 1}):
            out = _local_world_gather(duck_pt, W, H, spp, bounces, world, root)
            assert np.array_equal(bits(out["image"]), bits(want)), (W, H, world, root)
            assert all(out[("ms", k)] >= 0.0 for k in range(world))
    out = _local_world_gather(duck_pt, 200, 150, 3, 3, world, world - 1, loopback=True)   # the root's own shard through the staging area as well
    whole, _ = _renderer(duck_pt, 200, 150, 3, 3)
    whole.render(3)
    assert np.array_equal(bits(out["image"]), bits(whole.read_accumulation()[0]))
    whole.close()


def test_local_transport_reports_a_missing_rank_instead_of_hanging(duck_pt, monkeypatch):
    """A world of 2 in which rank 1 never arrives: the root's receive gives up after RF_COMM_TIMEOUT_S with an error that names the rank."""
    monkeypatch.setenv("RF_COMM_TRANSPORT", "local")
    monkeypatch.setenv("RF_COMM_TIMEOUT_S", "2")
    r, _ = _renderer(duck_pt, 96, 64, 1, 1)
    r.set_tile_shard(0, 2)
    comm = rf.TileComm(rf.comm_unique_id(), 0, 2, 0)
    r.render(1)
    with pytest.raises(rf.RayfinderError, match="local transport: a peer's send did not happen within 2 s on rank 0"):
        r.gather_frame(comm, root=0)
    comm.close()
    r.close()


# ------------------------------------------------------------------ round 2: the analytic scenes on the GPU
def _pt_from_rects(rects):
    import analytic_scene as an
    P, N, T, I, tex = an.scene_arrays(rects)
    return rf.PtFormat.from_triangles(P, N, T, I, tex)


def test_analytic_corner_scene_gpu_vs_independent_integrator_and_oracle():
    """The two-walls-and-a-floor scene of tests/test_oracle_pins.py: the GPU's f32 sum image equals the oracle's bit for
    bit, and single samples agree with the independent float64 integrator (tests/analytic_scene.py) where its
    decisions are robust -- multi-bounce throughput order, NEE before the last-bounce break, unclamped cosine."""
    import analytic_scene as an
    from test_oracle_pins import corner_rects
    rects = corner_rects()
    pt = _pt_from_rects(rects)
    sc, _ = oracle_scene_from_pt(pt)
    W = H = 24
    cam = rf.create_camera((-1.5, 2.0, 2.0), (0.3, 0.0, -0.3), 0.0, 1.0, float(orc.degrees_to_radians(50.0)), 1.0)
    sky = rf.make_sky(1.0, (1.0, 1.0, 1.0), 30.0, 35.0)
    sky40 = rf.aligned_sky_state(sky)
    for bounces in (1, 2, 3):
        # one sample per pixel and spp = 1: the accumulation image IS the sample (n = frame % 1 = 0)
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, 1, bounces, sky, 1.0), pt.scene())
        r.render(1)
        img, _ = r.read_accumulation()
        r.close()
        ref, _ = orc.render(sc, orc.make_render_params(W, H, rf.camera_to_array(cam), 1, bounces, 1.0, sky40), 0, 1)
        assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
        checked = 0
        for y in range(H):
            for x in range(W):
                u = orc.animated_blue_noise(x, y, 0, 1).astype(np.float64)
                want, robust, _ = an.path_sample(rects, rf.camera_to_array(cam), sky40, W, H, x, y, u, bounces)
                if robust:
                    checked += 1
                    assert np.allclose(img[y, x, :3].astype(np.float64), want, rtol=3e-4, atol=2e-5 * max(1.0, float(np.abs(want).max()))), (x, y, bounces)
        assert checked > 0.8 * W * H


def test_sealed_room_is_exactly_black_on_the_gpu():
    """Closed white room, camera inside: every sun sample occluded, no path reaches the sky -> every pixel exactly 0
    after 6 bounces x 8 spp; any traversal leak (missed far child, wrong offset side, dropped stack entry) would light one."""
    from test_oracle_pins import box_rects
    pt = _pt_from_rects(box_rects(True))
    W, H, spp, bounces = 96, 64, 8, 6
    cam = rf.create_camera((0.2, -0.3, 0.4), (0.9, 0.2, -0.8), 0.0, 1.0, float(orc.degrees_to_radians(80.0)), W / H)
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, bounces, rf.make_sky(), 1.0), pt.scene())
    r.render(spp)
    img, _ = r.read_accumulation()
    s = r.stats()
    r.close()
    assert not img[..., :3].any()
    assert s["closest_rays"] == W * H * spp * bounces == s["shadow_rays"]


def test_accumulation_restart_keeps_frame_count(duck_pt, duck_oracle):
    """setRenderParameters() with a change restarts the accumulation but frameCount keeps counting
    (reference_path_tracer.cpp:556-563,577-591): the restarted image holds samples n = frameCount % spp in a rotated
    order -- what bench.py's timed region is, and what its parity crop compares with."""
    W, H, spp, bounces = 96, 64, 6, 3
    r, params = _renderer(duck_pt, W, H, spp, bounces)
    r.render(4)                                                   # frames 0..3
    r.set_render_parameters(rf.make_render_parameters(W, H, params.camera, spp, bounces, params.sky, 0.5))
    r.render(spp + 2)                                             # frames 4..9 accumulate, 10..11 only advance frameCount
    img, acc = r.read_accumulation()
    assert acc == spp
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), spp, bounces, 0.5, rf.aligned_sky_state(params.sky))
    ref, _ = orc.render(duck_oracle.scene, rp, 4, spp + 2, accumulated_start=0)
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
    r.close()


# ------------------------------------------------------------------ deferred-lighting variant
def test_deferred_lighting_variant_vs_oracle(duck_pt, duck_oracle):
    """Lighting pass (deferred_renderer_lighting_pass.wgsl:96-186: fixed 2 bounces, solar disk in the sky term, the
    1/16384 + 1024 offsets) + exponential resolve (deferred_renderer_resolve_pass.wgsl:33-54) over a primary-ray
    G-buffer: sample buffer, accumulation buffer and BGRA output == the oracle's restatement, frame after frame."""
    W, H = 136, 96
    r, params = _renderer(duck_pt, W, H, 1, 2, exposure=0.25)
    rp = orc.make_render_params(W, H, rf.camera_to_array(params.camera), 1, 2, 0.25, rf.aligned_sky_state(params.sky))
    for frames in (1, 4):
        r.reset_deferred()
        r.render_deferred(frames)
        sample, accum, bgra, n = r.read_deferred()
        assert n == frames
        ws, wa, wsrgb, st = orc.deferred_frames(duck_oracle.scene, rp, frames)
        assert np.array_equal(bits(sample), bits(ws))
        assert np.array_equal(bits(accum), bits(wa))
        want = np.floor(wsrgb * np.float32(255.0) + np.float32(0.5)).astype(np.int64)
        got = np.stack([(bgra >> 16) & 255, (bgra >> 8) & 255, bgra & 255], axis=-1).astype(np.int64)
        assert np.array_equal(got, want)
    # frame 0 initialises the accumulation; later frames move it by a tenth of the difference
    assert np.isfinite(accum).all() and (sample[..., 2] > 0).any()
    # sky pixels that look at the sun carry the solar disk term
    s = r.stats()
    assert s["shadow_rays"] > 0
    r.close()


# ------------------------------------------------------------------ round 2: a Sponza-shaped asset end to end
def test_courtyard_asset_bake_render_and_bench(tmp_path):
    """tools/make_test_asset.py (multi-mesh, multi-material .gltf with external PNG / progressive-JPEG URIs, node TRS):
    rf-pt-format-tool with the host builder and with --gpu-bvh write the same .pt; the render of it equals the oracle's
    bit for bit; bench.py --scene runs on it."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_test_asset
    path = make_test_asset.write_courtyard(str(tmp_path / "courtyard"))
    tool = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-pt-format-tool")
    subprocess.check_call([tool, path], stdout=subprocess.DEVNULL)
    host = open(path.replace(".gltf", ".pt"), "rb").read()
    subprocess.check_call([tool, "--gpu-bvh", path], stdout=subprocess.DEVNULL)
    gpu = open(path.replace(".gltf", ".pt"), "rb").read()
    assert host == gpu
    pt = rf.PtFormat.load(path.replace(".gltf", ".pt"))
    sc, _ = oracle_scene_from_pt(pt)
    W, H, spp, bounces = 160, 96, 4, 4
    cam = rf.create_camera((4.0, 2.5, 4.5), (1.0, 0.8, -1.0), 0.0, 1.0, float(orc.degrees_to_radians(55.0)), W / H)
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, bounces, rf.make_sky(), 0.25), pt.scene())
    r.render(spp)
    img, _ = r.read_accumulation()
    s = r.stats()
    r.close()
    ref, st = orc.render(sc, orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(rf.make_sky())), 0, spp)
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
    assert s["closest_rays"] == st.closestRays and s["shadow_rays"] == st.shadowRays > 0
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", path.replace(".gltf", ".pt"), "--width", "320", "--height", "192",
                                   "--steps", "1", "--warmup", "1", "--bounces", "4", "--cpu-seconds", "1", "--no-live-counters"], stderr=subprocess.DEVNULL)
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["value"] > 0 and line["nan_pixels"] == 0 and line["parity_crop"]["verdict"] == "bit-identical"
    assert "courtyard.pt" in line["config"]["workload"]


# ------------------------------------------------------------------ round 2: scheduling choices never change a pixel
def test_slot_order_batching_and_accumulation_variants_give_identical_images(duck_pt, duck_oracle):
    """Path-slot order (sample-major / pixel groups of 1, 4, 64 pixels), direction-sorted samples, the LDS accumulation
    kernel, kShade's grid cap and the batch size are scheduling decisions: the f32 sum image is the oracle's, bit for
    bit, under every combination (odd frame size, spp that is not a multiple of anything, several batches)."""
    W, H, spp, bounces = 150, 90, 23, 4
    cam = rf.fly_camera(W, H)
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(rf.make_sky()))
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    NQ = dict(quad_from_bounce=0, quad_shadow_from_bounce=0)      # without the quad records (the default since round 3) the other layouts run
    variants = [dict(), dict(slot_group_shift=-1), dict(slot_group_shift=2), dict(slot_group_shift=6, sample_sort=0), dict(sample_sort=0),
                dict(accumulate_runs=0), dict(shade_blocks=7), dict(slot_group_shift=10, accumulate_runs=0),
                dict(NQ, compact_from_bounce=0, compact_shadow_from_bounce=0), dict(NQ, compact_from_bounce=1, compact_shadow_from_bounce=1),
                dict(NQ, compact_from_bounce=1, compact_shadow_from_bounce=1, uniform_fetch=0), dict(NQ),
                dict(NQ, hot_from_bounce=1, hot_shadow_from_bounce=1), dict(NQ, hot_from_bounce=2, hot_shadow_from_bounce=1, uniform_fetch=0),
                dict(quad_from_bounce=2, quad_shadow_from_bounce=1, uniform_fetch=0), dict(quad_from_bounce=1, quad_shadow_from_bounce=3, quad_except_mask=2, quad_shadow_except_mask=8),
                dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0), dict(quad_half_from_bounce=2, quad_half_shadow_from_bounce=1, uniform_fetch=0),
                dict(quad_half_from_bounce=3, quad_half_shadow_from_bounce=2, shade_sort_from_bounce=0), dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=1),
                dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=1, quad_local_shadow_from_bounce=1),
                dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=2, quad_local_shadow_from_bounce=1, uniform_fetch=0, shadow_sign_order=0),
                dict(shade_sort_from_bounce=1), dict(shade_sort_from_bounce=2, shade_blocks=5), dict(shade_sort_from_bounce=0)]
    for opts in variants:
        for max_paths in (0, 5 * 15 * 1024):                  # default batch (all 23 samples at once) / 5 samples per batch -> 5, 5, 5, 4, 4
            r, _ = _renderer(duck_pt, W, H, spp, bounces, cam=cam, max_paths_in_flight=max_paths)
            for k, v in opts.items():
                r.set_option(k, v)
            r.render(spp)
            img, acc = r.read_accumulation()
            r.close()
            assert acc == spp
            assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3])), (opts, max_paths)


@pytest.mark.parametrize("packet_bounces", [1, 99])
def test_lockstep_packet_traversal_gives_identical_images(duck_pt, duck_oracle, packet_bounces):
    """kTracePacket (one wave = one packet walking the tree in lockstep: shared stack, scalar-cache fetches, per-lane rayTMax and
    active bit) is a scheduling choice too: the image is the oracle's bit for bit, at bounce 1 only (where its waves are one
    pixel's samples) and at every bounce (incoherent packets, mixed direction signs: several passes per wave); a tree deeper
    than the packet's 24-entry shared stack hands its rays to the scalar traversal."""
    W, H, spp, bounces = 150, 90, 23, 4
    cam = rf.fly_camera(W, H)
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(rf.make_sky()))
    ref, _ = orc.render(duck_oracle.scene, rp, 0, spp)
    r, _ = _renderer(duck_pt, W, H, spp, bounces, cam=cam)
    r.set_option("packet_bounces", packet_bounces)
    r.render(spp)
    img, acc = r.read_accumulation()
    r.close()
    assert acc == spp
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
    # the 60-deep chain: every far child of the chain is pending at once
    nodes, tris, attrs = _chain_scene(60)
    sc = rf.scene_from_arrays(nodes, tris, attrs, [(np.array([0xFFFFFFFF], np.uint32), 1, 1)])
    cam2 = rf.create_camera((0.2, 0.2, -5.0), (0.2, 0.2, 0.0), 0.0, 1.0, np.radians(20.0), 1.0)
    images = []
    for pk in (0, packet_bounces):
        r = rf.ReferencePathTracer(rf.make_render_parameters(64, 64, cam2, 2, 2, rf.make_sky(), 1.0), sc)
        r.set_option("packet_bounces", pk)
        r.render(2)
        images.append(r.read_accumulation()[0])
        s = r.stats()
        assert s["abandoned_rays"] == 0 and s["scalar_redo_rays"] > 0
        r.close()
    assert np.array_equal(bits(images[0]), bits(images[1]))


def test_rf_render_cli_four_ranks_on_one_gpu_through_the_local_transport(duck_pt, tmp_path):
    """`rf-render --gpus N` (one host thread per rank: renderer, tile shard, communicator, frame-end gather, the root's tonemap of the gathered image) had never run with N > 1: RCCL
    refuses two ranks per device.  Under RF_COMM_TRANSPORT=local the same binary runs its 4 (and 3) ranks on the one GPU there is; PNG and PFM must equal the --gpus 1 files byte for byte."""
    import subprocess
    from conftest import ROOT
    scene = tmp_path / "Duck.pt"
    duck_pt.save(scene)
    exe = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-render")
    W, H, spp, bounces = 200, 150, 4, 3                  # ragged right / bottom tiles
    outs = {}
    for gpus in (1, 4, 3):
        png, pfm = tmp_path / f"g{gpus}.png", tmp_path / f"g{gpus}.pfm"
        env = dict(os.environ, RF_COMM_TRANSPORT="local", RF_COMM_TIMEOUT_S="120")
        txt = subprocess.check_output([exe, str(scene), "--width", str(W), "--height", str(H), "--spp", str(spp), "--bounces", str(bounces), "--out", str(png), "--pfm", str(pfm),
                                       "--gpus", str(gpus)], env=env, timeout=300).decode()
        assert f"on {gpus} GPU(s)" in txt
        outs[gpus] = (open(png, "rb").read(), open(pfm, "rb").read(), txt.split("(")[-1])
    for gpus in (4, 3):
        assert outs[gpus][0] == outs[1][0], f"PNG of --gpus {gpus} differs"
        assert outs[gpus][1] == outs[1][1], f"PFM of --gpus {gpus} differs"
        assert outs[gpus][2] == outs[1][2], (outs[gpus][2], outs[1][2])            # the same rays in total


def test_rf_render_cli_writes_the_tonemapped_image(duck_pt, tmp_path):
    """rf-render (the offline twin of the `pt` app, --gpus 1 path): its PNG holds exactly rf_renderer_read_tonemapped's texels."""
    import subprocess
    import zlib
    from conftest import ROOT
    scene = tmp_path / "Duck.pt"
    duck_pt.save(scene)
    out = tmp_path / "duck.png"
    W, H, spp, bounces = 96, 64, 4, 3
    txt = subprocess.check_output([os.path.join(ROOT, "rayfinder_amd", "bin", "rf-render"), str(scene), "--width", str(W), "--height", str(H), "--spp", str(spp),
                                   "--bounces", str(bounces), "--out", str(out), "--gpus", "1"]).decode()
    assert f"{W}x{H}, {spp} spp, {bounces} bounces on 1 GPU(s)" in txt
    data = open(out, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    idat = b""
    off = 8
    while off < len(data):
        n = int.from_bytes(data[off:off + 4], "big"); typ = data[off + 4:off + 8]
        if typ == b"IDAT":
            idat += data[off + 8:off + 8 + n]
        off += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + 4 * W)
    assert (raw[:, 0] == 0).all()
    rgba = raw[:, 1:].reshape(H, W, 4)
    r, _ = _renderer(duck_pt, W, H, spp, bounces)       # rf-render's defaults: fly camera, default sky, exposure 2 stops = 0.25
    r.render(spp)
    bgra = r.read_tonemapped()
    r.close()
    want = np.stack([(bgra >> 16) & 255, (bgra >> 8) & 255, bgra & 255, bgra >> 24], -1).astype(np.uint8)
    assert np.array_equal(rgba, want)


# ---------------------------------------------------------------- bvh-visualizer: the file the tool writes (VERDICT r2 missing 2)
def test_bvh_visualizer_tool_png_is_the_grey_map_of_the_node_visit_pass(duck_pt, tmp_path):
    """rf-bvh-visualizer (GPU mode) writes grey = u32(min(0.01 * nodesVisited, 1) * 255), alpha 255
    (src/bvh-visualizer/main.cpp:73-84) of trace_primary_stats' node-visit map; --cpu writes the SAME bytes from the host pass;
    GPU pass == host pass (rf_bvh_visualizer_pass) on every output."""
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-bvh-visualizer")
    a = duck_pt.arrays()
    pt_path = tmp_path / "Duck.pt"
    duck_pt.save(pt_path)
    r0, _ = _renderer(duck_pt, 64, 64, 1, 1)
    for (w, h, args) in ((1280, 720, []), (256, 256, ["256", "256"]), (333, 77, ["333", "77"])):
        cam = rf.bvh_visualizer_camera(a["bvhNodes"], np.float32(np.float32(w) / np.float32(h)))
        gpu = r0.trace_primary_stats(cam, w, h)
        host = rf.bvh_visualizer_pass(cam, w, h, a["bvhNodes"], a["bvhPositionAttributes"])
        for k in ("nodesVisited", "hit", "triTests"):
            assert np.array_equal(gpu[k], host[k]), k
        assert np.array_equal(bits(gpu["t"]), bits(host["t"]))
        files = []
        for mode in ([], ["--cpu"]):
            out = tmp_path / f"viz_{w}_{len(mode)}.png"
            p = subprocess.run([exe] + mode + ["--out", str(out), str(pt_path)] + args, capture_output=True, text=True, cwd=tmp_path)
            assert p.returncode == 0, p.stderr
            files.append(out.read_bytes())
            img = np.array(Image.open(out))
            assert img.shape == (h, w, 4) and (img[..., 3] == 255).all()
            grey = rf.bvh_visualizer_grey(gpu["nodesVisited"]).reshape(h, w)
            assert np.array_equal(img[..., 0], grey) and np.array_equal(img[..., 1], grey) and np.array_equal(img[..., 2], grey)
        assert files[0] == files[1]


# ------------------------------------------------------------------ round 4: bench.py as its own launcher; product inputs of the checker
def _bench_line(extra, tmp_path):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    duck = str(tmp_path / "Duck.pt")
    if not os.path.exists(duck):
        rf.PtFormat.from_gltf(DUCK).save(duck)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", duck, "--width", "320", "--height", "192", "--steps", "2", "--warmup", "1",
                        "--bounces", "4", "--cpu-seconds", "1", "--repeat", "3"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout"
    return json.loads(lines[0]), p.stderr.decode()


def test_bench_line_shape_is_the_same_however_the_rank_was_started(tmp_path):
    """`python bench.py --gpus 1`, the same through bench.py's own launcher (torch.distributed.run --standalone, what
    `--gpus N > 1` does by itself), and one rank through the whole N > 1 plumbing (RCCL process group, the product's
    communicator, frame-end exchange to itself, device un-tile): one JSON line each, same keys, n_gpus = 1, the same rays,
    and the median-of-3 bookkeeping."""
    plain, _ = _bench_line(["--gpus", "1"], tmp_path)                         # (with the live counter passes: the default)
    launched, err = _bench_line(["--gpus", "1", "--launch", "--no-live-counters"], tmp_path)
    assert "launching 1 ranks" in err
    looped, _ = _bench_line(["--gpus", "1", "--launch", "--exchange-at-world-1"], tmp_path)
    # the HBM figures of a workload without a committed counter profile come from this run's own rocprofv3 passes (8 = 2 steps' one batch... bounces = 4 launches)
    tl = plain["roofline"].get("traffic_live")
    assert tl and tl["live"] and tl["launches_profiled"] == 4 and plain["roofline"]["traffic"] > 0 and 0 < plain["roofline"]["frac"] < 1
    assert "traffic_live" not in launched["roofline"] and launched["roofline"]["traffic"] is None
    for line in (plain, launched, looped):
        assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["unit"] == "Mrays/s" and line["nan_pixels"] == 0
        rep = line["repeats"]
        assert rep["count"] == 3 and len(rep["value"]) == 3 and rep["min"] <= line["value"] <= rep["max"]
        assert abs(line["value"] - sorted(rep["value"])[1]) <= 0.11
        assert abs(line["ms_per_step"] * 2 - sorted(rep["timed_region_s"])[1] * 1e3) < 0.11      # (the list is rounded to 0.1 ms)
        assert line["parity_crop"]["verdict"] == "bit-identical"            # the LAST repeat's frame against the oracle
    assert set(plain) == set(launched) == set(looped)            # (top-level keys; the roofline block differs by `traffic_live`)
    assert plain["rays"] == launched["rays"] == looped["rays"]
    assert plain["rccl_ranks"] == 0 and plain["exchange"] == "none"
    assert looped["rccl_ranks"] == 1 and "C++ RCCL exchange" in looped["exchange"] and "FALLBACK" not in looped["exchange"]
    # the exchange is priced in every line that runs it (round 5): HIP events around the rank's sends / receives + un-tile, one figure per repeat
    assert plain["exchange_ms"] is None and launched["exchange_ms"] is None
    ex = looped["exchange_ms"]
    assert len(ex["per_repeat_max_over_ranks"]) == 3 and all(0.0 < v < 1000.0 for v in ex["per_repeat_max_over_ranks"]) and ex["median"] == sorted(ex["per_repeat_max_over_ranks"])[1]


def test_product_computed_checker_inputs_equal_the_oracles_own():
    """The parity tests above give the oracle sky states, cameras and baked arrays the PRODUCT computed (rf.aligned_sky_state,
    rf.fly_camera / create_camera, oracle_scene_from_pt): here each of them is compared with the oracle's own computation -- and the
    sky with the reference-compiled vectors -- inside the driver-run suite (tests/checker_inputs.py; the CPU suite runs the same)."""
    import checker_inputs
    assert checker_inputs.check_sky_states() == (30, 144)
    assert checker_inputs.check_cameras() >= 50
    nodes, info = checker_inputs.check_atrium_bake()
    assert nodes == 459645 and info["triangles"] == 265024
    # the other scenes the GPU tests bake with the product and hand to the oracle: Duck (glTF ingest + builder) and the courtyard asset
    d = DuckOracle()
    a = rf.PtFormat.from_gltf(DUCK).arrays()
    assert a["bvhNodes"].tobytes() == d.nodes.tobytes() and a["trianglePositionAttributes"].tobytes() == d.pos48.tobytes()
    assert a["triangleVertexAttributes"].tobytes() == np.ascontiguousarray(d.attr80).tobytes()
    px, w, h = a["baseColorTextures"][0]
    assert (w, h) == tuple(int(x) for x in d.descs[0][:2]) and np.array_equal(px, d.texels[:px.size])


def test_leaves_that_share_a_first_triangle_keep_the_exact_record_layouts(duck_pt, duck_oracle):
    """A hand-made .pt may point two leaves at the same triangle (validateScene checks ranges, not disjointness).  The
    half-precision / local-grid kernels cull a leaf by the exact box kept in its FIRST triangle's record, which can hold only one
    box: for such a scene those layouts stay off (rf_wide.hpp, leafBoxesIntoTriangles), whatever the options say, and the image is
    the reference-ordered oracle's on every layout that is left."""
    a = duck_pt.arrays()
    nodes = a["bvhNodes"].copy()
    leaves = np.nonzero(nodes["triangleCount"] == 1)[0]
    rng = np.random.default_rng(3)
    for k in rng.choice(len(leaves) - 1, 400, replace=False):       # 400 leaves now test a triangle that lies in ANOTHER leaf's box
        nodes[leaves[k + 1]]["trianglesOffset"] = nodes[leaves[k]]["trianglesOffset"]
    tex = [(px, w, h) for (px, w, h) in a["baseColorTextures"]]
    sc = rf.scene_from_arrays(nodes, a["trianglePositionAttributes"], a["triangleVertexAttributes"], tex)
    W, H, spp, bounces = 160, 120, 3, 4
    cam = rf.fly_camera(W, H)
    osc = orc.OracleScene(nodes, a["trianglePositionAttributes"], a["triangleVertexAttributes"], duck_oracle.descs, duck_oracle.texels)
    ref, _ = orc.render(osc, orc.make_render_params(W, H, orc.default_pt_camera(W, H), spp, bounces, 0.25, orc.aligned_sky_state()), 0, spp)
    for opts in (dict(), dict(quad_half_from_bounce=1, quad_half_shadow_from_bounce=1), dict(quad_local_from_bounce=1, quad_local_shadow_from_bounce=1),
                 dict(quad_from_bounce=0, quad_shadow_from_bounce=0)):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, spp, bounces, rf.make_sky(), 0.25), sc)
        for k, v in opts.items():
            r.set_option(k, v)
        r.render(spp)
        img, _ = r.read_accumulation()
        r.close()
        assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3])), opts


def test_clutter_atrium_crops_vs_oracle_and_its_traversal_statistics():
    """The harder stand-in (scenes.atrium(detail="clutter"): cloth, displaced spheres, chains, diagonal cables, plants of overlapping leaves) under every
    record layout the renderer may pick for it: crops bit-identical to the oracle, and the counting build reproduces the oracle's node visits and triangle
    tests -- 78 visits and 7.5 triangle tests per closest-hit ray against the plain atrium's 62 / 3.0."""
    from rayfinder_amd import scenes
    pt, info = scenes.atrium(1, "clutter")
    sc, _ = oracle_scene_from_pt(pt)
    W, H, spp, bounces = 480, 270, 4, 8
    crops = [(200, 100, 264, 148), (0, 0, 64, 32), (400, 200, 480, 270)]
    rp = orc.make_render_params(W, H, orc.default_pt_camera(W, H), spp, bounces, 0.25, orc.aligned_sky_state())
    for opts in (dict(), dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=1, quad_local_shadow_from_bounce=1),
                 dict(quad_half_from_bounce=0, quad_half_shadow_from_bounce=0, quad_local_from_bounce=0, quad_local_shadow_from_bounce=0)):
        r, _ = _renderer(pt, W, H, spp, bounces)
        for k, v in opts.items():
            r.set_option(k, v)
        r.render(spp)
        img, _ = r.read_accumulation()
        s = r.stats()
        r.close()
        assert s["abandoned_rays"] == 0
        for (x0, y0, x1, y1) in crops:
            ref, _ = orc.render(sc, rp, 0, spp, x0, y0, x1, y1)
            assert np.array_equal(bits(img[y0:y1, x0:x1, :3]), bits(ref[y0:y1, x0:x1, :3])), (opts, x0, y0)
    # the reference's visit counts on a frame small enough for the oracle: counting build == oracle
    w2, h2 = 160, 90
    r, _ = _renderer(pt, w2, h2, 2, bounces)
    r.set_counting(True)
    r.render(2)
    s = r.stats()
    r.close()
    _, st = orc.render(sc, orc.make_render_params(w2, h2, orc.default_pt_camera(w2, h2), 2, bounces, 0.25, orc.aligned_sky_state()), 0, 2)
    assert s["closest_rays"] == st.closestRays and s["closest_node_visits"] == st.closestNodeVisits and s["closest_triangle_tests"] == st.closestTriTests
    assert s["shadow_rays"] == st.shadowRays
    assert st.closestNodeVisits / st.closestRays > 70 and st.closestTriTests / st.closestRays > 5.0


@pytest.mark.parametrize("distance", [10.0, 1000.0])
def test_camera_far_outside_the_scene_keeps_primary_rays_off_the_scalar_path(duck_pt, duck_oracle, distance):
    """A camera beyond the conservative records' origin bound (4 R + 1): its primary launch reads the exact quad records instead of sending every ray to
    the scalar traversal (round 4: that launch took twice as long); later bounces start on surfaces and keep the 64-byte records.  Bit-identical either way."""
    a = duck_pt.arrays()
    lo, hi = np.array(a["bvhNodes"][0]["min"][:3]), np.array(a["bvhNodes"][0]["max"][:3])
    centre, size = 0.5 * (lo + hi), float(np.max(hi - lo))
    W, H, spp, bounces = 320, 240, 4, 4
    eye = centre + np.array([0.6, 0.4, 0.7]) / np.linalg.norm([0.6, 0.4, 0.7]) * distance * size
    cam = rf.create_camera(eye, centre, 0.0, 1.0, float(2.0 * np.arctan(0.75 / distance)), W / H)
    r, _ = _renderer(duck_pt, W, H, spp, bounces, cam=cam)
    r.render(spp)
    img, _ = r.read_accumulation()
    s = r.stats()
    r.close()
    assert s["scalar_redo_rays"] < 0.01 * s["primary_rays"]
    ref, st = orc.render(duck_oracle.scene, orc.make_render_params(W, H, orc.create_camera(eye, centre, 0.0, 1.0, float(2.0 * np.arctan(0.75 / distance)), W / H), spp, bounces, 0.25,
                                                                   orc.aligned_sky_state()), 0, spp)
    assert s["closest_rays"] == st.closestRays and st.closestRays > W * H * spp      # the duck is in frame: paths bounce
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
