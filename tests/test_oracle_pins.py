"""Pins the CPU oracle (oracle/rf_oracle.c) before anything trusts it.

1. The reference's own unit tests for this path, replayed against the oracle:
   src/tests/aabb.cpp:8-132, src/tests/intersection.cpp:9-28, src/tests/bvh.cpp:34-102.
2. Outputs of the reference's own hw_skymodel.c (compiled unmodified into oracle/_ref) -- live when
   the .so is present, and through the committed vectors tests/golden/sky_ref.npz always.
3. The probe numbers SURVEY.md 8(c) recorded from the reference's C++ (nodes, leaves, node-visit
   sums): reproduced exactly.
"""
import hashlib
import struct
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits
from oracle import orc

FLT_MAX = np.finfo(np.float32).max


def _f(*v):
    return np.array(v, np.float32)


# ---------------------------------------------------------------- src/tests/aabb.cpp
def test_default_aabb_merge_with_point_yields_point():
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    orc.lib().orc_aabb_merge_point(orc._p(_f(FLT_MAX, FLT_MAX, FLT_MAX)), orc._p(_f(-FLT_MAX, -FLT_MAX, -FLT_MAX)), orc._p(_f(0, 0, 0)), orc._p(lo), orc._p(hi))
    assert np.array_equal(lo, _f(0, 0, 0)) and np.array_equal(hi, _f(0, 0, 0))


def test_default_aabb_merge_with_aabb_yields_that_aabb():
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    orc.lib().orc_aabb_merge(orc._p(_f(FLT_MAX, FLT_MAX, FLT_MAX)), orc._p(_f(-FLT_MAX, -FLT_MAX, -FLT_MAX)), orc._p(_f(-1, -1, -1)), orc._p(_f(1, 1, 1)), orc._p(lo), orc._p(hi))
    assert np.array_equal(lo, _f(-1, -1, -1)) and np.array_equal(hi, _f(1, 1, 1))


def test_max_dimension_ties_return_z_and_largest_otherwise():
    assert orc.lib().orc_aabb_max_dimension(orc._p(_f(-1, -1, -1)), orc._p(_f(1, 1, 1))) == 2
    assert orc.lib().orc_aabb_max_dimension(orc._p(_f(-3, -2, -1)), orc._p(_f(1, 1, 1))) == 0


def test_surface_area_of_2_cube_is_24():
    assert orc.lib().orc_aabb_surface_area(orc._p(_f(-1, -1, -1)), orc._p(_f(1, 1, 1))) == pytest.approx(24.0)


@pytest.mark.parametrize("origin,direction,lo,hi,expected", [
    ((-2, 0, 0), (1, 0, 0), (-1, -1, -1), (1, 1, 1), True),     # x slab (axis-parallel: inf invDir on y,z)
    ((0, -1, 0), (0, 1, 0), (-1, 0, -1), (1, 1, 1), True),      # y slab
    ((0, 0, -1), (0, 0, 1), (-1, -1, 0), (1, 1, 1), True),      # z slab
    ((-1, -1, -1), (1, 1, 1), (-1, -1, -1), (1, 1, 1), True),   # corner graze
    ((-2, 0, -1), (0, 1, 0), (-1, -1, -1), (1, 1, 1), False),   # miss
])
def test_ray_aabb_cases(origin, direction, lo, hi, expected):
    with np.errstate(all="ignore"):
        got = orc.lib().orc_ray_intersect_aabb(orc._p(_f(*origin)), orc._p(_f(*direction)), orc._p(_f(*lo)), orc._p(_f(*hi)), 100.0)
    assert bool(got) == expected


# ---------------------------------------------------------------- src/tests/intersection.cpp
def test_ray_intersects_triangle_known_answer():
    p = np.zeros(3, np.float32)
    t = np.zeros(1, np.float32)
    tri = _f(0, 0, 1, 1, 0, 1, 0, 1, 1)
    hit = orc.lib().orc_ray_intersect_triangle(orc._p(_f(0, 0, 0)), orc._p(_f(0, 0, 1)), orc._p(tri), 1000.0, orc._p(p), orc._p(t))
    assert hit
    assert abs(p[0]) < 1e-3 and abs(p[1]) < 1e-3 and p[2] == pytest.approx(1.0, rel=1e-3)
    assert t[0] == pytest.approx(1.0)


# ---------------------------------------------------------------- src/tests/bvh.cpp
def test_bvh_intersection_matches_brute_force_on_duck(duck_oracle):
    d = duck_oracle
    assert len(d.nodes) > 0 and len(d.idx) > 0
    cam = orc.bvh_test_camera(d.tris36)
    mism, hits = 0, 0
    for i in range(64):
        u = np.float32(i) / np.float32(64)
        for j in range(64):
            v = np.float32(j) / np.float32(64)
            ray = orc.generate_camera_ray(cam, u, v)
            did, t = orc.brute_force(d.tris36, ray, 1000.0)
            r = orc.intersect_bvh_batch(d.nodes, d.tris36, ray[None, :], 1000.0)
            assert bool(r["hit"][0]) == did
            if did:
                hits += 1
                assert r["t"][0] == pytest.approx(t)
                mism += int(bits(r["t"][:1])[0] != bits(np.array([t]))[0])
    assert hits == 1216
    assert mism == 0         # t is even bit-identical


# ---------------------------------------------------------------- golden: Duck
def test_duck_golden_and_survey_probe_numbers(duck_oracle):
    d = duck_oracle
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    assert len(d.nodes) == 8383 == int(g["num_nodes"])
    assert hashlib.sha256(d.nodes.tobytes()).hexdigest() == str(g["nodes_sha256"])
    leaves = d.nodes[d.nodes["triangleCount"] > 0]
    assert len(leaves) == 4192
    assert list(np.bincount(leaves["triangleCount"])) == [0, 4182, 1, 8, 1]
    assert d.nodes[0]["secondChildOffset"] == 4006 and d.nodes[0]["splitAxis"] == 0
    assert np.allclose(d.nodes[0]["min"], [-0.692985, 0.0992937, -0.613282], rtol=1e-6)
    assert np.allclose(d.nodes[0]["max"], [0.961799, 1.6397, 0.539252], rtol=1e-6)
    # leaf/interior encoding of bvh.cpp:31-55; pads are zero (byte-deterministic nodes)
    inner = d.nodes[d.nodes["triangleCount"] == 0]
    assert (leaves["splitAxis"] == 0xFFFFFFFF).all() and (leaves["secondChildOffset"] == 0).all()
    assert (inner["trianglesOffset"] == 0).all() and (inner["splitAxis"] <= 2).all()
    assert (d.nodes["pad0"] == 0).all() and (d.nodes["pad1"] == 0).all()

    cam = orc.bvh_visualizer_camera(d.nodes, np.float32(1.0))
    viz = orc.bvh_visualize(d.nodes, d.tris36, cam, 256, 256)
    assert np.array_equal(viz["nodesVisited"], g["viz256_nodes_visited"].astype(np.uint32))
    assert int(viz["hit"].sum()) == 19462
    assert int(viz["nodesVisited"].sum()) == 1209382 and int(viz["nodesVisited"].max()) == 157
    assert int(viz["triTests"].sum()) == 69097 and int(viz["stackHigh"].max()) == 12

    cam = orc.bvh_visualizer_camera(d.nodes, np.float32(np.float32(1280) / np.float32(720)))
    viz = orc.bvh_visualize(d.nodes, d.tris36, cam, 1280, 720)
    assert int(viz["hit"].sum()) == 154034
    assert int(viz["nodesVisited"].sum()) == 9979946 and int(viz["nodesVisited"].max()) == 159
    assert np.array_equal(viz["nodesVisited"].reshape(720, 1280).sum(axis=1), g["viz720_row_sums"])


def test_cpu_positions36_and_gpu_positions48_traverse_identically(duck_oracle):
    d = duck_oracle
    cam = orc.bvh_visualizer_camera(d.nodes, np.float32(1.0))
    a = orc.bvh_visualize(d.nodes, d.tris36, cam, 64, 64)
    b = orc.bvh_visualize(d.nodes, d.pos48, cam, 64, 64)
    assert np.array_equal(a["nodesVisited"], b["nodesVisited"]) and np.array_equal(bits(a["t"]), bits(b["t"]))


def test_bvh_visualizer_grey_value():
    # main.cpp:73-76: p = u32(min(0.01*nodesVisited, 1) * 255)
    assert orc.lib().orc_bvh_visualizer_pixel(0) == 0xFF000000
    assert orc.lib().orc_bvh_visualizer_pixel(50) & 0xFF == 127
    assert orc.lib().orc_bvh_visualizer_pixel(157) & 0xFF == 255


# ---------------------------------------------------------------- sky model vs the reference's C
def _check_sky(state_new, radiance):
    g = np.load(os.path.join(GOLDEN, "sky_ref.npz"))
    for case, want, rc in zip(g["cases"], g["states"], g["rcs"]):
        got_rc, got = state_new(case[0], case[1], case[2:5])
        assert got_rc == rc == 0
        assert np.array_equal(bits(got), bits(want)), case
    for case, rc in zip(g["bad"], g["bad_rc"]):
        assert state_new(case[0], case[1], case[2:5])[0] == rc != 0
    for i, th, gm, ch, want in g["samples"]:
        got = radiance(g["states"][int(i)], np.float32(th), np.float32(gm), int(ch))
        assert bits(np.array([got]))[0] == bits(np.array([want], np.float32))[0]


def test_oracle_sky_matches_reference_vectors():
    _check_sky(orc.sky_state_new, orc.sky_state_radiance)


@pytest.mark.skipif(not orc.RefSky.available(), reason="oracle/_ref not built (reference mount absent)")
def test_oracle_sky_matches_live_reference_build():
    ref = orc.RefSky()
    rng = np.random.default_rng(7)
    for _ in range(200):
        elev = np.float32(rng.uniform(0, 1.5707)); turb = np.float32(rng.uniform(1, 10)); alb = rng.uniform(0, 1, 3).astype(np.float32)
        rc1, s1 = orc.sky_state_new(elev, turb, alb)
        rc2, s2 = ref.state_new(elev, turb, alb)
        assert rc1 == rc2 and np.array_equal(bits(s1), bits(s2))
        th = np.float32(rng.uniform(0, 1.57)); gm = np.float32(rng.uniform(0, 3.14)); ch = int(rng.integers(0, 3))
        assert bits(np.array([orc.sky_state_radiance(s1, th, gm, ch)]))[0] == bits(np.array([ref.radiance(s2, th, gm, ch)]))[0]


def test_default_sky_known_values():
    # default sky turbidity 1, albedo 1, zenith 30 deg
    s = orc.aligned_sky_state()
    assert np.allclose(s[30:33], [796325.938, 503392.188, 234451.922], rtol=1e-7)
    assert np.allclose(s[27:30], [9.67027664, 16.4482307, 27.4176598], rtol=1e-7)
    assert np.allclose(s[0:3], [-1.08006394, -0.166454494, 2.705446], rtol=1e-6)
    assert np.allclose(s[36:39], [0.5, 0.8660254, 0.0], atol=1e-7)
    assert orc.sky_state_radiance(s[:33], 0.0, orc.degrees_to_radians(30.0), 0) == pytest.approx(2.99965715, rel=1e-6)
    # WGSL skyRadiance = sky_state_radiance without the solar disk term (gamma outside the disk)
    for ch in range(3):
        a = orc.wgsl_sky_radiance(s, 0.7, 1.1, ch); b = orc.sky_state_radiance(s[:33], 0.7, 1.1, ch)
        assert a == pytest.approx(b, rel=2e-6)


# ---------------------------------------------------------------- shading: analytic known answers
def test_solar_constants_bit_patterns():
    import math
    import struct
    f32 = np.float32
    rad = f32(0.255) * (f32(3.1415927) / f32(180))
    c = f32(math.cos(float(rad)))
    inv = f32(2) * f32(3.1415927) * (f32(1) - c)
    assert struct.unpack("<I", struct.pack("<f", rad))[0] == 0x3B91D640
    assert struct.unpack("<I", struct.pack("<f", c))[0] == 0x3F7FFF5A
    assert struct.unpack("<I", struct.pack("<f", inv))[0] == 0x38826048


def test_blue_noise_table_is_the_references():
    """The 128x128 RG8 table (src/pt/blue_noise.c): pinned by digest always, and against the reference's own
    blue_noise.c compiled unmodified (oracle/_ref/libbluenoise_ref.so) when that was built."""
    import hashlib
    import rayfinder_amd as rf
    table = np.asarray(orc.blue_noise_table(), np.uint8).reshape(-1)
    assert table.size == 128 * 128 * 2
    assert hashlib.sha256(table.tobytes()).hexdigest() == "49394f182237718fd4d0927f879d213f176c6f93db9afea7ecb4cbfee68075fd"
    with open(os.path.join(os.path.dirname(rf.__file__), "data", "blue_noise_128x128_rg8.bin"), "rb") as f:
        assert f.read() == table.tobytes()          # the product embeds the same bytes
    ref = orc.ref_blue_noise_table()
    if ref is not None:
        values, w, h = ref
        assert (w, h) == (128, 128) and np.array_equal(values, table)


def test_animated_blue_noise_definition():
    table = orc.blue_noise_table()
    # sample 0 is the raw table texel; 255 -> 1.0 -> fract -> 0 (H18)
    for (x, y) in [(0, 0), (5, 9), (127, 127), (130, 257)]:
        o = orc.animated_blue_noise(x, y, 0, 64)
        idx = (y % 128) * 128 + (x % 128)
        want = np.array([table[2 * idx], table[2 * idx + 1]], np.float32) / np.float32(255)
        want = want - np.floor(want)
        assert np.array_equal(o, want)
    # frame index wraps at spp: n = frameIdx % spp
    assert np.array_equal(orc.animated_blue_noise(3, 4, 70, 64), orc.animated_blue_noise(3, 4, 6, 64))
    o = orc.animated_blue_noise(3, 4, 6, 64)
    assert (o >= 0).all() and (o < 1).all()


def test_single_quad_nee_known_answer():
    """One white floor quad, camera looking straight down, 1 bounce: radiance of a hit pixel is
    exactly solarRadiance * (albedo/pi) * cos(n, l) * invPdf (wgsl:194-203), visibility 1."""
    import math
    # winding chosen so that normalize(cross(e1, e2)) points up: offsetRay pushes the hit point
    # along the GEOMETRIC normal whatever side the ray came from (wgsl:514-516), so a floor wound
    # the other way shadows itself -- reference behaviour, reproduced, not "fixed".
    P = np.array([[-2, 0, -2, 2, 0, 2, 2, 0, -2], [-2, 0, -2, -2, 0, 2, 2, 0, 2]], np.float32)
    nodes, idx, _ = orc.build_bvh(P)
    tris = orc.reorder(P, idx)
    pos48 = np.zeros((2, 12), np.float32); pos48[:, 0:3] = tris[:, 0:3]; pos48[:, 4:7] = tris[:, 3:6]; pos48[:, 8:11] = tris[:, 6:9]
    att = np.zeros((2, 20), np.float32); att[:, 1] = att[:, 5] = att[:, 9] = 1.0
    sc = orc.OracleScene(nodes, pos48, att, np.array([[1, 1, 0]], np.uint32), np.array([0xFFFFFFFF], np.uint32))
    cam = orc.create_camera([0, 3, 0.001], [0, 0, 0], 0.0, 1.0, orc.degrees_to_radians(40.0), 1.0)
    sky = orc.aligned_sky_state()
    rp = orc.make_render_params(16, 16, cam, 4, 1, 1.0, sky)
    rgb, st = orc.pixel_sample(sc, rp, 8, 8, 0)
    u = orc.animated_blue_noise(8, 8, 0, 4)
    cos_max = np.frombuffer(np.uint32(0x3F7FFF5A).tobytes(), np.float32)[0]
    # light direction is within 0.255 deg of the sun direction: n.l ~ cos(30 deg)
    expected = sky[30:33] * (1.0 / math.pi) * math.cos(math.radians(30.0)) * 6.216817e-05
    assert st.shadowRays == 1 and st.closestRays == 1
    assert np.allclose(rgb, expected, rtol=5e-3), (rgb, expected)
    assert cos_max < 1.0 and u.shape == (2,)


def test_miss_returns_sky_radiance():
    P = np.array([[-2, 0, -2, 2, 0, -2, 2, 0, 2]], np.float32)
    nodes, idx, _ = orc.build_bvh(P)
    pos48 = np.zeros((1, 12), np.float32)
    att = np.zeros((1, 20), np.float32)
    sc = orc.OracleScene(nodes, pos48, att, np.array([[1, 1, 0]], np.uint32), np.array([0xFFFFFFFF], np.uint32))
    cam = orc.create_camera([0, 1, 0], [0, 2, 0.001], 0.0, 1.0, orc.degrees_to_radians(40.0), 1.0)  # looking up
    sky = orc.aligned_sky_state()
    rp = orc.make_render_params(8, 8, cam, 1, 4, 1.0, sky)
    rgb, st = orc.pixel_sample(sc, rp, 4, 4, 0)
    ray = orc.wgsl_camera_ray(rp, 4, 4, 0)
    theta = np.float32(np.arccos(np.float64(ray[4])))
    d = np.float32(np.float32(ray[3] * sky[36] + ray[4] * sky[37]) + ray[5] * sky[38])
    gamma = np.float32(np.arccos(np.float64(np.clip(d, -1, 1))))
    want = np.array([orc.wgsl_sky_radiance(sky, theta, gamma, c) for c in range(3)], np.float32)
    assert st.shadowRays == 0 and np.array_equal(bits(rgb), bits(want))


def test_texture_lookup_semantics(duck_oracle):
    d = duck_oracle
    px = d.texels
    # texel (row i, col j) for uv -> j = u32(fract(u)*w), i = u32(fract(v)*h); fract wraps negatives
    for (u, v) in [(0.25, 0.5), (1.25, -0.5), (0.999, 0.001), (-0.3, 2.7)]:
        fu, fv = np.float32(u) - np.floor(np.float32(u)), np.float32(v) - np.floor(np.float32(v))
        j, i = int(np.float32(fu * np.float32(512))), int(np.float32(fv * np.float32(512)))
        bgra = int(px[i * 512 + j])
        srgb = np.array([(bgra >> 16) & 255, (bgra >> 8) & 255, bgra & 255], np.float32) / np.float32(255)
        want = np.array([np.float32(float(c) ** float(np.float32(2.2))) for c in srgb], np.float32)
        got = orc.texture_lookup(d.scene, 0, u, v)
        assert np.allclose(got, want, rtol=1e-6)


# ---------------------------------------------------------------- sampling helpers: analytic pins
# (nothing under src/pt is tested by the reference; these are the properties its
#  WGSL helpers must have, checked on the restatement)
def test_pixar_onb_is_orthonormal_and_right_handed():
    """wgsl:309-319 (Duff et al.): for unit n, (u, v, n) is an orthonormal right-handed frame, including
    the n.z < 0 branch and the poles."""
    rng = np.random.default_rng(3)
    ns = rng.normal(size=(300, 3))
    ns /= np.linalg.norm(ns, axis=1, keepdims=True)
    ns = np.concatenate([ns, [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, -1, 0], [0.6, 0.8, 0.0], [0.0, 0.6, -0.8]]]).astype(np.float32)
    for n in ns:
        u, v = orc.pixar_onb(n)
        u64, v64, n64 = u.astype(np.float64), v.astype(np.float64), n.astype(np.float64)
        assert abs(np.dot(u64, u64) - 1) < 1e-5 and abs(np.dot(v64, v64) - 1) < 1e-5
        assert abs(np.dot(u64, v64)) < 1e-5 and abs(np.dot(u64, n64)) < 1e-5 and abs(np.dot(v64, n64)) < 1e-5
        assert np.allclose(np.cross(u64, v64), n64, atol=1e-5)


def test_cosine_hemisphere_and_cone_samples():
    """wgsl:582-592: (cos 2 pi u.y sqrt(1-u.x), sin 2 pi u.y sqrt(1-u.x), sqrt(u.x)) is unit with z = sqrt(u.x);
    wgsl:568-579: the cone sample is unit with cos(theta) = 1 - u.x (1 - cosThetaMax) >= cosThetaMax."""
    rng = np.random.default_rng(4)
    cmax = struct.unpack("<f", struct.pack("<I", 0x3F7FFF5A))[0]
    for ux, uy in np.concatenate([rng.uniform(0, 1, (200, 2)), [[0, 0], [0, 0.25], [0.999999, 0.5], [0.5, 0.999999]]]).astype(np.float32):
        d = orc.direction_in_cosine_weighted_hemisphere(ux, uy).astype(np.float64)
        assert abs(np.dot(d, d) - 1) < 2e-6
        assert d[2] == np.float64(np.sqrt(np.float32(ux)))
        c = orc.direction_in_cone(ux, uy, cmax).astype(np.float64)
        assert abs(np.dot(c, c) - 1) < 2e-6
        assert c[2] == np.float64(np.float32(1) - np.float32(ux) * (np.float32(1) - np.float32(cmax))) and c[2] >= cmax - 1e-7
    # u.x = 0 (blue-noise byte 255 on sample 0, Appendix A H18): a grazing bounce direction, z exactly 0
    assert orc.direction_in_cosine_weighted_hemisphere(0.0, 0.3)[2] == 0.0


def test_offset_ray_known_answers():
    """ray_intersection.cpp:17-35 == wgsl:523-544 (Ray Tracing Gems ch. 6): away from the origin the float's bit
    pattern moves by int(256 n) ulps, away from zero when the offset points outwards; within 1/32 of the origin
    the offset is n / 65536 instead; per component."""
    def ulps(a, b):
        return int(np.float32(b).view(np.int32)) - int(np.float32(a).view(np.int32))
    p = np.array([2.0, -3.5, 0.01], np.float32)
    n = np.array([1.0, 0.0, 0.0], np.float32)
    o = orc.offset_ray(p, n)
    assert ulps(p[0], o[0]) == 256 and o[1] == p[1] and o[2] == p[2] + np.float32(0.0) / 65536
    o = orc.offset_ray(p, np.array([0.0, 1.0, 0.0], np.float32))
    assert ulps(p[1], o[1]) == -256 and o[1] > p[1]             # negative coordinate: the bit pattern decreases, the value moves up
    o = orc.offset_ray(p, np.array([0.0, -0.5, 1.0], np.float32))
    assert ulps(p[1], o[1]) == 128 and o[1] < p[1]
    assert o[2] == np.float32(p[2] + np.float32(1.0 / 65536.0) * np.float32(1.0))   # |p.z| < 1/32
    # int() truncates toward zero: |256 n| < 1 moves nothing
    o = orc.offset_ray(p, np.array([0.003, 0.0, 0.0], np.float32))
    assert o[0] == p[0]


def test_aces_tonemap_known_answers():
    """wgsl:277-285 + 59-63: acesFilmic(x) = x(2.51x+0.03)/(x(2.43x+0.59)+0.14) clamped to [0,1], then ^(1/2.2)."""
    img = np.array([[0, 0, 0, 0], [1, 1, 1, 0], [4, 0.5, 100, 0], [0.18, 0.18, 0.18, 0]], np.float32)
    out = orc.tonemap(img, 1, 1.0)
    assert np.array_equal(out[0], [0, 0, 0])
    x = np.float64([1, 4, 0.5, 100, 0.18])
    want = np.clip(x * (2.51 * x + 0.03) / (x * (2.43 * x + 0.59) + 0.14), 0, 1) ** (1 / 2.2)
    got = np.float64([out[1][0], out[2][0], out[2][1], out[2][2], out[3][0]])
    assert np.allclose(got, want, rtol=2e-6)
    assert out[2][2] == 1.0                                   # saturates
    # exposure and sample count enter as exposure * sum / n
    assert np.array_equal(orc.tonemap(img * 8, 4, 0.5), orc.tonemap(img, 1, 1.0))


# ---------------------------------------------------------------- multi-bounce known answers (round 2)
# Nothing in the reference tests rayColor (wgsl:180-234).  tests/analytic_scene.py is a second, independent
# restatement (float64, analytic rectangles, no BVH / Moller-Trumbore, no code shared with the oracle); the two
# must agree wherever the analytic one's decisions are robust.
def _oracle_scene_from_rects(rects):
    import analytic_scene as an
    P, N, T, I, tex = an.scene_arrays(rects)
    nodes, idx, _ = orc.build_bvh(P)
    Pr, Nr, Tr, Ir = orc.reorder(P, idx), orc.reorder(N, idx), orc.reorder(T, idx), orc.reorder(I, idx)
    n = len(Pr)
    pos48 = np.zeros((n, 12), np.float32); pos48[:, 0:3] = Pr[:, 0:3]; pos48[:, 4:7] = Pr[:, 3:6]; pos48[:, 8:11] = Pr[:, 6:9]
    att = np.zeros((n, 20), np.float32)
    att[:, 0:3] = Nr[:, 0:3]; att[:, 4:7] = Nr[:, 3:6]; att[:, 8:11] = Nr[:, 6:9]; att[:, 12:18] = Tr
    att.view(np.uint32)[:, 18] = Ir
    descs = np.array([[1, 1, k] for k in range(len(tex))], np.uint32)
    texels = np.concatenate([px for px, _, _ in tex])
    return orc.OracleScene(nodes, pos48, att, descs, texels)


def corner_rects():
    """Floor + two walls.  The x = 1 wall is wound AWAY from the scene (geometric normal +x) while its shading normal
    faces the scene (-x): hit points on it are pushed behind it (wgsl:514-516), so its sun sample is unoccluded with a
    NEGATIVE cosine -- the unclamped NEE term of wgsl:201."""
    import analytic_scene as an
    return [an.Rect(1, 0.0, (-3, -3), (3, 3), (0, 1, 0), +1, (200, 120, 60)),
            an.Rect(0, 1.0, (0, -3), (1.5, 3), (-1, 0, 0), +1, (90, 200, 160)),
            an.Rect(2, -1.2, (-3, 0), (1, 2.0), (0, 0, 1), +1, (230, 230, 230))]


@pytest.mark.parametrize("bounces", [1, 2, 3])
def test_multi_bounce_paths_match_an_independent_analytic_integrator(bounces):
    import analytic_scene as an
    rects = corner_rects()
    sc = _oracle_scene_from_rects(rects)
    W = H = 24
    cam = orc.create_camera([-1.5, 2.0, 2.0], [0.3, 0.0, -0.3], 0.0, 1.0, orc.degrees_to_radians(50.0), 1.0)
    sky = orc.aligned_sky_state(azimuth_deg=35.0)      # sun (0.41, 0.87, -0.29): grazes none of the three planes
    spp = 4
    rp = orc.make_render_params(W, H, cam, spp, bounces, 1.0, sky)
    kinds, checked = set(), 0
    for frame in range(spp):
        for y in range(H):
            for x in range(W):
                u = orc.animated_blue_noise(x, y, frame, spp).astype(np.float64)
                want, robust, trace = an.path_sample(rects, cam, sky, W, H, x, y, u, bounces)
                if not robust:
                    continue
                got, st = orc.pixel_sample(sc, rp, x, y, frame)
                scale = max(1.0, float(np.abs(want).max()))
                assert np.allclose(got.astype(np.float64), want, rtol=3e-4, atol=2e-5 * scale), (x, y, frame, got, want, trace)
                checked += 1
                assert st.closestRays == len(trace) and st.shadowRays == sum(1 for t in trace if t[0] == "hit")
                for t in trace:
                    kinds.add(("sky", t[1]) if t[0] == "sky" else ("hit", t[1], t[3] > 0, t[4] < 0))
    assert checked > 0.8 * W * H * spp
    assert ("hit", 1, True, False) in kinds                     # lit first vertex
    if bounces >= 2:
        assert ("sky", 2) in kinds                              # escaped after one bounce: throughput = albedo of vertex 1
        assert any(k[0] == "hit" and k[1] == 2 for k in kinds)  # NEE at the second vertex, added before the `bounce == numBounces` break
        assert any(k[0] == "hit" and k[2] and k[3] for k in kinds)      # unoccluded sun sample with a negative cosine (no clamp)
        assert any(k[0] == "hit" and not k[2] for k in kinds)   # a shadowed vertex
    if bounces >= 3:
        assert ("sky", 3) in kinds and any(k[0] == "hit" and k[1] == 3 for k in kinds)


def box_rects(sealed, srgb=(255, 255, 255)):
    """A room [-1, 1]^3 seen from inside; faces overlap at the edges (each spans [-1.5, 1.5]) so that no ray can slip
    between two faces; normals (geometric and shading) point inwards."""
    import analytic_scene as an
    faces = []
    for axis in range(3):
        for side in (-1, 1):
            if axis == 1 and side == 1 and not sealed:
                continue                                        # open to the sky
            n = [0, 0, 0]; n[axis] = -side
            faces.append(an.Rect(axis, float(side), (-1.5, -1.5), (1.5, 1.5), n, -side, srgb))
    return faces


def test_sealed_white_room_is_black_and_open_room_respects_the_energy_bound():
    """Closed scene: every sun sample is occluded and no path reaches the sky, so the estimate is EXACTLY zero
    whatever the albedo and bounce count -- any leak (self-intersection, a missed far child, a wrong offset side)
    shows up as a non-zero pixel.  Open top, albedo rho: every sample lies within
    +- sum_b rho^b * solar * invPdf / pi  +  rho^k * max sky radiance (wgsl:203,228 with |n.l| <= 1)."""
    W = H = 32
    cam = orc.create_camera([0.2, -0.3, 0.4], [0.9, 0.2, -0.8], 0.0, 1.0, orc.degrees_to_radians(80.0), 1.0)
    sky = orc.aligned_sky_state()
    sc = _oracle_scene_from_rects(box_rects(True))
    img, st = orc.render(sc, orc.make_render_params(W, H, cam, 4, 6, 1.0, sky), 0, 4)
    assert st.closestRays == W * H * 4 * 6 and st.shadowRays == st.closestRays     # every ray hit a wall, every bounce
    assert not img[..., :3].any()
    srgb = (186, 186, 186)
    rho = (186 / 255.0) ** 2.2
    sc = _oracle_scene_from_rects(box_rects(False, srgb))
    bounces, spp = 5, 8
    img, st = orc.render(sc, orc.make_render_params(W, H, cam, spp, bounces, 1.0, sky), 0, spp)
    solar, inv_pdf = sky[30:33].astype(np.float64), 6.216817e-05
    import analytic_scene as an
    max_sky = max(an.sky_radiance(sky, t, g).max() for t in np.linspace(0, np.pi / 2, 40) for g in np.linspace(0, np.pi, 40))
    bound = spp * (sum(rho ** b for b in range(1, bounces + 1)) * solar.max() * inv_pdf / np.pi + max_sky)
    assert np.isfinite(img).all() and np.abs(img[..., :3]).max() <= bound
    assert img[..., :3].max() > 0 and st.shadowRays < st.closestRays              # light does get in, some paths leave


def test_deferred_variant_analytic_properties():
    """The deferred-lighting restatement (oracle side): frame 0 initialises the accumulation, later frames are the 0.1 / 0.9
    blend (resolve_pass.wgsl:41-49); a pixel that sees only sky carries skyRadiance (+ the solar disk inside 0.255 deg);
    a floor pixel's first light sample equals the path tracer's single-bounce NEE term up to the different offset."""
    import analytic_scene as an
    rects = [an.Rect(1, 0.0, (-2, -2), (2, 2), (0, 1, 0), +1, (255, 255, 255))]
    sc = _oracle_scene_from_rects(rects)
    W = H = 16
    cam = orc.create_camera([0, 3, 0.001], [0, 0, 0], 0.0, 1.0, orc.degrees_to_radians(40.0), 1.0)
    sky = orc.aligned_sky_state()
    rp = orc.make_render_params(W, H, cam, 1, 2, 1.0, sky)
    s1, a1, _, _ = orc.deferred_frames(sc, rp, 1)
    assert np.array_equal(bits(s1), bits(a1))
    s2, a2, _, _ = orc.deferred_frames(sc, rp, 2)
    # frame 1's blend: replay frame 0 for the previous value
    assert np.array_equal(bits(a2), bits(np.float32(0.1) * s2 + np.float32(0.9) * a1))
    # an isolated floor: the bounce ray always escapes, so colour = NEE(floor) + albedo * sky(wi) (+ sun if wi looks at it)
    solar, inv_pdf = sky[30:33].astype(np.float64), 6.216817e-05
    nee = solar / np.pi * np.cos(np.radians(30.0)) * inv_pdf
    px = s1[8, 8].astype(np.float64)
    assert (px > nee * 0.99).all() and (px < nee * 1.01 + 60.0).all()
    # looking up: sky only, and the pixel that looks into the sun includes the solar radiance
    up = orc.create_camera([0, 1, 0], [0.5, 1 + 0.8660254, 0.0], 0.0, 1.0, orc.degrees_to_radians(2.0), 1.0)   # towards the sun (zenith 30 deg, azimuth 0)
    rp2 = orc.make_render_params(W, H, up, 1, 2, 1.0, sky)
    s, _, _, st = orc.deferred_frames(sc, rp2, 1)
    assert st.shadowRays == 0
    assert s[..., 0].max() > 0.5 * sky[30] and s[..., 0].min() < 0.01 * sky[30]     # inside / outside the 0.255-degree disk


def test_oracle_stack_bound_is_the_products_96_entries():
    """The oracle's traversal stack (oracle/rf_oracle.c ORC_STACK) is aligned with the product's (rf_device.hpp: 24 + 72 = 96):
    a chain with 70 or 96 pending far children is traversed to the end (the reference's 32-entry array would be overrun), one with
    97 is abandoned with what was found up to there -- nothing -- and says so through stackHigh = 97.  The product's HOST query
    (rf_intersect_bvh, growable pending list) agrees wherever the bound is not reached."""
    import rayfinder_amd as rf
    for depth in (31, 70, 96, 97, 150):
        P = np.zeros((depth + 1, 9), np.float32)
        for i in range(depth + 1):
            z = np.float32(-1.0 - i)
            P[i] = [-1, -1, z, 1, -1, z, 0, 1, z]
        nodes = np.zeros(2 * depth + 1, rf.NODE_DTYPE)
        for k in range(depth):
            nodes[k]["min"] = [-1, -1, -1.0 - depth]; nodes[k]["max"] = [1, 1, -1.0 - k]
            nodes[k]["splitAxis"] = 0
            nodes[k]["secondChildOffset"] = 2 * depth - k
        def leaf(at, tri):
            nodes[at]["min"] = [-1, -1, P[tri][2]]; nodes[at]["max"] = [1, 1, P[tri][2]]
            nodes[at]["trianglesOffset"] = tri; nodes[at]["triangleCount"] = 1; nodes[at]["splitAxis"] = 0xFFFFFFFF
        leaf(depth, depth)
        for k in range(depth):
            leaf(2 * depth - k, k)
        ray = np.array([[0.0, 0.0, 5.0, 1e-3, 1e-3, -1.0]], np.float32)   # +x: first child first, every second child pending
        o = orc.intersect_bvh_batch(nodes, P, ray, 1e30)
        vis = orc.shadow_batch(nodes, P, ray, 1e30)
        if depth <= 96:
            hit, rec, st = rf.intersect_bvh(ray[0], nodes, P, 1e30)
            assert o["hit"][0] == 1 and hit and o["tri"][0] == rec["triangle"] == 0 and o["t"][0] == rec["t"]
            assert o["stackHigh"][0] == depth == st["stack_high_water"] and o["nodesVisited"][0] == 2 * depth + 1 == st["nodes_visited"]
            assert vis[0] == 0.0
        else:
            assert o["hit"][0] == 0 and o["stackHigh"][0] == 97          # abandoned before any leaf was reached
            assert vis[0] == 1.0


def test_wgsl_builtins_are_the_specified_f64_evaluations():
    """sin / cos / acos / exp of the WGSL restatement are the f32 roundings of SPECIFIED f64 sequences (fdlibm kernels, rf_oracle.c d_exp
    ...; the product runs the same sequences on the GPU): within a few ulp(f64) of the C library over the argument ranges the renderer
    uses -- so the f32 rounding is the correctly rounded value except about once in 2^29 calls -- and the IEEE special cases hold."""
    import ctypes as C
    import math
    L = orc.lib()
    for n in ("orc_d_exp", "orc_d_sin", "orc_d_cos", "orc_d_acos"):
        f = getattr(L, n); f.restype = C.c_double; f.argtypes = [C.c_double]
    rng = np.random.default_rng(3)
    def worst(f, g, xs, absolute_floor=0.0):
        w = 0.0
        for x in xs:
            a, b = f(float(x)), g(float(x))
            w = max(w, abs(a - b) / max(math.ulp(b), absolute_floor))
        return w
    ex = np.concatenate([rng.uniform(-90, 90, 40000), rng.uniform(-1, 1, 20000) * 10.0 ** rng.uniform(-8, 1, 20000)]).astype(np.float32)
    assert worst(L.orc_d_exp, math.exp, ex) <= 1.5
    an = np.concatenate([rng.uniform(0, 2 * np.pi, 60000), rng.uniform(-7, 7, 20000)]).astype(np.float32)
    # (near the zeros of sin / cos the error is a few ulp of the tiny RESULT, i.e. ~1e-17 absolute: far below an f32's half ulp)
    assert worst(L.orc_d_sin, math.sin, an, absolute_floor=2.0 ** -54) <= 2.0
    assert worst(L.orc_d_cos, math.cos, an, absolute_floor=2.0 ** -54) <= 2.0
    ac = np.concatenate([rng.uniform(-1, 1, 60000), [1.0, -1.0, 0.0, 0.5, -0.5, np.nextafter(np.float32(1), np.float32(0)), np.float32(0.49999997)]]).astype(np.float32)
    assert worst(L.orc_d_acos, math.acos, ac) <= 1.5
    assert math.isnan(L.orc_d_acos(1.5)) and math.isnan(L.orc_d_acos(float("nan"))) and math.isnan(L.orc_d_sin(float("inf")))
    assert L.orc_d_exp(-800.0) == 0.0 and L.orc_d_exp(800.0) == float("inf") and L.orc_d_exp(0.0) == 1.0
    assert L.orc_d_acos(1.0) == 0.0 and L.orc_d_cos(0.0) == 1.0 and L.orc_d_sin(0.0) == 0.0


def test_reference_angle_test_vectors():
    """src/tests/angle.cpp:8-29 replayed on the oracle's degrees -> radians (the conversion behind the camera's vfov / yaw / pitch and the sky's zenith / azimuth):
    90 degrees == 0.5 pi, 90 + 90 == pi, within Catch::Approx's default tolerance (100 float epsilons, relative)."""
    eps = 100.0 * float(np.finfo(np.float32).eps)
    half_pi, pi = np.float32(0.5) * np.float32(np.pi), np.float32(np.pi)
    assert abs(float(orc.degrees_to_radians(90.0)) - float(half_pi)) <= eps * float(half_pi)
    assert abs(float(orc.degrees_to_radians(90.0 + 90.0)) - float(pi)) <= eps * float(pi)
    # and the round trip the reference checks (asDegrees of what was given in radians)
    assert abs(float(orc.degrees_to_radians(90.0)) * 180.0 / float(pi) - 90.0) <= eps * 90.0
