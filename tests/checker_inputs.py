"""What the GPU parity tests hand to the CHECKER that the PRODUCT computed: sky states (rf.aligned_sky_state), cameras
(rf.fly_camera / rf.create_camera) and baked scene arrays (conftest.oracle_scene_from_pt).  If one of those were wrong, product and
oracle would be wrong together and `-m gpu` would stay green.  These checks compare each of them with the oracle's own computation of
the same thing (and the sky with the vectors the reference's C produced, tests/golden/sky_ref.npz).  They need no GPU; they run in
the CPU suite (test_host_core.py) AND in the driver-run `-m gpu` suite (test_gpu_parity.py), which is where the hole was."""
import os

import numpy as np

import rayfinder_amd as rf
from conftest import GOLDEN, bits
from oracle import gltf_ref, orc

# every non-default sky a GPU test renders with (test_sky_dome..., test_lens_sampling..., test_set_render_parameters...,
# test_analytic_..., the randomized differential test's generator with its seeds)
SKIES = [dict(), dict(turbidity=9.5, albedo=(0.9, 0.1, 0.4), sun_zenith_degrees=82.0, sun_azimuth_degrees=200.0),
         dict(turbidity=4.5, albedo=(0.3, 0.5, 0.2), sun_zenith_degrees=75.0, sun_azimuth_degrees=200.0),
         dict(turbidity=1.0, albedo=(1.0, 1.0, 1.0), sun_zenith_degrees=30.0, sun_azimuth_degrees=35.0),
         dict(turbidity=2.0, albedo=(0.0, 0.0, 0.0), sun_zenith_degrees=0.0, sun_azimuth_degrees=0.0),
         dict(turbidity=10.0, albedo=(1.0, 0.0, 1.0), sun_zenith_degrees=89.0, sun_azimuth_degrees=359.0)]


def random_skies(n=24):
    for seed in range(n):
        rng = np.random.default_rng(1000 + seed)
        yield dict(turbidity=float(rng.uniform(1, 10)), albedo=tuple(float(x) for x in rng.uniform(0, 1, 3)),
                   sun_zenith_degrees=float(rng.uniform(0, 89)), sun_azimuth_degrees=float(rng.uniform(0, 360)))


def check_sky_states():
    for kw in SKIES + list(random_skies()):
        got = rf.aligned_sky_state(rf.make_sky(**kw))
        want = orc.aligned_sky_state(kw.get("turbidity", 1.0), kw.get("albedo", (1.0, 1.0, 1.0)), kw.get("sun_zenith_degrees", 30.0), kw.get("sun_azimuth_degrees", 0.0))
        assert np.array_equal(bits(got), bits(want)), kw
    # ... and both against what the reference's hw_skymodel.c produced (sky_state_new -> the 33 floats AlignedSkyState copies)
    g = np.load(os.path.join(GOLDEN, "sky_ref.npz"))
    for case, want, rc in zip(g["cases"], g["states"], g["rcs"]):
        for state_new in (rf.sky_state_new, orc.sky_state_new):
            got_rc, got = state_new(case[0], case[1], case[2:5])
            assert got_rc == rc == 0 and np.array_equal(bits(got), bits(want)), case
    # AlignedSkyState's layout (aligned_sky_state.hpp:34-71: params 0..26, sky 27..29, solar 30..32, pad, sun direction 36..38) over a
    # state the reference-compiled vectors hold: the default sky's first 33 floats must be sky_state_new's for its elevation
    sky = rf.aligned_sky_state(rf.make_sky())
    rc, st = rf.sky_state_new(np.float32(np.float32(0.5) * np.float32(np.pi)) - orc.degrees_to_radians(30.0), 1.0, (1.0, 1.0, 1.0))
    assert rc == 0 and np.array_equal(bits(sky[:33]), bits(st)) and not sky[33:36].any() and sky[39] == 0
    return len(SKIES) + 24, int(len(g["cases"]))


def check_cameras(frame_sizes=((1920, 1080), (800, 600), (3840, 2160), (96, 64), (320, 192), (150, 90), (256, 256), (128, 96), (64, 64), (33, 17))):
    n = 0
    for (w, h) in frame_sizes:
        for kw in (dict(), dict(aperture=0.2, focus_distance=2.0, vfov_degrees=50.0)):
            a = rf.camera_to_array(rf.fly_camera(w, h, **kw))
            b = orc.default_pt_camera(w, h, vfov_degrees=kw.get("vfov_degrees", 70.0), aperture=kw.get("aperture", 0.0), focus=kw.get("focus_distance", 10.0))
            assert np.array_equal(bits(a), bits(b)), (w, h, kw)
            n += 1
    deg = orc.degrees_to_radians
    explicit = [((0.0, 0.2, 0.0), (0.3, 1.0, 0.2), 0.0, 1.0, float(deg(110.0)), 96 / 64), ((0.2, 0.2, -5.0), (0.2, 0.2, 0.0), 0.0, 1.0, float(np.radians(20.0)), 1.0),
                ((0, 3, 0.001), (0, 0, 0), 0.0, 1.0, float(deg(40.0)), 1.0), ((0.3, 0.4, 3.0), (0, 0, 0), 0.0, 1.0, float(deg(60.0)), 4 / 3),
                ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.0, 1.0, float(deg(12.0)), 1.5), ((-1.5, 2.0, 2.0), (0.3, 0.0, -0.3), 0.0, 1.0, float(deg(50.0)), 1.0),
                ((0.2, -0.3, 0.4), (0.9, 0.2, -0.8), 0.0, 1.0, float(deg(80.0)), 160 / 96), ((4.0, 2.5, 4.5), (1.0, 0.8, -1.0), 0.0, 1.0, float(deg(55.0)), 160 / 96),
                ((0.2, 0.3, 3.0), (0.0, 0.0, 0.0), 0.0, 1.0, float(np.radians(60.0)), 1.25)]
    for seed in range(24):      # the randomized differential test's cameras
        rng = np.random.default_rng(1000 + seed)
        pos = rng.uniform(-3, 3, 3); pos[1] = abs(pos[1]) + 0.3
        explicit.append((pos, rng.uniform(-1, 1, 3), float(rng.choice([0.0, 0.05, 0.3])), float(rng.uniform(1, 6)), float(deg(float(rng.uniform(30, 100)))), float(rng.uniform(0.5, 2.0))))
    for (o, at, ap, fo, vf, asp) in explicit:
        a = rf.camera_to_array(rf.create_camera(o, at, ap, fo, vf, asp))
        b = orc.create_camera(o, at, ap, fo, vf, asp)
        assert np.array_equal(bits(a), bits(b)), (o, at)
        n += 1
    return n


def check_bake_against_oracle_builder(P, N, UV, T, pt):
    """The product's bake of a triangle soup (PtFormat.from_triangles: host builder + reorder + GPU layouts) against the ORACLE's
    builder and layout code on the same soup: node bytes, bvh positions, 48-byte positions, 80-byte attributes."""
    a = pt.arrays()
    nodes, idx, _ = orc.build_bvh(P)
    assert a["bvhNodes"].tobytes() == nodes.tobytes(), "node bytes differ from the oracle builder's"
    tris36 = orc.reorder(np.ascontiguousarray(P, np.float32), idx)
    assert a["bvhPositionAttributes"].tobytes() == tris36.tobytes()
    pos48, attr80 = gltf_ref.gpu_layout(tris36, orc.reorder(np.ascontiguousarray(N, np.float32), idx), orc.reorder(np.ascontiguousarray(UV, np.float32), idx),
                                        orc.reorder(np.ascontiguousarray(T, np.uint32), idx))
    assert a["trianglePositionAttributes"].tobytes() == pos48.tobytes()
    assert a["triangleVertexAttributes"].tobytes() == np.ascontiguousarray(attr80).tobytes()
    return len(nodes)


def check_atrium_bake():
    from rayfinder_amd import scenes
    pt, info = scenes.atrium(1)
    P, N, UV, T = scenes.atrium_triangles(1)
    n = check_bake_against_oracle_builder(P, N, UV, T, pt)
    tex = scenes.atrium_textures()
    got = pt.arrays()["baseColorTextures"]
    assert len(got) == len(tex) == 25
    for (px, w, h), (qx, w2, h2) in zip(got, tex):
        assert (w, h) == (w2, h2) and np.array_equal(px, qx)
    return n, info
