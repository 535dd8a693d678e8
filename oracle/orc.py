"""ctypes binding of the CPU oracle (oracle/librf_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package rayfinder_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "librf_oracle.so")
_REF_PATH = os.path.join(HERE, "_ref", "libhwsky_ref.so")
DATA = os.path.join(HERE, "..", "rayfinder_amd", "data")

NODE_DTYPE = np.dtype([("min", "<f4", 3), ("pad0", "<f4"), ("max", "<f4", 3), ("pad1", "<f4"),
                       ("trianglesOffset", "<u4"), ("secondChildOffset", "<u4"),
                       ("triangleCount", "<u4"), ("splitAxis", "<u4")])
assert NODE_DTYPE.itemsize == 48


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(HERE, "rf_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])


class Stats(C.Structure):
    _fields_ = [("closestRays", C.c_uint64), ("shadowRays", C.c_uint64), ("closestNodeVisits", C.c_uint64),
                ("shadowNodeVisits", C.c_uint64), ("closestTriTests", C.c_uint64), ("shadowTriTests", C.c_uint64),
                ("texelOobClamps", C.c_uint64), ("nanPixels", C.c_uint64), ("stackHigh", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Scene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("positions", C.c_void_p), ("attrs", C.c_void_p), ("texDescs", C.c_void_p),
                ("texels", C.c_void_p), ("numTexels", C.c_uint64), ("blueNoise", C.c_void_p)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("camera", C.c_float * 19),
                ("numSamplesPerPixel", C.c_uint32), ("numBounces", C.c_uint32), ("exposure", C.c_float),
                ("sky", C.c_float * 40)]


_lib = None
_tables = None


def lib():
    global _lib, _tables
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_build_bvh.restype = C.c_uint64
        _lib.orc_aabb_surface_area.restype = C.c_float
        _lib.orc_degrees_to_radians.restype = C.c_float
        _lib.orc_degrees_to_radians.argtypes = [C.c_float]
        _lib.orc_sky_state_radiance.restype = C.c_float
        _lib.orc_sky_state_radiance.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
        _lib.orc_wgsl_sky_radiance.restype = C.c_float
        _lib.orc_wgsl_sky_radiance.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint32]
        _lib.orc_sky_state_new.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib.orc_aligned_sky_state.argtypes = [C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_create_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_fly_camera.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p]
        _lib.orc_generate_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_bvh_visualizer_camera.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        _lib.orc_ray_intersect_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        _lib.orc_ray_intersect_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        _lib.orc_intersect_bvh_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_float] + [C.c_void_p] * 8
        _lib.orc_shadow_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        _lib.orc_brute_force_intersect.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_float, C.c_void_p]
        _lib.orc_bvh_test_camera.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
        _lib.orc_bvh_visualize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        _lib.orc_bvh_visualizer_pixel.restype = C.c_uint32
        _lib.orc_bvh_visualizer_pixel.argtypes = [C.c_uint32]
        _lib.orc_texture_lookup.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_animated_blue_noise.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.orc_wgsl_camera_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.orc_pixel_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.orc_render.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 6 + [C.c_void_p, C.c_void_p]
        _lib.orc_render_from.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p, C.c_void_p]
        _lib.orc_deferred_frame.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p] * 4
        _lib.orc_render_threads.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 7 + [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _lib.orc_render_threads.restype = C.c_int
        _lib.orc_bvh_visualize_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
        _lib.orc_bvh_visualize_threads.restype = C.c_int
        _lib.orc_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        _lib.orc_pixar_onb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_direction_in_cone.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        _lib.orc_direction_in_cosine_weighted_hemisphere.argtypes = [C.c_float, C.c_float, C.c_void_p]
        _lib.orc_offset_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_reorder_attributes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        _lib.orc_build_bvh.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        assert _lib.orc_sizeof_stats() == C.sizeof(Stats)
        assert _lib.orc_sizeof_scene() == C.sizeof(Scene)
        assert _lib.orc_sizeof_render_params() == C.sizeof(RenderParams)
        _tables = np.fromfile(os.path.join(DATA, "hw_sky_tables.bin"), dtype="<f4")
        assert _tables.size == 3630
        _lib.orc_set_sky_tables(_tables.ctypes.data_as(C.c_void_p))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def blue_noise_table():
    return np.fromfile(os.path.join(DATA, "blue_noise_128x128_rg8.bin"), dtype=np.uint8)


# ---------------------------------------------------------------------------------- BVH
def build_bvh(positions36):
    """positions36: (N,9) f32 -> (nodes[NODE_DTYPE], triangleIndices[u64], depth)"""
    tris = f32(positions36).reshape(-1, 9)
    n = tris.shape[0]
    nodes = np.zeros(2 * n, dtype=NODE_DTYPE)
    idx = np.zeros(n, dtype=np.uint64)
    depth = C.c_int(0)
    cnt = lib().orc_build_bvh(_p(tris), n, _p(nodes), _p(idx), C.byref(depth))
    return nodes[:cnt].copy(), idx, depth.value


def reorder(attrs, triangle_indices):
    a = np.ascontiguousarray(attrs)
    out = np.empty_like(a)
    stride = a.strides[0]
    lib().orc_reorder_attributes(_p(a), _p(out), a.shape[0], stride, _p(np.ascontiguousarray(triangle_indices, dtype=np.uint64)))
    return out


def intersect_bvh_batch(nodes, tris, rays, tmax):
    tris = f32(tris); rays = f32(rays).reshape(-1, 6)
    stride = tris.shape[1]
    n = rays.shape[0]
    out = dict(hit=np.zeros(n, np.uint8), t=np.zeros(n, np.float32), p=np.zeros((n, 3), np.float32),
               tri=np.zeros(n, np.uint32), uv=np.zeros((n, 2), np.float32), nodesVisited=np.zeros(n, np.uint32),
               triTests=np.zeros(n, np.uint32), stackHigh=np.zeros(n, np.uint32))
    lib().orc_intersect_bvh_batch(_p(nodes), _p(tris), stride, _p(rays), n, tmax, _p(out["hit"]), _p(out["t"]), _p(out["p"]),
                                  _p(out["tri"]), _p(out["uv"]), _p(out["nodesVisited"]), _p(out["triTests"]), _p(out["stackHigh"]))
    return out


def shadow_batch(nodes, tris, rays, tmax):
    tris = f32(tris); rays = f32(rays).reshape(-1, 6)
    vis = np.zeros(rays.shape[0], np.float32)
    lib().orc_shadow_batch(_p(nodes), _p(tris), tris.shape[1], _p(rays), rays.shape[0], tmax, _p(vis))
    return vis


def brute_force(tris, ray, tmax):
    tris = f32(tris); ray = f32(ray)
    t = C.c_float(0)
    did = lib().orc_brute_force_intersect(_p(tris), tris.shape[1], tris.shape[0], _p(ray), tmax, C.byref(t))
    return bool(did), np.float32(t.value)


def bvh_visualize(nodes, tris, cam19, W, H, row0=0, row1=None):
    tris = f32(tris); cam19 = f32(cam19)
    row1 = H if row1 is None else row1
    nv = np.zeros(W * H, np.uint32); hit = np.zeros(W * H, np.uint8); t = np.zeros(W * H, np.float32)
    tt = np.zeros(W * H, np.uint32); sh = np.zeros(W * H, np.uint32)
    lib().orc_bvh_visualize(_p(nodes), _p(tris), tris.shape[1], _p(cam19), W, H, row0, row1, _p(nv), _p(hit), _p(t), _p(tt), _p(sh))
    return dict(nodesVisited=nv, hit=hit, t=t, triTests=tt, stackHigh=sh)


# ---------------------------------------------------------------------------------- camera
def degrees_to_radians(d):
    return np.float32(lib().orc_degrees_to_radians(np.float32(d)))


def create_camera(origin, look_at, aperture, focus, vfov_rad, aspect):
    cam = np.zeros(19, np.float32)
    lib().orc_create_camera(_p(f32(origin)), _p(f32(look_at)), aperture, focus, vfov_rad, aspect, _p(cam))
    return cam


def bvh_visualizer_camera(nodes, aspect):
    cam = np.zeros(19, np.float32)
    lib().orc_bvh_visualizer_camera(_p(nodes), aspect, _p(cam))
    return cam


def bvh_test_camera(tris):
    tris = f32(tris)
    cam = np.zeros(19, np.float32)
    lib().orc_bvh_test_camera(_p(tris), tris.shape[1], tris.shape[0], _p(cam))
    return cam


def generate_camera_ray(cam19, u, v):
    r = np.zeros(6, np.float32)
    lib().orc_generate_camera_ray(_p(f32(cam19)), u, v, _p(r))
    return r


def default_pt_camera(width, height, vfov_degrees=70.0, aperture=0.0, focus=10.0,
                      position=(1.22, 1.25, -1.25), yaw_deg=129.64, pitch_deg=-13.73):
    """fly_camera_controller.hpp:47-52 pose, .cpp:12-22 getCamera, .cpp:138-148 orientation;
    vfov 70 deg is the UI default that overrides the controller's 80 (pt/main.cpp:49,314).
    aspect = float(width)/float(height) (common/extent.hpp:38-42)."""
    cam = np.zeros(19, np.float32)
    aspect = np.float32(np.float32(width) / np.float32(height))
    lib().orc_fly_camera(_p(f32(position)), np.float32(yaw_deg), np.float32(pitch_deg), np.float32(vfov_degrees),
                         np.float32(aperture), np.float32(focus), aspect, _p(cam))
    return cam


# ---------------------------------------------------------------------------------- sky
def sky_state_new(elevation, turbidity, albedo):
    st = np.zeros(33, np.float32)
    rc = lib().orc_sky_state_new(np.float32(elevation), np.float32(turbidity), _p(f32(albedo)), _p(st))
    return rc, st


def aligned_sky_state(turbidity=1.0, albedo=(1.0, 1.0, 1.0), zenith_deg=30.0, azimuth_deg=0.0):
    out = np.zeros(40, np.float32)
    rc = lib().orc_aligned_sky_state(np.float32(turbidity), _p(f32(albedo)), np.float32(zenith_deg), np.float32(azimuth_deg), _p(out))
    assert rc == 0, rc
    return out


def sky_state_radiance(state33, theta, gamma, channel):
    return np.float32(lib().orc_sky_state_radiance(_p(f32(state33)), np.float32(theta), np.float32(gamma), channel))


def wgsl_sky_radiance(sky40, theta, gamma, channel):
    return np.float32(lib().orc_wgsl_sky_radiance(_p(f32(sky40)), np.float32(theta), np.float32(gamma), channel))


_REF_BLUE_NOISE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libbluenoise_ref.so")


def ref_blue_noise_table():
    """The reference's own src/pt/blue_noise.c, compiled unmodified (oracle/_ref): (bytes[32768], width, height), or None."""
    if not os.path.exists(_REF_BLUE_NOISE_PATH):
        return None
    lib_ = C.CDLL(_REF_BLUE_NOISE_PATH)
    arr = (C.c_uint8 * 32768).in_dll(lib_, "blueNoiseValues")
    return (np.frombuffer(bytes(arr), np.uint8).copy(), C.c_size_t.in_dll(lib_, "blueNoiseWidth").value,
            C.c_size_t.in_dll(lib_, "blueNoiseHeight").value)


class RefSky:
    """The reference's own hw_skymodel.c, compiled unmodified (oracle/_ref)."""

    class Params(C.Structure):
        _fields_ = [("elevation", C.c_float), ("turbidity", C.c_float), ("albedo", C.c_float * 3)]

    class State(C.Structure):
        _fields_ = [("params", C.c_float * 27), ("sky_radiances", C.c_float * 3), ("solar_radiances", C.c_float * 3)]

    def __init__(self):
        self.lib = C.CDLL(_REF_PATH)
        self.lib.sky_state_radiance.restype = C.c_float
        self.lib.sky_state_radiance.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]

    @staticmethod
    def available():
        return os.path.exists(_REF_PATH)

    def state_new(self, elevation, turbidity, albedo):
        p = self.Params(elevation, turbidity, (C.c_float * 3)(*albedo))
        s = self.State()
        rc = self.lib.sky_state_new(C.byref(p), C.byref(s))
        return rc, np.array(list(s.params) + list(s.sky_radiances) + list(s.solar_radiances), np.float32)

    def radiance(self, state33, theta, gamma, channel):
        s = self.State()
        C.memmove(C.byref(s), f32(state33).ctypes.data, 33 * 4)
        return np.float32(self.lib.sky_state_radiance(C.byref(s), np.float32(theta), np.float32(gamma), channel))


# ---------------------------------------------------------------------------------- renderer
class OracleScene:
    """Holds numpy arrays alive and exposes the C struct."""

    def __init__(self, nodes, positions48, attrs80, tex_descs, texels, blue_noise=None):
        self.nodes = np.ascontiguousarray(nodes)
        self.positions = np.ascontiguousarray(positions48)
        self.attrs = np.ascontiguousarray(attrs80)
        self.tex_descs = np.ascontiguousarray(tex_descs, dtype=np.uint32)
        self.texels = np.ascontiguousarray(texels, dtype=np.uint32)
        self.blue_noise = blue_noise_table() if blue_noise is None else np.ascontiguousarray(blue_noise, np.uint8)
        self.c = Scene(self.nodes.ctypes.data, self.positions.ctypes.data, self.attrs.ctypes.data,
                       self.tex_descs.ctypes.data, self.texels.ctypes.data, self.texels.size, self.blue_noise.ctypes.data)


def make_render_params(width, height, cam19, spp, bounces, exposure, sky40):
    rp = RenderParams()
    rp.width, rp.height = width, height
    rp.camera[:] = [float(x) for x in f32(cam19)]
    rp.numSamplesPerPixel, rp.numBounces = spp, bounces
    rp.exposure = exposure
    rp.sky[:] = [float(x) for x in f32(sky40)]
    return rp


def render(scene, rp, first_frame, num_frames, x0=0, y0=0, x1=None, y1=None, image=None, accumulated_start=None):
    """num_frames render() calls from frameCount = first_frame.  accumulated_start: the accumulated sample count at that
    point (None: a fresh renderer, == first_frame); 0 after a setRenderParameters() change (reference_path_tracer.cpp:556-563)."""
    lib()
    x1 = rp.width if x1 is None else x1
    y1 = rp.height if y1 is None else y1
    if image is None:
        image = np.zeros((rp.height, rp.width, 4), np.float32)
    st = Stats()
    lib().orc_render_from(C.byref(scene.c), C.byref(rp), first_frame, first_frame if accumulated_start is None else accumulated_start, num_frames,
                          x0, y0, x1, y1, _p(image), C.byref(st))
    return image, st


def render_threads(scene, rp, first_frame, num_frames, x0, y0, x1, y1, num_threads, rows_per_block=1, image=None, accumulated_start=None):
    """render() on num_threads C-side threads (pthreads inside the oracle: rows dealt in static blocks, block b to thread
    b % num_threads) -- bench.py's all-cores cpu_baseline.  -> (image, stats, threads actually started)."""
    lib()
    if image is None:
        image = np.zeros((rp.height, rp.width, 4), np.float32)
    st = Stats()
    started = lib().orc_render_threads(C.byref(scene.c), C.byref(rp), first_frame, first_frame if accumulated_start is None else accumulated_start, num_frames,
                                       x0, y0, x1, y1, _p(image), C.byref(st), num_threads, rows_per_block)
    return image, st, started


def bvh_visualize_threads(nodes, tris, cam19, W, H, num_threads, rows_per_block=1):
    tris = f32(tris); cam19 = f32(cam19)
    nv = np.zeros(W * H, np.uint32)
    started = lib().orc_bvh_visualize_threads(_p(nodes), _p(tris), tris.shape[1], _p(cam19), W, H, _p(nv), num_threads, rows_per_block)
    return nv, started


def deferred_frames(scene, rp, num_frames, x0=0, y0=0, x1=None, y1=None):
    """The deferred-lighting variant (lighting + resolve pass over a primary-ray G-buffer), frames 0..num_frames-1 of a fresh
    renderer -> (last sample buffer (H,W,3), accumulation buffer (H,W,3), resolve output srgb (H,W,3), stats)."""
    x1 = rp.width if x1 is None else x1
    y1 = rp.height if y1 is None else y1
    sample = np.zeros((rp.height, rp.width, 3), np.float32); accum = np.zeros_like(sample); srgb = np.zeros_like(sample)
    st = Stats()
    for f in range(num_frames):
        lib().orc_deferred_frame(C.byref(scene.c), C.byref(rp), f, x0, y0, x1, y1, _p(sample), _p(accum), _p(srgb), C.byref(st))
    return sample, accum, srgb, st


def pixel_sample(scene, rp, x, y, frame):
    rgb = np.zeros(3, np.float32)
    st = Stats()
    lib().orc_pixel_sample(C.byref(scene.c), C.byref(rp), x, y, frame, _p(rgb), C.byref(st))
    return rgb, st


def wgsl_camera_ray(rp, x, y, frame, blue_noise=None):
    bn = blue_noise_table() if blue_noise is None else blue_noise
    r = np.zeros(6, np.float32)
    lib().orc_wgsl_camera_ray(C.byref(rp), _p(bn), x, y, frame, _p(r))
    return r


def animated_blue_noise(x, y, frame, spp, blue_noise=None):
    bn = blue_noise_table() if blue_noise is None else blue_noise
    o = np.zeros(2, np.float32)
    lib().orc_animated_blue_noise(_p(bn), x, y, frame, spp, _p(o))
    return o


def texture_lookup(scene, desc_idx, u, v):
    rgb = np.zeros(3, np.float32)
    lib().orc_texture_lookup(C.byref(scene.c), desc_idx, np.float32(u), np.float32(v), _p(rgb))
    return rgb


def pixar_onb(n):
    n = f32(n); u = np.zeros(3, np.float32); v = np.zeros(3, np.float32)
    lib().orc_pixar_onb(_p(n), _p(u), _p(v))
    return u, v


def direction_in_cone(ux, uy, cos_theta_max):
    d = np.zeros(3, np.float32)
    lib().orc_direction_in_cone(C.c_float(ux), C.c_float(uy), C.c_float(cos_theta_max), _p(d))
    return d


def direction_in_cosine_weighted_hemisphere(ux, uy):
    d = np.zeros(3, np.float32)
    lib().orc_direction_in_cosine_weighted_hemisphere(C.c_float(ux), C.c_float(uy), _p(d))
    return d


def offset_ray(p, n):
    p = f32(p); n = f32(n); o = np.zeros(3, np.float32)
    lib().orc_offset_ray(_p(p), _p(n), _p(o))
    return o


def tonemap(image, acc, exposure):
    img = f32(image).reshape(-1, 4)
    out = np.zeros((img.shape[0], 3), np.float32)
    lib().orc_tonemap(_p(img), img.shape[0], acc, np.float32(exposure), _p(out))
    return out
