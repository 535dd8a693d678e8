"""rayfinder_amd -- MI355X-native offline path-tracing core (thin Python host over the C ABI).

The product is librayfinder_amd.so (C++20 host + hand-written gfx950 HIP kernels, see
include/rayfinder_amd.h).  This package only marshals numpy arrays across the C ABI so that tests,
bench.py and torch.distributed plumbing can drive it; it mirrors the reference's seams
(src/pt/reference_path_tracer.hpp:59-76, src/pt-format/pt_format.hpp:18-43,
src/common/bvh.hpp:33, src/common/camera.hpp:24-34) by name.  No CPU fallback exists.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import RayfinderError, check, lib  # noqa: F401

NODE_DTYPE = np.dtype([("min", "<f4", 3), ("pad0", "<f4"), ("max", "<f4", 3), ("pad1", "<f4"),
                       ("trianglesOffset", "<u4"), ("secondChildOffset", "<u4"),
                       ("triangleCount", "<u4"), ("splitAxis", "<u4")])
TILE = 32


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def version():
    return lib.rf_version().decode()


# ------------------------------------------------------------------------------------- camera / sky
def camera_to_array(cam):
    return np.frombuffer(bytes(cam), dtype=np.float32).copy()


def camera_from_array(a):
    cam = _ffi.Camera()
    C.memmove(C.byref(cam), _f32(a).ctypes.data, 76)
    return cam


def create_camera(origin, look_at, aperture, focus_distance, vfov_radians, aspect_ratio):
    """createCamera (src/common/camera.cpp:7-42) -> _ffi.Camera"""
    cam = _ffi.Camera()
    check(lib.rf_create_camera(_ptr(_f32(origin)), _ptr(_f32(look_at)), aperture, focus_distance, vfov_radians, aspect_ratio, C.byref(cam)))
    return cam


def fly_camera(width, height, position=(1.22, 1.25, -1.25), yaw_degrees=129.64, pitch_degrees=-13.73, vfov_degrees=70.0,
               aperture=0.0, focus_distance=10.0):
    """The interactive app's default camera (fly_camera_controller.hpp:47-52, main.cpp:49,314)."""
    cam = _ffi.Camera()
    aspect = np.float32(np.float32(width) / np.float32(height))
    check(lib.rf_fly_camera(_ptr(_f32(position)), yaw_degrees, pitch_degrees, vfov_degrees, aperture, focus_distance, aspect, C.byref(cam)))
    return cam


def bvh_visualizer_camera(nodes, aspect_ratio):
    cam = _ffi.Camera()
    check(lib.rf_bvh_visualizer_camera(_ptr(np.ascontiguousarray(nodes[:1])), np.float32(aspect_ratio), C.byref(cam)))
    return cam


def make_sky(turbidity=1.0, albedo=(1.0, 1.0, 1.0), sun_zenith_degrees=30.0, sun_azimuth_degrees=0.0):
    return _ffi.Sky(turbidity, (C.c_float * 3)(*albedo), sun_zenith_degrees, sun_azimuth_degrees)


def sky_state_new(elevation, turbidity, albedo):
    st = np.zeros(33, np.float32)
    rc = lib.rf_sky_state_new(np.float32(elevation), np.float32(turbidity), _ptr(_f32(albedo)), _ptr(st))
    return rc, st


def sky_state_radiance(state33, theta, gamma, channel):
    return np.float32(lib.rf_sky_state_radiance(_ptr(_f32(state33)), np.float32(theta), np.float32(gamma), channel))


def aligned_sky_state(sky):
    out = np.zeros(40, np.float32)
    check(lib.rf_aligned_sky_state(C.byref(sky), _ptr(out)))
    return out


def make_render_parameters(width, height, camera, spp=128, bounces=4, sky=None, exposure=1.0):
    return _ffi.RenderParameters(width, height, camera, spp, bounces, sky if sky is not None else make_sky(), exposure)


# ------------------------------------------------------------------------------------- BVH (host)
def build_bvh(positions36):
    """buildBvh (src/common/bvh.hpp:33) -> (nodes[NODE_DTYPE], triangleIndices[u64], depth)"""
    tris = _f32(positions36).reshape(-1, 9)
    n = tris.shape[0]
    nodes = np.zeros(max(2 * n, 1), dtype=NODE_DTYPE)
    idx = np.zeros(n, np.uint64)
    cnt = C.c_uint64(0)
    depth = C.c_int32(0)
    check(lib.rf_build_bvh(_ptr(tris), n, _ptr(nodes), C.byref(cnt), _ptr(idx), C.byref(depth)))
    return nodes[:cnt.value].copy(), idx, depth.value


# ------------------------------------------------------------------------------------- BVH queries on the host
INTERSECTION_DTYPE = np.dtype([("p", "<f4", 3), ("t", "<f4"), ("triangle", "<u4"), ("u", "<f4"), ("v", "<f4")])
BVH_STATS_DTYPE = np.dtype([("nodes_visited", "<u4"), ("triangle_tests", "<u4"), ("stack_high_water", "<u4")])


def _positions(tris):
    """(N, 9) f32 Positions (36 B) or (N, 12) PositionAttribute (48 B) -> (array, stride)"""
    tris = _f32(tris)
    tris = tris.reshape(-1, 12 if tris.ndim == 2 and tris.shape[1] == 12 else 9)
    return tris, tris.shape[1] * 4


def intersect_bvh(ray6, nodes, positions, t_max):
    """rayIntersectBvh (src/common/ray_intersection.hpp:43-49) on the host -> (hit: bool, intersection record, stats record)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    tris, stride = _positions(positions)
    out = np.zeros(1, INTERSECTION_DTYPE); st = np.zeros(1, BVH_STATS_DTYPE)
    hit = C.c_int(0)
    check(lib.rf_intersect_bvh(_ptr(_f32(ray6)), _ptr(nodes), nodes.shape[0], _ptr(tris), stride, tris.shape[0], np.float32(t_max), _ptr(out), _ptr(st), C.byref(hit)))
    return bool(hit.value), out[0], st[0]


def intersect_bvh_batch(rays6, nodes, positions, t_max, threads=0):
    """-> dict(hit u8, tri, t, uv, p, nodesVisited, triTests, stackHigh), one entry per ray; host threads, no GPU."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    tris, stride = _positions(positions)
    rays = _f32(rays6).reshape(-1, 6)
    n = rays.shape[0]
    hit = np.zeros(n, np.uint8); out = np.zeros(n, INTERSECTION_DTYPE); st = np.zeros(n, BVH_STATS_DTYPE)
    check(lib.rf_intersect_bvh_batch(_ptr(rays), n, _ptr(nodes), nodes.shape[0], _ptr(tris), stride, tris.shape[0], np.float32(t_max), threads, _ptr(hit), _ptr(out), _ptr(st)))
    return dict(hit=hit, tri=out["triangle"].copy(), t=out["t"].copy(), uv=np.stack([out["u"], out["v"]], axis=1), p=out["p"].copy(),
                nodesVisited=st["nodes_visited"].copy(), triTests=st["triangle_tests"].copy(), stackHigh=st["stack_high_water"].copy())


def bvh_visualizer_pass(camera, width, height, nodes, positions, threads=0, row_begin=0, row_end=None):
    """The bvh-visualizer pixel loop (src/bvh-visualizer/main.cpp:60-78) on the host; the CPU twin of
    ReferencePathTracer.trace_primary_stats -> dict(nodesVisited, hit, t, triTests), each width*height."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    tris, stride = _positions(positions)
    n = width * height
    nv = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); tt = np.zeros(n, np.uint32)
    check(lib.rf_bvh_visualizer_pass(C.byref(camera), width, height, row_begin, height if row_end is None else row_end, _ptr(nodes), nodes.shape[0], _ptr(tris), stride,
                                     tris.shape[0], threads, _ptr(nv), _ptr(hit), _ptr(t), _ptr(tt)))
    return dict(nodesVisited=nv, hit=hit, t=t, triTests=tt)


def bvh_visualizer_grey(nodes_visited):
    """src/bvh-visualizer/main.cpp:73-76: grey level u32(min(0.01f * nodesVisited, 1) * 255) per pixel (f32 arithmetic)."""
    x = np.float32(0.01) * np.asarray(nodes_visited).astype(np.float32)
    return (np.minimum(x, np.float32(1.0)) * np.float32(255.0)).astype(np.uint32).astype(np.uint8)


def check_wide_layouts(nodes):
    """rf_check_wide_layouts: the render path's record layouts of a flattened tree decode to the same child planes (host only).
    -> {"regular": bool, "compact": bool, "hot": bool}; raises on a mismatch."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    flags = C.c_uint32(0)
    check(lib.rf_check_wide_layouts(_ptr(nodes), nodes.shape[0], C.byref(flags)))
    return {"regular": bool(flags.value & 1), "compact": bool(flags.value & 2), "hot": bool(flags.value & 4), "quad": bool(flags.value & 8), "quad_half": bool(flags.value & 16), "quad_local": bool(flags.value & 32), "oct": bool(flags.value & 64)}


def wide_layout_stats(nodes):
    """Flags of check_wide_layouts plus the surface-area ratio of the half-precision quad boxes (the renderer uses them by default
    closest-hit: half-precision records up to 1.075, local-grid beyond; shadow: local-grid up to 1.10, exact beyond)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    flags = C.c_uint32(0); ratio = C.c_float(0.0)
    check(lib.rf_wide_layout_stats(_ptr(nodes), nodes.shape[0], C.byref(flags), C.byref(ratio)))
    return {"flags": flags.value, "quad_half_area_ratio": float(ratio.value)}


def texture_from_memory(data):
    """Texture::fromMemory (texture.cpp:12-54): PNG / JPEG bytes -> (BGRA u32 texels [h*w], width, height)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(0), C.c_uint32(0)
    check(lib.rf_texture_from_memory(_ptr(buf), buf.size, C.byref(w), C.byref(h), None))
    px = np.zeros(w.value * h.value, np.uint32)
    check(lib.rf_texture_from_memory(_ptr(buf), buf.size, C.byref(w), C.byref(h), _ptr(px)))
    return px, w.value, h.value


def set_bake_bvh_builder(gpu_device=None):
    """BVH builder of PtFormat.from_gltf / from_triangles: None = host (default), int = GPU builder on that device."""
    check(lib.rf_pt_format_set_bvh_builder(-1 if gpu_device is None else int(gpu_device)))


def build_bvh_gpu(positions36, device_ordinal=0):
    """GPU build of the same tree (rf_bvh_gpu.hip) -> (nodes, triangleIndices, depth, build_ms)."""
    tris = _f32(positions36).reshape(-1, 9)
    n = tris.shape[0]
    nodes = np.zeros(max(2 * n, 1), dtype=NODE_DTYPE)
    idx = np.zeros(n, np.uint64)
    cnt = C.c_uint64(0)
    depth = C.c_int32(0)
    ms = C.c_float(0)
    check(lib.rf_build_bvh_gpu(_ptr(tris), n, _ptr(nodes), C.byref(cnt), _ptr(idx), C.byref(depth), device_ordinal, C.byref(ms)))
    return nodes[:cnt.value].copy(), idx, depth.value, ms.value


def tiles_for_rank(width, height, rank, world_size):
    n = C.c_uint32(0)
    check(lib.rf_tiles_for_rank(width, height, rank, world_size, None, C.byref(n)))
    tiles = np.zeros(n.value, np.uint32)
    check(lib.rf_tiles_for_rank(width, height, rank, world_size, _ptr(tiles), C.byref(n)))
    return tiles


def untile(compact, tile_ids, width, height, image=None):
    """compact: (numTiles*1024, 4) f32 tile-major -> (H, W, 4) row-major (other pixels untouched)."""
    compact = _f32(compact).reshape(-1, 4)
    tile_ids = np.ascontiguousarray(tile_ids, np.uint32)
    if image is None:
        image = np.zeros((height, width, 4), np.float32)
    check(lib.rf_untile(_ptr(compact), _ptr(tile_ids), tile_ids.size, width, height, _ptr(image)))
    return image


# ------------------------------------------------------------------------------------- .pt files
class PtFormat:
    """nlrs::PtFormat (src/pt-format/pt_format.hpp:18-43) held by the native library."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.rf_pt_format_destroy(self._h)
            self._h = None

    @classmethod
    def from_gltf(cls, path):
        h = C.c_void_p()
        check(lib.rf_pt_format_from_gltf(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        check(lib.rf_pt_format_load(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def deserialize(cls, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        h = C.c_void_p()
        check(lib.rf_pt_format_deserialize(_ptr(buf), buf.size, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_triangles(cls, positions36, normals36, tex_coords24, texture_indices, textures):
        """textures: list of (pixels u32 array, width, height)"""
        p = _f32(positions36).reshape(-1, 9)
        n = _f32(normals36).reshape(-1, 9)
        t = _f32(tex_coords24).reshape(-1, 6)
        ti = np.ascontiguousarray(texture_indices, np.uint32)
        keep = [np.ascontiguousarray(px, np.uint32) for px, _, _ in textures]
        arr = (_ffi.Texture * max(len(textures), 1))()
        for i, (px, w, h) in enumerate(textures):
            arr[i] = _ffi.Texture(keep[i].ctypes.data, w, h)
        h = C.c_void_p()
        check(lib.rf_pt_format_from_triangles(_ptr(p), _ptr(n), _ptr(t), _ptr(ti), p.shape[0], arr, len(textures), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        check(lib.rf_pt_format_save(self._h, str(path).encode()))

    def serialize(self):
        size = C.c_uint64(0)
        check(lib.rf_pt_format_serialize(self._h, None, C.byref(size)))
        buf = np.zeros(size.value, np.uint8)
        check(lib.rf_pt_format_serialize(self._h, _ptr(buf), C.byref(size)))
        return buf.tobytes()

    def view(self):
        v = _ffi.PtFormatView()
        check(lib.rf_pt_format_view_get(self._h, C.byref(v)))
        return v

    def _array(self, ptr, count, dtype, cols=None):
        if count == 0:
            return np.zeros((0,) if cols is None else (0, cols), dtype)
        dt = np.dtype(dtype)
        n = count * (cols or 1)
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * dt.itemsize,)).view(dt).copy()
        return a if cols is None else a.reshape(count, cols)

    def arrays(self):
        """Copies of every array as numpy (names as in the reference struct)."""
        v = self.view()
        out = dict(
            bvhNodes=self._array(v.bvh_nodes, v.num_bvh_nodes, NODE_DTYPE),
            bvhPositionAttributes=self._array(v.bvh_position_attributes, v.num_bvh_position_attributes, np.float32, 9),
            trianglePositionAttributes=self._array(v.triangle_position_attributes, v.num_triangle_position_attributes, np.float32, 12),
            triangleVertexAttributes=self._array(v.triangle_vertex_attributes, v.num_triangle_vertex_attributes, np.float32, 20),
            vertexPositions=self._array(v.vertex_positions, v.num_vertex_positions, np.float32, 4),
            vertexNormals=self._array(v.vertex_normals, v.num_vertex_normals, np.float32, 4),
            vertexTexCoords=self._array(v.vertex_tex_coords, v.num_vertex_tex_coords, np.float32, 2),
            vertexIndices=self._array(v.vertex_indices, v.num_vertex_indices, np.uint32),
            modelVertexPositions=self._array(v.model_vertex_positions, v.num_model_vertex_positions, np.uint64, 2),
            modelVertexNormals=self._array(v.model_vertex_normals, v.num_model_vertex_normals, np.uint64, 2),
            modelVertexTexCoords=self._array(v.model_vertex_tex_coords, v.num_model_vertex_tex_coords, np.uint64, 2),
            modelVertexIndices=self._array(v.model_vertex_indices, v.num_model_vertex_indices, np.uint64, 2),
            modelBaseColorTextureIndices=self._array(v.model_base_color_texture_indices, v.num_model_base_color_texture_indices, np.uint32),
        )
        out["baseColorTextures"] = [self.texture(i) for i in range(v.num_textures)]
        return out

    def texture(self, i):
        t = _ffi.Texture()
        check(lib.rf_pt_format_texture(self._h, i, C.byref(t)))
        px = self._array(t.pixels, t.width * t.height, np.uint32)
        return px, t.width, t.height

    def scene(self):
        """nlrs::Scene spans (src/pt/main.cpp:150-157); valid while this PtFormat is alive."""
        v = self.view()
        textures = (_ffi.Texture * max(int(v.num_textures), 1))()
        sc = _ffi.Scene()
        check(lib.rf_pt_format_scene(self._h, C.byref(sc), textures))
        sc._keepalive = (textures, self)
        return sc


def scene_from_arrays(nodes, positions48, attrs80, textures):
    """Build an rf_scene from numpy arrays; textures = list of (pixels u32, w, h)."""
    nodes = np.ascontiguousarray(nodes)
    pos = _f32(positions48).reshape(-1, 12)
    att = np.ascontiguousarray(attrs80, dtype=np.float32).reshape(-1, 20)
    keep = [np.ascontiguousarray(px, np.uint32) for px, _, _ in textures]
    arr = (_ffi.Texture * max(len(textures), 1))()
    for i, (px, w, h) in enumerate(textures):
        arr[i] = _ffi.Texture(keep[i].ctypes.data, w, h)
    sc = _ffi.Scene(nodes.ctypes.data, nodes.shape[0], pos.ctypes.data, att.ctypes.data, pos.shape[0], arr, len(textures))
    sc._keepalive = (nodes, pos, att, keep, arr)
    return sc


# ------------------------------------------------------------------------------------- multi-GPU frame exchange
RF_GATHER_LOOPBACK = 1


def gather_layout(width, height, world_size):
    """Staging layout of the frame-end gather: (rank_first_tile[world+1], tile_slot[tiles], tile_owner[tiles])."""
    n = ((width + 31) // 32) * ((height + 31) // 32)
    first = np.zeros(world_size + 1, np.uint32); slot = np.zeros(n, np.uint32); owner = np.zeros(n, np.uint32)
    check(lib.rf_gather_layout(width, height, world_size, _ptr(first), _ptr(slot), _ptr(owner)))
    return first, slot, owner


def gather_plan(width, height, world_size, rank, root=0, loopback=False):
    """The point-to-point operations `rank` posts for a frame-end gather to `root` (what rf_renderer_gather_frame executes):
    (n, 4) u32 rows {is_send, peer, offset_tiles, count_tiles}.  Host arithmetic, no GPU."""
    n = C.c_uint32(0)
    flags = RF_GATHER_LOOPBACK if loopback else 0
    check(lib.rf_gather_plan(width, height, world_size, rank, root, flags, None, C.byref(n)))
    ops = np.zeros((n.value, 4), np.uint32)
    check(lib.rf_gather_plan(width, height, world_size, rank, root, flags, _ptr(ops), C.byref(n)))
    return ops


def device_count():
    """hipGetDeviceCount as the library sees it (0 without a GPU)."""
    n = C.c_int32(0)
    check(lib.rf_device_count(C.byref(n)))
    return n.value


def comm_unique_id():
    """ncclGetUniqueId: 128 bytes that rank 0 hands to the other ranks before TileComm(...)."""
    buf = np.zeros(128, np.uint8)
    check(lib.rf_comm_unique_id(_ptr(buf)))
    return buf.tobytes()


class TileComm:
    """One RCCL communicator per rank (one process per GPU); collective constructor."""

    def __init__(self, unique_id, rank, world_size, device_ordinal=0):
        buf = np.frombuffer(bytes(unique_id), np.uint8).copy()
        assert buf.size == 128
        self._h = C.c_void_p()
        self.rank, self.world_size = rank, world_size
        check(lib.rf_comm_create(_ptr(buf), rank, world_size, device_ordinal, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.rf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def info(self):
        """What RCCL reports for this communicator: dict(rccl_ranks, rccl_rank, device)."""
        n, r, d = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        check(lib.rf_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return dict(rccl_ranks=n.value, rccl_rank=r.value, device=d.value)

    def local_transport(self):
        """True: this communicator runs on the local TEST transport (RF_COMM_TRANSPORT=local when its id was made), not on RCCL."""
        v = C.c_uint32(0)
        check(lib.rf_comm_transport(self._h, C.byref(v)))
        return bool(v.value)

    def last_exchange_ms(self):
        """Device time of this rank's last gather_frame (HIP events around the sends / receives + the root's un-tile); -1 before the first."""
        v = C.c_double(-1.0)
        check(lib.rf_comm_last_exchange_ms(self._h, C.byref(v)))
        return v.value

    def read_frame(self, renderer, width, height):
        """Root: the gathered row-major (H, W, 4) float image."""
        img = np.zeros((height, width, 4), np.float32)
        check(lib.rf_comm_read_frame(self._h, renderer._h, _ptr(img)))
        return img

    def all_reduce_max(self, value, renderer=None):
        v = C.c_double(value)
        check(lib.rf_comm_all_reduce_max(self._h, renderer._h if renderer is not None else None, C.byref(v)))
        return v.value


# ------------------------------------------------------------------------------------- renderer
class ReferencePathTracer:
    """Host-side mirror of nlrs::ReferencePathTracer (src/pt/reference_path_tracer.hpp:59-76).

    ctor copies the scene into HBM; set_render_parameters resets accumulation on any change;
    render(n) advances n frames (one sample each) without host round trips.
    """

    def __init__(self, render_params, scene, max_width=0, max_height=0, device_ordinal=0, max_paths_in_flight=0):
        desc = _ffi.RendererDescriptor(render_params, max_width, max_height, device_ordinal, max_paths_in_flight)
        self._h = C.c_void_p()
        self._params = render_params
        check(lib.rf_renderer_create(C.byref(desc), C.byref(scene), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:   # lib is None once the interpreter is shutting down
            lib.rf_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_render_parameters(self, params):
        check(lib.rf_renderer_set_render_parameters(self._h, C.byref(params)))
        self._params = params

    def render(self, num_frames=1):
        check(lib.rf_renderer_render(self._h, num_frames))

    def synchronize(self):
        check(lib.rf_renderer_synchronize(self._h))

    def average_renderpass_duration_ms(self):
        return lib.rf_renderer_average_renderpass_duration_ms(self._h)

    def render_progress_percentage(self):
        return lib.rf_renderer_render_progress_percentage(self._h)

    def read_accumulation(self):
        """-> ((H, W, 4) f32 sum image, accumulated sample count)"""
        img = np.zeros((self._params.height, self._params.width, 4), np.float32)
        acc = C.c_uint32(0)
        check(lib.rf_renderer_read_accumulation(self._h, _ptr(img), C.byref(acc)))
        return img, acc.value

    def read_tonemapped(self):
        img = np.zeros((self._params.height, self._params.width), np.uint32)
        check(lib.rf_renderer_read_tonemapped(self._h, _ptr(img)))
        return img

    # deferred-lighting variant (nlrs::DeferredRenderer's lighting + resolve passes over a primary-ray G-buffer)
    def render_deferred(self, num_frames=1):
        check(lib.rf_renderer_render_deferred(self._h, num_frames))

    def reset_deferred(self):
        check(lib.rf_renderer_reset_deferred(self._h))

    def read_deferred(self):
        """-> (sample buffer (H,W,3), accumulation buffer (H,W,3), BGRA8 (H,W), frames rendered)"""
        w, h = self._params.width, self._params.height
        sample = np.zeros((h, w, 3), np.float32); accum = np.zeros((h, w, 3), np.float32); bgra = np.zeros((h, w), np.uint32)
        n = C.c_uint32(0)
        check(lib.rf_renderer_read_deferred(self._h, _ptr(sample), _ptr(accum), _ptr(bgra), C.byref(n)))
        return sample, accum, bgra, n.value

    def set_counting(self, enabled):
        check(lib.rf_renderer_set_counting(self._h, int(enabled)))

    def set_timing(self, enabled):
        check(lib.rf_renderer_set_timing(self._h, int(enabled)))

    def set_option(self, name, value):
        check(lib.rf_renderer_set_option(self._h, name.encode(), int(value)))

    def reset_stats(self):
        check(lib.rf_renderer_reset_stats(self._h))

    def stats(self):
        s = _ffi.Stats()
        check(lib.rf_renderer_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    LAYOUT_NAMES = ("binary", "compact", "hot", "quad", "quad_half", "quad_local", "oct", "scalar", "packet")

    def layout_info(self, bounces=8):
        """rf_renderer_layout_info: the record layout this renderer reads in the closest-hit / any-hit launch of bounce 1..n (what it picked by itself for the
        scene + any option set since), whether that any-hit launch starts at the occluder cache, and the per-scene parameters behind the choice."""
        raw = np.zeros(48 + 4 + 2 + 2, np.uint32)      # 3 x 16 words, 4 words, 2 floats, one u64
        check(lib.rf_renderer_layout_info(self._h, _ptr(raw)))
        n = min(bounces, 16)
        return dict(closest=[self.LAYOUT_NAMES[v] for v in raw[:n]], shadow=[self.LAYOUT_NAMES[v] for v in raw[16:16 + n]], shadow_cached=[bool(v) for v in raw[32:32 + n]],
                    occluder_hint_levels=int(raw[48]), shadow_first_look_from_bounce=int(raw[49]), dense_leaf_min=int(raw[50]), legacy_layouts_compiled=bool(raw[51]),
                    quad_half_area_ratio=float(raw[52:53].view(np.float32)[0]), tree_bytes=int(raw[54:56].view(np.uint64)[0]))

    def memory_info(self):
        """Device memory held by the handle: dict(path_state_bytes, paths_allocated, max_paths_per_batch, scene_bytes)."""
        v = [C.c_uint64(0) for _ in range(4)]
        check(lib.rf_renderer_memory_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("path_state_bytes", "paths_allocated", "max_paths_per_batch", "scene_bytes"), (x.value for x in v)))

    def bounce_stats(self):
        """Per-bounce queue occupancy (rays traced) and traversal kernel ms since the last reset."""
        cap = 32
        cr = np.zeros(cap, np.uint64); sr = np.zeros(cap, np.uint64); mc = np.zeros(cap, np.float64); ms = np.zeros(cap, np.float64)
        n = C.c_uint32(0)
        check(lib.rf_renderer_get_bounce_stats(self._h, cap, _ptr(cr), _ptr(sr), _ptr(mc), _ptr(ms), C.byref(n)))
        k = n.value
        return dict(closest_rays=cr[:k], shadow_rays=sr[:k], ms_closest=mc[:k], ms_shadow=ms[:k])

    # multi-GPU tile sharding
    def set_tile_shard(self, rank, world_size):
        check(lib.rf_renderer_set_tile_shard(self._h, rank, world_size))

    def shard_tiles(self):
        n = C.c_uint32(0)
        check(lib.rf_renderer_shard_tiles(self._h, None, C.byref(n)))
        tiles = np.zeros(n.value, np.uint32)
        check(lib.rf_renderer_shard_tiles(self._h, _ptr(tiles), C.byref(n)))
        return tiles

    def accumulation_device_buffer(self):
        p = C.c_void_p()
        n = C.c_uint64(0)
        check(lib.rf_renderer_accumulation_device_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def gather_frame(self, comm, root=0, loopback=False):
        """Frame-end RCCL exchange (collective; enqueued on the handle's stream).  Root: device pointer of the
        row-major W*H float4 image; other ranks: None."""
        p = C.c_void_p()
        check(lib.rf_renderer_gather_frame(self._h, comm._h, root, RF_GATHER_LOOPBACK if loopback else 0, C.byref(p)))
        return p.value

    def tonemap_device_image(self, device_ptr, width, height, samples):
        out = np.zeros((height, width), np.uint32)
        check(lib.rf_renderer_tonemap_device_image(self._h, C.c_void_p(device_ptr), width * height, samples, _ptr(out)))
        return out

    def bind_accumulation_buffer(self, device_ptr, nbytes):
        check(lib.rf_renderer_bind_accumulation_buffer(self._h, C.c_void_p(device_ptr), nbytes))

    # BVH queries
    def trace_primary_stats(self, camera, width, height):
        n = width * height
        nv = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8); t = np.zeros(n, np.float32); tt = np.zeros(n, np.uint32)
        check(lib.rf_renderer_trace_primary_stats(self._h, C.byref(camera), width, height, _ptr(nv), _ptr(hit), _ptr(t), _ptr(tt)))
        return dict(nodesVisited=nv, hit=hit, t=t, triTests=tt)

    def intersect_rays(self, rays6, t_max):
        rays = _f32(rays6).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(tri=np.zeros(n, np.uint32), t=np.zeros(n, np.float32), uv=np.zeros((n, 2), np.float32),
                   p=np.zeros((n, 3), np.float32), nodesVisited=np.zeros(n, np.uint32), triTests=np.zeros(n, np.uint32))
        check(lib.rf_renderer_intersect_rays(self._h, _ptr(rays), n, np.float32(t_max), _ptr(out["tri"]), _ptr(out["t"]), _ptr(out["uv"]),
                                             _ptr(out["p"]), _ptr(out["nodesVisited"]), _ptr(out["triTests"])))
        out["hit"] = (out["tri"] != 0xFFFFFFFF).astype(np.uint8)
        return out

    def occluded_rays(self, rays6, t_max):
        rays = _f32(rays6).reshape(-1, 6)
        vis = np.zeros(rays.shape[0], np.float32)
        check(lib.rf_renderer_occluded_rays(self._h, _ptr(rays), rays.shape[0], np.float32(t_max), _ptr(vis)))
        return vis
