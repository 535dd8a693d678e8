"""ctypes declarations for librayfinder_amd.so (the C ABI in include/rayfinder_amd.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails
loudly if the shared object has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `make -C rayfinder_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYFINDER_AMD_LIB: an experiment build of the same library (make EXP=... LIBNAME=...), for A/B measurements
LIB_PATH = os.environ.get("RAYFINDER_AMD_LIB") or os.path.join(_HERE, "librayfinder_amd.so")
if os.environ.get("RAYFINDER_AMD_LIB"):
    import sys
    print(f"[rayfinder_amd] RAYFINDER_AMD_LIB overrides the packaged library: loading {LIB_PATH}", file=sys.stderr, flush=True)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first (make -C rayfinder_amd/csrc). "
        "rayfinder_amd has no CPU fallback.")

# Load order: the PyTorch wheel bundles its own HIP + HSA runtimes (torch/lib/libamdhip64.so, no
# SONAME), this library links ROCm's (libamdhip64.so.7).  Both can live in one process, but only if
# PyTorch's copies are mapped first -- with this library first, whichever runtime initialises second
# reports "no ROCm-capable device" (seen on the MI355X box: `pytest tests/test_gpu_parity.py` alone
# failed while `pytest tests` passed, because another test module imported torch earlier).  So when
# torch is installed it is imported before the dlopen; the C++ CLI tools never see PyTorch at all.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass
lib = C.CDLL(LIB_PATH)

RF_OK = 0
RF_ERROR_INVALID_ARGUMENT = 1
RF_ERROR_RUNTIME = 2
RF_ERROR_NO_DEVICE = 3
RF_ERROR_OUT_OF_RANGE = 4


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("lens_radius", C.c_float)]


class Sky(C.Structure):
    _fields_ = [("turbidity", C.c_float), ("albedo", C.c_float * 3), ("sun_zenith_degrees", C.c_float),
                ("sun_azimuth_degrees", C.c_float)]


class RenderParameters(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("camera", Camera),
                ("num_samples_per_pixel", C.c_uint32), ("num_bounces", C.c_uint32), ("sky", Sky), ("exposure", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("bvh_nodes", C.c_void_p), ("num_bvh_nodes", C.c_uint64), ("position_attributes", C.c_void_p),
                ("vertex_attributes", C.c_void_p), ("num_triangles", C.c_uint64), ("base_color_textures", C.POINTER(Texture)),
                ("num_textures", C.c_uint64)]


class RendererDescriptor(C.Structure):
    _fields_ = [("render_params", RenderParameters), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("device_ordinal", C.c_int32), ("max_paths_in_flight", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("closest_node_visits", C.c_uint64), ("closest_triangle_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_triangle_tests", C.c_uint64), ("paths", C.c_uint64),
                ("stack_high_water", C.c_uint32), ("batch_samples_used", C.c_uint32),
                ("ms_raygen", C.c_double), ("ms_closest", C.c_double), ("ms_shade", C.c_double), ("ms_shadow", C.c_double),
                ("ms_accumulate", C.c_double),
                ("launches_raygen", C.c_uint32), ("launches_closest", C.c_uint32), ("launches_shade", C.c_uint32),
                ("launches_shadow", C.c_uint32), ("launches_accumulate", C.c_uint32), ("batches_traced", C.c_uint32),
                ("closest_record_fetches", C.c_uint64), ("shadow_record_fetches", C.c_uint64),
                ("abandoned_rays", C.c_uint64), ("scalar_redo_rays", C.c_uint64), ("shadow_rays_hint_answered", C.c_uint64), ("shadow_rays_self_answered", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class PtFormatView(C.Structure):
    _fields_ = [("bvh_nodes", C.c_void_p), ("num_bvh_nodes", C.c_uint64),
                ("bvh_position_attributes", C.c_void_p), ("num_bvh_position_attributes", C.c_uint64),
                ("triangle_position_attributes", C.c_void_p), ("num_triangle_position_attributes", C.c_uint64),
                ("triangle_vertex_attributes", C.c_void_p), ("num_triangle_vertex_attributes", C.c_uint64),
                ("vertex_positions", C.c_void_p), ("num_vertex_positions", C.c_uint64),
                ("vertex_normals", C.c_void_p), ("num_vertex_normals", C.c_uint64),
                ("vertex_tex_coords", C.c_void_p), ("num_vertex_tex_coords", C.c_uint64),
                ("vertex_indices", C.c_void_p), ("num_vertex_indices", C.c_uint64),
                ("model_vertex_positions", C.c_void_p), ("num_model_vertex_positions", C.c_uint64),
                ("model_vertex_normals", C.c_void_p), ("num_model_vertex_normals", C.c_uint64),
                ("model_vertex_tex_coords", C.c_void_p), ("num_model_vertex_tex_coords", C.c_uint64),
                ("model_vertex_indices", C.c_void_p), ("num_model_vertex_indices", C.c_uint64),
                ("model_base_color_texture_indices", C.c_void_p), ("num_model_base_color_texture_indices", C.c_uint64),
                ("num_textures", C.c_uint64)]


# Every symbol include/rayfinder_amd.h declares, with its signature.
SIGNATURES = {
    "rf_last_error_message": (C.c_char_p, []),
    "rf_version": (C.c_char_p, []),
    "rf_renderer_create": (C.c_int, [C.POINTER(RendererDescriptor), C.POINTER(Scene), C.POINTER(C.c_void_p)]),
    "rf_renderer_destroy": (None, [C.c_void_p]),
    "rf_renderer_set_render_parameters": (C.c_int, [C.c_void_p, C.POINTER(RenderParameters)]),
    "rf_renderer_render": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rf_renderer_synchronize": (C.c_int, [C.c_void_p]),
    "rf_renderer_average_renderpass_duration_ms": (C.c_float, [C.c_void_p]),
    "rf_renderer_render_progress_percentage": (C.c_float, [C.c_void_p]),
    "rf_renderer_read_accumulation": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_renderer_read_tonemapped": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rf_renderer_render_deferred": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rf_renderer_reset_deferred": (C.c_int, [C.c_void_p]),
    "rf_renderer_read_deferred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_renderer_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "rf_renderer_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rf_renderer_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "rf_renderer_reset_stats": (C.c_int, [C.c_void_p]),
    "rf_renderer_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rf_renderer_set_tile_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rf_renderer_shard_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_renderer_accumulation_device_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "rf_renderer_bind_accumulation_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "rf_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "rf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rf_comm_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rf_comm_destroy": (None, [C.c_void_p]),
    "rf_renderer_gather_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rf_renderer_tonemap_device_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "rf_comm_read_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_comm_all_reduce_max": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "rf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "rf_renderer_layout_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rf_renderer_memory_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4),
    "rf_gather_plan": (C.c_int, [C.c_uint32] * 6 + [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_gather_layout": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_tiles_for_rank": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_untile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rf_renderer_trace_primary_stats": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_renderer_get_bounce_stats": (C.c_int, [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5),
    "rf_renderer_intersect_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float] + [C.c_void_p] * 6),
    "rf_renderer_occluded_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]),
    "rf_intersect_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "rf_intersect_bvh_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_bvh_visualizer_pass": (C.c_int, [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_build_bvh": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_int32)]),
    "rf_comm_last_exchange_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "rf_comm_transport": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rf_check_wide_layouts": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]),
    "rf_wide_layout_stats": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "rf_build_bvh_gpu": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rf_create_camera": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]),
    "rf_fly_camera": (C.c_int, [C.c_void_p] + [C.c_float] * 6 + [C.POINTER(Camera)]),
    "rf_bvh_visualizer_camera": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(Camera)]),
    "rf_sky_state_new": (C.c_int, [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "rf_sky_state_radiance": (C.c_float, [C.c_void_p, C.c_float, C.c_float, C.c_int]),
    "rf_aligned_sky_state": (C.c_int, [C.POINTER(Sky), C.c_void_p]),
    "rf_texture_from_memory": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_pt_format_set_bvh_builder": (C.c_int, [C.c_int32]),
    "rf_pt_format_from_gltf": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rf_pt_format_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rf_pt_format_deserialize": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rf_pt_format_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rf_pt_format_serialize": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "rf_pt_format_from_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Texture), C.c_uint64, C.POINTER(C.c_void_p)]),
    "rf_pt_format_view_get": (C.c_int, [C.c_void_p, C.POINTER(PtFormatView)]),
    "rf_pt_format_texture": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(Texture)]),
    "rf_pt_format_destroy": (None, [C.c_void_p]),
    "rf_pt_format_scene": (C.c_int, [C.c_void_p, C.POINTER(Scene), C.POINTER(Texture)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class RayfinderError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def check(status):
    if status != RF_OK:
        raise RayfinderError(status, lib.rf_last_error_message().decode("utf-8", "replace"))
