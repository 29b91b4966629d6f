"""ctypes binding of libvhr_amd.so (the C ABI declared in include/vhr_amd.h).

The shared library is the product; this module only marshals numpy arrays and Python callbacks across the
boundary, mirroring the reference's class names (ResourceManager / RenderGraph / execution contexts:
/root/reference/src/rendering_backend/resource_manager.h:16-78, src/render_graph/render_graph.h:5-60,
compute_execution_context.h:7-43, raytracing_execution_context.h:4-20).  There is no CPU fallback: if the
library is missing or no HIP device exists, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvhr_amd.so")

# every symbol include/vhr_amd.h declares (tests/test_abi.py checks the .so exports each of them)
EXPORTS = [
    "vhr_create", "vhr_destroy", "vhr_resize", "vhr_last_error", "vhr_synchronize", "vhr_version", "vhr_abi_struct_sizes",
    "vhr_default_trace_params", "vhr_update_geometry", "vhr_upload_texture_from_data", "vhr_upload_new_storage_image",
    "vhr_destroy_storage_image", "vhr_update_per_frame_ubo", "vhr_set_trace_params", "vhr_graph_destroy_resources",
    "vhr_graph_add_graphics_pass", "vhr_graph_add_raytracing_pass", "vhr_graph_add_compute_pass", "vhr_graph_build",
    "vhr_graph_execute", "vhr_graph_gather_performance_statistics", "vhr_graph_get_pass_time_ms",
    "vhr_graph_get_execution_order", "vhr_graph_contains_image", "vhr_graph_get_image_format",
    "vhr_graph_set_pass_epilogue", "vhr_graph_bind_external_image", "vhr_trace_rays", "vhr_compute_get_display_size",
    "vhr_compute_dispatch", "vhr_compute_blit_image_storage_to_transient", "vhr_compute_blit_image_transient_to_storage",
    "vhr_compute_blit_image_storage_to_storage", "vhr_hybrid_create", "vhr_hybrid_destroy", "vhr_hybrid_build",
    "vhr_hybrid_rebuild", "vhr_hybrid_get_push_constants", "vhr_hybrid_last_error", "vhr_hybrid_state_size", "vhr_hybrid_save_state",
    "vhr_hybrid_load_state", "vhr_get_last_per_frame_ubo", "vhr_get_display_size",
    "vhr_get_transient_image", "vhr_get_storage_image", "vhr_upload_transient_image", "vhr_download_transient_image",
    "vhr_upload_storage_image", "vhr_download_storage_image", "vhr_standin_gbuffer", "vhr_standin_gbuffer_with_albedo", "vhr_standin_composition", "vhr_standin_shadow_map", "vhr_set_strip", "vhr_set_tile",
    "vhr_standin_raytraced_composition", "vhr_raytraced_create", "vhr_raytraced_destroy", "vhr_raytraced_build", "vhr_raytraced_rebuild",
    "vhr_raytraced_last_error",
    "vhr_set_ray_statistics", "vhr_get_ray_statistics", "vhr_get_bvh_statistics", "vhr_get_current_stream", "vhr_get_bvh_builder", "vhr_get_bvh_presplit_level", "vhr_get_bvh_frame", "vhr_get_bvh_form_checks", "vhr_get_bvh_fingerprint", "vhr_get_bvh_tree_fingerprint", "vhr_set_kernel_timing",
    "vhr_get_kernel_time", "vhr_set_option", "vhr_get_option", "vhr_option_count", "vhr_option_info", "vhr_get_traversal_statistics", "vhr_source_fingerprint", "vhr_debug_wave_lifetimes", "vhr_get_reflection_statistics", "vhr_get_binary64_statistics", "vhr_debug_ray_triangle", "vhr_get_traversal_cycles", "vhr_get_drain_statistics", "vhr_get_build_times", "vhr_atrous_overlap", "vhr_atrous_output_extent", "vhr_strip_plan_make",
    "vhr_strip_plan_exchanges", "vhr_tile_grid_choose", "vhr_tile_plan_make", "vhr_tile_plan_make_weighted", "vhr_get_tile_cost_map", "vhr_tile_plan_exchanges", "vhr_tile_plan_replan", "vhr_comm_replan", "vhr_comm_get_unique_id", "vhr_comm_use_library", "vhr_comm_library", "vhr_comm_create", "vhr_comm_create_tiled", "vhr_comm_destroy", "vhr_comm_last_error", "vhr_comm_exchange_raytraced",
    "vhr_comm_start_frame_exchanges", "vhr_comm_finish_frame_exchanges",
    "vhr_calibration_stream_read",
]


class VhrError(RuntimeError):
    pass


class CreateInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("stream", C.c_void_p),
                ("flags", C.c_uint32)]


class TransientImage(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_int32),
                ("binding", C.c_uint32), ("clear_value", C.c_float * 4), ("multisampled", C.c_int32)]


class TransientResource(C.Structure):
    _fields_ = [("type", C.c_int32), ("name", C.c_char_p), ("image", TransientImage)]


class HitShader(C.Structure):
    _fields_ = [("closest_hit", C.c_char_p), ("any_hit", C.c_char_p)]


class RaytracingPipelineDescription(C.Structure):
    _fields_ = [("name", C.c_char_p), ("raygen_shader", C.c_char_p), ("miss_shaders", C.POINTER(C.c_char_p)),
                ("miss_shader_count", C.c_uint32), ("hit_shaders", C.POINTER(HitShader)), ("hit_shader_count", C.c_uint32)]


class ComputePipelineDescription(C.Structure):
    _fields_ = [("kernels", C.POINTER(C.c_char_p)), ("kernel_count", C.c_uint32), ("push_constant_size", C.c_uint32)]


class ImageInfo(C.Structure):
    _fields_ = [("device_ptr", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_int32),
                ("bytes_per_pixel", C.c_uint32)]


class CompositionDesc(C.Structure):
    _fields_ = [("shadow_mode", C.c_int32), ("ambient_occlusion_mode", C.c_int32), ("reflection_mode", C.c_int32),
                ("albedo_image", C.c_char_p), ("normals_image", C.c_char_p), ("motion_image", C.c_char_p), ("depth_image", C.c_char_p),
                ("shadow_ao_image", C.c_char_p), ("reflections_image", C.c_char_p), ("output_storage_image", C.c_int32),
                ("ssao_image", C.c_char_p), ("shadow_map_image", C.c_char_p)]


class StripPlanC(C.Structure):
    """vhr_strip_plan (include/vhr_amd.h): the C planner's row strip, field for field tiling.StripPlan."""
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
                ("overlap", C.c_uint32), ("halo", C.c_uint32)]


class RowExchangeC(C.Structure):
    _fields_ = [("peer", C.c_int32), ("send_begin", C.c_uint32), ("send_end", C.c_uint32), ("recv_begin", C.c_uint32), ("recv_end", C.c_uint32)]


class TilePlanC(C.Structure):
    """vhr_tile_plan (include/vhr_amd.h): the C planner's screen tile, field for field tiling.TilePlan."""
    _fields_ = [(n, C.c_uint32) for n in ("rank", "world", "width", "height", "grid_rows", "grid_cols", "col_begin", "col_end", "row_begin", "row_end",
                                          "overlap", "halo_rows", "halo_cols")] + [("col_cut", C.c_uint32 * 17), ("row_cut", (C.c_uint32 * 17) * 16)]      # VHR_TILE_MAX_GRID = 16


class RectC(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("x1", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32)]


class RectExchangeC(C.Structure):
    _fields_ = [("peer", C.c_int32), ("send", RectC), ("recv", RectC)]


class HybridSettings(C.Structure):
    _fields_ = [("shadow_mode", C.c_int32), ("ambient_occlusion_mode", C.c_int32), ("reflection_mode", C.c_int32),
                ("denoise_shadow_and_ao", C.c_int32), ("atrous_steps", C.c_int32)]


EXTERNAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
RAYTRACING_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
COMPUTE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)

_lib = None


def option_table():
    """vhr_option_info for every option of the library's table: {name: (default, min, max)}."""
    L = load()
    out = {}
    for i in range(L.vhr_option_count()):
        name, d, lo, hi = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
        if L.vhr_option_info(i, C.byref(name), C.byref(d), C.byref(lo), C.byref(hi)) != 0:
            raise VhrError(f"vhr_option_info({i})")
        out[name.value.decode()] = (int(d.value), int(lo.value), int(hi.value))
    return out


def source_fingerprint():
    """vhr_source_fingerprint(): the hash of the sources the loaded library was built from."""
    return load().vhr_source_fingerprint().decode()


def load():
    """Load libvhr_amd.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VhrError(f"{LIB_PATH} is missing: build it with `make -C vulkanhybridrenderer_amd/csrc` "
                       "(there is no CPU or pure-Python fallback)")
    # One HIP runtime (and one RCCL) per process.  PyTorch's wheel bundles its own libamdhip64 / librccl; libvhr_amd.so links
    # the ROCm installation's.  The loader merges them by soname only when torch's copies are loaded FIRST -- the other way round
    # the process ends up with two HIP runtimes, of which the second finds no device (measured: ncclCommInitRank fails with "no
    # ROCm-capable device", and the process aborts at exit).  torch is this package's plumbing for device memory and
    # torch.distributed anyway, so it goes first whenever it is installed; a C / C++ host without torch has one runtime by itself.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
    L.vhr_create.argtypes = [C.POINTER(CreateInfo), C.POINTER(vp)]
    L.vhr_destroy.argtypes = [vp]
    L.vhr_destroy.restype = None
    L.vhr_last_error.argtypes = [vp]
    L.vhr_last_error.restype = C.c_char_p
    L.vhr_synchronize.argtypes = [vp]
    L.vhr_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.vhr_version.restype = C.c_char_p
    L.vhr_abi_struct_sizes.argtypes = [C.POINTER(u32)]
    L.vhr_default_trace_params.argtypes = [vp]
    L.vhr_default_trace_params.restype = None
    L.vhr_update_geometry.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    L.vhr_upload_texture_from_data.argtypes = [vp, u32, u32, vp, i32, vp]
    L.vhr_upload_new_storage_image.argtypes = [vp, u32, u32, i32]
    L.vhr_destroy_storage_image.argtypes = [vp, i32]
    L.vhr_update_per_frame_ubo.argtypes = [vp, u32, vp]
    L.vhr_set_trace_params.argtypes = [vp, vp]
    L.vhr_graph_destroy_resources.argtypes = [vp]
    L.vhr_graph_add_graphics_pass.argtypes = [vp, C.c_char_p, C.POINTER(TransientResource), u32,
                                              C.POINTER(TransientResource), u32, EXTERNAL_CB, vp]
    L.vhr_graph_add_raytracing_pass.argtypes = [vp, C.c_char_p, C.POINTER(TransientResource), u32,
                                                C.POINTER(TransientResource), u32,
                                                C.POINTER(RaytracingPipelineDescription), RAYTRACING_CB, vp]
    L.vhr_graph_add_compute_pass.argtypes = [vp, C.c_char_p, C.POINTER(TransientResource), u32,
                                             C.POINTER(TransientResource), u32, C.POINTER(ComputePipelineDescription),
                                             COMPUTE_CB, vp]
    L.vhr_graph_build.argtypes = [vp]
    L.vhr_graph_execute.argtypes = [vp, u32, u32]
    L.vhr_graph_gather_performance_statistics.argtypes = [vp]
    L.vhr_graph_get_pass_time_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.vhr_graph_get_execution_order.argtypes = [vp, C.c_char_p, u32]
    L.vhr_graph_contains_image.argtypes = [vp, C.c_char_p]
    L.vhr_graph_get_image_format.argtypes = [vp, C.c_char_p]
    L.vhr_graph_set_pass_epilogue.argtypes = [vp, C.c_char_p, EXTERNAL_CB, vp]
    L.vhr_graph_bind_external_image.argtypes = [vp, C.c_char_p, vp]
    L.vhr_trace_rays.argtypes = [vp, u32, u32]
    L.vhr_compute_get_display_size.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.vhr_compute_dispatch.argtypes = [vp, C.c_char_p, u32, u32, u32, vp, u32]
    L.vhr_compute_blit_image_storage_to_transient.argtypes = [vp, i32, C.c_char_p]
    L.vhr_compute_blit_image_transient_to_storage.argtypes = [vp, C.c_char_p, i32]
    L.vhr_compute_blit_image_storage_to_storage.argtypes = [vp, i32, i32]
    L.vhr_hybrid_create.argtypes = [vp, C.POINTER(HybridSettings), EXTERNAL_CB, vp, EXTERNAL_CB, vp, C.POINTER(vp)]
    L.vhr_hybrid_destroy.argtypes = [vp]
    L.vhr_hybrid_destroy.restype = None
    L.vhr_hybrid_build.argtypes = [vp]
    L.vhr_hybrid_rebuild.argtypes = [vp, C.POINTER(HybridSettings)]
    L.vhr_hybrid_get_push_constants.argtypes = [vp, vp]
    L.vhr_hybrid_last_error.argtypes = [vp]
    L.vhr_hybrid_last_error.restype = C.c_char_p
    L.vhr_hybrid_state_size.argtypes = [vp, C.POINTER(u64)]
    L.vhr_hybrid_save_state.argtypes = [vp, vp, u64]
    L.vhr_hybrid_load_state.argtypes = [vp, vp, u64, vp]
    L.vhr_get_last_per_frame_ubo.argtypes = [vp, vp]
    L.vhr_get_display_size.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.vhr_get_transient_image.argtypes = [vp, C.c_char_p, C.POINTER(ImageInfo)]
    L.vhr_get_storage_image.argtypes = [vp, i32, C.POINTER(ImageInfo)]
    L.vhr_upload_transient_image.argtypes = [vp, C.c_char_p, vp, u64]
    L.vhr_download_transient_image.argtypes = [vp, C.c_char_p, vp, u64]
    L.vhr_upload_storage_image.argtypes = [vp, i32, vp, u64]
    L.vhr_download_storage_image.argtypes = [vp, i32, vp, u64]
    L.vhr_standin_gbuffer.argtypes = [vp, u32, C.c_char_p, C.c_char_p, C.c_char_p]
    L.vhr_standin_gbuffer_with_albedo.argtypes = [vp, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.vhr_standin_composition.argtypes = [vp, u32, C.POINTER(CompositionDesc)]
    L.vhr_standin_raytraced_composition.argtypes = [vp, C.c_char_p, i32]
    L.vhr_standin_shadow_map.argtypes = [vp, u32, C.c_char_p]
    L.vhr_raytraced_create.argtypes = [vp, i32, EXTERNAL_CB, vp, C.POINTER(vp)]
    L.vhr_raytraced_destroy.argtypes = [vp]
    L.vhr_raytraced_destroy.restype = None
    L.vhr_raytraced_build.argtypes = [vp]
    L.vhr_raytraced_rebuild.argtypes = [vp, i32]
    L.vhr_raytraced_last_error.argtypes = [vp]
    L.vhr_raytraced_last_error.restype = C.c_char_p
    L.vhr_set_strip.argtypes = [vp, u32, u32, u32, u32]
    L.vhr_set_tile.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32]
    L.vhr_set_ray_statistics.argtypes = [vp, i32]
    L.vhr_get_ray_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_bvh_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_bvh_form_checks.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_current_stream.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.vhr_get_bvh_builder.argtypes = [vp, C.POINTER(i32)]
    L.vhr_get_bvh_presplit_level.argtypes = [vp, C.POINTER(i32)]
    L.vhr_get_bvh_frame.argtypes = [vp, C.POINTER(C.c_float)]
    L.vhr_get_bvh_fingerprint.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_bvh_tree_fingerprint.argtypes = [vp, C.POINTER(u64)]
    L.vhr_set_option.argtypes = [vp, C.c_char_p, i32]
    L.vhr_get_traversal_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_traversal_cycles.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_reflection_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_binary64_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_debug_ray_triangle.argtypes = [vp, C.c_void_p, u32, C.c_void_p, C.c_void_p]
    L.vhr_debug_wave_lifetimes.argtypes = [vp, C.POINTER(u32), u32, C.POINTER(u32)]
    L.vhr_source_fingerprint.restype = C.c_char_p
    L.vhr_source_fingerprint.argtypes = []
    L.vhr_get_drain_statistics.argtypes = [vp, C.POINTER(u64)]
    L.vhr_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i32)]
    L.vhr_option_count.restype = i32
    L.vhr_option_count.argtypes = []
    L.vhr_option_info.argtypes = [i32, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.vhr_get_build_times.argtypes = [vp, C.POINTER(C.c_double)]
    L.vhr_atrous_overlap.restype = u32
    L.vhr_atrous_overlap.argtypes = [u32]
    L.vhr_atrous_output_extent.restype = u32
    L.vhr_atrous_output_extent.argtypes = [u32, u32]
    L.vhr_strip_plan_make.argtypes = [u32, u32, u32, u32, u32, C.POINTER(StripPlanC)]
    L.vhr_strip_plan_exchanges.argtypes = [C.POINTER(StripPlanC), u32, C.POINTER(RowExchangeC)]
    L.vhr_tile_grid_choose.argtypes = [u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    L.vhr_tile_plan_make.argtypes = [u32, u32, u32, u32, u32, u32, u32, u32, u32, C.POINTER(TilePlanC)]
    L.vhr_tile_plan_make_weighted.argtypes = [u32, u32, u32, u32, u32, u32, u32, u32, u32, vp, u32, u32, u32, C.POINTER(TilePlanC)]
    L.vhr_tile_plan_exchanges.argtypes = [C.POINTER(TilePlanC), u32, u32, C.POINTER(RectExchangeC), u32]
    L.vhr_tile_plan_replan.argtypes = [C.POINTER(TilePlanC), C.POINTER(TilePlanC), C.POINTER(RectExchangeC), u32]
    L.vhr_comm_create_tiled.argtypes = [vp, C.POINTER(TilePlanC), C.c_char_p, C.POINTER(vp)]
    L.vhr_comm_get_unique_id.argtypes = [C.c_char_p]
    L.vhr_comm_create.argtypes = [vp, C.POINTER(StripPlanC), C.c_char_p, C.POINTER(vp)]
    L.vhr_comm_destroy.argtypes = [vp]
    L.vhr_comm_destroy.restype = None
    L.vhr_comm_last_error.argtypes = [vp]
    L.vhr_comm_last_error.restype = C.c_char_p
    L.vhr_comm_exchange_raytraced.argtypes = [vp, C.c_char_p]
    L.vhr_comm_start_frame_exchanges.argtypes = [vp, i32, i32, C.c_char_p, i32, vp]
    L.vhr_comm_finish_frame_exchanges.argtypes = [vp]
    L.vhr_comm_replan.argtypes = [vp, C.POINTER(TilePlanC), i32, i32, i32]
    L.vhr_calibration_stream_read.argtypes = [vp, i32, u32]
    L.vhr_set_kernel_timing.argtypes = [vp, i32]
    L.vhr_get_kernel_time.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64), i32]
    _lib = L
    return L


# image / resource names of the hybrid path (hybrid_render_path.cpp:16-19,104-111,265-273)
NORMALS = "World Space Normals and Object IDs"
MOTION = "Motion Vectors and Metallic Roughness"
DEPTH = "Depth"
RAYTRACED = "Raytraced Shadows and Ambient Occlusion"
REFLECTIONS = "Raytraced Reflections"
DENOISED = "Denoised Raytraced Shadows and Ambient Occlusion"
ALBEDO = "Albedo"
SVGF_SHADER = "hybrid_render_path/svgf.comp"
ATROUS_SHADER = "hybrid_render_path/svgf_atrous_filter.comp"
RAYTRACED_OUTPUT = "RaytracedOutput"        # raytraced_render_path.cpp:15
SHADOW_MAP = "Shadow Map"                           # hybrid_render_path.cpp:62
SSAO_RAW = "Screen Space Ambient Occlusion Raw"     # hybrid_render_path.cpp:149
SSAO = "Screen Space Ambient Occlusion"             # :176
SSR = "Screen Space Reflections"                    # :219
SSAO_SHADER = "hybrid_render_path/ssao.comp"
SSAO_BLUR_SHADER = "hybrid_render_path/ssao_blur.comp"
SSR_SHADER = "hybrid_render_path/ssr.comp"

ATTACHMENT_IMAGE, SAMPLED_IMAGE, STORAGE_IMAGE = 0, 1, 2


def transient(name, fmt, binding, kind=STORAGE_IMAGE, width=0, height=0, clear=(0, 0, 0, 0)):
    """VkUtils::CreateTransient{Attachment,Sampled,Storage}Image (vulkan_utils.h:363-453)."""
    r = TransientResource()
    r.type = 0
    r.name = name.encode()
    r.image.type = kind
    r.image.width, r.image.height = width, height
    r.image.format = fmt
    r.image.binding = binding
    r.image.clear_value = (C.c_float * 4)(*clear)
    return r


def render_output(binding=0):
    return transient("RENDER_OUTPUT", 0, binding, ATTACHMENT_IMAGE)


_NP_FORMATS = {abi.FORMAT_R16G16B16A16_SFLOAT: (np.uint16, 4), abi.FORMAT_R16G16_SFLOAT: (np.uint16, 2),
               abi.FORMAT_D32_SFLOAT: (np.float32, 1), abi.FORMAT_B8G8R8A8_UNORM: (np.uint8, 4),
               abi.FORMAT_R8G8B8A8_UNORM: (np.uint8, 4), abi.FORMAT_R8G8B8A8_SRGB: (np.uint8, 4),
               abi.FORMAT_B8G8R8A8_SRGB: (np.uint8, 4)}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """vhr_context: device + ResourceManager + RenderGraph state for one GPU."""

    def __init__(self, width, height, device=0, stream=None, host_only=False, internal_stream=False):
        """stream: a hipStream_t handle used as given (None / 0 = the device's default stream)."""
        self.L = load()
        self.handle = C.c_void_p()
        info = CreateInfo(device, width, height, stream, (1 if host_only else 0) | (2 if internal_stream else 0))
        rc = self.L.vhr_create(C.byref(info), C.byref(self.handle))
        if rc < 0:
            self.handle = None
            raise VhrError("vhr_create: " + self.L.vhr_last_error(None).decode())
        self.width, self.height = width, height
        self._keep = []        # callbacks and strings the C side points at
        self._callback_error = None

    def close(self):
        if getattr(self, "handle", None):
            self.L.vhr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc < 0:
            raise VhrError(f"{what}: {self.L.vhr_last_error(self.handle).decode()}")
        return rc

    def synchronize(self):
        self.check(self.L.vhr_synchronize(self.handle), "synchronize")

    def display_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        self.check(self.L.vhr_get_display_size(self.handle, C.byref(w), C.byref(h)), "display_size")
        return int(w.value), int(h.value)

    def resize(self, width, height):
        """VulkanContext::Resize (renderer.cpp:113-118): the new display extent; the graph and the storage pool's images are released, geometry, the
        acceleration structure, textures and options stay.  Follow with the render path's build()."""
        self.check(self.L.vhr_resize(self.handle, int(width), int(height)), "resize")
        self.width, self.height = int(width), int(height)

    # ---- ResourceManager ----
    def update_geometry(self, vertices, indices, primitives):
        v = np.ascontiguousarray(vertices)
        i = np.ascontiguousarray(indices, np.uint32)
        p = np.ascontiguousarray(primitives)
        assert v.dtype == abi.vertex_dtype and p.dtype == abi.primitive_dtype
        self.check(self.L.vhr_update_geometry(self.handle, _p(v), len(v), _p(i), len(i), _p(p), len(p)), "UpdateGeometry")

    def upload_scene(self, scene):
        for t in scene.textures:
            self.upload_texture_from_data(t["rgba8"], t["format"], (t["mag"], t["min"], t["address_u"], t["address_v"]))
        self.update_geometry(scene.vertices, scene.indices, scene.primitives)

    def upload_texture_from_data(self, rgba8, fmt=abi.FORMAT_R8G8B8A8_UNORM, sampler=None):
        img = np.ascontiguousarray(rgba8, np.uint8)
        s = (C.c_int32 * 4)(*sampler) if sampler is not None else None
        return self.check(self.L.vhr_upload_texture_from_data(self.handle, img.shape[1], img.shape[0], _p(img), fmt, s),
                          "UploadTextureFromData")

    def upload_new_storage_image(self, width, height, fmt):
        r = self.L.vhr_upload_new_storage_image(self.handle, width, height, fmt)
        if r < -1:        # -1 = the reference's "pool exhausted" sentinel, returned as is; every failure is below it
            self.check(r, "UploadNewStorageImage")
        return r

    def destroy_storage_image(self, idx):
        self.check(self.L.vhr_destroy_storage_image(self.handle, idx), "DestroyStorageImage")

    def update_per_frame_ubo(self, resource_idx, pfd):
        a = np.ascontiguousarray(pfd)
        assert a.dtype == abi.per_frame_dtype
        self.check(self.L.vhr_update_per_frame_ubo(self.handle, resource_idx, _p(a)), "UpdatePerFrameUBO")

    def set_trace_params(self, tp):
        a = np.ascontiguousarray(tp)
        assert a.dtype == abi.trace_params_dtype
        self.check(self.L.vhr_set_trace_params(self.handle, _p(a)), "set_trace_params")

    def set_strip(self, row_begin, row_end, overlap=0, halo=None):
        halo = overlap if halo is None else halo
        self.check(self.L.vhr_set_strip(self.handle, row_begin, row_end, overlap, halo), "set_strip")

    def set_tile(self, col_begin, col_end, row_begin, row_end, overlap=0, halo_rows=None, halo_cols=None):
        halo_rows = overlap if halo_rows is None else halo_rows
        halo_cols = overlap if halo_cols is None else halo_cols
        self.check(self.L.vhr_set_tile(self.handle, col_begin, col_end, row_begin, row_end, overlap, halo_rows, halo_cols), "set_tile")

    # ---- RenderGraph ----
    def destroy_resources(self):
        self.check(self.L.vhr_graph_destroy_resources(self.handle), "DestroyResources")
        self._keep.clear()

    @staticmethod
    def _resources(lst):
        arr = (TransientResource * max(1, len(lst)))(*lst)
        return arr, len(lst)

    def add_graphics_pass(self, name, dependencies, outputs, callback=None):
        cb = EXTERNAL_CB(lambda user, ctx: self._guard(lambda: callback(self))) if callback else EXTERNAL_CB()
        d, nd = self._resources(dependencies)
        o, no = self._resources(outputs)
        self._keep += [cb, d, o, dependencies, outputs]
        self.check(self.L.vhr_graph_add_graphics_pass(self.handle, name.encode(), d, nd, o, no, cb, None), "AddGraphicsPass")

    def add_raytracing_pass(self, name, dependencies, outputs, callback, raygen="hybrid_render_path/raygen.rgen",
                            miss=("hybrid_render_path/miss.rmiss", "hybrid_render_path/reflection_miss.rmiss"),
                            closest_hit=("hybrid_render_path/reflection_hit.rchit",), pipeline_name="Raytrace Pipeline", any_hit=None):
        cb = RAYTRACING_CB(lambda user, exec_: self._guard(lambda: callback(RaytracingExecutionContext(self, exec_))))
        d, nd = self._resources(dependencies)
        o, no = self._resources(outputs)
        miss_arr = (C.c_char_p * len(miss))(*[m.encode() for m in miss])
        any_hit = any_hit or (None,) * len(closest_hit)
        hits = (HitShader * len(closest_hit))(*[HitShader(h.encode(), ah.encode() if ah else None) for h, ah in zip(closest_hit, any_hit)])
        desc = RaytracingPipelineDescription(pipeline_name.encode(), raygen.encode(), miss_arr, len(miss), hits, len(closest_hit))
        self._keep += [cb, d, o, dependencies, outputs, miss_arr, hits, desc]
        self.check(self.L.vhr_graph_add_raytracing_pass(self.handle, name.encode(), d, nd, o, no, C.byref(desc), cb, None),
                   "AddRaytracingPass")

    def add_compute_pass(self, name, dependencies, outputs, kernels, push_constant_size, callback):
        cb = COMPUTE_CB(lambda user, exec_: self._guard(lambda: callback(ComputeExecutionContext(self, exec_))))
        d, nd = self._resources(dependencies)
        o, no = self._resources(outputs)
        k = (C.c_char_p * len(kernels))(*[s.encode() for s in kernels])
        desc = ComputePipelineDescription(k, len(kernels), push_constant_size)
        self._keep += [cb, d, o, dependencies, outputs, k, desc]
        self.check(self.L.vhr_graph_add_compute_pass(self.handle, name.encode(), d, nd, o, no, C.byref(desc), cb, None),
                   "AddComputePass")

    def set_pass_epilogue(self, name, callback):
        cb = EXTERNAL_CB(lambda user, ctx: self._guard(lambda: callback(self))) if callback else EXTERNAL_CB()
        self._keep.append(cb)
        self.check(self.L.vhr_graph_set_pass_epilogue(self.handle, name.encode(), cb, None), "set_pass_epilogue")

    def build(self):
        self.check(self.L.vhr_graph_build(self.handle), "Build")

    def execute(self, resource_idx=0, image_idx=0):
        self._callback_error = None
        rc = self.L.vhr_graph_execute(self.handle, resource_idx, image_idx)
        if self._callback_error is not None:
            raise self._callback_error
        self.check(rc, "Execute")

    def gather_performance_statistics(self):
        self.check(self.L.vhr_graph_gather_performance_statistics(self.handle), "GatherPerformanceStatistics")

    def pass_time_ms(self, name):
        ema, last = C.c_double(), C.c_double()
        self.check(self.L.vhr_graph_get_pass_time_ms(self.handle, name.encode(), C.byref(ema), C.byref(last)), "pass_time_ms")
        return ema.value, last.value

    def execution_order(self):
        buf = C.create_string_buffer(4096)
        n = self.check(self.L.vhr_graph_get_execution_order(self.handle, buf, 4096), "execution_order")
        return buf.value.decode().split("\n") if n else []

    def contains_image(self, name):
        return bool(self.L.vhr_graph_contains_image(self.handle, name.encode()))

    def image_format(self, name):
        return self.L.vhr_graph_get_image_format(self.handle, name.encode())

    def bind_external_image(self, name, device_ptr):
        self.check(self.L.vhr_graph_bind_external_image(self.handle, name.encode(), device_ptr), "bind_external_image")

    # ---- harness access ----
    def transient_info(self, name):
        info = ImageInfo()
        self.check(self.L.vhr_get_transient_image(self.handle, name.encode(), C.byref(info)), "get_transient_image")
        return info

    def storage_info(self, idx):
        info = ImageInfo()
        self.check(self.L.vhr_get_storage_image(self.handle, idx, C.byref(info)), "get_storage_image")
        return info

    @staticmethod
    def _host_array(info):
        dt, ch = _NP_FORMATS[info.format]
        shape = (info.height, info.width, ch) if ch > 1 else (info.height, info.width)
        return np.zeros(shape, dt)

    def download(self, name_or_idx):
        if isinstance(name_or_idx, str):
            info = self.transient_info(name_or_idx)
            out = self._host_array(info)
            self.check(self.L.vhr_download_transient_image(self.handle, name_or_idx.encode(), _p(out), out.nbytes), "download")
        else:
            info = self.storage_info(name_or_idx)
            out = self._host_array(info)
            self.check(self.L.vhr_download_storage_image(self.handle, name_or_idx, _p(out), out.nbytes), "download")
        return out

    def upload(self, name_or_idx, array):
        a = np.ascontiguousarray(array)
        if isinstance(name_or_idx, str):
            self.check(self.L.vhr_upload_transient_image(self.handle, name_or_idx.encode(), _p(a), a.nbytes), "upload")
        else:
            self.check(self.L.vhr_upload_storage_image(self.handle, name_or_idx, _p(a), a.nbytes), "upload")

    def standin_gbuffer(self, resource_idx=0, normals=NORMALS, motion=MOTION, depth=DEPTH):
        self.check(self.L.vhr_standin_gbuffer(self.handle, resource_idx, normals.encode(), motion.encode(), depth.encode()),
                   "standin_gbuffer")

    def standin_gbuffer_with_albedo(self, resource_idx=0, albedo=ALBEDO, normals=NORMALS, motion=MOTION, depth=DEPTH):
        self.check(self.L.vhr_standin_gbuffer_with_albedo(self.handle, resource_idx, albedo.encode(), normals.encode(), motion.encode(),
                                                          depth.encode()), "standin_gbuffer_with_albedo")

    def standin_composition(self, output_storage_image, shadow_mode=0, ao_mode=0, reflection_mode=0, shadow_ao=DENOISED,
                            reflections=REFLECTIONS, resource_idx=0, ssao=None, shadow_map=None):
        """composition.frag stand-in.  ao_mode 1 reads `ssao` (default SSAO), reflection_mode 1 expects `reflections` = SSR."""
        if ao_mode == 1 and ssao is None:
            ssao = SSAO
        if shadow_mode == 1 and shadow_map is None:
            shadow_map = SHADOW_MAP
        d = CompositionDesc(shadow_mode, ao_mode, reflection_mode, ALBEDO.encode(), NORMALS.encode(), MOTION.encode(), DEPTH.encode(),
                            shadow_ao.encode(), reflections.encode() if reflections else None, output_storage_image,
                            ssao.encode() if ssao else None, shadow_map.encode() if shadow_map else None)
        self.check(self.L.vhr_standin_composition(self.handle, resource_idx, C.byref(d)), "standin_composition")

    def standin_shadow_map(self, resource_idx=0, shadow_map=SHADOW_MAP):
        self.check(self.L.vhr_standin_shadow_map(self.handle, resource_idx, shadow_map.encode()), "standin_shadow_map")

    def standin_raytraced_composition(self, output_storage_image, raytraced_output=RAYTRACED_OUTPUT):
        self.check(self.L.vhr_standin_raytraced_composition(self.handle, raytraced_output.encode(), output_storage_image),
                   "standin_raytraced_composition")

    def set_ray_statistics(self, enable):
        self.check(self.L.vhr_set_ray_statistics(self.handle, int(enable)), "set_ray_statistics")

    def ray_statistics(self):
        out = (C.c_uint64 * 4)()
        self.check(self.L.vhr_get_ray_statistics(self.handle, out), "ray_statistics")
        return dict(unique_rays=out[0], reference_issued_rays=out[1], covered_pixels=out[2], stack_overflows=out[3])

    def calibration_stream_read(self, storage_image, bytes_per_lane):
        self.check(self.L.vhr_calibration_stream_read(self.handle, storage_image, bytes_per_lane), "calibration_stream_read")

    def traversal_statistics(self):
        out = (C.c_uint64 * 4)()
        self.check(self.L.vhr_get_traversal_statistics(self.handle, out), "traversal_statistics")
        d = dict(node_visits=out[0], leaf_visits=out[1], triangle_tests=out[2], wave_iterations=out[3])
        d["active_lane_utilisation"] = (out[0] + out[2]) / (64.0 * out[3]) if out[3] else 0.0
        return d

    def reflection_statistics(self):
        """The mirror-ray launch's counters (reflection_queue_kernel with statistics enabled)."""
        out = (C.c_uint64 * 10)()
        self.check(self.L.vhr_get_reflection_statistics(self.handle, out), "reflection_statistics")
        d = dict(rays=out[0], second_bounce_rays=out[1], node_visits=out[2], leaf_visits=out[3], triangle_tests=out[4], wave_iterations=out[5],
                 refills=out[6], waves=out[7], cycles_total=out[8], cycles_walk=out[9])
        d["active_lane_utilisation"] = (out[2] + out[4]) / (64.0 * out[5]) if out[5] else 0.0
        return d

    def ray_triangle(self, pairs):
        """Decision (vi) as the device computes it, on explicit pairs: pairs = (n, 17) float32 (o, d, v0, e1, e2, tmin, tmax) -> (hit (n,) bool, tuv (n, 3) float32)."""
        import numpy as np
        pairs = np.ascontiguousarray(pairs, dtype=np.float32).reshape(-1, 17)
        hit = np.zeros(len(pairs), np.uint32)
        tuv = np.zeros((len(pairs), 3), np.float32)
        self.check(self.L.vhr_debug_ray_triangle(self.handle, pairs.ctypes.data, len(pairs), hit.ctypes.data, tuv.ctypes.data), "ray_triangle")
        return hit != 0, tuv

    def binary64_statistics(self):
        """Decision (vi)'s binary64 half in the queue kernels of the last vhr_trace_rays (statistics enabled): pixels the any-hit launch and the mirror
        ray's launch computed again by the per-pixel code."""
        out = (C.c_uint64 * 4)()
        self.check(self.L.vhr_get_binary64_statistics(self.handle, out), "binary64_statistics")
        return dict(pixels_again=int(out[0]), mirror_pixels_again=int(out[1]))

    def wave_lifetimes(self, capacity=1 << 20):
        """Lifetimes (shader clock ticks) of the last ray-tracing launch's waves (what raygen_cost_order sorts by)."""
        import numpy as np
        out = np.zeros(capacity, dtype=np.uint32)
        n = C.c_uint32(0)
        self.check(self.L.vhr_debug_wave_lifetimes(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32)), capacity, C.byref(n)), "wave_lifetimes")
        return out[:n.value].copy()

    def tile_cost_map(self):
        """vhr_get_tile_cost_map: the last frame's wave lifetimes (any-hit + mirror-ray launch) per 8 x 8-pixel cell, (ceil(H / 8), ceil(W / 8)) uint32."""
        import numpy as np
        rows, cols = (self.height + 7) // 8, (self.width + 7) // 8
        out = np.zeros((rows, cols), np.uint32)
        self.L.vhr_get_tile_cost_map.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        self.check(self.L.vhr_get_tile_cost_map(self.handle, out.ctypes.data_as(C.c_void_p), cols, rows), "tile_cost_map")
        return out

    def traversal_cycles(self):
        out = (C.c_uint64 * 8)()
        self.check(self.L.vhr_get_traversal_cycles(self.handle, out), "traversal_cycles")
        return dict(total=out[0], setup=out[1], refill=out[2], nodes=out[3], leaves=out[4], refills=out[5], waves=out[6], drain_iterations=out[7])

    def bvh_form_checks(self):
        """Host-side self-check of the last build: (boxes checked, CH violations, 48-byte-node violations, compact-node violations)."""
        out = (C.c_uint64 * 4)()
        self.check(self.L.vhr_get_bvh_form_checks(self.handle, out), "bvh_form_checks")
        return tuple(int(v) for v in out)

    def bvh_frame(self):
        """Option "bvh_frame": the frame the current tree's boxes are in, a 3 x 3 array (row i = axis i in world coordinates; the identity = the world axes)."""
        import numpy as np
        out = (C.c_float * 9)()
        self.check(self.L.vhr_get_bvh_frame(self.handle, out), "get_bvh_frame")
        return np.array(list(out), np.float32).reshape(3, 3)

    def bvh_presplit_level(self):
        """Option "bvh_presplit": the grid level the current tree's references were split on, -1 = one reference per triangle."""
        out = C.c_int32()
        self.check(self.L.vhr_get_bvh_presplit_level(self.handle, C.byref(out)), "get_bvh_presplit_level")
        return int(out.value)

    def bvh_builder_used(self):
        """Which builder made the current tree: 1 = the device's (option "bvh_builder" 1, the default), 0 = the host's."""
        out = C.c_int32()
        self.check(self.L.vhr_get_bvh_builder(self.handle, C.byref(out)), "get_bvh_builder")
        return int(out.value)

    def bvh_fingerprint(self):
        out = C.c_uint64()
        self.check(self.L.vhr_get_bvh_fingerprint(self.handle, C.byref(out)), "bvh_fingerprint")
        return int(out.value)

    def bvh_tree_fingerprint(self):
        """A hash of the tree, not of its arrays: equal for the host's and the device's build of a scene."""
        out = C.c_uint64()
        self.check(self.L.vhr_get_bvh_tree_fingerprint(self.handle, C.byref(out)), "bvh_tree_fingerprint")
        return int(out.value)

    def build_times_ms(self):
        """K0: (host BVH build, upload of scene + tree) of the last upload_scene, milliseconds."""
        out = (C.c_double * 2)()
        self.check(self.L.vhr_get_build_times(self.handle, out), "build_times")
        return float(out[0]), float(out[1])

    def drain_statistics(self):
        out = (C.c_uint64 * 4)()
        self.check(self.L.vhr_get_drain_statistics(self.handle, out), "drain_statistics")
        return dict(cut_entries=out[0], drain_trips_le4=out[1], drain_trips_le8=out[2], drain_trips_le16=out[3])

    def get_option(self, key):
        out = C.c_int32()
        self.check(self.L.vhr_get_option(self.handle, key.encode(), C.byref(out)), f"get_option({key})")
        return int(out.value)

    def bvh_statistics(self):
        out = (C.c_uint64 * 5)()
        self.check(self.L.vhr_get_bvh_statistics(self.handle, out), "bvh_statistics")
        return dict(nodes=out[0], triangles=out[1], max_depth=out[2], node_bytes=out[3], triangle_bytes=out[4])

    KERNEL_KINDS = {"raygen": 0, "svgf_temporal": 1, "svgf_atrous": 2, "blit": 3, "reflection": 4, "ssao": 5, "ssao_blur": 6, "ssr": 7, "svgf_atrous_async": 8}

    def current_stream(self):
        """hipStream_t (as an int) the library is enqueueing on right now: inside a pass callback, the stream that pass is ordered on."""
        out = C.c_void_p()
        self.check(self.L.vhr_get_current_stream(self.handle, C.byref(out)), "get_current_stream")
        return int(out.value or 0)

    def set_option(self, key, value):
        self.check(self.L.vhr_set_option(self.handle, key.encode(), int(value)), "set_option")

    def set_kernel_timing(self, kinds):
        """kinds: True (all), False (off) or an iterable of kind names from KERNEL_KINDS."""
        if kinds is True:
            mask = 0xFF
        elif not kinds:
            mask = 0
        else:
            mask = sum(1 << self.KERNEL_KINDS[k] for k in kinds)
        self.check(self.L.vhr_set_kernel_timing(self.handle, mask), "set_kernel_timing")

    def kernel_time(self, kind, reset=False):
        """(total_ms, launches) of a kernel kind, measured with HIP events on the context stream."""
        ms, n = C.c_double(), C.c_uint64()
        self.check(self.L.vhr_get_kernel_time(self.handle, self.KERNEL_KINDS[kind], C.byref(ms), C.byref(n), int(reset)), "kernel_time")
        return ms.value, n.value

    def _guard(self, fn):
        """Run a Python pass body; park exceptions (they must not unwind through C) for execute() to re-raise."""
        try:
            fn()
        except Exception as e:   # noqa: BLE001
            self._callback_error = e


class RaytracingExecutionContext:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def trace_rays(self, width, height):
        self.ctx.check(self.ctx.L.vhr_trace_rays(self.handle, width, height), "TraceRays")


class ComputeExecutionContext:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def get_display_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        self.ctx.check(self.ctx.L.vhr_compute_get_display_size(self.handle, C.byref(w), C.byref(h)), "GetDisplaySize")
        return w.value, h.value

    def dispatch(self, shader, x_groups, y_groups, z_groups, push_constants=None):
        if push_constants is None:
            rc = self.ctx.L.vhr_compute_dispatch(self.handle, shader.encode(), x_groups, y_groups, z_groups, None, 0)
        else:
            a = np.ascontiguousarray(push_constants)
            rc = self.ctx.L.vhr_compute_dispatch(self.handle, shader.encode(), x_groups, y_groups, z_groups, _p(a), a.nbytes)
        self.ctx.check(rc, "Dispatch")

    def blit_image_storage_to_transient(self, src, dst):
        self.ctx.check(self.ctx.L.vhr_compute_blit_image_storage_to_transient(self.handle, src, dst.encode()), "BlitImageStorageToTransient")

    def blit_image_transient_to_storage(self, src, dst):
        self.ctx.check(self.ctx.L.vhr_compute_blit_image_transient_to_storage(self.handle, src.encode(), dst), "BlitImageTransientToStorage")

    def blit_image_storage_to_storage(self, src, dst):
        self.ctx.check(self.ctx.L.vhr_compute_blit_image_storage_to_storage(self.handle, src, dst), "BlitImageStorageToStorage")


class HybridRenderPath:
    """vhr_hybrid_*: the C++ re-host of HybridRenderPath (csrc/hybrid_render_path.cpp)."""

    def __init__(self, ctx, shadow_mode=0, ambient_occlusion_mode=2, reflection_mode=2, denoise=False, atrous_steps=5,
                 gbuffer_pass=None, composition_pass=None):
        self.ctx = ctx
        self.settings = HybridSettings(shadow_mode, ambient_occlusion_mode, reflection_mode, int(denoise), atrous_steps)
        self._g = EXTERNAL_CB(lambda user, c: ctx._guard(lambda: gbuffer_pass(ctx))) if gbuffer_pass else EXTERNAL_CB()
        self._c = EXTERNAL_CB(lambda user, c: ctx._guard(lambda: composition_pass(ctx))) if composition_pass else EXTERNAL_CB()
        self.handle = C.c_void_p()
        ctx.check(ctx.L.vhr_hybrid_create(ctx.handle, C.byref(self.settings), self._g, None, self._c, None, C.byref(self.handle)),
                  "vhr_hybrid_create")

    def _check(self, rc, what):
        if rc < 0:
            raise VhrError(f"{what}: {self.ctx.L.vhr_hybrid_last_error(self.handle).decode()}")

    def build(self):
        self._check(self.ctx.L.vhr_hybrid_build(self.handle), "HybridRenderPath::Build")

    def rebuild(self, **changes):
        for k, v in changes.items():
            setattr(self.settings, k, int(v))
        self._check(self.ctx.L.vhr_hybrid_rebuild(self.handle, C.byref(self.settings)), "HybridRenderPath::Rebuild")

    def push_constants(self):
        pc = np.zeros((), abi.svgf_push_constants_dtype)
        self.ctx.check(self.ctx.L.vhr_hybrid_get_push_constants(self.handle, _p(pc)), "get_push_constants")
        return pc

    def save_state(self):
        """vhr_hybrid_save_state: the path's cross-frame state (five SVGF images + the last frame's PerFrameData) as a uint8 array."""
        n = C.c_uint64()
        self._check(self.ctx.L.vhr_hybrid_state_size(self.handle, C.byref(n)), "vhr_hybrid_state_size")
        blob = np.empty(int(n.value), np.uint8)
        self._check(self.ctx.L.vhr_hybrid_save_state(self.handle, _p(blob), n.value), "vhr_hybrid_save_state")
        return blob

    def load_state(self, blob):
        """vhr_hybrid_load_state; returns the PerFrameData of the frame the blob was saved behind."""
        blob = np.ascontiguousarray(blob, np.uint8)
        last = np.zeros((), abi.per_frame_dtype)
        self._check(self.ctx.L.vhr_hybrid_load_state(self.handle, _p(blob), blob.size, _p(last)), "vhr_hybrid_load_state")
        return last

    def destroy(self):
        if self.handle:
            self.ctx.L.vhr_hybrid_destroy(self.handle)
            self.handle = None


class RaytracedRenderPath:
    """vhr_raytraced_*: the C++ re-host of RaytracedRenderPath (csrc/raytraced_render_path.cpp)."""

    def __init__(self, ctx, use_anyhit_shader=False, composition_pass=None):
        self.ctx = ctx
        self.use_anyhit_shader = bool(use_anyhit_shader)
        self._c = EXTERNAL_CB(lambda user, c: ctx._guard(lambda: composition_pass(ctx))) if composition_pass else EXTERNAL_CB()
        self.handle = C.c_void_p()
        ctx.check(ctx.L.vhr_raytraced_create(ctx.handle, int(self.use_anyhit_shader), self._c, None, C.byref(self.handle)),
                  "vhr_raytraced_create")

    def _check(self, rc, what):
        if rc < 0:
            raise VhrError(f"{what}: {self.ctx.L.vhr_raytraced_last_error(self.handle).decode()}")

    def build(self):
        self._check(self.ctx.L.vhr_raytraced_build(self.handle), "RaytracedRenderPath::Build")

    def rebuild(self, use_anyhit_shader):
        self.use_anyhit_shader = bool(use_anyhit_shader)
        self._check(self.ctx.L.vhr_raytraced_rebuild(self.handle, int(self.use_anyhit_shader)), "RaytracedRenderPath::Rebuild")

    def destroy(self):
        if self.handle:
            self.ctx.L.vhr_raytraced_destroy(self.handle)
            self.handle = None


def strip_plan(height, world, rank, max_motion_rows, atrous_steps=5):
    """vhr_strip_plan_make: the C planner (csrc/comm.cpp).  None when the strips are thinner than the history halo."""
    p = StripPlanC()
    rc = load().vhr_strip_plan_make(height, world, rank, max_motion_rows, atrous_steps, C.byref(p))
    if rc == -4:                      # VHR_ERROR_OUT_OF_SLOTS
        return None
    if rc != 0:
        raise VhrError(f"vhr_strip_plan_make: {rc}")
    return p


def strip_plan_exchanges(plan, n_rows):
    out = (RowExchangeC * 2)()
    n = load().vhr_strip_plan_exchanges(C.byref(plan), n_rows, out)
    return [(out[i].peer, (out[i].send_begin, out[i].send_end), (out[i].recv_begin, out[i].recv_end)) for i in range(n)]


def tile_grid(width, height, world, overlap):
    """vhr_tile_grid_choose: (grid_rows, grid_cols) of the C planner."""
    r, c = C.c_uint32(), C.c_uint32()
    rc = load().vhr_tile_grid_choose(width, height, world, overlap, C.byref(r), C.byref(c))
    if rc != 0:
        raise VhrError(f"vhr_tile_grid_choose: {rc}")
    return int(r.value), int(c.value)


def tile_plan(width, height, world, rank, grid_rows=0, grid_cols=0, max_motion_rows=0, max_motion_cols=0, atrous_steps=5, cost=None, cost_cell=8):
    """vhr_tile_plan_make / _make_weighted: the C planner (csrc/comm.cpp).  None when a tile is thinner than its history halo.  cost: a 2-D uint32
    array, one value per cost_cell x cost_cell pixel block -- the grid is cut at equal cost."""
    p = TilePlanC()
    if cost is None:
        rc = load().vhr_tile_plan_make(width, height, world, rank, grid_rows, grid_cols, max_motion_rows, max_motion_cols, atrous_steps, C.byref(p))
    else:
        cm = np.ascontiguousarray(cost, np.uint32)
        rc = load().vhr_tile_plan_make_weighted(width, height, world, rank, grid_rows, grid_cols, max_motion_rows, max_motion_cols, atrous_steps,
                                                cm.ctypes.data_as(C.c_void_p), cm.shape[1], cm.shape[0], cost_cell, C.byref(p))
    if rc == -4:                      # VHR_ERROR_OUT_OF_SLOTS
        return None
    if rc != 0:
        raise VhrError(f"vhr_tile_plan_make: {rc}")
    return p


def tile_plan_exchanges(plan, halo_rows, halo_cols):
    out = (RectExchangeC * 64)()
    n = load().vhr_tile_plan_exchanges(C.byref(plan), halo_rows, halo_cols, out, 64)
    if n < 0:
        raise VhrError(f"vhr_tile_plan_exchanges: {n}")
    rect = lambda r: (r.x0, r.x1, r.y0, r.y1)
    return [(out[i].peer, rect(out[i].send), rect(out[i].recv)) for i in range(n)]


def tile_plan_replan(old, new):
    """vhr_tile_plan_replan: [(peer, send rect, recv rect)] that carry the cross-frame SVGF state from `old`'s rectangles to `new`'s (tiling.replan_transfers)."""
    out = (RectExchangeC * 256)()
    n = load().vhr_tile_plan_replan(C.byref(old), C.byref(new), out, 256)
    if n < 0:
        raise VhrError(f"vhr_tile_plan_replan: {n}")
    rect = lambda r: (r.x0, r.x1, r.y0, r.y1)   # noqa: E731
    return [(out[i].peer, rect(out[i].send), rect(out[i].recv)) for i in range(n)]


def comm_use_library(path):
    """vhr_comm_use_library: the RCCL library csrc/comm.cpp loads (before the process's first vhr_comm_* call); None = the default resolution."""
    L = load()
    L.vhr_comm_use_library.argtypes = [C.c_char_p]
    rc = L.vhr_comm_use_library(path.encode() if path else None)
    if rc != 0:
        raise VhrError(f"vhr_comm_use_library: {rc} (the RCCL entry points are already bound)")


def comm_library():
    """vhr_comm_library: the file the RCCL entry points came from, or the reason they could not be resolved."""
    L = load()
    L.vhr_comm_library.restype = C.c_char_p
    return L.vhr_comm_library().decode()


def comm_use_library_from_environment():
    """Test and bench tooling: VHR_RCCL_LIBRARY=<path> (tests/rccl_shim's stand-in, a site's own build) is honoured HERE, by the host program's explicit call
    -- the library itself reads no environment variable.  Returns the path or None."""
    path = os.environ.get("VHR_RCCL_LIBRARY") or None
    if path:
        comm_use_library(path)
    return path


class Comm:
    """vhr_comm_*: the strip / tile exchanges inside the library (RCCL).  One per context and process."""

    def __init__(self, ctx, plan, unique_id):
        self.ctx = ctx
        self.handle = C.c_void_p()
        if isinstance(plan, TilePlanC):
            ctx.check(ctx.L.vhr_comm_create_tiled(ctx.handle, C.byref(plan), unique_id, C.byref(self.handle)), "vhr_comm_create_tiled")
        else:
            ctx.check(ctx.L.vhr_comm_create(ctx.handle, C.byref(plan), unique_id, C.byref(self.handle)), "vhr_comm_create")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = load().vhr_comm_get_unique_id(buf)
        if rc != 0:
            raise VhrError(f"vhr_comm_get_unique_id: {rc} (RCCL not available?)")
        return buf.raw

    def _check(self, rc, what):
        if rc != 0:
            raise VhrError(f"{what}: {self.ctx.L.vhr_comm_last_error(self.handle).decode()}")

    def exchange_raytraced(self, image=RAYTRACED):
        self._check(self.ctx.L.vhr_comm_exchange_raytraced(self.handle, image.encode()), "vhr_comm_exchange_raytraced")

    def start_frame_exchanges(self, history, moments, denoised=None, root=0, gathered_frame_ptr=None):
        self._check(self.ctx.L.vhr_comm_start_frame_exchanges(self.handle, history, moments, denoised.encode() if denoised else None, root,
                                                              gathered_frame_ptr), "vhr_comm_start_frame_exchanges")

    def finish_frame_exchanges(self):
        self._check(self.ctx.L.vhr_comm_finish_frame_exchanges(self.handle), "vhr_comm_finish_frame_exchanges")

    def replan(self, new_plan, history, moments, prev_normals):
        """vhr_comm_replan: the communicator and the context take `new_plan` (a TilePlanC), the three cross-frame storage images follow their pixels."""
        self._check(self.ctx.L.vhr_comm_replan(self.handle, C.byref(new_plan), int(history), int(moments), int(prev_normals)), "vhr_comm_replan")

    def destroy(self):
        if self.handle:
            self.ctx.L.vhr_comm_destroy(self.handle)
            self.handle = None
