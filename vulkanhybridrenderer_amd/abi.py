"""Data ABI of the hot path: numpy mirrors of the structs the reference shares between C++ and GLSL.

Layouts follow /root/reference/src/rendering_backend/glsl_common.h:22-99 (GLSL `scalar` layout ==
packed C++): Vertex 56 B, Material 44 B, Primitive 120 B, DirectionalLight 112 B, PerFrameData 584 B,
SVGFPushConstants 24 B.  The same layouts are declared for C callers in include/vhr_types.h.

Matrices are glm/GLSL column-major: a 4x4 math matrix M (M[row, col]) is stored as M.T.flatten().
"""
import numpy as np

# VkFormat values used by the path (hybrid_render_path.cpp:16-19,104-111,247-261); the C ABI accepts
# the Vulkan enum values unchanged so reference-side code passes its VkFormat straight through.
FORMAT_R8G8B8A8_UNORM = 37
FORMAT_R8G8B8A8_SRGB = 43
FORMAT_B8G8R8A8_UNORM = 44
FORMAT_B8G8R8A8_SRGB = 50
FORMAT_R16G16_SFLOAT = 83
FORMAT_R16G16B16A16_SFLOAT = 97
FORMAT_D32_SFLOAT = 126

FORMAT_STRIDE = {
    FORMAT_R8G8B8A8_UNORM: 4, FORMAT_R8G8B8A8_SRGB: 4, FORMAT_B8G8R8A8_UNORM: 4, FORMAT_B8G8R8A8_SRGB: 4,
    FORMAT_R16G16_SFLOAT: 4, FORMAT_R16G16B16A16_SFLOAT: 8, FORMAT_D32_SFLOAT: 4,
}

# VkFilter / VkSamplerAddressMode values (vulkan_common.h:21-26 SamplerInfo)
FILTER_NEAREST, FILTER_LINEAR = 0, 1
ADDRESS_REPEAT, ADDRESS_MIRRORED_REPEAT, ADDRESS_CLAMP_TO_EDGE = 0, 1, 2

vertex_dtype = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 4),
                         ("uv0", "<f4", 2), ("uv1", "<f4", 2)])
material_dtype = np.dtype([("base_color", "<f4", 4), ("base_color_texture", "<i4"),
                           ("metallic_roughness_texture", "<i4"), ("normal_map", "<i4"),
                           ("metallic_factor", "<f4"), ("roughness_factor", "<f4"),
                           ("alpha_mask", "<i4"), ("alpha_cutoff", "<f4")])
primitive_dtype = np.dtype([("transform", "<f4", 16), ("material", material_dtype),
                            ("vertex_offset", "<u4"), ("index_offset", "<u4"), ("index_count", "<u4")])
directional_light_dtype = np.dtype([("projview", "<f4", 16), ("direction", "<f4", 4),
                                    ("color", "<f4", 4), ("intensity", "<f4", 4)])
per_frame_dtype = np.dtype([("camera_view", "<f4", 16), ("camera_proj", "<f4", 16),
                            ("camera_view_inverse", "<f4", 16), ("camera_proj_inverse", "<f4", 16),
                            ("camera_viewproj_inverse", "<f4", 16), ("camera_view_prev_frame", "<f4", 16),
                            ("camera_proj_prev_frame", "<f4", 16), ("directional_light", directional_light_dtype),
                            ("display_size", "<f4", 2), ("display_size_inverse", "<f4", 2),
                            ("frame_index", "<u4"), ("blue_noise_texture_index", "<i4")])
svgf_push_constants_dtype = np.dtype([("integrated_shadow_and_ao", "<i4", 2),
                                      ("prev_frame_normals_and_object_ids", "<i4"),
                                      ("shadow_and_ao_history", "<i4"),
                                      ("shadow_and_ao_moments_history", "<i4"), ("atrous_step", "<i4")])
# parameters raygen.rgen hard-codes (include/vhr_types.h: vhr_trace_params)
trace_params_dtype = np.dtype([("shadow_enable", "<u4"), ("ao_spp", "<u4"), ("ao_tmax", "<f4"),
                               ("reflections", "<u4"), ("cone_cos_max", "<f4"), ("normal_bias", "<f4"),
                               ("tmin", "<f4"), ("tmax", "<f4")])

assert vertex_dtype.itemsize == 56
assert material_dtype.itemsize == 44
assert primitive_dtype.itemsize == 120
assert directional_light_dtype.itemsize == 112
assert per_frame_dtype.itemsize == 584
assert svgf_push_constants_dtype.itemsize == 24
assert trace_params_dtype.itemsize == 32


def default_trace_params(shadow=True, ao_spp=2, reflections=True):
    """raygen.rgen:31-65 constants: 1 cone shadow sample (cos_max 0.999995), 2 AO rays (tmax 5),
    one mirror bounce, origin bias 0.1 * N, tmin 0.01, tmax 1e4."""
    p = np.zeros((), trace_params_dtype)
    p["shadow_enable"] = 1 if shadow else 0
    p["ao_spp"] = ao_spp
    p["ao_tmax"] = 5.0
    p["reflections"] = int(reflections)          # True / 1 = the reference's single bounce, 2 = the two-bounce extension
    p["cone_cos_max"] = 0.999995
    p["normal_bias"] = 0.1
    p["tmin"] = 0.01
    p["tmax"] = 10000.0
    return p


def mat_to_glm(m):
    """4x4 math matrix (row, col) -> 16 floats, column-major."""
    return np.asarray(m, dtype=np.float64).T.reshape(16).astype(np.float32)


def glm_to_mat(a):
    return np.asarray(a, dtype=np.float64).reshape(4, 4).T
