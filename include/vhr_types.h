/*
 * vhr_types.h -- plain-C data ABI of the hybrid ray-tracing hot path.
 *
 * Layouts are those the reference shares between C++ and GLSL
 * (src/rendering_backend/glsl_common.h:22-99; GLSL `scalar` layout == packed C++) and the POD pass
 * descriptions of src/rendering_backend/vulkan_common.h:236-341.  Matrices are glm column-major
 * (m[col*4 + row]).  Sizes are static-asserted below (56/44/120/112/584/24 bytes).
 */
#ifndef VHR_TYPES_H
#define VHR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- glsl_common.h:79-85 ---- */
typedef struct vhr_vertex {
    float pos[3];
    float normal[3];
    float tangent[4];
    float uv0[2];
    float uv1[2];
} vhr_vertex;

/* ---- glsl_common.h:87-97 ---- */
typedef struct vhr_material {
    float   base_color[4];
    int32_t base_color_texture;          /* -1 = none */
    int32_t metallic_roughness_texture;  /* -1 = none */
    int32_t normal_map;
    float   metallic_factor;
    float   roughness_factor;
    int32_t alpha_mask;
    float   alpha_cutoff;
} vhr_material;

/* ---- glsl_common.h:99-105 ---- */
typedef struct vhr_primitive {
    float        transform[16];
    vhr_material material;
    uint32_t     vertex_offset;
    uint32_t     index_offset;
    uint32_t     index_count;
} vhr_primitive;

/* ---- glsl_common.h:56-61 ---- */
typedef struct vhr_directional_light {
    float projview[16];
    float direction[4];
    float color[4];
    float intensity[4];
} vhr_directional_light;

/* ---- glsl_common.h:63-76 ---- */
typedef struct vhr_per_frame_data {
    float camera_view[16];
    float camera_proj[16];
    float camera_view_inverse[16];
    float camera_proj_inverse[16];
    float camera_viewproj_inverse[16];
    float camera_view_prev_frame[16];
    float camera_proj_prev_frame[16];
    vhr_directional_light directional_light;
    float    display_size[2];
    float    display_size_inverse[2];
    uint32_t frame_index;
    int32_t  blue_noise_texture_index;
} vhr_per_frame_data;

/* ---- glsl_common.h:31-39 ---- */
typedef struct vhr_svgf_push_constants {
    int32_t integrated_shadow_and_ao[2];           /* ping-pong storage-image indices (.x, .y) */
    int32_t prev_frame_normals_and_object_ids;
    int32_t shadow_and_ao_history;
    int32_t shadow_and_ao_moments_history;
    int32_t atrous_step;
} vhr_svgf_push_constants;

/* ---- glsl_common.h:41-50: push constants of the screen-space alternatives ---- */
typedef struct vhr_ssr_push_constants {
    float   ray_distance;     /* 25.0  (hybrid_render_path.cpp:203-208) */
    float   step_size;        /* 0.1 */
    float   thickness;        /* 0.5 */
    int32_t bsearch_steps;    /* 10 */
} vhr_ssr_push_constants;
typedef struct vhr_ssao_push_constants {
    float radius;             /* 0.75 (hybrid_render_path.cpp:139-141) */
} vhr_ssao_push_constants;

/* Constants raygen.rgen hard-codes (data/shaders/hybrid_render_path/raygen.rgen:29-65).  Defaults
 * (vhr_default_trace_params) reproduce the shader; other values are documented extensions used by
 * BASELINE.json configs 3 and 5 (ao_spp 4 / 16). */
typedef struct vhr_trace_params {
    uint32_t shadow_enable;   /* 1 */
    uint32_t ao_spp;          /* 2 */
    float    ao_tmax;         /* 5.0 */
    uint32_t reflections;     /* 1 = one mirror bounce shaded by reflection_hit.rchit, 0 = off, 2 = two bounces (extension, BASELINE config 5) */
    float    cone_cos_max;    /* 0.999995 */
    float    normal_bias;     /* 0.1 */
    float    tmin;            /* 0.01 */
    float    tmax;            /* 10000.0 */
} vhr_trace_params;

/* VkFormat values (passed through unchanged from reference-side code) */
enum {
    VHR_FORMAT_UNDEFINED           = 0,
    VHR_FORMAT_R8G8B8A8_UNORM      = 37,
    VHR_FORMAT_R8G8B8A8_SRGB       = 43,
    VHR_FORMAT_B8G8R8A8_UNORM      = 44,
    VHR_FORMAT_B8G8R8A8_SRGB       = 50,
    VHR_FORMAT_R16G16_SFLOAT       = 83,
    VHR_FORMAT_R16G16B16A16_SFLOAT = 97,
    VHR_FORMAT_D32_SFLOAT          = 126
};

/* vulkan_common.h:21-26 SamplerInfo (VkFilter / VkSamplerAddressMode values) */
typedef struct vhr_sampler_info {
    int32_t mag_filter;       /* 0 nearest, 1 linear */
    int32_t min_filter;
    int32_t address_mode_u;   /* 0 repeat, 1 mirrored repeat, 2 clamp to edge */
    int32_t address_mode_v;
} vhr_sampler_info;

/* vulkan_common.h:236-268 TransientResource / TransientImage */
enum { VHR_TRANSIENT_RESOURCE_IMAGE = 0, VHR_TRANSIENT_RESOURCE_BUFFER = 1 };
enum { VHR_TRANSIENT_ATTACHMENT_IMAGE = 0, VHR_TRANSIENT_SAMPLED_IMAGE = 1, VHR_TRANSIENT_STORAGE_IMAGE = 2 };

typedef struct vhr_transient_image {
    int32_t  type;            /* VHR_TRANSIENT_*_IMAGE */
    uint32_t width;           /* 0 with height 0 = swapchain (display) sized, render_graph.cpp:960-964 */
    uint32_t height;
    int32_t  format;          /* VkFormat value */
    uint32_t binding;         /* set 3 binding within the pass */
    float    clear_value[4];  /* colour, or [0] = depth */
    int32_t  multisampled;
} vhr_transient_image;

typedef struct vhr_transient_resource {
    int32_t     type;         /* VHR_TRANSIENT_RESOURCE_* */
    const char *name;         /* resources are identified by name across passes; "RENDER_OUTPUT" is the sink */
    vhr_transient_image image;
} vhr_transient_resource;

/* vulkan_common.h:284-296 */
typedef struct vhr_hit_shader {
    const char *closest_hit;
    const char *any_hit;      /* NULL if none */
} vhr_hit_shader;

typedef struct vhr_raytracing_pipeline_description {
    const char *name;
    const char *raygen_shader;
    const char *const *miss_shaders;
    uint32_t miss_shader_count;
    const vhr_hit_shader *hit_shaders;
    uint32_t hit_shader_count;
} vhr_raytracing_pipeline_description;

/* vulkan_common.h:311-318 + PushConstantDescription */
typedef struct vhr_compute_pipeline_description {
    const char *const *kernels;       /* shader names, e.g. "hybrid_render_path/svgf.comp" */
    uint32_t kernel_count;
    uint32_t push_constant_size;
} vhr_compute_pipeline_description;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
static_assert(sizeof(vhr_vertex) == 56, "Vertex");
static_assert(sizeof(vhr_material) == 44, "Material");
static_assert(sizeof(vhr_primitive) == 120, "Primitive");
static_assert(sizeof(vhr_directional_light) == 112, "DirectionalLight");
static_assert(sizeof(vhr_per_frame_data) == 584, "PerFrameData");
static_assert(sizeof(vhr_svgf_push_constants) == 24, "SVGFPushConstants");
static_assert(sizeof(vhr_ssr_push_constants) == 16 && sizeof(vhr_ssao_push_constants) == 4, "SSRPushConstants / SSAOPushConstants");
static_assert(sizeof(vhr_trace_params) == 32, "vhr_trace_params");
#else
_Static_assert(sizeof(vhr_vertex) == 56, "Vertex");
_Static_assert(sizeof(vhr_material) == 44, "Material");
_Static_assert(sizeof(vhr_primitive) == 120, "Primitive");
_Static_assert(sizeof(vhr_directional_light) == 112, "DirectionalLight");
_Static_assert(sizeof(vhr_per_frame_data) == 584, "PerFrameData");
_Static_assert(sizeof(vhr_svgf_push_constants) == 24, "SVGFPushConstants");
_Static_assert(sizeof(vhr_ssr_push_constants) == 16 && sizeof(vhr_ssao_push_constants) == 4, "SSRPushConstants / SSAOPushConstants");
_Static_assert(sizeof(vhr_trace_params) == 32, "vhr_trace_params");
#endif

#endif /* VHR_TYPES_H */
