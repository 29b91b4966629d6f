/*
 * vhr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11 + OpenMP) of the hybrid ray-tracing hot path of
 * RMichelsen/VulkanHybridRenderer.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (vulkanhybridrenderer_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and
 * cannot be built here (Win32 + Vulkan SDK + GLSL->SPIR-V toolchain absent; its
 * ray/triangle + BVH arithmetic lives in the Vulkan driver).  The restatement is
 * pinned only by known-answer values derived by hand from the reference's
 * formulas (tests/golden/kat_*.json) and by an independent numpy restatement.
 * What IS pinned against the reference (round 2, tests/test_reference_pins.py): the
 * struct layouts below against src/rendering_backend/glsl_common.h itself (compiled
 * with the reference's glm by oracle/ref_probes/, `make -C oracle ref`, outputs in
 * tests/golden/ref_abi_layout.json) -- orc_struct_sizes() is part of that test -- and,
 * on the scene-host side, glm / cgltf / stb_image outputs.  The arithmetic of the
 * shaders restated in vhr_oracle.c stays unpinned.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef VHR_ORACLE_H
#define VHR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data ABI: src/rendering_backend/glsl_common.h:22-99 (scalar layout) ---- */
typedef struct { float pos[3]; float normal[3]; float tangent[4]; float uv0[2]; float uv1[2]; } orc_vertex;      /* 56 B */
typedef struct {
    float base_color[4];
    int32_t base_color_texture, metallic_roughness_texture, normal_map;
    float metallic_factor, roughness_factor;
    int32_t alpha_mask;
    float alpha_cutoff;
} orc_material;                                                                                                  /* 44 B */
typedef struct { float transform[16]; orc_material material; uint32_t vertex_offset, index_offset, index_count; } orc_primitive; /* 120 B */
typedef struct { float projview[16]; float direction[4]; float color[4]; float intensity[4]; } orc_directional_light;      /* 112 B */
typedef struct {
    float camera_view[16], camera_proj[16], camera_view_inverse[16], camera_proj_inverse[16];
    float camera_viewproj_inverse[16], camera_view_prev_frame[16], camera_proj_prev_frame[16];
    orc_directional_light directional_light;
    float display_size[2], display_size_inverse[2];
    uint32_t frame_index;
    int32_t blue_noise_texture_index;
} orc_per_frame_data;                                                                                            /* 584 B */

/* Parameters the reference hard-codes inside raygen.rgen (spp, tmax, ...). The
 * defaults reproduce the shader; other values are documented extensions. */
typedef struct {
    uint32_t shadow_enable;   /* raygen.rgen:31-41 (1) */
    uint32_t ao_spp;          /* raygen.rgen:45 (2) */
    float    ao_tmax;         /* raygen.rgen:52 (5.0) */
    uint32_t reflections;     /* raygen.rgen:59-65 (1 = one bounce, 0 = off; 2 = the two-bounce extension of BASELINE config 5) */
    float    cone_cos_max;    /* raygen.rgen:34 (0.999995) */
    float    normal_bias;     /* raygen.rgen:29 (0.1) */
    float    tmin;            /* raygen.rgen:40 (0.01) */
    float    tmax;            /* raygen.rgen:40 (10000.0) */
} orc_trace_params;

typedef struct orc_scene orc_scene;

/* ---- known-answer level entry points ---- */
uint32_t orc_seed_thread(uint32_t seed);
uint32_t orc_random(uint32_t *state);
float    orc_random01(uint32_t *state);
uint32_t orc_random_range(uint32_t *state, uint32_t lower, uint32_t upper);
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);
void     orc_sincos(float phi, float *s, float *c);
void     orc_uniform_sample_cone(float u0, float u1, float cos_theta_max, float out[3]);
void     orc_cosine_hemisphere(float u0, float u1, float out[3]);
void     orc_onb(const float n[3], float M[9]);
int      orc_ray_triangle(const float o[3], const float d[3], const float v0[3], const float e1[3],
                          const float e2[3], float tmin, float tmax, float *t, float *u, float *v);
void     orc_infinite_reverse_depth_projection(float yfov, float aspect, float znear, float out[16]);
void     orc_mat4_inverse(const float m[16], float out[16]);
void     orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void     orc_default_trace_params(orc_trace_params *p);
int      orc_struct_sizes(uint32_t out[8]);

/* ---- scene (resource_manager.cpp:593-718 semantics: world-space two-sided opaque soup) ---- */
orc_scene *orc_scene_create(const orc_vertex *vertices, uint32_t nv, const uint32_t *indices, uint32_t ni,
                            const orc_primitive *primitives, uint32_t np);
void     orc_scene_destroy(orc_scene *s);
/* format: 43 = R8G8B8A8_SRGB, 37 = R8G8B8A8_UNORM; filters 0 nearest / 1 linear;
 * address modes 0 repeat / 1 mirrored repeat / 2 clamp to edge (Vk enum values). */
int      orc_scene_add_texture(orc_scene *s, uint32_t w, uint32_t h, const uint8_t *rgba8, int format,
                               int mag_filter, int min_filter, int address_u, int address_v);
uint32_t orc_scene_triangle_count(const orc_scene *s);

/* any-hit / closest-hit queries for tests: use_bvh = 0 brute force, 1 oracle BVH */
int      orc_scene_occluded(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax, int use_bvh);
int      orc_scene_closest(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax, int use_bvh,
                           float *t, float *u, float *v, uint32_t *prim, uint32_t *tri_in_prim);

/* ---- stand-in G-buffer producer (gbuf.vert:19-28, gbuf.frag:17-59 incl. alpha discard :27-32 and normal mapping
 * :35-41; primary rays that step past discarded fragments) ---- */
void orc_gbuffer(const orc_scene *s, const orc_per_frame_data *pfd, uint32_t W, uint32_t H,
                 uint16_t *normals_ids /*RGBA16F*/, uint16_t *motion_mr /*RGBA16F*/, float *depth /*D32F*/);
/* same, plus the albedo attachment (B8G8R8A8_UNORM: bytes b, g, r, a; gbuf.frag:19-33); albedo may be NULL */
void orc_gbuffer_albedo(const orc_scene *s, const orc_per_frame_data *pfd, uint32_t W, uint32_t H,
                        uint16_t *normals_ids, uint16_t *motion_mr, float *depth, uint8_t *albedo_bgra8);

/* ---- next row f3: composition.vert:5-8 + composition.frag:60-161 ----
 * modes: 0 ray traced, 1 the raster-side alternative (shadow map with the 16-tap PCF of :81-107 / SSAO / SSR), 2 off.  shadow_ao: RGBA16F (denoised) if
 * shadow_ao_channels == 4 else RG16F.  Output: B8G8R8A8_SRGB swapchain texels (bytes b, g, r, a), row 0 = top of the
 * presented image = G-buffer row H-1 (the composition viewport is flipped, pipeline.cpp:175-178). */
void orc_composition(const orc_per_frame_data *pfd, uint32_t W, uint32_t H, int shadow_mode, int ao_mode, int reflection_mode,
                     const uint8_t *albedo_bgra8, const uint16_t *normals_ids, const uint16_t *motion_mr, const float *depth,
                     const uint16_t *shadow_ao, int shadow_ao_channels, const uint16_t *reflections /* mode 0: ray traced, 1: SSR */,
                     const uint16_t *ssao /* "Screen Space Ambient Occlusion" for ao_mode 1, else may be NULL */,
                     const float *shadow_map /* D32F, shadow_map_size^2, for shadow_mode 1, else may be NULL */, uint32_t shadow_map_size,
                     uint8_t *out_bgra8_srgb);
/* Stand-in for the rasterised "Shadow Map Pass" (hybrid_render_path.cpp:58-99): decision (xiv), orthographic closest-hit rays
 * through the texel centres of directional_light.projview's frustum, depth = 1 - t (reverse Z), clear value 0. */
void orc_shadow_map(const orc_scene *s, const orc_per_frame_data *pfd, uint32_t size, uint32_t row_begin, uint32_t row_end,
                    float *shadow_map, int use_bvh);

/* ---- K1 + K2: raygen.rgen:14-66, miss.rmiss, reflection_miss.rmiss, reflection_hit.rchit ---- */
/* vis_mask (optional, may be NULL): per pixel bit0 = shadow ray missed (lit), bit(1+i) = AO ray i missed,
 * bit7 = reflection ray hit, 0x40 = sky pixel. rays_out (optional): [0] unique rays traced. */
void orc_raygen(const orc_scene *s, const orc_per_frame_data *pfd, const orc_trace_params *tp,
                uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
                const uint16_t *normals_ids, const float *depth,
                uint16_t *shadow_ao /*RG16F*/, uint16_t *reflections /*RGBA16F*/,
                uint8_t *vis_mask, uint64_t *rays_out, int use_bvh);

/* ---- next row f4: the raytraced render path (raytraced_render_path.cpp:11-76) ----
 * raytraced_render_path/raygen.rgen + miss.rmiss + shadow_miss.rmiss + closesthit.rchit, or with use_anyhit_shader != 0
 * raygen_test_alpha.rgen + closesthit_test_alpha.rchit + shadow_anyhit.rahit.  Output: "RaytracedOutput", B8G8R8A8_UNORM
 * texels (bytes b, g, r, a).  rays_out (optional): primary + shadow rays traced. */
void orc_raytraced(const orc_scene *s, const orc_per_frame_data *pfd, uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
                   int use_anyhit_shader, uint8_t *out_bgra8, uint64_t *rays_out, int use_bvh);
/* raytraced_render_path/composition.frag:11-13 through the flipped presentation viewport -> B8G8R8A8_SRGB */
void orc_raytraced_composition(uint32_t W, uint32_t H, const uint8_t *raytraced_bgra8, uint8_t *out_bgra8_srgb);

/* ---- next row f4, second half: the screen-space alternatives (hybrid_render_path.cpp:138-243) ----
 * ssao.comp:14-53 -> "Screen Space Ambient Occlusion Raw"; ssao_blur.comp:11-26 -> "Screen Space Ambient Occlusion";
 * ssr.comp:16-137 -> "Screen Space Reflections" (all RGBA16F although the shaders say r16f: every component is stored).
 * radius: the reference never pushes SSAOPushConstants to ssao.comp (the SSAO Pass declares no push constants and
 * dispatches without them, hybrid_render_path.cpp:151-167; the blur pass gets them instead, :182-197), so what the shader
 * reads is undefined in Vulkan; the oracle takes the value as a parameter (0.75 = the value the host initialises, :139-141).
 * kind for orc_sample_linear_repeat: 0 RGBA16F, 1 D32F, 2 B8G8R8A8_UNORM. */
void orc_sample_linear_repeat(int kind, const void *img, uint32_t W, uint32_t H, float u, float v, float out[4]);
void orc_ssao(const orc_per_frame_data *pfd, uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
              const uint16_t *normals_ids, const float *depth, float radius, uint16_t *ssao_raw /*RGBA16F*/);
void orc_ssao_blur(const orc_per_frame_data *pfd, uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
                   const uint16_t *ssao_raw, uint16_t *ssao_blurred /*RGBA16F*/);
void orc_ssr(const orc_per_frame_data *pfd, uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end,
             const uint8_t *albedo_bgra8, const uint16_t *normals_ids, const uint16_t *motion_mr, const float *depth,
             float ray_distance, float step_size, float thickness, int32_t bsearch_steps, uint16_t *ssr_out /*RGBA16F*/);

/* ---- K3: svgf.comp:41-145 ---- */
void orc_svgf_temporal(const orc_per_frame_data *pfd, uint32_t W, uint32_t H,
                       const uint16_t *normals_ids, const uint16_t *motion_mr, const uint16_t *raytraced /*RG16F*/,
                       const uint16_t *prev_normals_ids, const uint16_t *history /*RGBA16F*/,
                       const uint16_t *moments_in /*RG16F snapshot*/,
                       uint16_t *integrated_out /*RGBA16F*/, uint16_t *moments_out /*RG16F*/);

/* ---- K4: svgf_atrous_filter.comp:53-103 ---- */
void orc_svgf_atrous(const orc_per_frame_data *pfd, uint32_t W, uint32_t H, const uint16_t *normals_ids,
                     const uint16_t *integrated_in, uint16_t *integrated_out, int32_t step);

/* ---- host schedule: hybrid_render_path.cpp:245-331 (persistent images + per-frame dispatch order) ---- */
typedef struct orc_svgf_state orc_svgf_state;
orc_svgf_state *orc_svgf_create(uint32_t W, uint32_t H);
void orc_svgf_destroy(orc_svgf_state *st);
void orc_svgf_frame(orc_svgf_state *st, const orc_per_frame_data *pfd, const uint16_t *normals_ids,
                    const uint16_t *motion_mr, const uint16_t *raytraced, uint16_t *denoised_out /*RGBA16F*/);
/* which: 0 integrated.x, 1 integrated.y, 2 prev normals, 3 history (RGBA16F each), 4 moments (RG16F) */
const uint16_t *orc_svgf_image(const orc_svgf_state *st, int which);

/* ---- the audit of decision (vi) (round 6): exact-arithmetic arbiter behind the fp32 ray / triangle decision ----
 * orc_ray_triangle_exact: the decision of Moeller-Trumbore's comparisons without rounding on the same fp32 inputs (vhr_exact.h):
 * 1 hit, 0 miss, -1 undecided in binary64 (settle with tests/exact_rational.py).  out (optional): det, u, v, t as binary64 values.
 * orc_ray_triangle_rules: bit 0 = the fp32 comparisons pass, bit 1 = round 5's residual rule accepts, bit 2 = the box rule (weighed in round 6,
 * dropped) accepts, bit 3 = the rule in force accepts (consistent, or confirmed in binary64).
 * orc_audit_begin / _end: while open, every non-alpha-tested ray the oracle traces is also decided pair by pair in exact arithmetic
 * (brute_force != 0: against every triangle; else through the padded boxes walked in binary64).  Not re-entrant; one audit at a time. */
enum { AUD_RULES = 4 };
typedef struct {
    uint64_t rays[2];                 /* [any-hit, closest-hit] rays audited */
    uint64_t rays_undecided;          /* rays with at least one pair the binary64 filter left undecided: left out of the per-ray counts */
    uint64_t rays_not_finite;
    uint64_t pairs, undecided;        /* (ray, triangle) pairs decided / left undecided */
    uint64_t exact_hits, mt_hits;     /* pairs that are exact hits / whose fp32 comparisons pass */
    uint64_t mt_miss_exact_hit;       /* class E: fp32 Moeller-Trumbore itself misses an exact hit */
    uint64_t cls[AUD_RULES][4];       /* of the pairs whose fp32 comparisons pass, per rule: A accepted & exact hit, B rejected & exact miss,
                                         C rejected & exact hit, D accepted & exact miss */
    uint64_t any_leak[AUD_RULES], any_spurious[AUD_RULES];   /* any-hit rays: fp32 unoccluded & exactly occluded / the converse */
    uint64_t closest_hit_miss[AUD_RULES];                    /* closest-hit rays: one side hits, the other misses */
    uint64_t closest_differs[AUD_RULES], closest_differs_far[AUD_RULES];   /* another triangle wins / and its t is > 1e-4 max(1, t) away */
    uint64_t records_dropped;
    uint64_t escalated;               /* pairs whose fp32 solution contradicted itself and went to binary64 */
} orc_audit_counts;
typedef struct {
    float o[3], d[3], tmin, tmax, v0[3], e1[3], e2[3];
    float t, u, v, det;               /* fp32 Moeller-Trumbore's values (t, u, v only where it got that far) */
    double xdet, xu, xv, xt;          /* binary64 values */
    uint32_t flat;
    uint8_t mt, pass_mask, any_hit;
    int8_t exact;
    char cls;                         /* 'B' 'C' 'D' 'E' as above, 'U' undecided */
    char pad_[3];
} orc_audit_record;
void     orc_audit_begin(int brute_force, uint32_t max_records);
uint32_t orc_audit_end(orc_audit_counts *out, orc_audit_record *records /* room for max_records */);
int      orc_ray_triangle_exact(const float o[3], const float d[3], const float v0[3], const float e1[3], const float e2[3],
                                float tmin, float tmax, double out_det_u_v_t[4]);
int      orc_ray_triangle_rules(const float o[3], const float d[3], const float v0[3], const float e1[3], const float e2[3],
                                float tmin, float tmax, float out_t_u_v_det[4]);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
