/*
 * vhr_amd.h -- C ABI of libvhr_amd.so: the MI355X-native (HIP / gfx950) implementation of the
 * ray-traced shadow / AO / mirror-reflection pass and the SVGF denoiser of
 * RMichelsen/VulkanHybridRenderer, behind the reference's render-graph pass API.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  All functions return 0 on success and a negative code on failure unless noted;
 * vhr_last_error() returns the message.  The reference has no error returns (VK_CHECK asserts,
 * vulkan_common.h:4-7); pool exhaustion returns -1 like resource_manager.cpp:847-848,876-877.
 *
 * Threading and streams: like the reference (renderer.cpp:184-235) one host thread drives one context.  Everything a frame PUBLISHES is
 * issued in order on the context's stream (the one given at creation, or an internal one with VHR_CREATE_INTERNAL_STREAM): after
 * vhr_graph_execute returns, work the caller enqueues on that stream sees every image the graph's passes declare as outputs.  Two things
 * leave that stream, all owned and joined by the library:
 *   - "svgf_async_unread" (default 1): a compute pass's a-trous dispatch whose output nothing reads (the reference's fifth iteration,
 *     hybrid_render_path.cpp:299-328) runs on a library-owned SIDE stream beside whatever the context's stream does next; the context's
 *     stream waits for it before the next compute pass, before storage-image uploads / downloads / vhr_get_storage_image and in
 *     vhr_synchronize.  A caller that reads that dispatch's storage image itself on the context's stream must call vhr_synchronize first
 *     or set the option to 0 (every dispatch in recorded order on the one stream).
 *   - "reflection_async" (default 1): the mirror ray's launch runs on a second library-owned stream behind the shadow / AO launch, beside the
 *     SVGF pass; the context's stream waits for it before a pass epilogue (not with value 2), before the frame's next external pass, at the
 *     end of vhr_graph_execute (so the sentence above holds for the Reflections image too), before image uploads, before downloads of and
 *     vhr_get_transient_image on the Reflections image, in vhr_get_current_stream (not with value 2: a caller who asks for the stream is
 *     about to enqueue kernels of its own, which may read Reflections or rewrite the G-buffer the launch reads) and in vhr_synchronize.
 *     With value 2 (the multi-GPU harness) a caller's own kernels that touch Reflections or the G-buffer inside a frame must call
 *     vhr_get_transient_image on Reflections (or vhr_synchronize) first.
 *   - "frames_in_flight" 2 / 3 (opt-in): the front of a frame (up to its last ray-tracing pass) runs on a second stream; vhr_get_current_stream
 *     tells an external pass which stream to enqueue on.
 * Pass time stamps ("pass_timestamps", vhr_graph_gather_performance_statistics) cover what the CONTEXT'S stream executes between a pass's
 * first kernel and the next kernel behind it: "Raytrace Pass" covers the shadow / AO launch (+ the mirror ray's with "reflection_async" 0);
 * with the side stream on, "SVGF Denoise Pass" covers 1 temporal + 4 a-trous dispatches + the
 * blits where the reference's vkCmdWriteTimestamp pair covers five a-trous dispatches (render_graph.cpp:167-182); with "svgf_async_unread" 0
 * it covers all five.  In mode 1 the END of a pass is stored by the next LIBRARY kernel on the stream, so GPU work an external (graphics)
 * pass's callback enqueues on the stream right behind a library pass is charged to that pass; mode 2 closes the pass in front of every such
 * callback and at the end of vhr_graph_execute (a one-thread kernel, +6 us each).
 * One context per GPU / per process for multi-GPU (screen tiles, vhr_set_tile; row strips, vhr_set_strip).
 */
#ifndef VHR_AMD_H
#define VHR_AMD_H

#include "vhr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vhr_context vhr_context;
typedef struct vhr_raytracing_execution_context vhr_raytracing_execution_context;
typedef struct vhr_compute_execution_context vhr_compute_execution_context;

enum {
    VHR_OK = 0,
    VHR_ERROR_INVALID_ARGUMENT = -1,
    VHR_ERROR_DEVICE = -2,
    VHR_ERROR_NOT_FOUND = -3,
    VHR_ERROR_OUT_OF_SLOTS = -4,
    VHR_ERROR_GRAPH = -5,
    VHR_ERROR_NO_DEVICE = -6,
    VHR_ERROR_UNSUPPORTED = -7      /* what the index-returning calls use for invalid arguments: their -1 is the reference's "exhausted" sentinel */
};

/* ---------------------------------------------------------------------------------------------
 * Context (replaces VulkanContext + the device half of ResourceManager:
 * src/rendering_backend/vulkan_context.cpp:44-89, resource_manager.cpp:15-70)
 * ------------------------------------------------------------------------------------------- */
/* vhr_create_info.flags */
enum {
    /* Pass registry / execution order / SanityCheck only: no HIP call is made, nothing can be uploaded,
     * executed or downloaded.  Lets the host-side graph logic be checked on a machine without a GPU. */
    VHR_CREATE_HOST_ONLY = 1,
    /* Ignore vhr_create_info.stream and create an internal (non-blocking) HIP stream. */
    VHR_CREATE_INTERNAL_STREAM = 2
};

typedef struct vhr_create_info {
    int32_t  device;          /* HIP device ordinal */
    uint32_t width;           /* display ("swapchain") size: context.swapchain.extent */
    uint32_t height;
    void    *stream;          /* hipStream_t to issue all work on, used as given (NULL = the default stream) */
    uint32_t flags;           /* 0 or VHR_CREATE_* bits */
} vhr_create_info;

int  vhr_create(const vhr_create_info *info, vhr_context **out);
void vhr_destroy(vhr_context *ctx);
/* VulkanContext::Resize (vulkan_context.cpp:118-120), the first half of the reference's one recovery route -- renderer.cpp:113-118,146-154:
 * `context->Resize(); active_render_path->Build();` when the swapchain is out of date.  The display extent changes; what was sized by the old one
 * is released: the graph (transient images, pass registry: RenderPath::Build would destroy it first anyway, render_path.cpp:14-20) and every image
 * of the storage pool (the render path's SVGF history, which its RegisterPath allocates again).  Geometry, the acceleration structure (K0 is NOT
 * rebuilt: vhr_get_build_times keeps its values), textures, options, trace parameters and kernel timers stay; the screen tile (vhr_set_tile) is the
 * whole image again.  Then the path registers and builds at the new extent: vhr_hybrid_build / vhr_raytraced_build (which read the extent from
 * the context) or RenderPath::Build() behind DeviceContext::Resize() (vhr_render_graph.hpp).  Frames after that equal a fresh context's bit for bit. */
int  vhr_resize(vhr_context *ctx, uint32_t width, uint32_t height);
const char *vhr_last_error(const vhr_context *ctx);   /* ctx may be NULL: error of the last failed vhr_create */
int  vhr_synchronize(vhr_context *ctx);               /* hipStreamSynchronize on the context stream */
/* The HIP stream (hipStream_t) the library is enqueueing on right now.  Outside vhr_graph_execute that is the stream given to
 * vhr_create.  Inside a pass or epilogue callback it is the stream that pass is ordered on: with "frames_in_flight" > 1 the passes up to
 * the last ray-tracing pass run on a second, library-owned stream, and an external graphics pass (the G-buffer producer) MUST enqueue its
 * work there -- or make that stream wait for its own -- for the Raytrace Pass to see a complete G-buffer (render_graph.cpp:722-796 orders
 * the same hand-over with image barriers).  The call also makes the stream wait for a held-back ray-tracing launch ("fuse_temporal") and for
 * the mirror ray's pending launch ("reflection_async" 1): whatever the caller enqueues behind it sees the frame's images as the single-stream
 * schedule would leave them. */
int  vhr_get_current_stream(vhr_context *ctx, void **stream);
const char *vhr_version(void);
int  vhr_abi_struct_sizes(uint32_t out[8]);            /* vertex, material, primitive, light, per-frame, push constants, trace params, 0 */
void vhr_default_trace_params(vhr_trace_params *out); /* raygen.rgen:29-65 constants */

/* ---------------------------------------------------------------------------------------------
 * ResourceManager (src/rendering_backend/resource_manager.h:16-78)
 * ------------------------------------------------------------------------------------------- */
/* UpdateGeometry (resource_manager.h:34, .cpp:291-360) + UpdateBLAS (.cpp:593-701) + UpdateTLAS
 * (.cpp:703-801): uploads the flat vertex / index / primitive arrays and builds the acceleration
 * structure -- here a host-built binned-SAH BVH2 over the world-space (transform-baked), two-sided,
 * all-opaque triangle soup.  gl_GeometryIndexEXT == primitive index, gl_PrimitiveID == triangle index
 * within the primitive.  On a VHR_CREATE_HOST_ONLY context the arrays are validated and the tree is built and checked
 * (vhr_get_bvh_statistics, vhr_get_bvh_form_checks, vhr_get_build_times), nothing is uploaded. */
int vhr_update_geometry(vhr_context *ctx, const vhr_vertex *vertices, uint32_t vertex_count,
                        const uint32_t *indices, uint32_t index_count,
                        const vhr_primitive *primitives, uint32_t primitive_count);
/* UploadTextureFromData (resource_manager.h:26, .cpp:152-196): RGBA8 texels, format
 * VHR_FORMAT_R8G8B8A8_{SRGB,UNORM}, sampler NULL = default sampler (LINEAR / REPEAT, .cpp:58-69).
 * Returns the bindless texture index (>= 0) or a negative error. */
int32_t vhr_upload_texture_from_data(vhr_context *ctx, uint32_t width, uint32_t height, const uint8_t *data,
                                     int32_t format, const vhr_sampler_info *sampler_info);
/* UploadNewStorageImage (resource_manager.h:28, .cpp:230-263): returns the first free slot index
 * (< 2048) of the bindless storage-image pool, or -1 when exhausted.  Contents are zero-initialised
 * (the reference leaves them undefined). */
int32_t vhr_upload_new_storage_image(vhr_context *ctx, uint32_t width, uint32_t height, int32_t format);
/* DestroyStorageImage (resource_manager.h:29, .cpp:265-269) */
int vhr_destroy_storage_image(vhr_context *ctx, int32_t id);
/* UpdatePerFrameUBO (resource_manager.h:35, .cpp:362-364); resource_idx < 3 (MAX_FRAMES_IN_FLIGHT) */
int vhr_update_per_frame_ubo(vhr_context *ctx, uint32_t resource_idx, const vhr_per_frame_data *per_frame_data);
/* extension: the constants raygen.rgen hard-codes */
int vhr_set_trace_params(vhr_context *ctx, const vhr_trace_params *params);

/* ---------------------------------------------------------------------------------------------
 * RenderGraph (src/render_graph/render_graph.h:10-21)
 * ------------------------------------------------------------------------------------------- */
typedef void (*vhr_external_pass_callback)(void *user, vhr_context *ctx);
typedef void (*vhr_raytracing_pass_callback)(void *user, vhr_raytracing_execution_context *exec);
typedef void (*vhr_compute_pass_callback)(void *user, vhr_compute_execution_context *exec);

/* DestroyResources (render_graph.h:8, .cpp:16-68): drops passes, pipelines and transient images */
int vhr_graph_destroy_resources(vhr_context *ctx);
/* AddGraphicsPass (render_graph.h:10-12, .cpp:70-85).  Raster passes (G-buffer, composition) stay with
 * the integrator: the callback runs at the pass's place in the execution order and is expected to
 * produce / consume the named transient images (through imported memory, vhr_graph_bind_external_image,
 * or vhr_upload_image).  `callback` may be NULL. */
int vhr_graph_add_graphics_pass(vhr_context *ctx, const char *render_pass_name,
                                const vhr_transient_resource *dependencies, uint32_t dependency_count,
                                const vhr_transient_resource *outputs, uint32_t output_count,
                                vhr_external_pass_callback callback, void *user);
/* AddRaytracingPass (render_graph.h:13-15, .cpp:87-101).  The shader names select HIP kernels:
 * raygen "hybrid_render_path/raygen.rgen", miss[0] "hybrid_render_path/miss.rmiss", miss[1]
 * "hybrid_render_path/reflection_miss.rmiss", hit[0].closest_hit
 * "hybrid_render_path/reflection_hit.rchit" (hybrid_render_path.cpp:112-124). */
int vhr_graph_add_raytracing_pass(vhr_context *ctx, const char *render_pass_name,
                                  const vhr_transient_resource *dependencies, uint32_t dependency_count,
                                  const vhr_transient_resource *outputs, uint32_t output_count,
                                  const vhr_raytracing_pipeline_description *pipeline,
                                  vhr_raytracing_pass_callback callback, void *user);
/* AddComputePass (render_graph.h:16-18, .cpp:103-116).  Known kernels:
 * "hybrid_render_path/svgf.comp", "hybrid_render_path/svgf_atrous_filter.comp" (SVGFPushConstants, 24 bytes), and the
 * screen-space alternatives of hybrid_render_path.cpp:138-243: "hybrid_render_path/ssao.comp" (bindings 0 normals, 1 depth,
 * 2 output; dispatched WITHOUT push constants, as the reference does although the shader reads SSAOPushConstants.radius --
 * the kernel uses the SSAOPushConstants last pushed on this context by any dispatch, 0.75 before the first),
 * "hybrid_render_path/ssao_blur.comp" (0 input, 1 output; SSAOPushConstants, 4 bytes, which the shader ignores) and
 * "hybrid_render_path/ssr.comp" (0 albedo, 1 normals, 2 motion / metallic-roughness, 3 depth, 4 output; SSRPushConstants,
 * 16 bytes).  Their inputs must be whole images (they sample anywhere on screen, REPEAT-wrapped); with row strips only the
 * owned output rows are computed (ssao.comp six more on either side for the blur).  A shader name is a
 * global key: registering it in two passes fails (render_graph.cpp:677). */
int vhr_graph_add_compute_pass(vhr_context *ctx, const char *render_pass_name,
                               const vhr_transient_resource *dependencies, uint32_t dependency_count,
                               const vhr_transient_resource *outputs, uint32_t output_count,
                               const vhr_compute_pipeline_description *pipeline,
                               vhr_compute_pass_callback callback, void *user);
/* Build (render_graph.h:20, .cpp:118-149): creates every named transient image, derives the execution
 * order by BFS from the single writer of "RENDER_OUTPUT" (.cpp:686-720) and runs SanityCheck
 * (.cpp:980-1021). */
int vhr_graph_build(vhr_context *ctx);
/* Execute (render_graph.h:21, .cpp:151-187): runs the pass callbacks in order on the context stream,
 * bracketing each pass with a pair of timing events (the reference's timestamp queries). */
int vhr_graph_execute(vhr_context *ctx, uint32_t resource_idx, uint32_t image_idx);
/* GatherPerformanceStatistics (render_graph.h:22, .cpp:189-201): blocks on the last Execute and folds
 * the per-pass times into the reference's EMA (0.95 * old + 0.05 * new). */
int vhr_graph_gather_performance_statistics(vhr_context *ctx);
/* pass_timestamps[name] (render_graph.cpp:199): EMA and last sample, milliseconds */
int vhr_graph_get_pass_time_ms(vhr_context *ctx, const char *render_pass_name, double *ema_ms, double *last_ms);
/* execution_order (render_graph.h:47): pass names joined by '\n' into buf; returns the pass count */
int vhr_graph_get_execution_order(vhr_context *ctx, char *buf, uint32_t buf_size);
/* ContainsImage / GetImageFormat (render_graph.h:25-26) */
int vhr_graph_contains_image(vhr_context *ctx, const char *image_name);
int32_t vhr_graph_get_image_format(vhr_context *ctx, const char *image_name);

/* extension (multi-GPU / interop hook): called right after the named pass's callback returns, on the
 * host thread, with the pass's work already enqueued on the context stream */
int vhr_graph_set_pass_epilogue(vhr_context *ctx, const char *render_pass_name,
                                vhr_external_pass_callback callback, void *user);
/* extension (interop): point a transient image at externally owned device memory of the same extent and
 * format (what VK_KHR_external_memory import would provide), 16-byte aligned.  NULL restores the context-owned memory. */
int vhr_graph_bind_external_image(vhr_context *ctx, const char *image_name, void *device_ptr);

/* ---------------------------------------------------------------------------------------------
 * RaytracingExecutionContext (src/render_graph/raytracing_execution_context.h:13)
 * ------------------------------------------------------------------------------------------- */
/* TraceRays(width, height) -> vkCmdTraceRaysKHR(w, h, 1) (raytracing_execution_context.cpp:4-13) */
int vhr_trace_rays(vhr_raytracing_execution_context *exec, uint32_t width, uint32_t height);

/* ---------------------------------------------------------------------------------------------
 * ComputeExecutionContext (src/render_graph/compute_execution_context.h:17-31)
 * ------------------------------------------------------------------------------------------- */
/* GetDisplaySize (:17) */
int vhr_compute_get_display_size(vhr_compute_execution_context *exec, uint32_t *width, uint32_t *height);
/* Dispatch (:18) and Dispatch<T> (:20-27): push constants are copied at call time (vkCmdPushConstants),
 * push_constants_size must equal the size declared at registration (assert at :23). */
int vhr_compute_dispatch(vhr_compute_execution_context *exec, const char *shader, uint32_t x_groups,
                         uint32_t y_groups, uint32_t z_groups, const void *push_constants,
                         uint32_t push_constants_size);
/* BlitImageStorageToTransient / TransientToStorage / StorageToStorage (:29-31, .cpp:31-176): same-extent
 * VK_FILTER_NEAREST blits == copies */
int vhr_compute_blit_image_storage_to_transient(vhr_compute_execution_context *exec, int32_t src, const char *dst);
int vhr_compute_blit_image_transient_to_storage(vhr_compute_execution_context *exec, const char *src, int32_t dst);
int vhr_compute_blit_image_storage_to_storage(vhr_compute_execution_context *exec, int32_t src, int32_t dst);

/* ---------------------------------------------------------------------------------------------
 * HybridRenderPath (src/render_paths/hybrid_render_path.{h,cpp}, render_path.{h,cpp}) re-hosted on the
 * API above; these entry points expose the C++ re-host (csrc/hybrid_render_path.cpp) to C callers.
 * ------------------------------------------------------------------------------------------- */
typedef struct vhr_hybrid_render_path vhr_hybrid_render_path;
typedef struct vhr_hybrid_settings {
    int32_t shadow_mode;             /* 0 raytraced, 1 rasterized, 2 off   (hybrid_render_path.h:4-8)  */
    int32_t ambient_occlusion_mode;  /* 0 raytraced, 1 ssao, 2 off        (hybrid_render_path.h:10-14) */
    int32_t reflection_mode;         /* 0 raytraced, 1 ssr, 2 off         (hybrid_render_path.h:16-20) */
    int32_t denoise_shadow_and_ao;   /* bool                              (hybrid_render_path.h:35)    */
    int32_t atrous_steps;            /* 5 in the reference (hybrid_render_path.cpp:299)               */
} vhr_hybrid_settings;
/* g-buffer / composition are external passes: callbacks may be NULL */
int  vhr_hybrid_create(vhr_context *ctx, const vhr_hybrid_settings *settings,
                       vhr_external_pass_callback gbuffer_pass, void *gbuffer_user,
                       vhr_external_pass_callback composition_pass, void *composition_user,
                       vhr_hybrid_render_path **out);
void vhr_hybrid_destroy(vhr_hybrid_render_path *path);      /* DeregisterPath + free */
int  vhr_hybrid_build(vhr_hybrid_render_path *path);        /* RenderPath::Build   (render_path.cpp:14-20) */
int  vhr_hybrid_rebuild(vhr_hybrid_render_path *path, const vhr_hybrid_settings *settings); /* Rebuild (:22-27) */
int  vhr_hybrid_get_push_constants(vhr_hybrid_render_path *path, vhr_svgf_push_constants *out);
const char *vhr_hybrid_last_error(vhr_hybrid_render_path *path);
/* Checkpoint / resume of the path's cross-frame state (the reference has none; its only state that survives a frame is
 * the five persistent SVGF storage images of hybrid_render_path.cpp:247-262, the previous frame's view / projection matrices and
 * frame_index, renderer.cpp:187-190,202).  The blob is host memory: a header (extent, formats, byte counts), the PerFrameData the last
 * vhr_graph_execute ran with -- a restored renderer continues with view_prev / proj_prev = its view / proj and frame_index + 1, which is
 * what Renderer::Render's function-static carries --, then the images in the order integrated[0], integrated[1], previous normals, history,
 * moments history, each as the path's NEXT frame will see it (the moments double buffer's current side; the ping-pong pair in its
 * frame-start order), so a blob loads into any path of the same extent whatever pool indices that path was given.  A context restored
 * from a blob continues bit-identically (tests/test_gpu_svgf.py).  `last_frame` may be NULL. */
int  vhr_hybrid_state_size(vhr_hybrid_render_path *path, uint64_t *bytes);
int  vhr_hybrid_save_state(vhr_hybrid_render_path *path, void *blob, uint64_t bytes);
int  vhr_hybrid_load_state(vhr_hybrid_render_path *path, const void *blob, uint64_t bytes, vhr_per_frame_data *last_frame);
/* The PerFrameData the last vhr_graph_execute of the context ran with (zeros before the first). */
int  vhr_get_last_per_frame_ubo(vhr_context *ctx, vhr_per_frame_data *out);

/* ---------------------------------------------------------------------------------------------
 * RaytracedRenderPath (src/render_paths/raytraced_render_path.{h,cpp}) re-hosted on
 * the API above (csrc/raytraced_render_path.cpp): "Raytracing Pass" -> "RaytracedOutput" (B8G8R8A8_UNORM) ->
 * external "Composition Pass".  use_anyhit_shader = the "Alpha test for shadows" switch (raytraced_render_path.h:15).
 * ------------------------------------------------------------------------------------------- */
typedef struct vhr_raytraced_render_path vhr_raytraced_render_path;
int  vhr_raytraced_create(vhr_context *ctx, int32_t use_anyhit_shader, vhr_external_pass_callback composition_pass,
                          void *composition_user, vhr_raytraced_render_path **out);
void vhr_raytraced_destroy(vhr_raytraced_render_path *path);
int  vhr_raytraced_build(vhr_raytraced_render_path *path);                              /* RenderPath::Build (render_path.cpp:14-20) */
int  vhr_raytraced_rebuild(vhr_raytraced_render_path *path, int32_t use_anyhit_shader); /* toggle + Rebuild (raytraced_render_path.cpp:90-92) */
const char *vhr_raytraced_last_error(vhr_raytraced_render_path *path);

/* ---------------------------------------------------------------------------------------------
 * Harness / test access (no reference counterpart: the reference inspects images through its ImGui
 * debug-texture viewer, renderer.cpp:215-224)
 * ------------------------------------------------------------------------------------------- */
typedef struct vhr_image_info {
    void    *device_ptr;      /* linear, row-major, tightly packed */
    uint32_t width, height;
    int32_t  format;
    uint32_t bytes_per_pixel;
} vhr_image_info;
int vhr_get_display_size(vhr_context *ctx, uint32_t *width, uint32_t *height);   /* context.swapchain.extent */
int vhr_get_transient_image(vhr_context *ctx, const char *name, vhr_image_info *out);
int vhr_get_storage_image(vhr_context *ctx, int32_t id, vhr_image_info *out);
/* synchronous copies (stream-ordered, then waited) */
int vhr_upload_transient_image(vhr_context *ctx, const char *name, const void *host_data, uint64_t bytes);
int vhr_download_transient_image(vhr_context *ctx, const char *name, void *host_data, uint64_t bytes);
int vhr_upload_storage_image(vhr_context *ctx, int32_t id, const void *host_data, uint64_t bytes);
int vhr_download_storage_image(vhr_context *ctx, int32_t id, void *host_data, uint64_t bytes);

/* Stand-in producer for the untouched G-buffer stage (gbuf.vert:19-28, gbuf.frag:17-59): casts primary
 * rays through the same BVH and writes the three named transient images with gbuf.frag's encodings
 * (normals+object id RGBA16F, motion+metallic/roughness RGBA16F, reverse-Z depth D32F) and the clears of
 * hybrid_render_path.cpp:16-19.  Alpha-masked / fully transparent fragments are discarded like gbuf.frag:27-32 (the
 * primary ray steps past them, up to 32 layers) and normal maps perturb the normal like :35-41 (SURVEY.md section 8 row
 * f2).  Uses the per-frame data of resource_idx. */
int vhr_standin_gbuffer(vhr_context *ctx, uint32_t resource_idx, const char *normals_image,
                        const char *motion_image, const char *depth_image);
/* same, also writing the "Albedo" attachment (B8G8R8A8_UNORM, gbuf.frag:19-33) */
int vhr_standin_gbuffer_with_albedo(vhr_context *ctx, uint32_t resource_idx, const char *albedo_image, const char *normals_image,
                                    const char *motion_image, const char *depth_image);

/* Stand-in for the rasterised "Shadow Map Pass" (hybrid_render_path.cpp:58-99, depth_prepass.vert:16-19; BASELINE configs[0]'s
 * shadow map): fills the named square D32_SFLOAT transient image ("Shadow Map", 4096 x 4096) with the depth of the closest hit of
 * the orthographic ray through every texel centre of PerFrameData.directional_light.projview's frustum (reverse Z: 1 on the near
 * plane, clear value 0 where nothing is hit).  A ray caster on the path's own BVH, not a rasteriser. */
int vhr_standin_shadow_map(vhr_context *ctx, uint32_t resource_idx, const char *shadow_map_image);

/* Next row: stand-in for the untouched composition stage -- composition.vert:5-8 +
 * composition.frag:60-161: shadows ray traced (0), from the shadow map with the shader's 16-tap PCF (1, :81-107) or off (2);
 * ambient occlusion and reflections ray traced (0), screen space (1: ssao.comp + ssao_blur.comp / ssr.comp, row f4) or
 * off (2).  Reads the named transient images, writes swapchain-format texels (B8G8R8A8_SRGB, bytes
 * b g r a, presentation orientation: row 0 = top) into a storage image of 4-byte texels. */
typedef struct vhr_composition_desc {
    int32_t shadow_mode, ambient_occlusion_mode, reflection_mode;      /* the three specialization constants, :6-8 */
    const char *albedo_image, *normals_image, *motion_image, *depth_image;
    const char *shadow_ao_image;      /* "Denoised Raytraced Shadows and Ambient Occlusion" or the raw RG16F image (:353-355) */
    const char *reflections_image;    /* "Raytraced Reflections" (mode 0) / "Screen Space Reflections" (mode 1); may be NULL for mode 2 */
    int32_t output_storage_image;
    const char *ssao_image;           /* "Screen Space Ambient Occlusion" for ambient_occlusion_mode 1, else may be NULL */
    const char *shadow_map_image;     /* "Shadow Map" (square D32_SFLOAT) for shadow_mode 1, else may be NULL */
} vhr_composition_desc;
int vhr_standin_composition(vhr_context *ctx, uint32_t resource_idx, const vhr_composition_desc *desc);

/* Next row: the raytraced render path (raytraced_render_path.cpp:11-76).  Its "Raytracing Pass"
 * is registered through vhr_graph_add_raytracing_pass with the shader set
 *   raygen "raytraced_render_path/raygen.rgen", miss { ".../miss.rmiss", ".../shadow_miss.rmiss" },
 *   hit group 0 { closest_hit ".../closesthit.rchit" }                                   (:19-34, use_anyhit_shader == 0)
 * or raygen ".../raygen_test_alpha.rgen", the same miss shaders,
 *   hit group 0 { closest_hit ".../closesthit_test_alpha.rchit", any_hit ".../shadow_anyhit.rahit" }   (use_anyhit_shader == 1)
 * and one storage-image output at binding 0 ("RaytracedOutput", B8G8R8A8_UNORM, :15); vhr_trace_rays then launches the
 * primary-ray kernel.  This entry is the stand-in for the path's untouched composition stage
 * (raytraced_render_path/composition.vert:5-8, composition.frag:11-13): the named image sampled at the texel centres and
 * written as swapchain texels (B8G8R8A8_SRGB, presentation orientation) into a storage image of 4-byte texels. */
int vhr_standin_raytraced_composition(vhr_context *ctx, const char *raytraced_output_image, int32_t output_storage_image);

/* Multi-GPU row strips: this context owns rows [row_begin, row_end) of the
 * display.  Ray tracing runs on the owned rows; the SVGF kernels on the owned rows extended by `overlap`
 * rows on each side (recomputed instead of exchanged between a-trous iterations); the blits copy the owned
 * rows extended by `halo` rows (>= overlap: the rows next frame's temporal pass may read) -- a blit fused into
 * the a-trous launch that produced its source ("fuse_blits") stores that launch's rows only, which are the valid
 * ones.  All ranges are clamped to the image.  Default: whole image, overlap 0, halo 0.  The halo rows of the
 * history images are filled by the caller's neighbour exchange (vulkanhybridrenderer_amd/tiling.py over RCCL). */
int vhr_set_strip(vhr_context *ctx, uint32_t row_begin, uint32_t row_end, uint32_t overlap, uint32_t halo);
/* The same for a SCREEN TILE (BASELINE.json north_star: "the framebuffer shards by screen tile"): this context owns the rectangle
 * [col_begin, col_end) x [row_begin, row_end).  The ray queue kernels trace it (plus `overlap` pixels all round with "trace_overlap"),
 * svgf.comp and the a-trous launches compute it extended by `overlap` on both axes, the blits copy it extended by halo_rows /
 * halo_cols (>= overlap: what next frame's svgf.comp may read).  The kernels' A-B variants and the other paths' kernels compute whole
 * rows of the tile's row range instead: a superset, same results inside the rectangle.  vhr_set_strip is the all-columns case. */
int vhr_set_tile(vhr_context *ctx, uint32_t col_begin, uint32_t col_end, uint32_t row_begin, uint32_t row_end, uint32_t overlap,
                 uint32_t halo_rows, uint32_t halo_cols);

/* ---- C1 / C2: the row-strip decomposition's exchanges inside the library (RCCL point-to-point, one process per GPU) ----------
 * The reference is single-GPU (one queue, renderer.cpp:135); these calls exist for an integrator that shards the framebuffer by
 * row strips.  The row arithmetic follows the reference's SVGF schedule (hybrid_render_path.cpp:288-329:
 * the published image is the output of a-trous iteration n-2, iteration i reads +-2*2^i rows) and is the one
 * vulkanhybridrenderer_amd/tiling.py uses (tests/test_comm_plan.py compares the two).  RCCL is loaded on first use: the copy the process
 * already holds (a PyTorch process), else the ROCm installation's; vhr_comm_use_library(path), called before any other vhr_comm_* call,
 * names the library to load instead (a site's own build) -- the library reads no environment variable for this.  vhr_comm_library() says which
 * file the entry points came from.  N > 1 has not run on two DEVICES yet (rounds 1-6 had one GPU per box); world 2, 3 and 4 run on one GPU
 * through tests/rccl_shim (a stand-in for the eight RCCL entry points this file uses, handed over through vhr_comm_use_library), bit-identical to the
 * torch.distributed route and to the single context, with one injected failure per error path (tests/test_comm_shim.py).  The stand-in copies with
 * blocking host calls: the ORDER of this file's stream and event dependencies against real RCCL's asynchronous transport is what stays unverified.
 * Error handling: a failure inside a grouped batch closes the group, marks the communicator unusable and is reported; what was
 * enqueued before it is drained by vhr_comm_finish_frame_exchanges. */
typedef struct vhr_strip_plan {
    uint32_t rank, world, height;
    uint32_t row_begin, row_end;     /* owned rows [g*H/N, (g+1)*H/N) */
    uint32_t overlap;                /* E: rows the SVGF kernels recompute beyond the strip (30 for the reference's 5 iterations) */
    uint32_t halo;                   /* Hh = E + ceil(max |motion.y| * H) + 2: rows of history / moments fetched from each neighbour */
} vhr_strip_plan;
typedef struct vhr_row_exchange { int32_t peer; uint32_t send_begin, send_end, recv_begin, recv_end; } vhr_row_exchange;
/* Screen tiles: a grid of grid_rows x grid_cols rectangles, rank = tile_row * grid_cols + tile_col; tile (r, c) owns columns
 * [c*W/C, (c+1)*W/C) and rows [r*H/R, (r+1)*H/R).  Row strips are the one-column grid. */
#define VHR_TILE_MAX_GRID 16        /* tiles per axis */
typedef struct vhr_tile_plan {
    uint32_t rank, world, width, height;
    uint32_t grid_rows, grid_cols;
    uint32_t col_begin, col_end, row_begin, row_end;     /* the owned rectangle */
    uint32_t overlap;                /* E: pixels the SVGF kernels recompute beyond the rectangle, on every cut side */
    uint32_t halo_rows, halo_cols;   /* E + ceil(max |motion| * extent) + 2 on a cut axis (E on an axis that is not cut): the margin of
                                      * history / moments fetched from the neighbours */
    /* the grid's cut lines: tile (r, c) owns columns [col_cut[c], col_cut[c + 1]) and rows [row_cut[c][r], row_cut[c][r + 1]) -- every COLUMN of tiles has its
     * own row cuts (vhr_tile_plan_make: the same in every column, at equal pixels; vhr_tile_plan_make_weighted: the columns split the cost, then every column
     * splits its own).  col_cut[0] = row_cut[c][0] = 0, col_cut[grid_cols] = width, row_cut[c][grid_rows] = height.  Entries past the grid are 0. */
    uint32_t col_cut[VHR_TILE_MAX_GRID + 1], row_cut[VHR_TILE_MAX_GRID][VHR_TILE_MAX_GRID + 1];
} vhr_tile_plan;
typedef struct vhr_rect { uint32_t x0, x1, y0, y1; } vhr_rect;                 /* [x0, x1) x [y0, y1) */
typedef struct vhr_rect_exchange { int32_t peer; vhr_rect send, recv; } vhr_rect_exchange;     /* an empty rectangle is all zeros */
#define VHR_COMM_UNIQUE_ID_BYTES 128
typedef struct vhr_comm vhr_comm;

uint32_t vhr_atrous_overlap(uint32_t atrous_steps);
uint32_t vhr_atrous_output_extent(uint32_t overlap, uint32_t step);     /* what "strip_shrink_overlap" makes an a-trous launch compute */
/* VHR_ERROR_OUT_OF_SLOTS when the thinnest strip is thinner than the history halo (use fewer GPUs or a taller image) */
int vhr_strip_plan_make(uint32_t height, uint32_t world, uint32_t rank, uint32_t max_motion_rows, uint32_t atrous_steps, vhr_strip_plan *out);
/* the neighbours' row ranges of an n_rows-deep halo; returns their number (0..2) */
int vhr_strip_plan_exchanges(const vhr_strip_plan *plan, uint32_t n_rows, vhr_row_exchange out[2]);

/* (grid_rows, grid_cols) with grid_rows * grid_cols == world that makes a rank compute the fewest pixels, (W / cols + 2E) x (H / rows + 2E) */
int vhr_tile_grid_choose(uint32_t width, uint32_t height, uint32_t world, uint32_t overlap, uint32_t *grid_rows, uint32_t *grid_cols);
/* grid_rows == 0 or grid_cols == 0: vhr_tile_grid_choose picks the grid.  VHR_ERROR_OUT_OF_SLOTS when a tile is thinner than its halo. */
int vhr_tile_plan_make(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t grid_rows, uint32_t grid_cols, uint32_t max_motion_rows,
                       uint32_t max_motion_cols, uint32_t atrous_steps, vhr_tile_plan *out);
/* The same grid cut at EQUAL COST instead of equal pixels (round 6).  cost[cy * cost_cols + cx] = what the (cell x cell)-pixel block at
 * (cx * cell, cy * cell) costs to trace -- e.g. the any-hit queue kernel's wave lifetimes of a whole-image frame (vhr_debug_wave_lifetimes: one
 * word per 8 x 8 tile, cell = 8); cost_cols >= ceil(width / cell), cost_rows >= ceil(height / cell).  Column cuts split the column sums, row cuts the
 * cost inside each column of tiles (a column's row cuts are its own: tiles of neighbouring columns meet at different heights, which the exchanges -- rectangle
 * intersections with every peer -- do not mind): cut j is the first cell boundary at which the running sum reaches j / n of the total, moved as far as the
 * halos require (no tile thinner than its halo).  cost == NULL or an all-zero map: equal pixels.  Placement
 * only: the images are those of any other plan, bit for bit.  Every rank must be given the same map. */
int vhr_tile_plan_make_weighted(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t grid_rows, uint32_t grid_cols, uint32_t max_motion_rows,
                                uint32_t max_motion_cols, uint32_t atrous_steps, const uint32_t *cost, uint32_t cost_cols, uint32_t cost_rows, uint32_t cell,
                                vhr_tile_plan *out);
/* the rectangles a margin of (halo_rows, halo_cols) pixels takes from / gives to each peer (up to 8); returns their number or < 0 */
int vhr_tile_plan_exchanges(const vhr_tile_plan *plan, uint32_t halo_rows, uint32_t halo_cols, vhr_rect_exchange *out, uint32_t capacity);
/* A re-plan between two frames (the grid cut again, e.g. vhr_tile_plan_make_weighted on a cost map scaled by the ranks' measured frame times): what carries the
 * path's cross-frame state -- the storage images SVGFPushConstants names shadow_and_ao_history, shadow_and_ao_moments_history and
 * prev_frame_normals_and_object_ids (hybrid_render_path.cpp:247-262) -- to the new rectangles.  recv = the pixels of this rank's NEW rectangle grown by the new
 * halo that `peer` OWNED under the old plan (its values are exact there), send = the mirror image; what the rank owned itself stays where it is.  Both plans of
 * the same rank, world and image; up to world - 1 entries; returns their number or < 0.  After the transfers: vhr_set_tile with the new rectangle. */
int vhr_tile_plan_replan(const vhr_tile_plan *old_plan, const vhr_tile_plan *new_plan, vhr_rect_exchange *out, uint32_t capacity);

/* The RCCL library to load, instead of the process's own / the ROCm installation's: before the first other vhr_comm_* call of the process
 * (VHR_ERROR_GRAPH afterwards: the entry points are bound once); NULL or "" = the default resolution.  A path that does not load fails the next call. */
int vhr_comm_use_library(const char *path);
/* The file the RCCL entry points were resolved from (resolves them if nobody has yet), or the reason they could not be. */
const char *vhr_comm_library(void);
int vhr_comm_get_unique_id(uint8_t out[VHR_COMM_UNIQUE_ID_BYTES]);      /* ncclGetUniqueId on one rank; the caller hands it to the others */
/* ncclCommInitRank + vhr_set_strip(plan) / vhr_set_tile(plan): collective over the `world` processes of the plan.  The plan must be one
 * the planner returns (it is recomputed and compared: two ranks that disagree on a rectangle would hang RCCL).  vhr_comm_destroy gives the
 * context the whole image back. */
int vhr_comm_create(vhr_context *ctx, const vhr_strip_plan *plan, const uint8_t unique_id[VHR_COMM_UNIQUE_ID_BYTES], vhr_comm **out);
int vhr_comm_create_tiled(vhr_context *ctx, const vhr_tile_plan *plan, const uint8_t unique_id[VHR_COMM_UNIQUE_ID_BYTES], vhr_comm **out);
void vhr_comm_destroy(vhr_comm *comm);
const char *vhr_comm_last_error(const vhr_comm *comm);
/* Raytrace Pass epilogue, only with "trace_overlap" off: the overlap rows of the raw shadow / AO image from the neighbours, in the
 * context's stream order (exchange #1). */
int vhr_comm_exchange_raytraced(vhr_comm *comm, const char *raytraced_image);
/* SVGF pass epilogue: exchange #2 (halo rows of the history and of the moments history just written, for the NEXT frame's
 * svgf.comp) and, if denoised_image != NULL, the gather of every rank's owned rows into `gathered_frame` on `root` (a device
 * buffer of the whole image there).  Issued on the communicator's own stream behind the context's work so far -- beside the next
 * frame's ray tracing -- and not waited for. */
int vhr_comm_start_frame_exchanges(vhr_comm *comm, int32_t history_storage_image, int32_t moments_storage_image, const char *denoised_image,
                                   int32_t root, void *gathered_frame);
/* Raytrace Pass epilogue of the next frame: the context's stream waits for them (no host synchronisation). */
int vhr_comm_finish_frame_exchanges(vhr_comm *comm);
/* A re-plan between two frames through the library's own RCCL calls: the communicator and the context take `new_plan` (vhr_tile_plan_make_weighted on the map
 * every rank holds; same rank, world and image as the plan the communicator was created with), and the three storage images that carry the path's state from frame
 * to frame travel to the new rectangles (vhr_tile_plan_replan), one grouped batch in the context's stream order.  Collective: every rank, between the same two
 * frames.  The previous frame's exchanges are finished first; a plan of another rank / world / image is refused before anything is enqueued. */
int vhr_comm_replan(vhr_comm *comm, const vhr_tile_plan *new_plan, int32_t history_storage_image, int32_t moments_storage_image, int32_t prev_normals_storage_image);

/* Statistics of the last vhr_trace_rays: out[0] = unique rays traced, out[1] = rays the reference would
 * issue (4x duplicate shadow ray, raygen.rgen:38-40), out[2] = covered (non-sky) pixels, out[3] = traversal
 * stack overflows (must be 0).  Requires vhr_set_ray_statistics(ctx, 1) (costs one counter flush per pass). */
int vhr_set_ray_statistics(vhr_context *ctx, int32_t enable);
int vhr_get_ray_statistics(vhr_context *ctx, uint64_t out[4]);

/* Options.  One table in the library (csrc/vhr_internal.hpp: VHR_OPTION_TABLE) holds every option's name, default, smallest and largest value;
 * vhr_option_count / vhr_option_info enumerate it, vhr_set_option refuses a value outside the range (VHR_ERROR_INVALID_ARGUMENT) and an
 * unknown name (VHR_ERROR_NOT_FOUND).  Every setting publishes identical images (the literal forms of the a-trous filter and of the mirror
 * ray within the float tolerance of tests/); what changes is which kernel runs or what is launched where.
 *  Forms of a shader -- 0 = the literal one-pixel-per-thread form (the in-tree cross-check of the default), 1 = the default:
 *   "raygen_variant"     raygen.rgen's shadow + AO rays: raygen_kernel / raygen_queue_kernel (a ray queue per 8x8-pixel tile and wave)
 *   "reflection_variant" the mirror ray + reflection_hit.rchit: reflection_kernel / reflection_queue_kernel (closest-hit queue per 16x8 tile,
 *                        shading with the whole wave; one or two bounces)
 *   "raytraced_variant"  the raytraced render path's pass: raytraced_kernel / raytraced_queue_kernel
 *   "atrous_variant"     svgf_atrous_filter.comp: svgf_atrous_kernel (direct cached loads, product-form weights) / svgf_atrous_tile_kernel
 *                        (LDS comb tiles, weights in the exponent)
 *  The queue kernels:
 *   "refill_threshold"   idle lanes per wave that trigger a queue refill (default 16)
 *   "lds_stack_levels"   traversal-stack entries per lane kept in LDS, deeper entries spill to scratch (default 8)
 *   "raygen_early_exit"  n/16: the inner-node loop is left once the walking lanes have dropped to that fraction of those that entered it
 *                        (0 = only when all are done; default 6 -- with "raygen_steal" filling the lanes, leaving a little earlier pays: r4)
 *   "reflection_early_exit" the same fraction for the mirror ray's (closest-hit) walk (default 8)
 *   "raygen_waves_per_block" 1, 2 or 4 tiles (= waves) per workgroup (default 2)
 *   "compact_nodes"      1 (default) = the any-hit queue kernel walks 32-byte nodes: both child boxes as centre and half extent in IEEE halves
 *                        relative to the scene centre, widened until they contain the fp32 boxes in exact arithmetic (vhr_get_bvh_form_checks),
 *                        read straight into v_fma_mix_f32 -- two 16-byte loads per visit instead of three, no unpacking; 0 = the 48-byte
 *                        nodes (fp32 centres, truncated fp32 half extents).  Boxes only cull: bit-identical.  A scene whose extent does not
 *                        fit the half range around its centre has no such nodes and is walked on the 48-byte ones.
 *   "raygen_tile_rows"   rows of the 8-pixel-wide tile a wave owns: 0 (default) = 8, or 6 for a launch whose 8x8 tiles fill less than 70 % of
 *                        the chip's wave slots (a 1080p / 8 screen tile: -8 %); 1..8 = that many (4 and 2 lose on whole frames)
 *   "raygen_cost_order"  1 (default) = the workgroups of a queue kernel's launch start in the order of their lifetimes two launches ago, longest
 *                        first: every wave leaves its lifetime, the launch's first workgroup sorts the previous launch's into 8 classes of cost
 *                        (stable inside a class) before it turns to its own tile, the next launch of the same shape on the same stream reads the
 *                        order.  No kernel, stream or event of its own; launches of >= 2 048 workgroups only.  2 = any launch (tests), 0 = row-major.
 *   "raygen_steal"       n (default 8) = once a wave's ray queue is dry and at least n of its lanes are idle, every idle lane takes the lowest pending
 *                        stack entry of a busy lane and walks that subtree for the same ray: any hit is an OR over the subtrees a ray touches, in
 *                        any order and by any lane, and the pixel keeps one "blocked" bit per ray.  sponza_proc -3 %, bistro_proc -13 % of the
 *                        launch; images identical.  0 = a lane only ever walks the rays it fetched.
 *  The a-trous kernel:
 *   "atrous_small_tiles" -1 (default) = 4-row instead of 8-row tiles when the launch has < 32 8-row tiles per CU (a 1080p frame and every screen
 *                        tile use 4-row tiles, a 4K frame 8-row ones), 0 = never, 1 = always
 *  The frame's schedule:
 *   "reflection_async"   1 (default) = the mirror ray's launch (raygen.rgen:59-65; its image is not denoised, nothing of the SVGF pass reads it) runs
 *                        on a stream of the library's own behind the shadow / AO launch, beside the SVGF pass; the context's stream waits for it
 *                        before the frame's next external pass, at the end of vhr_graph_execute and wherever the library waits or hands an
 *                        image out.  Frame with the mirror ray -2 % (sponza_proc) / -6 % (bistro_proc); "Raytrace Pass" then times the
 *                        shadow / AO launch alone and the SVGF pass's time includes what it shares with the mirror ray.  0 = in the pass, in order;
 *                        2 = as 1, and pass epilogues (vhr_graph_set_pass_epilogue) do not wait for it: for owners whose hooks touch neither the
 *                        Reflections image nor the G-buffer (the multi-GPU harness: its hooks exchange visibility and SVGF history)
 *   "fuse_blits"         1 (default) = a compute pass records its dispatches and blits and issues them when its callback returns; a same-extent
 *                        blit whose source is the output (or the normals input) of a recorded a-trous dispatch becomes a second store of that
 *                        launch instead of a copy kernel (all three blits of hybrid_render_path.cpp:310-325); 0 = every blit is a copy
 *   "svgf_elide_unread"  1 = an a-trous dispatch whose output image nothing later in its pass reads or publishes is not launched -- the
 *                        reference's fifth iteration (hybrid_render_path.cpp:299-328 publishes the fourth).  Everything the pass
 *                        publishes is bit-identical; the skipped dispatch's storage image keeps older contents: opt-in (default 0)
 *   "svgf_async_unread"  1 (default) = that dispatch is issued last, on the side stream (see "Threading and streams" at the top); only with one
 *                        frame in flight, for dispatches of >= 900 000 pixels, and when the pass itself copies what the dispatch reads of the
 *                        G-buffer normals (hybrid_render_path.cpp:319); 2 = whatever the size; 0 = every dispatch in recorded order
 *   "fuse_temporal"      1 = svgf.comp runs in the ray-tracing kernel's tile epilogues when the SVGF pass's first command is that dispatch on the
 *                        launch's own images (whole-image work, one frame in flight, no mirror ray, no epilogue hooked to the ray-tracing pass);
 *                        the pass times shift by that dispatch: default 0
 *   "frames_in_flight"   1 (default), 2 or 3; read by vhr_graph_build.  n > 1: frame f uses resource index f mod n (vulkan_common.h:9,
 *                        renderer.cpp:103-146); the passes up to and including the last ray-tracing pass are issued on a second stream beside the
 *                        previous frame's remaining passes, every graph-owned transient image exists once per index, an external binding belongs
 *                        to the index being executed, the two streams are ordered by events derived from the pass declarations
 *  Screen tiles / row strips (one process per GPU; vhr_set_tile, vhr_set_strip):
 *   "trace_overlap"      1 = the shadow / AO rays of the overlap margin are traced by this context as well, so the raw visibility needs no
 *                        neighbour exchange before svgf.comp (default 0: owned pixels only)
 *   "strip_shrink_overlap" 1 = an a-trous launch with step s computes the owned rectangle extended by overlap - (4s - 2) instead of the full
 *                        overlap -- all that can be valid, and all that is needed, after the reference's doubling schedule 1, 2, 4, ... s
 *                        (hybrid_render_path.cpp:299-319); only for callers that run that schedule (default 0)
 *  Instrumentation:
 *   "pass_timestamps"    1 (default) = begin / end stamps per ray-tracing / compute pass, written by the kernels themselves (see "Threading and
 *                        streams"); 2 = + the one-thread stamp kernel in front of external passes and at the end of vhr_graph_execute; 3 = HIP
 *                        event pairs on the dispatch packets (16 us per frame; the only form with "frames_in_flight" > 1); 0 = off
 *   "kernel_timing_stride" n >= 1: with vhr_set_kernel_timing on, only every n-th launch of a kind carries its event pair (a timed dispatch
 *                        costs ~6 us that the next kernel waits for)
 *  Not in the table (they configure the next vhr_update_geometry): "bvh_leaf_triangles" 1..4 (default 2); "bvh_builder" 1 = binned SAH on the
 *  device (default), where the reference builds its BLAS / TLAS (resource_manager.cpp:650,692,792), 0 = the same algorithm on the host
 *  (csrc/bvh_build.cpp: 8 / 28x slower to build on the two test scenes, the same tree up to the order of leaves in memory; images
 *  bit-identical; also what a host-only context and a device build deeper than the walkers' stacks fall back to -- "bvh_device_max_depth"
 *  1..40 (default 40 = those stacks) lowers the depth at which the device builder gives up, for tests of that hand-over); "bvh_host_checks" 1 =
 *  a device-built tree is fetched and the host's containment checks repeated on it (default 0: they run on the device; vhr_get_bvh_form_checks
 *  reports either); "bvh_build_threads"
 *  (host builder) 0 = up to 16 host threads (default), 1 = serial -- the tree is the same whatever the count; "bvh_presplit" 0 (default) =
 *  one reference per triangle, n = 1..400: triangles whose box wastes more than 1 % of the scene box's half area enter the build once per
 *  grid cell they pass through, with at most about n % more references than triangles (csrc/presplit.hpp, both builders, the same tree;
 *  images bit-identical, vhr_get_bvh_statistics then counts references; on a scene with such triangles the any-hit launch gains and the
 *  mirror ray's closest-hit launch loses, profiles/r5_sponza_hard.txt; vhr_get_bvh_presplit_level tells what the last build did);
 *  "bvh_frame" 1 (default) = the boxes along the frame (a rotation, found by the builder: csrc/bvh_frame.hpp) that minimises the summed area
 *  of the triangles' boxes, if that beats the world axes by 5 % -- a scene whose dominant orientation is not the world's walks up to twice as
 *  fast for it, a scene along the world axes keeps them and its tree, bit for bit (the search costs such a scene ~1.5 ms of K0) --, 0 = the world
 *  axes whatever the scene; the walkers rotate a ray once for the box tests and intersect triangles in world space as ever: images
 *  bit-identical, both builders, the same tree (vhr_get_bvh_frame tells which frame the current tree uses; "bvh_presplit" is not combined with a rotated frame). */
int vhr_set_option(vhr_context *ctx, const char *key, int32_t value);
int vhr_get_option(vhr_context *ctx, const char *key, int32_t *value);
int32_t vhr_option_count(void);
int vhr_option_info(int32_t index, const char **name, int32_t *default_value, int32_t *min_value, int32_t *max_value);

/* Per-kernel timing with HIP event pairs attached to every launch of a kernel kind on the context stream (the events ride on
 * the dispatch packet -- hipExtLaunchKernelGGL's start / stop events, the dispatch's own begin / end timestamps -- instead of
 * hipEventRecord, whose barrier packet costs ~4 us of stream time per record)
 * kind: 0 = raygen (K1: shadow + AO rays; with raygen_variant 0 also the mirror ray), 1 = svgf.comp (K3),
 * 2 = svgf_atrous_filter.comp (K4) on the context's stream, 3 = blits (K5), 4 = the mirror-ray kernel (K1's reflection ray + K2),
 * 5 / 6 / 7 = ssao.comp / ssao_blur.comp / ssr.comp, 8 = K4 dispatches issued on the side stream ("svgf_async_unread").
 * kind_mask has bit (1 << kind) set for every kind to time (0 = off).  vhr_get_kernel_time synchronises, folds
 * the recorded pairs into (total milliseconds, launch count) and optionally resets the totals. */
int vhr_set_kernel_timing(vhr_context *ctx, int32_t kind_mask);
int vhr_get_kernel_time(vhr_context *ctx, int32_t kind, double *total_ms, uint64_t *launches, int32_t reset);

/* Traversal work of the last vhr_trace_rays (work-queue raygen, statistics enabled): out[0] = inner-node
 * visits summed over lanes, out[1] = leaf visits, out[2] = ray/triangle tests, out[3] = trips of the two inner
 * loops (node steps + triangle tests) counted once per wave, i.e. by the slowest lane of each round.
 * Active-lane utilisation of the traversal loops = (out[0] + out[2]) / (64 * out[3]). */
/* A hash of the sources this library was built from (16 hex digits).  profiles/pmc_*.json carry the fingerprint of the library their
 * counters were collected on; bench.py quotes them only when it equals the loaded library's. */
const char *vhr_source_fingerprint(void);
/* Diagnostics: the lifetimes (shader clock ticks) of the last ray-tracing launch's waves, as left for "raygen_cost_order"
 * (index = tile pair * waves per workgroup + wave); *count = entries written. */
int vhr_debug_wave_lifetimes(vhr_context *ctx, uint32_t *out, uint32_t capacity, uint32_t *count);
/* Decision (vi) -- the ray / triangle test every walker of the library makes (fp32 Moeller-Trumbore in a fixed order; a candidate whose solution contradicts
 * itself decided again in binary64; the reference leaves this to the driver's traceRayEXT, raygen.rgen:39,51,64) -- on explicit pairs, for parity tests:
 * pairs = count x 17 floats (o[3], d[3], v0[3], e1[3], e2[3], tmin, tmax), HOST memory; hit[count] = 0 / 1, tuv[count x 3] = (t, u, v) of a hit, else 0. */
int vhr_debug_ray_triangle(vhr_context *ctx, const float *pairs, uint32_t count, uint32_t *hit, float *tuv);
/* What the last frame's rays cost, where: those lifetimes -- the any-hit launch's and the mirror ray's -- summed into a map of 8 x 8-pixel cells,
 * out[cy * cols + cx], cols >= ceil(width / 8), rows >= ceil(height / 8).  The cost map vhr_tile_plan_make_weighted cuts a grid by (cell = 8).
 * VHR_ERROR_NOT_FOUND when no queue kernel has left lifetimes ("raygen_cost_order" 0, or 1 on launches below 2 048 workgroups: set it to 2). */
int vhr_get_tile_cost_map(vhr_context *ctx, uint32_t *out, uint32_t cols, uint32_t rows);

int vhr_get_traversal_statistics(vhr_context *ctx, uint64_t out[4]);
/* The same for the mirror-ray launch of the last vhr_trace_rays (raygen.rgen:59-65 + reflection_hit.rchit; the queue kernel, statistics
 * enabled): out[0] = rays walked (first + second bounce), out[1] = second-bounce rays among them, out[2] = inner-node visits summed over
 * lanes, out[3] = leaf visits, out[4] = ray/triangle tests, out[5] = trips of the two inner loops counted once per wave (lane
 * utilisation = (out[2] + out[4]) / (64 * out[5])), out[6] = queue refills, out[7] = waves, out[8] = s_memtime ticks summed over the
 * waves' lifetimes, out[9] = of those, inside the walks (the rest is ray set-up and shading). */
int vhr_get_reflection_statistics(vhr_context *ctx, uint64_t out[10]);
/* Decision (vi)'s binary64 half in the work-queue kernels of the last vhr_trace_rays (statistics enabled).  A candidate whose fp32 solution contradicts
 * itself is decided again in binary64 -- in the per-pixel kernels in place, in the queue kernels outside their loops: the pixel is marked and computed
 * again by the per-pixel code when its tile is done.  out[0] = such pixels of the shadow / AO launch, out[1] = of the mirror ray's launch, out[2..3] = 0. */
int vhr_get_binary64_statistics(vhr_context *ctx, uint64_t out[4]);

/* Where the waves of the last work-queue raygen launch spent their time (statistics enabled; s_memtime ticks summed
 * over waves): out[0] = whole kernel, out[1] = per-tile pixel setup, out[2] = queue refills (ray generation),
 * out[3] = inner-node loop, out[4] = leaf (triangle) stage, out[5] = refills, out[6] = waves, out[7] = 0. */
int vhr_get_traversal_cycles(vhr_context *ctx, uint64_t out[8]);

/* The drain of the tiles' queues in the last shadow / AO launch (statistics enabled): out[0] = entries of the tiles' tree cuts summed over
 * the launch's waves (vhr_get_traversal_cycles out[6]), out[1..3] = wave-level trips made after the tile's queue had run dry with at most
 * 4 / 8 / 16 of the wave's rays still in flight. */
int vhr_get_drain_statistics(vhr_context *ctx, uint64_t out[4]);

/* Profiling aid: streams a storage image once with 4, 8 or 16 bytes per lane (a read of exactly width * height *
 * bytes-per-pixel bytes), used to calibrate rocprofv3's FETCH_SIZE for the SVGF kernels' access widths. */
int vhr_calibration_stream_read(vhr_context *ctx, int32_t storage_image, uint32_t bytes_per_lane);

/* K0 cost of the last vhr_update_geometry (the reference builds its BLAS / TLAS on the device, resource_manager.cpp:650,692,792; so does
 * "bvh_builder" 1, the default): out[0] = the build, out[1] = the upload of the scene arrays (+ the tree, where the host built it), in
 * milliseconds of host time.  The containment checks of the node forms run where the tree is and are not part of either. */
int vhr_get_build_times(vhr_context *ctx, double out[2]);
/* Which builder made the current tree: 0 = the host's, 1 = the device's ("bvh_builder" 1, the default; it falls back to the host builder
 * for a scene of a single leaf and for a tree deeper than the walkers' stacks) */
int vhr_get_bvh_builder(vhr_context *ctx, int32_t *used);
/* "bvh_presplit": the grid level (cell edge = longest scene edge / 2^level) the current tree's references were split on, -1 = every
 * triangle is one reference (the option is off, no triangle qualified, or the budget allowed no level) */
int vhr_get_bvh_presplit_level(vhr_context *ctx, int32_t *level);
/* "bvh_frame": the frame the current tree's boxes are in, row-major, row i = axis i in world coordinates (the identity: the world axes) */
int vhr_get_bvh_frame(vhr_context *ctx, float out[9]);

/* BVH facts for reporting: out[0] = node count, out[1] = triangle count, out[2] = max depth,
 * out[3] = node bytes, out[4] = triangle bytes */
int vhr_get_bvh_statistics(vhr_context *ctx, uint64_t out[5]);
/* Self-check of the last build (exact arithmetic, on the host): out[0] = child boxes checked, out[1] = centre / half-extent boxes that
 * do not contain their (lo, hi) box, out[2] = 48-byte-node boxes that do not contain the centre / half-extent box (or whose links differ),
 * out[3] = half-precision ("compact_nodes") boxes that do not contain theirs.  All three must be 0: the walkers' bit-identity with the
 * oracle rests on box tests that only cull. */
int vhr_get_bvh_form_checks(vhr_context *ctx, uint64_t out[4]);
/* A 64-bit hash of the last build's nodes and leaf triangles in their final order: two builds of the same input must agree whatever
 * "bvh_build_threads" was. */
int vhr_get_bvh_fingerprint(vhr_context *ctx, uint64_t *out);
/* A 64-bit hash of the TREE rather than of its arrays: per inner node the bits of its two child boxes and its children's hashes, per leaf
 * the flat ids of its triangles in ascending order -- independent of the numbering of the nodes, of the order of the leaves in memory and
 * of the order of the triangles inside a leaf.  The host's and the device's build of a scene agree on it ("bvh_builder" 0 / 1: the same
 * algorithm, the same tree; the reference's BLAS / TLAS are the driver's, resource_manager.cpp:593-801). */
int vhr_get_bvh_tree_fingerprint(vhr_context *ctx, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* VHR_AMD_H */
