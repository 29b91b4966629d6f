// Host side of the hot path: which passes the hybrid path registers, what they read and
// write, which persistent SVGF images it owns, and the per-frame SVGF dispatch schedule.
// Reference: src/render_paths/hybrid_render_path.cpp -- RT pass :101-136, SSAO / SSR passes :138-243, SVGF images :245-262, SVGF pass and
// schedule :264-331, composition inputs :333-351, DeregisterPath :383-392.
//
// Written against the vhr:: facade only (no access to library internals), i.e. exactly what a maintainer of
// the reference would compile after swapping the Vulkan render graph for this one.
#include "render_paths.hpp"

#include <cstring>
#include <string>
#include <utility>

namespace vhr {

namespace {
constexpr const char *kNormals = "World Space Normals and Object IDs";
constexpr const char *kMotion = "Motion Vectors and Metallic Roughness";
constexpr const char *kDepth = "Depth";
constexpr const char *kAlbedo = "Albedo";
constexpr const char *kRaytraced = "Raytraced Shadows and Ambient Occlusion";
constexpr const char *kReflections = "Raytraced Reflections";
constexpr const char *kDenoised = "Denoised Raytraced Shadows and Ambient Occlusion";
constexpr const char *kSsaoRaw = "Screen Space Ambient Occlusion Raw";
constexpr const char *kSsao = "Screen Space Ambient Occlusion";
constexpr const char *kSsr = "Screen Space Reflections";
constexpr const char *kSsaoShader = "hybrid_render_path/ssao.comp";
constexpr const char *kSsaoBlurShader = "hybrid_render_path/ssao_blur.comp";
constexpr const char *kSsrShader = "hybrid_render_path/ssr.comp";
constexpr const char *kSvgfShader = "hybrid_render_path/svgf.comp";
constexpr const char *kAtrousShader = "hybrid_render_path/svgf_atrous_filter.comp";

inline uint32_t groups_of_8(uint32_t n) { return n / 8 + (n % 8 != 0); }
}  // namespace

void HybridRenderPath::RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) {
    const uint32_t display_w = context.swapchain.extent.width, display_h = context.swapchain.extent.height;
    const bool any_raytraced = shadow_mode == SHADOW_MODE_RAYTRACED || ambient_occlusion_mode == AMBIENT_OCCLUSION_MODE_RAYTRACED ||
                               reflection_mode == REFLECTION_MODE_RAYTRACED;

    // G-buffer stage (hybrid_render_path.cpp:13-56): untouched raster pass, declared so the graph knows who
    // produces the hot path's inputs and with which formats / clears.
    render_graph.AddGraphicsPass("G-Buffer Pass", {},
                                 { VkUtils::CreateTransientAttachmentImage(kAlbedo, VHR_FORMAT_B8G8R8A8_UNORM, 0, VkUtils::ClearColor(0, 0, 0, 0)),
                                   VkUtils::CreateTransientAttachmentImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 1, VkUtils::ClearColor(0, 0, 0, 0)),
                                   VkUtils::CreateTransientAttachmentImage(kMotion, VHR_FORMAT_R16G16B16A16_SFLOAT, 2, VkUtils::ClearColor(0, 0, -1, -1)),
                                   VkUtils::CreateTransientAttachmentImage(kDepth, VHR_FORMAT_D32_SFLOAT, 3, VkUtils::ClearDepth(0)) },
                                 gbuffer_pass);

    if (shadow_mode == SHADOW_MODE_RASTERIZED) {
        // :58-100 -- a rasterised shadow map suppresses the ray-tracing pass even when AO / reflections are
        // ray traced (the reference's if / else-if); the shadow-map pass itself is raster work and stays outside.
        render_graph.AddGraphicsPass("Shadow Map Pass", {},
                                     { VkUtils::CreateTransientAttachmentImage("Shadow Map", 4096, 4096, VHR_FORMAT_D32_SFLOAT, 0, VkUtils::ClearDepth(0)) },
                                     nullptr);
    } else if (any_raytraced) {
        RaytracingPipelineDescription pipeline;                               // :112-124
        pipeline.name = "Raytrace Pipeline";
        pipeline.raygen_shader = "hybrid_render_path/raygen.rgen";
        pipeline.miss_shaders = { "hybrid_render_path/miss.rmiss", "hybrid_render_path/reflection_miss.rmiss" };
        pipeline.hit_shaders = { HitShader{ "hybrid_render_path/reflection_hit.rchit", nullptr } };
        render_graph.AddRaytracingPass(
            "Raytrace Pass",
            { VkUtils::CreateTransientSampledImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 0),
              VkUtils::CreateTransientSampledImage(kDepth, VHR_FORMAT_D32_SFLOAT, 1) },
            { VkUtils::CreateTransientStorageImage(kRaytraced, VHR_FORMAT_R16G16_SFLOAT, 2),
              VkUtils::CreateTransientStorageImage(kReflections, VHR_FORMAT_R16G16B16A16_SFLOAT, 3) },
            pipeline,
            [display_w, display_h](ExecuteRaytracingCallback execute_pipeline) {
                execute_pipeline("Raytrace Pipeline", [=](RaytracingExecutionContext &execution_context) {
                    execution_context.TraceRays(display_w, display_h);        // :128-131
                });
            });
    }

    if (ambient_occlusion_mode == AMBIENT_OCCLUSION_MODE_SSAO) {              // :138-200
        ssao_push_constants = vhr_ssao_push_constants{ 0.75f };
        ComputePipelineDescription ssao;                                      // no push constant description (:151-157) ...
        ssao.kernels = { ComputeKernel{ kSsaoShader } };
        render_graph.AddComputePass(
            "SSAO Pass",
            { VkUtils::CreateTransientSampledImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 0),
              VkUtils::CreateTransientSampledImage(kDepth, VHR_FORMAT_D32_SFLOAT, 1) },
            { VkUtils::CreateTransientStorageImage(kSsaoRaw, VHR_FORMAT_R16G16B16A16_SFLOAT, 2) },
            ssao,
            [](ComputeExecutionContext &execution_context) {
                const uvec2 display_size = execution_context.GetDisplaySize();
                execution_context.Dispatch(kSsaoShader, groups_of_8(display_size.x), groups_of_8(display_size.y), 1);   // ... and none pushed (:161-166)
            });
        ComputePipelineDescription blur;                                      // :177-188: the blur gets the SSAO constants it does not read
        blur.kernels = { ComputeKernel{ kSsaoBlurShader } };
        blur.push_constant_description.size = sizeof(vhr_ssao_push_constants);
        render_graph.AddComputePass(
            "SSAO Blur Pass",
            { VkUtils::CreateTransientStorageImage(kSsaoRaw, VHR_FORMAT_R16G16B16A16_SFLOAT, 0) },
            { VkUtils::CreateTransientStorageImage(kSsao, VHR_FORMAT_R16G16B16A16_SFLOAT, 1) },
            blur,
            [this](ComputeExecutionContext &execution_context) {
                const uvec2 display_size = execution_context.GetDisplaySize();
                execution_context.Dispatch(kSsaoBlurShader, groups_of_8(display_size.x), groups_of_8(display_size.y), 1, ssao_push_constants);
            });
    }

    if (reflection_mode == REFLECTION_MODE_SSR) {                             // :202-243
        ssr_push_constants = vhr_ssr_push_constants{ 25.0f, 0.1f, 0.5f, 10 };
        ComputePipelineDescription ssr;
        ssr.kernels = { ComputeKernel{ kSsrShader } };
        ssr.push_constant_description.size = sizeof(vhr_ssr_push_constants);
        render_graph.AddComputePass(
            "SSR Pass",
            { VkUtils::CreateTransientSampledImage(kAlbedo, VHR_FORMAT_B8G8R8A8_UNORM, 0),
              VkUtils::CreateTransientSampledImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 1),
              VkUtils::CreateTransientSampledImage(kMotion, VHR_FORMAT_R16G16B16A16_SFLOAT, 2),
              VkUtils::CreateTransientSampledImage(kDepth, VHR_FORMAT_D32_SFLOAT, 3) },
            { VkUtils::CreateTransientStorageImage(kSsr, VHR_FORMAT_R16G16B16A16_SFLOAT, 4) },
            ssr,
            [this](ComputeExecutionContext &execution_context) {
                const uvec2 display_size = execution_context.GetDisplaySize();
                execution_context.Dispatch(kSsrShader, groups_of_8(display_size.x), groups_of_8(display_size.y), 1, ssr_push_constants);
            });
    }

    if (denoise_shadow_and_ao && any_raytraced) {
        // five persistent images, :247-262 (the moments history really is allocated R16G16)
        svgf_push_constants.integrated_shadow_and_ao[0] = int32_t(resource_manager.UploadNewStorageImage(display_w, display_h, VHR_FORMAT_R16G16B16A16_SFLOAT));
        svgf_push_constants.integrated_shadow_and_ao[1] = int32_t(resource_manager.UploadNewStorageImage(display_w, display_h, VHR_FORMAT_R16G16B16A16_SFLOAT));
        svgf_push_constants.prev_frame_normals_and_object_ids = int32_t(resource_manager.UploadNewStorageImage(display_w, display_h, VHR_FORMAT_R16G16B16A16_SFLOAT));
        svgf_push_constants.shadow_and_ao_history = int32_t(resource_manager.UploadNewStorageImage(display_w, display_h, VHR_FORMAT_R16G16B16A16_SFLOAT));
        svgf_push_constants.shadow_and_ao_moments_history = int32_t(resource_manager.UploadNewStorageImage(display_w, display_h, VHR_FORMAT_R16G16_SFLOAT));
        svgf_push_constants.atrous_step = 1;
        svgf_textures_created = true;

        ComputePipelineDescription pipeline;                                  // :275-286
        pipeline.kernels = { ComputeKernel{ kSvgfShader }, ComputeKernel{ kAtrousShader } };
        pipeline.push_constant_description.size = sizeof(SVGFPushConstants);
        render_graph.AddComputePass(
            "SVGF Denoise Pass",
            { VkUtils::CreateTransientStorageImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 0),
              VkUtils::CreateTransientStorageImage(kMotion, VHR_FORMAT_R16G16B16A16_SFLOAT, 1),
              VkUtils::CreateTransientSampledImage(kDepth, VHR_FORMAT_D32_SFLOAT, 2),
              VkUtils::CreateTransientStorageImage(kRaytraced, VHR_FORMAT_R16G16_SFLOAT, 3) },
            { VkUtils::CreateTransientStorageImage(kDenoised, VHR_FORMAT_R16G16B16A16_SFLOAT, 4) },
            pipeline,
            [this](ComputeExecutionContext &execution_context) {
                // Per-frame schedule, :288-329.  With (x, y) = the ping-pong pair at frame start:
                //   svgf.comp writes x; iteration i filters x -> y with step 2^i, then the pair swaps;
                //   after iteration 0 its output becomes the temporal history; after the loop the current
                //   normals become next frame's "previous" normals and image y -- which after an odd number
                //   of swaps is the output of the SECOND-TO-LAST iteration -- is what gets published as the
                //   denoised result (the reference's off-by-one: with 5 steps the step-16 pass is dead work).
                SVGFPushConstants &pc = svgf_push_constants;
                const uvec2 display_size = execution_context.GetDisplaySize();
                const uint32_t gx = groups_of_8(display_size.x), gy = groups_of_8(display_size.y);
                execution_context.Dispatch(kSvgfShader, gx, gy, 1, pc);
                for (int i = 0; i < atrous_steps; ++i) {
                    pc.atrous_step = 1 << i;
                    execution_context.Dispatch(kAtrousShader, gx, gy, 1, pc);
                    if (i == 0) execution_context.BlitImageStorageToStorage(pc.integrated_shadow_and_ao[1], pc.shadow_and_ao_history);
                    std::swap(pc.integrated_shadow_and_ao[0], pc.integrated_shadow_and_ao[1]);
                }
                execution_context.BlitImageTransientToStorage(kNormals, pc.prev_frame_normals_and_object_ids);
                execution_context.BlitImageStorageToTransient(pc.integrated_shadow_and_ao[1], kDenoised);
                std::swap(pc.integrated_shadow_and_ao[0], pc.integrated_shadow_and_ao[1]);   // ready for the next frame
            });
    }

    // Composition stage (:333-379): untouched raster pass; its dependency list is what pulls the passes above
    // into the execution order (FindExecutionOrder walks back from RENDER_OUTPUT).
    render_graph.AddGraphicsPass(
        "Composition Pass",
        { VkUtils::CreateTransientSampledImage(kAlbedo, VHR_FORMAT_B8G8R8A8_UNORM, 0),
          VkUtils::CreateTransientSampledImage(kNormals, VHR_FORMAT_R16G16B16A16_SFLOAT, 1),
          VkUtils::CreateTransientSampledImage(kMotion, VHR_FORMAT_R16G16B16A16_SFLOAT, 2),
          VkUtils::CreateTransientSampledImage(kDepth, VHR_FORMAT_D32_SFLOAT, 3),
          VkUtils::CreateTransientSampledImage("Shadow Map", 4096, 4096, VHR_FORMAT_D32_SFLOAT, 4),
          VkUtils::CreateTransientSampledImage(kSsao, VHR_FORMAT_R16G16B16A16_SFLOAT, 5),
          VkUtils::CreateTransientSampledImage(kSsr, VHR_FORMAT_R16G16B16A16_SFLOAT, 6),
          denoise_shadow_and_ao ? VkUtils::CreateTransientSampledImage(kDenoised, VHR_FORMAT_R16G16B16A16_SFLOAT, 7)
                                : VkUtils::CreateTransientSampledImage(kRaytraced, VHR_FORMAT_R16G16_SFLOAT, 7),
          VkUtils::CreateTransientSampledImage(kReflections, VHR_FORMAT_R16G16B16A16_SFLOAT, 8) },
        { VkUtils::CreateTransientRenderOutput(0) }, composition_pass);
}

void HybridRenderPath::DeregisterPath(DeviceContext &, RenderGraph &, ResourceManager &resource_manager) {   // :383-392
    if (!svgf_textures_created) return;
    resource_manager.DestroyStorageImage(uint32_t(svgf_push_constants.integrated_shadow_and_ao[0]));
    resource_manager.DestroyStorageImage(uint32_t(svgf_push_constants.integrated_shadow_and_ao[1]));
    resource_manager.DestroyStorageImage(uint32_t(svgf_push_constants.prev_frame_normals_and_object_ids));
    resource_manager.DestroyStorageImage(uint32_t(svgf_push_constants.shadow_and_ao_history));
    resource_manager.DestroyStorageImage(uint32_t(svgf_push_constants.shadow_and_ao_moments_history));
    svgf_textures_created = false;
}

}  // namespace vhr

// ---------------------------------------------------------------------------------------------------------
// C entry points (vhr_amd.h, "HybridRenderPath" section) for callers without a C++ toolchain
// ---------------------------------------------------------------------------------------------------------
struct vhr_hybrid_render_path {
    vhr::DeviceContext context;
    vhr::ResourceManager resource_manager;
    vhr::RenderGraph render_graph;
    vhr::HybridRenderPath path;
    vhr_external_pass_callback gbuffer_cb = nullptr, composition_cb = nullptr;
    void *gbuffer_user = nullptr, *composition_user = nullptr;
    std::string error;
    vhr_hybrid_render_path(vhr_context *ctx, uint32_t w, uint32_t h)
        : context(ctx), resource_manager(context), render_graph(context, resource_manager), path(context, render_graph, resource_manager) {
        context.swapchain.extent = { w, h };
    }
};

static void apply_settings(vhr_hybrid_render_path *p, const vhr_hybrid_settings *s) {
    p->path.shadow_mode = s->shadow_mode;
    p->path.ambient_occlusion_mode = s->ambient_occlusion_mode;
    p->path.reflection_mode = s->reflection_mode;
    p->path.denoise_shadow_and_ao = s->denoise_shadow_and_ao != 0;
    p->path.atrous_steps = s->atrous_steps > 0 ? s->atrous_steps : 5;
}

template <typename F>
static int guarded(vhr_hybrid_render_path *p, F &&f) {
    try {
        f();
        return VHR_OK;
    } catch (const std::exception &e) {
        p->error = e.what();
        return VHR_ERROR_GRAPH;
    }
}

extern "C" {

int vhr_hybrid_create(vhr_context *ctx, const vhr_hybrid_settings *settings, vhr_external_pass_callback gbuffer_pass, void *gbuffer_user,
                      vhr_external_pass_callback composition_pass, void *composition_user, vhr_hybrid_render_path **out) {
    if (!ctx || !settings || !out) return VHR_ERROR_INVALID_ARGUMENT;
    uint32_t w = 0, h = 0;
    if (vhr_get_display_size(ctx, &w, &h) < 0) return VHR_ERROR_INVALID_ARGUMENT;
    auto *p = new vhr_hybrid_render_path(ctx, w, h);
    apply_settings(p, settings);
    p->gbuffer_cb = gbuffer_pass; p->gbuffer_user = gbuffer_user;
    p->composition_cb = composition_pass; p->composition_user = composition_user;
    if (gbuffer_pass) p->path.gbuffer_pass = [p](vhr::DeviceContext &c) { p->gbuffer_cb(p->gbuffer_user, c.handle); };
    if (composition_pass) p->path.composition_pass = [p](vhr::DeviceContext &c) { p->composition_cb(p->composition_user, c.handle); };
    *out = p;
    return VHR_OK;
}

void vhr_hybrid_destroy(vhr_hybrid_render_path *p) {
    if (!p) return;
    try {
        p->path.DeregisterPath(p->context, p->render_graph, p->resource_manager);
        p->render_graph.DestroyResources();
    } catch (...) {
    }
    delete p;
}

int vhr_hybrid_build(vhr_hybrid_render_path *p) {
    if (!p) return VHR_ERROR_INVALID_ARGUMENT;
    // (the display extent as the context has it NOW: after vhr_resize this is the second half of the reference's resize route, renderer.cpp:113-118)
    uint32_t w = 0, h = 0;
    if (vhr_get_display_size(p->context.handle, &w, &h) < 0) return VHR_ERROR_INVALID_ARGUMENT;
    p->context.swapchain.extent = { w, h };
    return guarded(p, [&] { p->path.Build(); });
}

int vhr_hybrid_rebuild(vhr_hybrid_render_path *p, const vhr_hybrid_settings *settings) {
    if (!p) return VHR_ERROR_INVALID_ARGUMENT;
    return guarded(p, [&] {
        // the UI applies the new modes, then calls Rebuild() (hybrid_render_path.cpp:394-441)
        p->path.DeregisterPath(p->context, p->render_graph, p->resource_manager);
        if (settings) apply_settings(p, settings);
        p->path.Build();
    });
}

int vhr_hybrid_get_push_constants(vhr_hybrid_render_path *p, vhr_svgf_push_constants *out) {
    if (!p || !out) return VHR_ERROR_INVALID_ARGUMENT;
    *out = p->path.svgf_push_constants;
    return VHR_OK;
}

const char *vhr_hybrid_last_error(vhr_hybrid_render_path *p) { return p ? p->error.c_str() : ""; }

}  // extern "C"

// ---- checkpoint / resume of the SVGF state (vhr_amd.h; through the public C ABI only, like the rest of this file) ----
namespace {
struct SvgfStateHeader {
    char magic[8];                  // "VHRSVGF1"
    uint32_t version, width, height, image_count;
    int32_t format[5];
    uint64_t image_bytes[5];
    uint64_t total_bytes;
    vhr_per_frame_data last_frame;
};
constexpr char kStateMagic[8] = { 'V', 'H', 'R', 'S', 'V', 'G', 'F', '1' };

// the five images in blob order, as the NEXT frame addresses them (the frame-start order of the ping-pong pair, :328)
int svgf_state_images(vhr_context *ctx, const vhr::HybridRenderPath &path, int32_t ids[5], vhr_image_info info[5], std::string &error) {
    if (!path.svgf_textures_created) { error = "SVGF state: the path has no SVGF images (denoise off, or not built)"; return VHR_ERROR_GRAPH; }
    const vhr::SVGFPushConstants &pc = path.svgf_push_constants;
    ids[0] = pc.integrated_shadow_and_ao[0]; ids[1] = pc.integrated_shadow_and_ao[1]; ids[2] = pc.prev_frame_normals_and_object_ids;
    ids[3] = pc.shadow_and_ao_history; ids[4] = pc.shadow_and_ao_moments_history;
    for (int i = 0; i < 5; ++i) {
        const int rc = vhr_get_storage_image(ctx, ids[i], &info[i]);
        if (rc < 0) { error = std::string("SVGF state: ") + vhr_last_error(ctx); return rc; }
    }
    return VHR_OK;
}

int svgf_state_size(vhr_context *ctx, const vhr::HybridRenderPath &path, uint64_t *bytes, std::string &error) {
    int32_t ids[5]; vhr_image_info info[5];
    const int rc = svgf_state_images(ctx, path, ids, info, error);
    if (rc < 0) return rc;
    uint64_t total = sizeof(SvgfStateHeader);
    for (int i = 0; i < 5; ++i) total += uint64_t(info[i].width) * info[i].height * info[i].bytes_per_pixel;
    *bytes = total;
    return VHR_OK;
}

int svgf_state_save(vhr_context *ctx, const vhr::HybridRenderPath &path, void *blob, uint64_t bytes, std::string &error) {
    int32_t ids[5]; vhr_image_info info[5];
    int rc = svgf_state_images(ctx, path, ids, info, error);
    if (rc < 0) return rc;
    SvgfStateHeader h{};
    std::memcpy(h.magic, kStateMagic, 8);
    h.version = 1; h.width = info[0].width; h.height = info[0].height; h.image_count = 5;
    h.total_bytes = sizeof h;
    for (int i = 0; i < 5; ++i) {
        h.format[i] = info[i].format;
        h.image_bytes[i] = uint64_t(info[i].width) * info[i].height * info[i].bytes_per_pixel;
        h.total_bytes += h.image_bytes[i];
    }
    if (bytes != h.total_bytes) { error = "SVGF state: the blob must hold exactly vhr_hybrid_state_size bytes"; return VHR_ERROR_INVALID_ARGUMENT; }
    rc = vhr_get_last_per_frame_ubo(ctx, &h.last_frame);
    if (rc < 0) return rc;
    char *out = static_cast<char *>(blob);
    std::memcpy(out, &h, sizeof h);
    out += sizeof h;
    for (int i = 0; i < 5; ++i) {          // (a download waits for every stream of the context, the side stream's dead iteration included)
        rc = vhr_download_storage_image(ctx, ids[i], out, h.image_bytes[i]);
        if (rc < 0) { error = std::string("SVGF state: ") + vhr_last_error(ctx); return rc; }
        out += h.image_bytes[i];
    }
    return VHR_OK;
}

int svgf_state_load(vhr_context *ctx, const vhr::HybridRenderPath &path, const void *blob, uint64_t bytes, vhr_per_frame_data *last_frame, std::string &error) {
    int32_t ids[5]; vhr_image_info info[5];
    int rc = svgf_state_images(ctx, path, ids, info, error);
    if (rc < 0) return rc;
    SvgfStateHeader h;
    if (bytes < sizeof h) { error = "SVGF state: blob shorter than its header"; return VHR_ERROR_INVALID_ARGUMENT; }
    std::memcpy(&h, blob, sizeof h);
    if (std::memcmp(h.magic, kStateMagic, 8) != 0 || h.version != 1 || h.image_count != 5) { error = "SVGF state: not a version-1 state blob"; return VHR_ERROR_INVALID_ARGUMENT; }
    if (h.total_bytes != bytes) { error = "SVGF state: byte count differs from the header's"; return VHR_ERROR_INVALID_ARGUMENT; }
    uint64_t total = sizeof h;
    for (int i = 0; i < 5; ++i) {
        if (h.width != info[i].width || h.height != info[i].height || h.format[i] != info[i].format ||
            h.image_bytes[i] != uint64_t(info[i].width) * info[i].height * info[i].bytes_per_pixel) {
            error = "SVGF state: the blob was saved from a path of another extent or image format";
            return VHR_ERROR_INVALID_ARGUMENT;
        }
        total += h.image_bytes[i];
    }
    if (total != bytes) { error = "SVGF state: image byte counts do not add up to the blob"; return VHR_ERROR_INVALID_ARGUMENT; }
    const char *in = static_cast<const char *>(blob) + sizeof h;
    for (int i = 0; i < 5; ++i) {
        rc = vhr_upload_storage_image(ctx, ids[i], in, h.image_bytes[i]);
        if (rc < 0) { error = std::string("SVGF state: ") + vhr_last_error(ctx); return rc; }
        in += h.image_bytes[i];
    }
    if (last_frame) *last_frame = h.last_frame;
    return VHR_OK;
}
}  // namespace

// the facade's methods (csrc/render_paths.hpp): errors throw like every other call of the facade
std::vector<uint8_t> vhr::HybridRenderPath::SaveState() {
    std::string error;
    uint64_t n = 0;
    if (svgf_state_size(context.handle, *this, &n, error) < 0) throw std::runtime_error(error);
    std::vector<uint8_t> blob(n);
    if (svgf_state_save(context.handle, *this, blob.data(), n, error) < 0) throw std::runtime_error(error);
    return blob;
}
vhr::PerFrameData vhr::HybridRenderPath::LoadState(const std::vector<uint8_t> &blob) {
    std::string error;
    vhr::PerFrameData last{};
    if (svgf_state_load(context.handle, *this, blob.data(), blob.size(), &last, error) < 0) throw std::runtime_error(error);
    return last;
}

extern "C" {
int vhr_hybrid_state_size(vhr_hybrid_render_path *p, uint64_t *bytes) {
    if (!p || !bytes) return VHR_ERROR_INVALID_ARGUMENT;
    return svgf_state_size(p->context.handle, p->path, bytes, p->error);
}
int vhr_hybrid_save_state(vhr_hybrid_render_path *p, void *blob, uint64_t bytes) {
    if (!p || !blob) return VHR_ERROR_INVALID_ARGUMENT;
    return svgf_state_save(p->context.handle, p->path, blob, bytes, p->error);
}
int vhr_hybrid_load_state(vhr_hybrid_render_path *p, const void *blob, uint64_t bytes, vhr_per_frame_data *last_frame) {
    if (!p || !blob) return VHR_ERROR_INVALID_ARGUMENT;
    return svgf_state_load(p->context.handle, p->path, blob, bytes, last_frame, p->error);
}
}  // extern "C"
