// Host side of SURVEY.md section 8 row f4: the passes the raytraced render path registers.
// Reference: src/render_paths/raytraced_render_path.cpp -- "Raytracing Pass" :12-47, "Composition Pass" :49-76,
// DeregisterPath :79 (owns nothing), use_anyhit_shader toggle :81-93.
//
// Written against the vhr:: facade only, like hybrid_render_path.cpp.
#include "render_paths.hpp"

#include <string>
#include <utility>

namespace vhr {

namespace {
constexpr const char *kRaytracedOutput = "RaytracedOutput";
}

void RaytracedRenderPath::RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &) {
    const uint32_t display_w = context.swapchain.extent.width, display_h = context.swapchain.extent.height;

    RaytracingPipelineDescription pipeline;                                        // :17-35
    pipeline.name = "Raytracing Pipeline";
    pipeline.raygen_shader = use_anyhit_shader ? "raytraced_render_path/raygen_test_alpha.rgen" : "raytraced_render_path/raygen.rgen";
    pipeline.miss_shaders = { "raytraced_render_path/miss.rmiss", "raytraced_render_path/shadow_miss.rmiss" };
    pipeline.hit_shaders = { use_anyhit_shader
                                 ? HitShader{ "raytraced_render_path/closesthit_test_alpha.rchit", "raytraced_render_path/shadow_anyhit.rahit" }
                                 : HitShader{ "raytraced_render_path/closesthit.rchit", nullptr } };
    render_graph.AddRaytracingPass(
        "Raytracing Pass", {},
        { VkUtils::CreateTransientStorageImage(kRaytracedOutput, VHR_FORMAT_B8G8R8A8_UNORM, 0) },      // :13-16
        pipeline,
        [display_w, display_h](ExecuteRaytracingCallback execute_pipeline) {                            // :36-46
            execute_pipeline("Raytracing Pipeline", [display_w, display_h](RaytracingExecutionContext &execution_context) {
                execution_context.TraceRays(display_w, display_h);
            });
        });

    // :49-76 -- fullscreen triangle that samples RaytracedOutput into the swapchain image; raster work, declared so the
    // graph has its single RENDER_OUTPUT writer and the execution order of the reference
    render_graph.AddGraphicsPass("Composition Pass",
                                 { VkUtils::CreateTransientSampledImage(kRaytracedOutput, VHR_FORMAT_B8G8R8A8_UNORM, 0) },
                                 { VkUtils::CreateTransientRenderOutput(0) },
                                 composition_pass);
}

void RaytracedRenderPath::DeregisterPath(DeviceContext &, RenderGraph &, ResourceManager &) {}          // :79

}  // namespace vhr

// ---------------------------------------------------------------------------------------------------------
// C entry points (vhr_amd.h, "RaytracedRenderPath" section) for callers without a C++ toolchain
// ---------------------------------------------------------------------------------------------------------
struct vhr_raytraced_render_path {
    vhr::DeviceContext context;
    vhr::ResourceManager resource_manager;
    vhr::RenderGraph render_graph;
    vhr::RaytracedRenderPath path;
    vhr_external_pass_callback composition_cb = nullptr;
    void *composition_user = nullptr;
    std::string error;
    vhr_raytraced_render_path(vhr_context *ctx, uint32_t w, uint32_t h)
        : context(ctx), resource_manager(context), render_graph(context, resource_manager), path(context, render_graph, resource_manager) {
        context.swapchain.extent = { w, h };
    }
};

template <typename F>
static int guarded(vhr_raytraced_render_path *p, F &&f) {
    try {
        f();
        return VHR_OK;
    } catch (const std::exception &e) {
        p->error = e.what();
        return VHR_ERROR_GRAPH;
    }
}

extern "C" {

int vhr_raytraced_create(vhr_context *ctx, int32_t use_anyhit_shader, vhr_external_pass_callback composition_pass, void *composition_user,
                         vhr_raytraced_render_path **out) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    uint32_t w = 0, h = 0;
    if (vhr_get_display_size(ctx, &w, &h) < 0) return VHR_ERROR_INVALID_ARGUMENT;
    auto *p = new vhr_raytraced_render_path(ctx, w, h);
    p->path.use_anyhit_shader = use_anyhit_shader != 0;
    p->composition_cb = composition_pass;
    p->composition_user = composition_user;
    if (composition_pass) p->path.composition_pass = [p](vhr::DeviceContext &c) { p->composition_cb(p->composition_user, c.handle); };
    *out = p;
    return VHR_OK;
}

void vhr_raytraced_destroy(vhr_raytraced_render_path *p) {
    if (!p) return;
    try {
        p->path.DeregisterPath(p->context, p->render_graph, p->resource_manager);
        p->render_graph.DestroyResources();
    } catch (...) {
    }
    delete p;
}

int vhr_raytraced_build(vhr_raytraced_render_path *p) {
    if (!p) return VHR_ERROR_INVALID_ARGUMENT;
    // (the display extent as the context has it NOW: after vhr_resize this is the second half of the reference's resize route, renderer.cpp:113-118)
    uint32_t w = 0, h = 0;
    if (vhr_get_display_size(p->context.handle, &w, &h) < 0) return VHR_ERROR_INVALID_ARGUMENT;
    p->context.swapchain.extent = { w, h };
    return guarded(p, [&] { p->path.Build(); });
}

int vhr_raytraced_rebuild(vhr_raytraced_render_path *p, int32_t use_anyhit_shader) {
    if (!p) return VHR_ERROR_INVALID_ARGUMENT;
    return guarded(p, [&] {
        p->path.use_anyhit_shader = use_anyhit_shader != 0;      // the radio button, then Rebuild() (:90-92)
        p->path.Rebuild();
    });
}

const char *vhr_raytraced_last_error(vhr_raytraced_render_path *p) { return p ? p->error.c_str() : ""; }

}  // extern "C"
