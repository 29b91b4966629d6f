// vhr_render_graph.hpp -- header-only C++ facade over the C ABI (vhr_amd.h) that mirrors the reference's
// render-graph pass API, so a render path written against the reference compiles against this with a
// namespace prefix:
//
//   reference class / function                         (file:line)                         here
//   ------------------------------------------------------------------------------------------------------
//   RenderGraph::Add{Graphics,Raytracing,Compute}Pass  render_graph.h:10-18                vhr::RenderGraph
//   RenderGraph::Build / Execute / Gather...           render_graph.h:20-22                vhr::RenderGraph
//   RaytracingExecutionContext::TraceRays              raytracing_execution_context.h:13   vhr::RaytracingExecutionContext
//   ComputeExecutionContext::{GetDisplaySize,Dispatch, compute_execution_context.h:17-31   vhr::ComputeExecutionContext
//       BlitImage*}
//   ResourceManager::{UpdateGeometry,UploadTexture...,  resource_manager.h:26-35            vhr::ResourceManager
//       UploadNewStorageImage,DestroyStorageImage,UpdatePerFrameUBO}
//   VkUtils::CreateTransient*                          vulkan_utils.h:347-453              vhr::VkUtils
//   TransientResource, *PipelineDescription, callbacks vulkan_common.h:236-341             vhr::*
//   RenderPath::{Build,Rebuild,RegisterPath,...}       render_path.h:8-14                  vhr::RenderPath
//
// Error behaviour: the reference asserts (VK_CHECK, vulkan_common.h:4-7); the facade throws
// std::runtime_error carrying vhr_last_error() so a mis-declared pass fails loudly.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vhr_amd.h"

namespace vhr {

using Vertex = vhr_vertex;
using Material = vhr_material;
using Primitive = vhr_primitive;
using DirectionalLight = vhr_directional_light;
using PerFrameData = vhr_per_frame_data;
using SVGFPushConstants = vhr_svgf_push_constants;
using SamplerInfo = vhr_sampler_info;
using TransientResource = vhr_transient_resource;
using Format = int32_t;   // VkFormat values

struct uvec2 { uint32_t x, y; };

struct HitShader { const char *closest_hit = nullptr; const char *any_hit = nullptr; };
struct RaytracingPipelineDescription {
    const char *name = nullptr;
    const char *raygen_shader = nullptr;
    std::vector<const char *> miss_shaders;
    std::vector<HitShader> hit_shaders;
};
struct ComputeKernel { const char *shader; };
struct PushConstantDescription { uint32_t size = 0; uint32_t shader_stage = 0; };
struct ComputePipelineDescription {
    std::vector<ComputeKernel> kernels;
    PushConstantDescription push_constant_description;
};

inline void check(vhr_context *ctx, int rc, const char *what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + vhr_last_error(ctx));
}

namespace VkUtils {
inline TransientResource MakeImage(const char *name, int32_t type, uint32_t w, uint32_t h, Format format, uint32_t binding) {
    TransientResource r{};
    r.type = VHR_TRANSIENT_RESOURCE_IMAGE;
    r.name = name;
    r.image.type = type;
    r.image.width = w;
    r.image.height = h;
    r.image.format = format;
    r.image.binding = binding;
    return r;
}
struct ClearValue { float v[4]; };
inline ClearValue ClearColor(float r, float g, float b, float a) { return ClearValue{ { r, g, b, a } }; }
inline ClearValue ClearDepth(float d) { return ClearValue{ { d, 0, 0, 0 } }; }
// vulkan_utils.h:347-361
inline TransientResource CreateTransientRenderOutput(uint32_t binding, bool multisampled = false) {
    TransientResource r = MakeImage("RENDER_OUTPUT", VHR_TRANSIENT_ATTACHMENT_IMAGE, 0, 0, VHR_FORMAT_UNDEFINED, binding);
    r.image.multisampled = multisampled;
    return r;
}
// vulkan_utils.h:363-395
inline TransientResource CreateTransientAttachmentImage(const char *name, Format format, uint32_t binding, ClearValue clear, bool multisampled = false) {
    TransientResource r = MakeImage(name, VHR_TRANSIENT_ATTACHMENT_IMAGE, 0, 0, format, binding);
    for (int i = 0; i < 4; ++i) r.image.clear_value[i] = clear.v[i];
    r.image.multisampled = multisampled;
    return r;
}
inline TransientResource CreateTransientAttachmentImage(const char *name, uint32_t w, uint32_t h, Format format, uint32_t binding, ClearValue clear, bool multisampled = false) {
    TransientResource r = MakeImage(name, VHR_TRANSIENT_ATTACHMENT_IMAGE, w, h, format, binding);
    for (int i = 0; i < 4; ++i) r.image.clear_value[i] = clear.v[i];
    r.image.multisampled = multisampled;
    return r;
}
// vulkan_utils.h:397-425
inline TransientResource CreateTransientSampledImage(const char *name, Format format, uint32_t binding) { return MakeImage(name, VHR_TRANSIENT_SAMPLED_IMAGE, 0, 0, format, binding); }
inline TransientResource CreateTransientSampledImage(const char *name, uint32_t w, uint32_t h, Format format, uint32_t binding) { return MakeImage(name, VHR_TRANSIENT_SAMPLED_IMAGE, w, h, format, binding); }
// vulkan_utils.h:427-453
inline TransientResource CreateTransientStorageImage(const char *name, Format format, uint32_t binding) { return MakeImage(name, VHR_TRANSIENT_STORAGE_IMAGE, 0, 0, format, binding); }
inline TransientResource CreateTransientStorageImage(const char *name, uint32_t w, uint32_t h, Format format, uint32_t binding) { return MakeImage(name, VHR_TRANSIENT_STORAGE_IMAGE, w, h, format, binding); }
}  // namespace VkUtils

// "VulkanContext" stand-in: owns the vhr_context and the display extent (context.swapchain.extent)
class DeviceContext {
public:
    struct Extent { uint32_t width, height; };
    struct Swapchain { Extent extent; } swapchain;
    explicit DeviceContext(vhr_context *existing) : handle(existing), owned(false) {}
    DeviceContext(int device, uint32_t width, uint32_t height, void *stream = nullptr) : owned(true) {
        vhr_create_info info{ device, width, height, stream, 0 };
        swapchain.extent = { width, height };
        if (vhr_create(&info, &handle) < 0) throw std::runtime_error(std::string("vhr_create: ") + vhr_last_error(nullptr));
    }
    ~DeviceContext() { if (owned) vhr_destroy(handle); }
    // vulkan_context.cpp:118-120 (renderer.cpp:113-118: `context->Resize(); active_render_path->Build();`): the new display extent; the graph and the
    // storage pool's images are released, geometry / acceleration structure / textures stay.  Follow with RenderPath::Build().
    void Resize(uint32_t width, uint32_t height) {
        if (vhr_resize(handle, width, height) < 0) throw std::runtime_error(std::string("Resize: ") + vhr_last_error(handle));
        swapchain.extent = { width, height };
    }
    DeviceContext(const DeviceContext &) = delete;
    DeviceContext &operator=(const DeviceContext &) = delete;
    vhr_context *handle = nullptr;
private:
    bool owned;
};

class ResourceManager {
public:
    explicit ResourceManager(DeviceContext &context) : context(context) {}
    // resource_manager.h:34 (+ UpdateBLAS / UpdateTLAS)
    void UpdateGeometry(const std::vector<Vertex> &vertices, const std::vector<uint32_t> &indices, const std::vector<Primitive> &primitives) {
        check(context.handle, vhr_update_geometry(context.handle, vertices.data(), uint32_t(vertices.size()), indices.data(), uint32_t(indices.size()),
                                                  primitives.data(), uint32_t(primitives.size())), "UpdateGeometry");
    }
    // resource_manager.h:26
    uint32_t UploadTextureFromData(uint32_t width, uint32_t height, uint8_t *data, Format format = VHR_FORMAT_R8G8B8A8_UNORM, SamplerInfo *sampler_info = nullptr) {
        int32_t r = vhr_upload_texture_from_data(context.handle, width, height, data, format, sampler_info);
        check(context.handle, r, "UploadTextureFromData");
        return uint32_t(r);
    }
    // resource_manager.h:28 -- returns uint32_t(-1) when the pool is exhausted (resource_manager.cpp:876-877)
    uint32_t UploadNewStorageImage(uint32_t width, uint32_t height, Format format) {
        int32_t r = vhr_upload_new_storage_image(context.handle, width, height, format);
        if (r < -1) check(context.handle, r, "UploadNewStorageImage");
        return uint32_t(r);
    }
    // resource_manager.h:29
    void DestroyStorageImage(uint32_t id) { check(context.handle, vhr_destroy_storage_image(context.handle, int32_t(id)), "DestroyStorageImage"); }
    // resource_manager.h:35
    void UpdatePerFrameUBO(uint32_t resource_idx, PerFrameData &per_frame_data) {
        check(context.handle, vhr_update_per_frame_ubo(context.handle, resource_idx, &per_frame_data), "UpdatePerFrameUBO");
    }
    DeviceContext &context;
};

class RaytracingExecutionContext {
public:
    explicit RaytracingExecutionContext(vhr_raytracing_execution_context *exec, vhr_context *ctx) : exec(exec), ctx(ctx) {}
    void TraceRays(uint32_t width, uint32_t height) { check(ctx, vhr_trace_rays(exec, width, height), "TraceRays"); }   // raytracing_execution_context.h:13
private:
    vhr_raytracing_execution_context *exec;
    vhr_context *ctx;
};

class ComputeExecutionContext {
public:
    explicit ComputeExecutionContext(vhr_compute_execution_context *exec, vhr_context *ctx) : exec(exec), ctx(ctx) {}
    uvec2 GetDisplaySize() {                                                                                 // compute_execution_context.h:17
        uvec2 s{};
        check(ctx, vhr_compute_get_display_size(exec, &s.x, &s.y), "GetDisplaySize");
        return s;
    }
    void Dispatch(const char *entry, uint32_t x_groups, uint32_t y_groups, uint32_t z_groups) {              // :18
        check(ctx, vhr_compute_dispatch(exec, entry, x_groups, y_groups, z_groups, nullptr, 0), "Dispatch");
    }
    template <typename T>
    void Dispatch(const char *entry, uint32_t x_groups, uint32_t y_groups, uint32_t z_groups, T &push_constants) {   // :20-27
        check(ctx, vhr_compute_dispatch(exec, entry, x_groups, y_groups, z_groups, &push_constants, uint32_t(sizeof(T))), "Dispatch");
    }
    void BlitImageStorageToTransient(int src, const char *dst) { check(ctx, vhr_compute_blit_image_storage_to_transient(exec, src, dst), "BlitImageStorageToTransient"); }   // :29
    void BlitImageTransientToStorage(const char *src, int dst) { check(ctx, vhr_compute_blit_image_transient_to_storage(exec, src, dst), "BlitImageTransientToStorage"); }   // :30
    void BlitImageStorageToStorage(int src, int dst) { check(ctx, vhr_compute_blit_image_storage_to_storage(exec, src, dst), "BlitImageStorageToStorage"); }                 // :31
private:
    vhr_compute_execution_context *exec;
    vhr_context *ctx;
};

// callback shapes of vulkan_common.h:326-341.  Graphics passes stay with the integrator, so their callback
// receives the device context instead of a GraphicsExecutionContext.
using ExternalPassCallback = std::function<void(DeviceContext &)>;
using RaytracingExecutionCallback = std::function<void(RaytracingExecutionContext &)>;
using ExecuteRaytracingCallback = std::function<void(std::string, RaytracingExecutionCallback)>;
using RaytracingPassCallback = std::function<void(ExecuteRaytracingCallback)>;
using ComputePassCallback = std::function<void(ComputeExecutionContext &)>;

class RenderGraph {
public:
    RenderGraph(DeviceContext &context, ResourceManager &resource_manager) : context(context), resource_manager(resource_manager) {}
    ~RenderGraph() { if (context.handle) vhr_graph_destroy_resources(context.handle); }

    void DestroyResources() {                                                                                // render_graph.h:8
        check(context.handle, vhr_graph_destroy_resources(context.handle), "DestroyResources");
        thunks.clear();
    }
    void AddGraphicsPass(const char *render_pass_name, std::vector<TransientResource> dependencies,           // render_graph.h:10-12
                         std::vector<TransientResource> outputs, ExternalPassCallback callback) {
        auto t = std::make_unique<Thunk>();
        t->graph = this;
        t->external = std::move(callback);
        check(context.handle, vhr_graph_add_graphics_pass(context.handle, render_pass_name, dependencies.data(), uint32_t(dependencies.size()),
                                                          outputs.data(), uint32_t(outputs.size()), t->external ? &RenderGraph::external_thunk : nullptr, t.get()),
              "AddGraphicsPass");
        thunks.push_back(std::move(t));
    }
    void AddRaytracingPass(const char *render_pass_name, std::vector<TransientResource> dependencies,         // render_graph.h:13-15
                           std::vector<TransientResource> outputs, RaytracingPipelineDescription pipeline, RaytracingPassCallback callback) {
        auto t = std::make_unique<Thunk>();
        t->graph = this;
        t->raytracing = std::move(callback);
        t->pipeline_name = pipeline.name ? pipeline.name : "";
        std::vector<vhr_hit_shader> hits;
        for (auto &h : pipeline.hit_shaders) hits.push_back(vhr_hit_shader{ h.closest_hit, h.any_hit });
        vhr_raytracing_pipeline_description d{ pipeline.name, pipeline.raygen_shader, pipeline.miss_shaders.data(), uint32_t(pipeline.miss_shaders.size()),
                                               hits.data(), uint32_t(hits.size()) };
        check(context.handle, vhr_graph_add_raytracing_pass(context.handle, render_pass_name, dependencies.data(), uint32_t(dependencies.size()), outputs.data(),
                                                            uint32_t(outputs.size()), &d, &RenderGraph::raytracing_thunk, t.get()),
              "AddRaytracingPass");
        thunks.push_back(std::move(t));
    }
    void AddComputePass(const char *render_pass_name, std::vector<TransientResource> dependencies,            // render_graph.h:16-18
                        std::vector<TransientResource> outputs, ComputePipelineDescription pipeline, ComputePassCallback callback) {
        auto t = std::make_unique<Thunk>();
        t->graph = this;
        t->compute = std::move(callback);
        std::vector<const char *> kernels;
        for (auto &k : pipeline.kernels) kernels.push_back(k.shader);
        vhr_compute_pipeline_description d{ kernels.data(), uint32_t(kernels.size()), pipeline.push_constant_description.size };
        check(context.handle, vhr_graph_add_compute_pass(context.handle, render_pass_name, dependencies.data(), uint32_t(dependencies.size()), outputs.data(),
                                                         uint32_t(outputs.size()), &d, &RenderGraph::compute_thunk, t.get()),
              "AddComputePass");
        thunks.push_back(std::move(t));
    }
    void Build() { check(context.handle, vhr_graph_build(context.handle), "Build"); }                         // render_graph.h:20
    void Execute(uint32_t resource_idx, uint32_t image_idx) {                                                 // render_graph.h:21
        pending = nullptr;
        int rc = vhr_graph_execute(context.handle, resource_idx, image_idx);
        if (pending) std::rethrow_exception(pending);
        check(context.handle, rc, "Execute");
    }
    void GatherPerformanceStatistics() { check(context.handle, vhr_graph_gather_performance_statistics(context.handle), "GatherPerformanceStatistics"); }   // render_graph.h:22
    double PassTimeMs(const char *render_pass_name) {
        double ema = 0, last = 0;
        check(context.handle, vhr_graph_get_pass_time_ms(context.handle, render_pass_name, &ema, &last), "PassTimeMs");
        return last;
    }
    bool ContainsImage(std::string image_name) { return vhr_graph_contains_image(context.handle, image_name.c_str()) != 0; }   // render_graph.h:25
    Format GetImageFormat(std::string image_name) { return vhr_graph_get_image_format(context.handle, image_name.c_str()); }     // render_graph.h:26

    DeviceContext &context;
    ResourceManager &resource_manager;

private:
    struct Thunk {
        RenderGraph *graph = nullptr;
        ExternalPassCallback external;
        RaytracingPassCallback raytracing;
        ComputePassCallback compute;
        std::string pipeline_name;
    };
    // exceptions must not unwind through the C ABI: park them and rethrow from Execute()
    static void external_thunk(void *user, vhr_context *) {
        Thunk *t = static_cast<Thunk *>(user);
        try { t->external(t->graph->context); } catch (...) { t->graph->pending = std::current_exception(); }
    }
    static void raytracing_thunk(void *user, vhr_raytracing_execution_context *exec) {
        Thunk *t = static_cast<Thunk *>(user);
        try {
            // ExecuteRaytracingPass, render_graph.cpp:889-912: the pass callback is handed an "execute pipeline" functor
            t->raytracing([&](std::string pipeline_name, RaytracingExecutionCallback execute_pipeline) {
                if (pipeline_name != t->pipeline_name) throw std::runtime_error("unknown raytracing pipeline '" + pipeline_name + "'");
                RaytracingExecutionContext execution_context(exec, t->graph->context.handle);
                execute_pipeline(execution_context);
            });
        } catch (...) { t->graph->pending = std::current_exception(); }
    }
    static void compute_thunk(void *user, vhr_compute_execution_context *exec) {
        Thunk *t = static_cast<Thunk *>(user);
        try {
            ComputeExecutionContext execution_context(exec, t->graph->context.handle);      // render_graph.cpp:917-918
            t->compute(execution_context);
        } catch (...) { t->graph->pending = std::current_exception(); }
    }
    std::vector<std::unique_ptr<Thunk>> thunks;
    std::exception_ptr pending;
};

// render_path.h:5-20, render_path.cpp:14-27
class RenderPath {
public:
    RenderPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager)
        : context(context), render_graph(render_graph), resource_manager(resource_manager) {}
    virtual ~RenderPath() = default;
    void Build() {
        check(context.handle, vhr_synchronize(context.handle), "vkDeviceWaitIdle");
        render_graph.DestroyResources();
        RegisterPath(context, render_graph, resource_manager);
        render_graph.Build();
    }
    void Rebuild() {
        check(context.handle, vhr_synchronize(context.handle), "vkDeviceWaitIdle");
        DeregisterPath(context, render_graph, resource_manager);
        Build();
    }
    virtual void RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) = 0;
    virtual void DeregisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) = 0;

protected:
    DeviceContext &context;
    RenderGraph &render_graph;
    ResourceManager &resource_manager;
};

}  // namespace vhr
