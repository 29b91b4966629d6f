// RenderGraph half of the C ABI: pass registry, transient images, execution order, per-pass timing, and the
// two execution contexts the pass callbacks drive.  Reference: src/render_graph/render_graph.{h,cpp},
// raytracing_execution_context.{h,cpp}, compute_execution_context.{h,cpp}.
//
// Where the reference records Vulkan commands and submits once per frame (renderer.cpp:135), callbacks here
// enqueue HIP work directly on the context's single in-order stream: the order of enqueue is the order of
// execution, which gives the sequential semantics the reference intends (it omits the compute->compute
// barriers between its SVGF dispatches, compute_execution_context.cpp:12-29).
#include <algorithm>
#include <cstring>
#include <deque>

#include "vhr_internal.hpp"

using namespace vhr;

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return (ctx)->fail(VHR_ERROR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static const char *kRenderOutput = "RENDER_OUTPUT";
static const char *kRaygen = "hybrid_render_path/raygen.rgen";
static const char *kMiss = "hybrid_render_path/miss.rmiss";
static const char *kReflectionMiss = "hybrid_render_path/reflection_miss.rmiss";
static const char *kReflectionHit = "hybrid_render_path/reflection_hit.rchit";
// the raytraced render path's shader sets (raytraced_render_path.cpp:19-34), SURVEY.md section 8 row f4
static const char *kRtRaygen = "raytraced_render_path/raygen.rgen";
static const char *kRtRaygenAlpha = "raytraced_render_path/raygen_test_alpha.rgen";
static const char *kRtMiss = "raytraced_render_path/miss.rmiss";
static const char *kRtShadowMiss = "raytraced_render_path/shadow_miss.rmiss";
static const char *kRtClosestHit = "raytraced_render_path/closesthit.rchit";
static const char *kRtClosestHitAlpha = "raytraced_render_path/closesthit_test_alpha.rchit";
static const char *kRtShadowAnyHit = "raytraced_render_path/shadow_anyhit.rahit";
static const char *kSvgf = "hybrid_render_path/svgf.comp";
static const char *kAtrous = "hybrid_render_path/svgf_atrous_filter.comp";
// the screen-space alternatives (hybrid_render_path.cpp:138-243), SURVEY.md section 8 row f4
static const char *kSsao = "hybrid_render_path/ssao.comp";
static const char *kSsaoBlur = "hybrid_render_path/ssao_blur.comp";
static const char *kSsr = "hybrid_render_path/ssr.comp";
static bool known_compute_shader(const char *k) {
    return !std::strcmp(k, kSvgf) || !std::strcmp(k, kAtrous) || !std::strcmp(k, kSsao) || !std::strcmp(k, kSsaoBlur) || !std::strcmp(k, kSsr);
}

static void free_pass_events(PassDescription &p) {
    if (p.ev_begin) hipEventDestroy(p.ev_begin);
    if (p.ev_end) hipEventDestroy(p.ev_end);
    p.ev_begin = p.ev_end = nullptr;
}

static int add_pass_common(vhr_context *ctx, PassDescription &p, const char *name, const vhr_transient_resource *deps, uint32_t ndeps,
                           const vhr_transient_resource *outs, uint32_t nouts) {
    if (!name || (!deps && ndeps) || (!outs && nouts)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Add*Pass: null argument");
    if (ctx->pass_descriptions.count(name))        // assert(!pass_descriptions.contains(name)), render_graph.cpp:84,99,114
        return ctx->fail(VHR_ERROR_GRAPH, std::string("render pass '") + name + "' is already registered");
    p.name = name;
    p.resource_names.reserve(ndeps + nouts);
    auto take = [&](const vhr_transient_resource *src, uint32_t n, std::vector<vhr_transient_resource> &dst) -> int {
        for (uint32_t i = 0; i < n; ++i) {
            if (!src[i].name) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Add*Pass: transient resource without a name");
            if (src[i].type != VHR_TRANSIENT_RESOURCE_IMAGE) return ctx->fail(VHR_ERROR_GRAPH, "transient buffers are not supported (render_graph.cpp:1015-1018)");
            p.resource_names.emplace_back(src[i].name);   // the reference stores the caller's pointer; we own a copy
            dst.push_back(src[i]);
        }
        return VHR_OK;
    };
    int rc = take(deps, ndeps, p.dependencies);
    if (rc) return rc;
    rc = take(outs, nouts, p.outputs);
    if (rc) return rc;
    size_t k = 0;
    for (auto &r : p.dependencies) r.name = p.resource_names[k++].c_str();
    for (auto &r : p.outputs) r.name = p.resource_names[k++].c_str();
    return VHR_OK;
}

static void commit_pass(vhr_context *ctx, PassDescription &&p) {
    const std::string name = p.name;
    auto &slot = ctx->pass_descriptions[name];
    slot = std::move(p);
    // resource name pointers must follow the moved strings
    size_t k = 0;
    for (auto &r : slot.dependencies) r.name = slot.resource_names[k++].c_str();
    for (auto &r : slot.outputs) r.name = slot.resource_names[k++].c_str();
    ctx->registration_order.push_back(name);
    ctx->built = false;
}

static const vhr_transient_resource *find_binding(const PassDescription &p, uint32_t binding) {
    for (auto &r : p.dependencies) if (r.image.binding == binding) return &r;
    for (auto &r : p.outputs) if (r.image.binding == binding) return &r;
    return nullptr;
}

static Image *pass_image(vhr_context *ctx, const PassDescription &p, uint32_t binding, int32_t expect_format, const char *what) {
    const vhr_transient_resource *r = find_binding(p, binding);
    if (!r) { ctx->fail(VHR_ERROR_GRAPH, std::string("pass '") + p.name + "' declares no resource at binding " + std::to_string(binding) + " (" + what + ")"); return nullptr; }
    auto it = ctx->images.find(r->name);
    if (it == ctx->images.end()) { ctx->fail(VHR_ERROR_GRAPH, std::string("transient image '") + r->name + "' was not created (graph not built?)"); return nullptr; }
    if (expect_format && it->second.format != expect_format) {
        ctx->fail(VHR_ERROR_GRAPH, std::string("transient image '") + r->name + "' has an unexpected format for " + what);
        return nullptr;
    }
    return &it->second;
}

extern "C" {

int vhr_graph_destroy_resources(vhr_context *ctx) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) {
        hipSetDevice(ctx->device);
        ctx->sync_streams();
        for (auto &kv : ctx->pass_descriptions) free_pass_events(kv.second);
        for (auto &kv : ctx->images) {
            Image &im = kv.second;
            if (im.slot_owned[0]) { for (void *q : im.slot_owned) hipFree(q); }      // one instance per frame slot
            else hipFree(im.owned);
        }
        for (int i = 0; i < 3; ++i) {
            if (ctx->front_done[i]) hipEventDestroy(ctx->front_done[i]);
            if (ctx->back_done[i]) hipEventDestroy(ctx->back_done[i]);
            ctx->front_done[i] = ctx->back_done[i] = nullptr;
            ctx->back_pending[i] = false;
        }
    }
    ctx->front_passes = 0;
    ctx->cur_slot = 0;
    ctx->pass_descriptions.clear();
    ctx->registration_order.clear();
    ctx->execution_order.clear();
    ctx->images.clear();
    ctx->compute_kernel_owner.clear();
    ctx->built = false;
    return VHR_OK;
}

int vhr_graph_add_graphics_pass(vhr_context *ctx, const char *name, const vhr_transient_resource *deps, uint32_t ndeps,
                                const vhr_transient_resource *outs, uint32_t nouts, vhr_external_pass_callback cb, void *user) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    PassDescription p;
    p.kind = PassKind::Graphics;
    int rc = add_pass_common(ctx, p, name, deps, ndeps, outs, nouts);
    if (rc) return rc;
    p.external_cb = cb;
    p.user = user;
    commit_pass(ctx, std::move(p));
    return VHR_OK;
}

int vhr_graph_add_raytracing_pass(vhr_context *ctx, const char *name, const vhr_transient_resource *deps, uint32_t ndeps,
                                  const vhr_transient_resource *outs, uint32_t nouts,
                                  const vhr_raytracing_pipeline_description *pipeline, vhr_raytracing_pass_callback cb, void *user) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (!pipeline || !pipeline->raygen_shader || !cb) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "AddRaytracingPass: pipeline description and callback are required");
    // Shader names select the HIP kernels: the hybrid path's set and the raytraced path's two sets exist here.
    const bool hybrid = !std::strcmp(pipeline->raygen_shader, kRaygen);
    const bool rt_opaque = !std::strcmp(pipeline->raygen_shader, kRtRaygen), rt_alpha = !std::strcmp(pipeline->raygen_shader, kRtRaygenAlpha);
    if (!hybrid && !rt_opaque && !rt_alpha)
        return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no HIP kernel for raygen shader '") + pipeline->raygen_shader + "'");
    PassDescription p;
    p.kind = PassKind::Raytracing;
    int rc = add_pass_common(ctx, p, name, deps, ndeps, outs, nouts);
    if (rc) return rc;
    p.pipeline_name = pipeline->name ? pipeline->name : "";
    p.raygen = pipeline->raygen_shader;
    for (uint32_t i = 0; i < pipeline->miss_shader_count; ++i) p.miss.emplace_back(pipeline->miss_shaders[i] ? pipeline->miss_shaders[i] : "");
    for (uint32_t i = 0; i < pipeline->hit_shader_count; ++i) {
        if (pipeline->hit_shaders[i].any_hit && !rt_alpha)
            return ctx->fail(VHR_ERROR_NOT_FOUND, "an any-hit shader only exists for the raytraced path's alpha test (raytraced_render_path.cpp:25-29); "
                                                  "the hybrid path's geometry is all opaque (resource_manager.cpp:633)");
        p.closest_hit.emplace_back(pipeline->hit_shaders[i].closest_hit ? pipeline->hit_shaders[i].closest_hit : "");
        p.any_hit.emplace_back(pipeline->hit_shaders[i].any_hit ? pipeline->hit_shaders[i].any_hit : "");
    }
    if (hybrid) {
        // miss index 0 = visibility, 1 = reflection (raygen.rgen:39,64); hit group 0 = reflection closest hit
        if (p.miss.size() < 2 || p.miss[0] != kMiss || p.miss[1] != kReflectionMiss || p.closest_hit.empty() || p.closest_hit[0] != kReflectionHit)
            return ctx->fail(VHR_ERROR_NOT_FOUND, "raytracing pipeline must name miss.rmiss, reflection_miss.rmiss and reflection_hit.rchit (hybrid_render_path.cpp:112-124)");
    } else {
        // miss index 0 = primary (raygen.rgen:20), 1 = shadow (closesthit.rchit:50); hit group 0 shades, its any-hit filters
        const char *want_hit = rt_alpha ? kRtClosestHitAlpha : kRtClosestHit;
        if (p.miss.size() < 2 || p.miss[0] != kRtMiss || p.miss[1] != kRtShadowMiss || p.closest_hit.empty() || p.closest_hit[0] != want_hit ||
            (rt_alpha && p.any_hit[0] != kRtShadowAnyHit))
            return ctx->fail(VHR_ERROR_NOT_FOUND, "raytracing pipeline must name the raytraced path's miss.rmiss, shadow_miss.rmiss and "
                                                  "closesthit.rchit, or with raygen_test_alpha.rgen closesthit_test_alpha.rchit + shadow_anyhit.rahit "
                                                  "(raytraced_render_path.cpp:19-34)");
    }
    p.rt_cb = cb;
    p.user = user;
    commit_pass(ctx, std::move(p));
    return VHR_OK;
}

int vhr_graph_add_compute_pass(vhr_context *ctx, const char *name, const vhr_transient_resource *deps, uint32_t ndeps,
                               const vhr_transient_resource *outs, uint32_t nouts, const vhr_compute_pipeline_description *pipeline,
                               vhr_compute_pass_callback cb, void *user) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (!pipeline || !cb || (!pipeline->kernels && pipeline->kernel_count)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "AddComputePass: pipeline description and callback are required");
    PassDescription p;
    p.kind = PassKind::Compute;
    int rc = add_pass_common(ctx, p, name, deps, ndeps, outs, nouts);
    if (rc) return rc;
    for (uint32_t i = 0; i < pipeline->kernel_count; ++i) {
        const char *k = pipeline->kernels[i];
        if (!k || !known_compute_shader(k))
            return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no HIP kernel for compute shader '") + (k ? k : "(null)") + "'");
        if (ctx->compute_kernel_owner.count(k))       // assert(!compute_pipelines.contains(kernel.shader)), render_graph.cpp:677
            return ctx->fail(VHR_ERROR_GRAPH, std::string("compute shader '") + k + "' is already registered by pass '" + ctx->compute_kernel_owner[k] + "'");
        p.kernels.emplace_back(k);
    }
    for (auto &k : p.kernels) ctx->compute_kernel_owner[k] = name;
    p.push_constant_size = pipeline->push_constant_size;
    p.compute_cb = cb;
    p.user = user;
    commit_pass(ctx, std::move(p));
    return VHR_OK;
}

// ActualizeResource, render_graph.cpp:921-977
static int actualize(vhr_context *ctx, const vhr_transient_resource &r) {
    if (!std::strcmp(r.name, kRenderOutput)) return VHR_OK;
    if (ctx->images.count(r.name)) return VHR_OK;
    Image im;
    im.width = (r.image.width == 0 && r.image.height == 0) ? ctx->width : r.image.width;      // swapchain sized, :960-964
    im.height = (r.image.width == 0 && r.image.height == 0) ? ctx->height : r.image.height;
    im.format = r.image.format;
    im.bpp = format_stride(r.image.format);
    if (!im.bpp || !im.width || !im.height) return ctx->fail(VHR_ERROR_GRAPH, std::string("transient image '") + r.name + "': unsupported format or empty extent");
    if (!ctx->host_only) {
        HIP_TRY(ctx, hipMalloc(&im.owned, im.bytes()));
        HIP_TRY(ctx, hipMemsetAsync(im.owned, 0, im.bytes(), ctx->stream));
        if (ctx->frames_in_flight > 1) {                      // one instance per frame slot
            im.slot_owned[0] = im.owned;
            for (int i = 1; i < ctx->frames_in_flight; ++i) {
                HIP_TRY(ctx, hipMalloc(&im.slot_owned[i], im.bytes()));
                HIP_TRY(ctx, hipMemsetAsync(im.slot_owned[i], 0, im.bytes(), ctx->stream));
            }
        }
    }
    im.ptr = im.owned;
    im.used = true;
    ctx->images[r.name] = im;
    return VHR_OK;
}

// FindExecutionOrder, render_graph.cpp:686-720: BFS from the single writer of RENDER_OUTPUT through
// writers[dependency], reversed, duplicates pruned keeping the first occurrence.
static int find_execution_order(vhr_context *ctx, std::map<std::string, std::vector<std::string>> &writers) {
    auto &w = writers[kRenderOutput];
    if (w.size() != 1) return ctx->fail(VHR_ERROR_GRAPH, "exactly one pass must write RENDER_OUTPUT (render_graph.cpp:687)");
    std::vector<std::string> order{ w[0] };
    std::deque<std::string> stack{ w[0] };
    size_t guard = 0;
    while (!stack.empty()) {
        const PassDescription &pass = ctx->pass_descriptions[stack.front()];
        stack.pop_front();
        for (const auto &dep : pass.dependencies)
            for (const auto &writer : writers[dep.name]) {
                order.push_back(writer);
                stack.push_back(writer);
            }
        if (++guard > 1000000) return ctx->fail(VHR_ERROR_GRAPH, "cyclic render graph");
    }
    std::reverse(order.begin(), order.end());
    std::vector<std::string> found;
    for (auto &n : order)
        if (std::find(found.begin(), found.end(), n) == found.end()) found.push_back(n);
    ctx->execution_order = found;
    return VHR_OK;
}

// SanityCheck, render_graph.cpp:980-1021
static int sanity_check(vhr_context *ctx) {
    std::map<std::string, std::vector<vhr_transient_resource>> participating;
    for (auto &name : ctx->execution_order) {
        const PassDescription &p = ctx->pass_descriptions[name];
        for (auto &r : p.dependencies) participating[r.name].push_back(r);
        for (auto &r : p.outputs) participating[r.name].push_back(r);
    }
    for (auto &kv : participating) {
        if (kv.first == kRenderOutput) continue;
        const auto &f = kv.second.front();
        for (auto &r : kv.second)
            if (r.image.width != f.image.width || r.image.height != f.image.height || r.image.format != f.image.format)
                return ctx->fail(VHR_ERROR_GRAPH, "SanityCheck: resource '" + kv.first + "' is declared with different extent/format by different passes");
    }
    return VHR_OK;
}

int vhr_graph_build(vhr_context *ctx) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->images.empty())                                  // the slot count of existing images stands (Build after DestroyResources changes it)
        ctx->frames_in_flight = ctx->host_only ? 1 : std::max(1, std::min(3, ctx->options[vhr::kOptFramesInFlight]));
    std::map<std::string, std::vector<std::string>> writers;
    int stamp_slots = 0;
    // the reference iterates an unordered_map; registration order is the deterministic choice here
    for (auto &name : ctx->registration_order) {
        PassDescription &p = ctx->pass_descriptions[name];
        for (auto &r : p.dependencies) { int rc = actualize(ctx, r); if (rc) return rc; }
        for (auto &r : p.outputs) { writers[r.name].push_back(p.name); int rc = actualize(ctx, r); if (rc) return rc; }
        if (!p.ev_begin && !ctx->host_only) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&p.ev_begin, hipEventDisableSystemFence));
            HIP_TRY(ctx, hipEventCreateWithFlags(&p.ev_end, hipEventDisableSystemFence));
        }
        p.stamp_index = stamp_slots < vhr::kMaxStampedPasses ? stamp_slots++ : -1;       // in-kernel time stamps (vhr::Stamps); beyond the table: events
    }
    int rc = find_execution_order(ctx, writers);
    if (rc) return rc;
    rc = sanity_check(ctx);
    if (rc) return rc;
    // frames in flight: the front of the frame = everything up to and including the last ray-tracing pass
    ctx->front_passes = 0;
    if (ctx->frames_in_flight > 1) {
        for (size_t i = 0; i < ctx->execution_order.size(); ++i)
            if (ctx->pass_descriptions[ctx->execution_order[i]].kind == PassKind::Raytracing) ctx->front_passes = i + 1;
        if (ctx->front_passes == ctx->execution_order.size()) ctx->front_passes = 0;       // nothing behind it to overlap with
        if (ctx->front_passes) {
            if (!ctx->front_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->front_stream, hipStreamNonBlocking));
            for (int i = 0; i < ctx->frames_in_flight; ++i) {
                if (!ctx->front_done[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->front_done[i], hipEventDisableTiming));
                if (!ctx->back_done[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->back_done[i], hipEventDisableTiming));
            }
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the instances' clears, before another stream may touch them
    }
    ctx->built = true;
    return VHR_OK;
}

int vhr_graph_execute(vhr_context *ctx, uint32_t resource_idx, uint32_t image_idx) {
    (void)image_idx;
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: Execute needs a device");
    if (!ctx->built) return ctx->fail(VHR_ERROR_GRAPH, "Execute before Build");
    if (resource_idx >= 3) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "resource_idx >= MAX_FRAMES_IN_FLIGHT");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->error.clear();
    ctx->last_resource_idx = resource_idx;
    const bool split = ctx->frames_in_flight > 1 && ctx->front_passes != 0;
    const uint32_t slot = ctx->frames_in_flight > 1 ? resource_idx % uint32_t(ctx->frames_in_flight) : 0u;
    hipStream_t const back = ctx->stream;
    if (ctx->frames_in_flight > 1) {
        ctx->cur_slot = slot;
        for (auto &kv : ctx->images) kv.second.select_slot(slot);
        // this frame's front rewrites the slot's images: the back of the frame that used the slot last has to be through with them
        if (split && ctx->back_pending[slot]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->front_stream, ctx->back_done[slot], 0));
    }
    struct RestoreStream { vhr_context *c; hipStream_t s; ~RestoreStream() { c->stream = s; } } restore{ ctx, back };
    for (size_t pi = 0; pi < ctx->execution_order.size(); ++pi) {
        PassDescription &p = ctx->pass_descriptions[ctx->execution_order[pi]];
        ctx->stream = (split && pi < ctx->front_passes) ? ctx->front_stream : back;
        // vkCmdWriteTimestamp x2 (render_graph.cpp:167-182): the stamps ride on the pass's first and last kernel dispatch
        // (vhr_context::dispatch_events).  External (graphics) passes enqueue nothing here and are timed by their owner's API.
        const bool stamps = ctx->options[vhr::kOptPassTimestamps] != 0 && p.kind != PassKind::Graphics;
        p.begin_stamped = p.end_on_last_dispatch = p.stamped_in_kernel = false;
        ctx->cur_pass = stamps ? &p : nullptr;
        if (p.kind == PassKind::Graphics) {
            { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) return jrc; }               // its owner may read the Reflections image (or rewrite the G-buffer the mirror ray reads)
            if (ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }      // its owner enqueues on the stream
            // An in-kernel END stamp still pending belongs to the library pass in front of this one; whatever the owner's callback enqueues
            // on the stream would be charged to that pass (the stamp is stored by the NEXT library kernel).  "pass_timestamps" 2 closes the
            // pass here with the one-thread stamp kernel (+6 us per external pass with work behind a library pass); mode 1 leaves it open --
            // right for callbacks that only bind images, as the harness's do.
            if (p.external_cb && ctx->pending_end && ctx->options[vhr::kOptPassTimestamps] == 2) vhr::launch_stamp(ctx);
            if (p.external_cb) p.external_cb(p.user, ctx);
        } else if (p.kind == PassKind::Raytracing) {
            vhr_raytracing_execution_context ec{ ctx, &p, resource_idx };      // ExecuteRaytracingPass, :889-912
            ctx->may_defer_raygen = !p.epilogue_cb;                            // "fuse_temporal": nobody waits for this pass's image at its end
            p.rt_cb(p.user, &ec);
            ctx->may_defer_raygen = false;
        } else {
            vhr_compute_execution_context ec{ ctx, &p, resource_idx };         // ExecuteComputePass, :914-919
            ctx->recording = true;                                             // the callback records, like a command buffer
            p.compute_cb(p.user, &ec);
            ctx->recording = false;
            const int frc = vhr::flush_recorded(ctx);                          // ... and the pass is issued when it returns
            if (frc != VHR_OK) { ctx->cur_pass = nullptr; return frc; }
        }
        ctx->cur_pass = nullptr;
        if (stamps && p.stamped_in_kernel) ctx->pending_end = &ctx->d_stamps[p.stamp_index].end;      // the next kernel on the stream stores it
        else if (stamps && p.begin_stamped && !p.end_on_last_dispatch) HIP_TRY(ctx, hipEventRecord(p.ev_end, ctx->stream));
        p.timed = stamps && p.begin_stamped;
        if (split && pi + 1 == ctx->front_passes) {                            // the back of the frame consumes what its front produced
            HIP_TRY(ctx, hipEventRecord(ctx->front_done[slot], ctx->front_stream));
            HIP_TRY(ctx, hipStreamWaitEvent(back, ctx->front_done[slot], 0));
        }
        // Epilogues run on the stream their consumers are on: the caller's, where the pass ran there or has just been handed over to
        // it (the last front pass); a front pass with more front passes behind it keeps the front stream, so that its epilogue is
        // ordered between the two (it used to land on the caller's stream, unordered against either).  Whichever it is,
        // vhr_get_current_stream tells the callback.
        ctx->stream = (split && pi + 1 < ctx->front_passes) ? ctx->front_stream : back;
        if (p.epilogue_cb && ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }
        // its owner expects the pass's images, the mirror ray's among them ("reflection_async" 2: the owners of epilogues promise not to touch
        // the Reflections image or the G-buffer -- the multi-GPU harness, whose hooks exchange visibility and SVGF history only)
        if (p.epilogue_cb && ctx->options[vhr::kOptReflectionAsync] != 2) { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) return jrc; }
        if (p.epilogue_cb) p.epilogue_cb(p.epilogue_user, ctx);
        ctx->stream = back;
        if (!ctx->error.empty()) return VHR_ERROR_GRAPH;                       // a callback's call failed: surface it
    }
    if (split) {
        HIP_TRY(ctx, hipEventRecord(ctx->back_done[slot], back));
        ctx->back_pending[slot] = true;
    }
    // A stamped pass with no kernel of the library behind it in the frame: its end is stored by the next kernel the stream runs -- the
    // next frame's first one while the host runs ahead of the GPU (the reference's loop does, renderer.cpp:103-146), or the one-thread
    // kernel every call that waits for the stream issues first (vhr_synchronize, GatherPerformanceStatistics, image downloads).
    // "pass_timestamps" 2 issues that kernel here instead, for hosts that neither run ahead nor wait (+6 us per frame).
    if (ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }      // no pass took it: issued as it is
    { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) return jrc; }        // "reflection_async": the frame ends with its mirror ray (the next frame's producer rewrites what it reads)
    if (ctx->pending_end && ctx->options[vhr::kOptPassTimestamps] == 2) vhr::launch_stamp(ctx);
    return VHR_OK;
}

int vhr_graph_gather_performance_statistics(vhr_context *ctx) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return VHR_OK;
    { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }  // VK_QUERY_RESULT_WAIT_BIT, render_graph.cpp:192-193
    vhr::PassStampPair host_stamps[vhr::kMaxStampedPasses];
    bool have_stamps = false;
    for (auto &name : ctx->execution_order) {
        PassDescription &p = ctx->pass_descriptions[name];
        if (!p.timed) continue;
        float ms = 0.0f;
        if (p.stamped_in_kernel) {
            if (!have_stamps) {
                if (hipMemcpy(host_stamps, ctx->d_stamps, sizeof(host_stamps), hipMemcpyDeviceToHost) != hipSuccess) continue;
                have_stamps = true;
            }
            const vhr::PassStampPair &sp = host_stamps[p.stamp_index];
            if (sp.end < sp.begin) continue;
            ms = float(double(sp.end - sp.begin) / ctx->wall_clock_khz);
        } else if (hipEventElapsedTime(&ms, p.ev_begin, p.ev_end) != hipSuccess) continue;
        p.last_ms = ms;
        p.ema_ms = p.ema_ms * 0.95 + double(ms) * 0.05;                        // render_graph.cpp:199
    }
    return VHR_OK;
}

int vhr_graph_get_pass_time_ms(vhr_context *ctx, const char *name, double *ema_ms, double *last_ms) {
    if (!ctx || !name) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->pass_descriptions.find(name);
    if (it == ctx->pass_descriptions.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no pass named '") + name + "'");
    if (ema_ms) *ema_ms = it->second.ema_ms;
    if (last_ms) *last_ms = it->second.last_ms;
    return VHR_OK;
}

int vhr_graph_get_execution_order(vhr_context *ctx, char *buf, uint32_t buf_size) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    std::string s;
    for (size_t i = 0; i < ctx->execution_order.size(); ++i) { if (i) s += '\n'; s += ctx->execution_order[i]; }
    if (buf && buf_size) { std::strncpy(buf, s.c_str(), buf_size - 1); buf[buf_size - 1] = 0; }
    return int(ctx->execution_order.size());
}

int vhr_graph_contains_image(vhr_context *ctx, const char *image_name) {
    return (ctx && image_name && ctx->images.count(image_name)) ? 1 : 0;
}
int32_t vhr_graph_get_image_format(vhr_context *ctx, const char *image_name) {
    if (!ctx || !image_name) return 0;
    auto it = ctx->images.find(image_name);
    return it == ctx->images.end() ? 0 : it->second.format;
}

int vhr_graph_set_pass_epilogue(vhr_context *ctx, const char *name, vhr_external_pass_callback cb, void *user) {
    if (!ctx || !name) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->pass_descriptions.find(name);
    if (it == ctx->pass_descriptions.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no pass named '") + name + "'");
    it->second.epilogue_cb = cb;
    it->second.epilogue_user = user;
    return VHR_OK;
}

int vhr_graph_bind_external_image(vhr_context *ctx, const char *image_name, void *device_ptr) {
    if (!ctx || !image_name) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->images.find(image_name);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + image_name + "'");
    if (reinterpret_cast<uintptr_t>(device_ptr) & 15u)           // the kernels read and write these images with up to 16-byte accesses
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, std::string("external memory for '") + image_name + "' must be 16-byte aligned");
    it->second.slot_external[ctx->cur_slot] = device_ptr;       // belongs to the frame slot being recorded (slot 0 without frames in flight)
    it->second.ptr = device_ptr ? device_ptr : it->second.owned;
    return VHR_OK;
}

// ---- RaytracingExecutionContext::TraceRays (raytracing_execution_context.cpp:4-13) ----
int vhr_trace_rays(vhr_raytracing_execution_context *exec, uint32_t width, uint32_t height) {
    if (!exec) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = exec->ctx;
    const PassDescription &p = *exec->pass;
    if (p.raygen == kRtRaygen || p.raygen == kRtRaygenAlpha) {
        // set 3 binding 0 of raytraced_render_path/raygen.rgen:6 (declared rgba8, created B8G8R8A8_UNORM, raytraced_render_path.cpp:15)
        Image *out = pass_image(ctx, p, 0, VHR_FORMAT_B8G8R8A8_UNORM, "output_image");
        if (!out) return VHR_ERROR_GRAPH;
        return launch_raytraced(ctx, ctx->per_frame[exec->resource_idx], width, height, *out, p.raygen == kRtRaygenAlpha);
    }
    // set 3 bindings of raygen.rgen:6-9
    Image *normals = pass_image(ctx, p, 0, VHR_FORMAT_R16G16B16A16_SFLOAT, "world_space_normals_and_object_ids");
    Image *depth = pass_image(ctx, p, 1, VHR_FORMAT_D32_SFLOAT, "depth");
    Image *shadow_ao = pass_image(ctx, p, 2, VHR_FORMAT_R16G16_SFLOAT, "raytraced_shadow_and_ambient_occlusion");
    if (!normals || !depth || !shadow_ao) return VHR_ERROR_GRAPH;
    Image *reflections = find_binding(p, 3) ? pass_image(ctx, p, 3, VHR_FORMAT_R16G16B16A16_SFLOAT, "raytraced_reflections") : nullptr;
    if (find_binding(p, 3) && !reflections) return VHR_ERROR_GRAPH;
    return launch_raygen(ctx, ctx->per_frame[exec->resource_idx], width, height, *normals, *depth, *shadow_ao, reflections);
}

// ---- ComputeExecutionContext (compute_execution_context.{h,cpp}) ----
int vhr_compute_get_display_size(vhr_compute_execution_context *exec, uint32_t *width, uint32_t *height) {
    if (!exec || !width || !height) return VHR_ERROR_INVALID_ARGUMENT;
    *width = exec->ctx->width;              // context.swapchain.extent, compute_execution_context.cpp:8-10
    *height = exec->ctx->height;
    return VHR_OK;
}

static Image *storage_image(vhr_context *ctx, int32_t id, const char *what) {
    if (id < 0 || uint32_t(id) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[id].used) {
        ctx->fail(VHR_ERROR_NOT_FOUND, std::string("storage image index ") + std::to_string(id) + " (" + what + ") is not allocated");
        return nullptr;
    }
    return &ctx->storage_images[id];
}

int vhr_compute_dispatch(vhr_compute_execution_context *exec, const char *shader, uint32_t x_groups, uint32_t y_groups,
                         uint32_t z_groups, const void *push_constants, uint32_t push_constants_size) {
    if (!exec || !shader) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = exec->ctx;
    const PassDescription &p = *exec->pass;
    if (std::find(p.kernels.begin(), p.kernels.end(), shader) == p.kernels.end())
        return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("compute shader '") + shader + "' is not part of pass '" + p.name + "'");
    if (push_constants_size != p.push_constant_size || (push_constants_size && !push_constants))     // assert, compute_execution_context.h:23
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: push constant size differs from the pipeline's declaration");
    if (z_groups != 1) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: z_groups must be 1");
    const vhr_per_frame_data &pfd = ctx->per_frame[exec->resource_idx];
    // ---- the screen-space alternatives.  Bindings: ssao.comp:8-10, ssao_blur.comp:8-9, ssr.comp:8-12 ----
    // (issued at once: anything the pass has recorded so far goes first, in order)
    if (std::strcmp(shader, kSvgf) != 0 && std::strcmp(shader, kAtrous) != 0 && !ctx->recorded.empty()) {
        const int frc = vhr::flush_recorded(ctx);
        if (frc != VHR_OK) return frc;
    }
    if (!std::strcmp(shader, kSsao)) {
        // The reference dispatches ssao.comp WITHOUT push constants (its pass declares none, hybrid_render_path.cpp:151-167) although
        // the shader reads pc.radius; a caller that does push SSAOPushConstants here is accepted too.
        if (push_constants_size != 0 && push_constants_size != sizeof(vhr_ssao_push_constants))
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: ssao.comp takes no push constants (as the reference dispatches it) or SSAOPushConstants (4 bytes)");
        if (push_constants_size) std::memcpy(&ctx->ssao_radius, push_constants, sizeof(float));
        Image *normals = pass_image(ctx, p, 0, VHR_FORMAT_R16G16B16A16_SFLOAT, "world_space_normals_and_object_ids");
        Image *depth = pass_image(ctx, p, 1, VHR_FORMAT_D32_SFLOAT, "depth");
        Image *out = pass_image(ctx, p, 2, VHR_FORMAT_R16G16B16A16_SFLOAT, "screen_space_ambient_occlusion");
        if (!normals || !depth || !out) return VHR_ERROR_GRAPH;
        return launch_ssao(ctx, pfd, *normals, *depth, *out, ctx->ssao_radius, x_groups, y_groups);
    }
    if (!std::strcmp(shader, kSsaoBlur)) {
        if (push_constants_size != sizeof(vhr_ssao_push_constants))
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: ssao_blur.comp is dispatched with SSAOPushConstants (4 bytes, hybrid_render_path.cpp:182-197)");
        std::memcpy(&ctx->ssao_radius, push_constants, sizeof(float));     // the shader ignores it; the next ssao.comp dispatch sees it
        Image *in = pass_image(ctx, p, 0, VHR_FORMAT_R16G16B16A16_SFLOAT, "screen_space_ambient_occlusion");
        Image *out = pass_image(ctx, p, 1, VHR_FORMAT_R16G16B16A16_SFLOAT, "screen_space_ambient_occlusion_blurred");
        if (!in || !out) return VHR_ERROR_GRAPH;
        return launch_ssao_blur(ctx, pfd, *in, *out, x_groups, y_groups);
    }
    if (!std::strcmp(shader, kSsr)) {
        if (push_constants_size != sizeof(vhr_ssr_push_constants))
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: ssr.comp takes SSRPushConstants (16 bytes)");
        vhr_ssr_push_constants spc;
        std::memcpy(&spc, push_constants, sizeof spc);
        Image *albedo = pass_image(ctx, p, 0, VHR_FORMAT_B8G8R8A8_UNORM, "albedo");
        Image *normals = pass_image(ctx, p, 1, VHR_FORMAT_R16G16B16A16_SFLOAT, "world_space_normals_and_object_ids");
        Image *motion = pass_image(ctx, p, 2, VHR_FORMAT_R16G16B16A16_SFLOAT, "motion_vectors_and_metallic_roughness");
        Image *depth = pass_image(ctx, p, 3, VHR_FORMAT_D32_SFLOAT, "depth");
        Image *out = pass_image(ctx, p, 4, VHR_FORMAT_R16G16B16A16_SFLOAT, "screen_space_reflections");
        if (!albedo || !normals || !motion || !depth || !out) return VHR_ERROR_GRAPH;
        return launch_ssr(ctx, pfd, *albedo, *normals, *motion, *depth, *out, spc, x_groups, y_groups);
    }
    if (push_constants_size != sizeof(vhr_svgf_push_constants))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "Dispatch: the SVGF kernels take SVGFPushConstants (24 bytes)");
    vhr_svgf_push_constants pc;
    std::memcpy(&pc, push_constants, sizeof pc);              // copied at call time, like vkCmdPushConstants
    // set 3 bindings of svgf.comp:8-12 / svgf_atrous_filter.comp:8-12
    Image *normals = pass_image(ctx, p, 0, VHR_FORMAT_R16G16B16A16_SFLOAT, "world_space_normals_and_object_ids");
    if (!normals) return VHR_ERROR_GRAPH;
    if (!std::strcmp(shader, kSvgf)) {
        Image *motion = pass_image(ctx, p, 1, VHR_FORMAT_R16G16B16A16_SFLOAT, "motion_vectors_and_metallic_roughness");
        Image *raytraced = pass_image(ctx, p, 3, VHR_FORMAT_R16G16_SFLOAT, "raytraced_shadow_and_ao_texture");
        Image *integrated = storage_image(ctx, pc.integrated_shadow_and_ao[0], "integrated_shadow_and_ao[0]");
        Image *prev_normals = storage_image(ctx, pc.prev_frame_normals_and_object_ids, "prev_frame_normals_and_object_ids");
        Image *history = storage_image(ctx, pc.shadow_and_ao_history, "shadow_and_ao_history");
        Image *moments = storage_image(ctx, pc.shadow_and_ao_moments_history, "shadow_and_ao_moments_history");
        if (!motion || !raytraced || !integrated || !prev_normals || !history || !moments) return VHR_ERROR_GRAPH;
        return launch_svgf_temporal(ctx, pfd, *normals, *motion, *raytraced, *prev_normals, *history, *moments, *integrated, x_groups, y_groups);
    }
    Image *in = storage_image(ctx, pc.integrated_shadow_and_ao[0], "integrated_shadow_and_ao[0]");
    Image *out = storage_image(ctx, pc.integrated_shadow_and_ao[1], "integrated_shadow_and_ao[1]");
    if (!in || !out) return VHR_ERROR_GRAPH;
    return launch_svgf_atrous(ctx, pfd, *normals, *in, *out, pc.atrous_step, x_groups, y_groups);
}

int vhr_compute_blit_image_storage_to_transient(vhr_compute_execution_context *exec, int32_t src, const char *dst) {
    if (!exec || !dst) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = exec->ctx;
    Image *s = storage_image(ctx, src, "blit source");
    auto it = ctx->images.find(dst);
    if (!s) return VHR_ERROR_GRAPH;
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + dst + "'");
    return copy_image_rows(ctx, *s, it->second);
}
int vhr_compute_blit_image_transient_to_storage(vhr_compute_execution_context *exec, const char *src, int32_t dst) {
    if (!exec || !src) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = exec->ctx;
    Image *d = storage_image(ctx, dst, "blit destination");
    auto it = ctx->images.find(src);
    if (!d) return VHR_ERROR_GRAPH;
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + src + "'");
    return copy_image_rows(ctx, it->second, *d);
}
int vhr_compute_blit_image_storage_to_storage(vhr_compute_execution_context *exec, int32_t src, int32_t dst) {
    if (!exec) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = exec->ctx;
    Image *s = storage_image(ctx, src, "blit source"), *d = storage_image(ctx, dst, "blit destination");
    if (!s || !d) return VHR_ERROR_GRAPH;
    return copy_image_rows(ctx, *s, *d);
}

}  // extern "C"
