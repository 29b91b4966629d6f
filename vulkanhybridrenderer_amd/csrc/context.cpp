// Context + ResourceManager half of the C ABI (include/vhr_amd.h): device memory for the global scene
// buffers, the bindless texture table, the storage-image pool and the per-frame uniform data.
// Reference: src/rendering_backend/resource_manager.{h,cpp}, vulkan_context.cpp.
#include <cmath>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>

#include "vhr_internal.hpp"
#include "fingerprint.h"      // build/fingerprint.h, written by the Makefile

namespace vhr {
int upload_srgb_lut(const float *lut);

uint32_t format_stride(int32_t format) {      // VkUtils::FormatStride, vulkan_utils.h:128-148
    switch (format) {
        case VHR_FORMAT_R8G8B8A8_UNORM:
        case VHR_FORMAT_R8G8B8A8_SRGB:
        case VHR_FORMAT_B8G8R8A8_UNORM:
        case VHR_FORMAT_B8G8R8A8_SRGB:
        case VHR_FORMAT_R16G16_SFLOAT:
        case VHR_FORMAT_D32_SFLOAT: return 4;
        case VHR_FORMAT_R16G16B16A16_SFLOAT: return 8;
        default: return 0;
    }
}
}  // namespace vhr

using namespace vhr;

static thread_local std::string g_create_error;     // vhr_create has no context to carry its message: per thread

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return (ctx)->fail(VHR_ERROR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

vhr::DeviceScene vhr_context::device_scene() const {
    DeviceScene s;
    s.nodes = d_nodes;
    s.nodes16 = d_nodes16;
    s.nodes_ch = d_nodes_ch;
    s.nodes48 = d_nodes48;
    s.centre[0] = bvh_centre[0]; s.centre[1] = bvh_centre[1]; s.centre[2] = bvh_centre[2];
    s.pad0 = 0.0f;
    s.tris = d_tris;
    s.vertices = d_vertices;
    s.indices = d_indices;
    s.primitives = d_primitives;
    s.normal_matrices = d_normal_matrices;
    s.textures = d_textures;
    s.node_count = node_count;
    s.tri_count = tri_count;
    s.primitive_count = primitive_count;
    s.texture_count = uint32_t(textures.size());
    for (int i = 0; i < 9; ++i) s.frame[i] = bvh_frame[i];
    s.frame_on = bvh_frame_on ? 1u : 0u;
    return s;
}

// ---- optional per-kernel event timing -------------------------------------------------------------
static constexpr size_t kTimerCapacity = 16384;     // events per kernel kind (8192 launches between drains)

// Events for the next kernel dispatch.  A kernel kind under vhr_set_kernel_timing gets its own (start, stop) pair; the pass
// whose callback is running gets its begin stamp on its first dispatch and its end stamp on every dispatch (the last one
// stands).  Where both want the same slot the kernel timer wins and the pass falls back to a recorded event.
void vhr_context::dispatch_events(hipEvent_t &start, hipEvent_t &stop) {
    start = stop = nullptr;
    if (timing_kind >= 0 && (kernel_timing_mask & (1u << timing_kind))) {
        KernelTimer &t = kernel_timers[timing_kind];
        // "kernel_timing_stride" n: only every n-th launch of the kind carries an event pair (a profiled dispatch costs ~6 us of
        // completion-signal and release-fence handling that the next kernel waits for; sampling keeps the measured loop honest)
        const uint64_t stride = uint64_t(std::max(1, options[vhr::kOptKernelTimingStride]));
        if ((t.seen++ % stride) == 0 && t.used + 2 <= kTimerCapacity) {
            while (t.events.size() < t.used + 2) {
                hipEvent_t e;
                if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) break;
                t.events.push_back(e);
            }
            if (t.events.size() >= t.used + 2) {
                start = t.events[t.used];
                stop = t.events[t.used + 1];
                t.used += 2;
            }
        }
    }
    if (cur_pass && options[vhr::kOptPassTimestamps] && cur_pass->ev_begin && !(in_kernel_stamps() && cur_pass->stamp_index >= 0)) {
        if (!cur_pass->begin_stamped) {
            if (!start) start = cur_pass->ev_begin;
            else hipEventRecord(cur_pass->ev_begin, stream);
            cur_pass->begin_stamped = true;
        }
        if (!stop) { stop = cur_pass->ev_end; cur_pass->end_on_last_dispatch = true; }
        else cur_pass->end_on_last_dispatch = false;
    }
}

// The stamps of the next kernel launch (vhr::launch): the end of the pass that finished last, the begin of the running pass's first kernel.
vhr::Stamps vhr_context::take_stamps() {
    vhr::Stamps st{ nullptr, nullptr };
    if (!in_kernel_stamps()) pending_end = nullptr;      // (stamps switched to event pairs while an end was pending: nothing will store it)
    if (no_stamps || !in_kernel_stamps()) return st;
    if (pending_end) { st.prev_end = pending_end; pending_end = nullptr; }
    if (cur_pass && options[vhr::kOptPassTimestamps] && cur_pass->stamp_index >= 0 && !cur_pass->begin_stamped) {
        st.begin = &d_stamps[cur_pass->stamp_index].begin;
        cur_pass->begin_stamped = true;
        cur_pass->stamped_in_kernel = true;
    }
    return st;
}

int vhr_context::sync_streams() {
    if (host_only) return VHR_OK;
    if (deferred_raygen) { const int drc = vhr::flush_deferred_raygen(this, nullptr); if (drc != VHR_OK) return drc; }
    if (pending_end) vhr::launch_stamp(this);          // the end of the last pass, before the host waits for it
    if (front_stream && hipStreamSynchronize(front_stream) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamSynchronize(front stream) failed");
    if (side_stream && hipStreamSynchronize(side_stream) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamSynchronize(side stream) failed");
    side_pending = false;
    if (refl_stream && hipStreamSynchronize(refl_stream) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamSynchronize(mirror-ray stream) failed");
    refl_pending = false;
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamSynchronize failed");     // (a null handle is the default stream)
    return VHR_OK;
}

int vhr_context::join_side() {
    if (!side_pending) return VHR_OK;
    side_pending = false;
    if (hipStreamWaitEvent(stream, side_done, 0) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamWaitEvent(side stream) failed");
    return VHR_OK;
}

int vhr_context::join_refl() {
    if (!refl_pending) return VHR_OK;
    refl_pending = false;
    if (hipStreamWaitEvent(stream, refl_done, 0) != hipSuccess) return fail(VHR_ERROR_DEVICE, "hipStreamWaitEvent(mirror-ray stream) failed");
    return VHR_OK;
}

extern "C" {

int vhr_set_kernel_timing(vhr_context *ctx, int32_t kind_mask) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    ctx->kernel_timing_mask = uint32_t(kind_mask);
    return VHR_OK;
}

int vhr_get_kernel_time(vhr_context *ctx, int32_t kind, double *total_ms, uint64_t *launches, int32_t reset) {
    if (!ctx || kind < 0 || kind >= kKernelKinds) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    KernelTimer &t = ctx->kernel_timers[kind];
    for (size_t i = 0; i + 1 < t.used; i += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, t.events[i], t.events[i + 1]) == hipSuccess) { t.total_ms += ms; ++t.launches; }
    }
    t.used = 0;
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    if (reset) { t.total_ms = 0.0; t.launches = 0; }
    return VHR_OK;
}

const char *vhr_version(void) { return "vhr_amd 0.1.0 (gfx950)"; }

int vhr_abi_struct_sizes(uint32_t out[8]) {
    out[0] = sizeof(vhr_vertex); out[1] = sizeof(vhr_material); out[2] = sizeof(vhr_primitive);
    out[3] = sizeof(vhr_directional_light); out[4] = sizeof(vhr_per_frame_data); out[5] = sizeof(vhr_svgf_push_constants);
    out[6] = sizeof(vhr_trace_params); out[7] = 0;
    return 7;
}

void vhr_default_trace_params(vhr_trace_params *p) {
    p->shadow_enable = 1;          // raygen.rgen:31-41
    p->ao_spp = 2;                 // raygen.rgen:45
    p->ao_tmax = 5.0f;             // raygen.rgen:52
    p->reflections = 1;            // raygen.rgen:59-65
    p->cone_cos_max = 0.999995f;   // raygen.rgen:34
    p->normal_bias = 0.1f;         // raygen.rgen:29
    p->tmin = 0.01f;               // raygen.rgen:40
    p->tmax = 10000.0f;
}

int vhr_create(const vhr_create_info *info, vhr_context **out) {
    if (!info || !out || info->width == 0 || info->height == 0) { g_create_error = "vhr_create: invalid arguments"; return VHR_ERROR_INVALID_ARGUMENT; }
    if (info->flags & VHR_CREATE_HOST_ONLY) {
        // graph bookkeeping only (pass registry, execution order, SanityCheck): no device is touched and
        // nothing can be executed, uploaded or downloaded
        vhr_context *ctx = new vhr_context();
        ctx->host_only = true;
        ctx->width = info->width;
        ctx->height = info->height;
        ctx->row_end = info->height;
        ctx->col_end = info->width;
        ctx->storage_images.resize(vhr_context::kMaxGlobalResources);
        vhr_default_trace_params(&ctx->trace_params);
        *out = ctx;
        return VHR_OK;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        // the product path never falls back to the CPU
        g_create_error = "vhr_create: no HIP device available (this library has no CPU path)";
        return VHR_ERROR_NO_DEVICE;
    }
    if (info->device < 0 || info->device >= count) { g_create_error = "vhr_create: device ordinal out of range"; return VHR_ERROR_INVALID_ARGUMENT; }
    if (hipSetDevice(info->device) != hipSuccess) { g_create_error = "vhr_create: hipSetDevice failed"; return VHR_ERROR_DEVICE; }
    vhr_context *ctx = new vhr_context();
    ctx->device = info->device;
    ctx->width = info->width;
    ctx->height = info->height;
    ctx->row_begin = 0;
    ctx->row_end = info->height;
    ctx->col_begin = 0;
    ctx->col_end = info->width;
    if (info->flags & VHR_CREATE_INTERNAL_STREAM) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            g_create_error = "vhr_create: hipStreamCreate failed";
            delete ctx;
            return VHR_ERROR_DEVICE;
        }
        ctx->own_stream = true;
    } else {
        // the caller's stream, used as given; NULL is the device's default stream (which is also what
        // torch.cuda.current_stream().cuda_stream reports for PyTorch's default stream)
        ctx->stream = static_cast<hipStream_t>(info->stream);
    }
    ctx->storage_images.resize(vhr_context::kMaxGlobalResources);
    vhr_default_trace_params(&ctx->trace_params);
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        lut[i] = float(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, info->device) == hipSuccess && prop.multiProcessorCount > 0) ctx->cu_count = prop.multiProcessorCount;
    if (upload_srgb_lut(lut) != 0 || hipMalloc(reinterpret_cast<void **>(&ctx->d_ray_stats), 2 * sizeof(RayStats)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_tile_counter), sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->d_stamps), sizeof(vhr::PassStampPair) * vhr::kMaxStampedPasses) != hipSuccess ||
        hipMemset(ctx->d_stamps, 0, sizeof(vhr::PassStampPair) * vhr::kMaxStampedPasses) != hipSuccess) {
        g_create_error = "vhr_create: device initialisation failed";
        vhr_destroy(ctx);
        return VHR_ERROR_DEVICE;
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, info->device) == hipSuccess && khz > 0) ctx->wall_clock_khz = double(khz);
    *out = ctx;
    return VHR_OK;
}

static void free_scene(vhr_context *ctx) {
    hipFree(ctx->d_vertices); hipFree(ctx->d_indices); hipFree(ctx->d_primitives); hipFree(ctx->d_normal_matrices);
    hipFree(ctx->d_nodes); hipFree(ctx->d_nodes16); hipFree(ctx->d_nodes_ch); hipFree(ctx->d_nodes48); hipFree(ctx->d_tris);
    ctx->d_vertices = nullptr; ctx->d_indices = nullptr; ctx->d_primitives = nullptr; ctx->d_normal_matrices = nullptr;
    ctx->d_nodes = nullptr; ctx->d_nodes16 = nullptr; ctx->d_nodes_ch = nullptr; ctx->d_nodes48 = nullptr; ctx->d_tris = nullptr;
    ctx->vertex_count = ctx->index_count = ctx->primitive_count = ctx->node_count = ctx->tri_count = 0;
}

void vhr_destroy(vhr_context *ctx) {
    if (!ctx) return;
    if (ctx->host_only) {
        vhr_graph_destroy_resources(ctx);
        delete ctx;
        return;
    }
    hipSetDevice(ctx->device);
    ctx->sync_streams();
    vhr_graph_destroy_resources(ctx);
    if (ctx->front_stream) hipStreamDestroy(ctx->front_stream);
    if (ctx->refl_stream) hipStreamDestroy(ctx->refl_stream);
    if (ctx->refl_ready) hipEventDestroy(ctx->refl_ready);
    if (ctx->refl_done) hipEventDestroy(ctx->refl_done);
    if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
    if (ctx->side_ready) hipEventDestroy(ctx->side_ready);
    if (ctx->side_done) hipEventDestroy(ctx->side_done);
    hipFree(ctx->d_stamps);
    for (auto &im : ctx->storage_images) {      // {ptr, alt} hold both allocations of a double-buffered image
        if (!im.used) continue;
        hipFree(im.ptr);
        if (im.alt && im.alt != im.ptr) hipFree(im.alt);
    }
    for (auto &t : ctx->textures) hipFree(t.texels);
    hipFree(ctx->d_textures);
    free_scene(ctx);
    hipFree(ctx->d_ray_stats);
    hipFree(ctx->d_tile_counter);
    for (vhr_context::CostOrder *co : { &ctx->cost_order_raygen, &ctx->cost_order_reflection, &ctx->cost_order_raytraced })
        for (int i = 0; i < 2; ++i) { hipFree(co->cost[i]); hipFree(co->order[i]); }
    for (auto &t : ctx->kernel_timers)
        for (hipEvent_t e : t.events) hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *vhr_last_error(const vhr_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int vhr_get_current_stream(vhr_context *ctx, void **stream) {
    if (!ctx || !stream) return VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }      // the caller is about to enqueue behind it
    // ... and its own kernels may read the Reflections image or rewrite the G-buffer the mirror ray's pending launch reads: the stream waits for
    // that launch here ("reflection_async" 2 = the multi-GPU hooks, which only touch SVGF images and join where they must, keep the overlap)
    if (!ctx->host_only && ctx->options[vhr::kOptReflectionAsync] != 2) { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) return jrc; }
    *stream = static_cast<void *>(ctx->stream);       // inside a pass callback of vhr_graph_execute: the stream that pass is ordered on
    return VHR_OK;
}

int vhr_synchronize(vhr_context *ctx) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return VHR_OK;
    if (!ctx->recorded.empty()) { const int rc = vhr::flush_recorded(ctx); if (rc != VHR_OK) return rc; }    // called from inside a compute pass
    { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    return VHR_OK;
}

// inverseTranspose(mat3(transform)) -- hybrid_render_path.cpp:44 (used by the stand-in G-buffer only)
static void normal_matrix3(const float *m, float out[9]) {
    double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
    double c00 = e * i - f * h, c01 = -(d * i - f * g), c02 = d * h - e * g;
    double c10 = -(b * i - c * h), c11 = a * i - c * g, c12 = -(a * h - b * g);
    double c20 = b * f - c * e, c21 = -(a * f - c * d), c22 = a * e - b * d;
    double det = a * c00 + b * c01 + c * c02;
    double id = det != 0.0 ? 1.0 / det : 0.0;
    out[0] = float(c00 * id); out[3] = float(c01 * id); out[6] = float(c02 * id);
    out[1] = float(c10 * id); out[4] = float(c11 * id); out[7] = float(c12 * id);
    out[2] = float(c20 * id); out[5] = float(c21 * id); out[8] = float(c22 * id);
}

// the device-built tree on the host (for "bvh_host_checks" and the fingerprint)
static int fetch_device_tree(vhr_context *ctx, HostBvh &bvh) {
    bvh.nodes.resize(ctx->node_count); bvh.nodes_ch.resize(ctx->node_count); bvh.nodes48.resize(ctx->node_count); bvh.nodes16.resize(ctx->node_count);
    bvh.tris.resize(ctx->tri_count);
    HIP_TRY(ctx, hipMemcpy(bvh.nodes.data(), ctx->d_nodes, sizeof(BvhNode) * ctx->node_count, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(bvh.nodes_ch.data(), ctx->d_nodes_ch, sizeof(BvhNodeCH) * ctx->node_count, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(bvh.nodes48.data(), ctx->d_nodes48, sizeof(BvhNode48) * ctx->node_count, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(bvh.nodes16.data(), ctx->d_nodes16, sizeof(BvhNode16) * ctx->node_count, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(bvh.tris.data(), ctx->d_tris, sizeof(BvhTri) * ctx->tri_count, hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; ++a) bvh.centre[a] = ctx->bvh_centre[a];
    bvh.nodes16_valid = nodes16_in_range(bvh);
    bvh.max_depth = ctx->bvh_depth;
    return VHR_OK;
}

int vhr_update_geometry(vhr_context *ctx, const vhr_vertex *vertices, uint32_t vertex_count, const uint32_t *indices,
                        uint32_t index_count, const vhr_primitive *primitives, uint32_t primitive_count) {
    if (!ctx || (!vertices && vertex_count) || (!indices && index_count) || (!primitives && primitive_count))
        return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: null array") : VHR_ERROR_INVALID_ARGUMENT;
    // (a host-only context validates, builds the tree and checks its node forms, and stops before the upload: the builder is host code,
    // and the CPU tests exercise it this way)
    // Validate every offset the kernels will dereference (an out-of-range index would fault the GPU).
    uint64_t total_triangles = 0;
    for (uint32_t p = 0; p < primitive_count; ++p) {
        const vhr_primitive &pr = primitives[p];
        if (uint64_t(pr.index_offset) + pr.index_count > index_count)
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: primitive " + std::to_string(p) + " index range exceeds the index buffer");
        for (uint32_t k = 0; k < pr.index_count; ++k)
            if (uint64_t(pr.vertex_offset) + indices[pr.index_offset + k] >= vertex_count)
                return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: primitive " + std::to_string(p) + " references a vertex beyond the vertex buffer");
        const int32_t tex[2] = { pr.material.base_color_texture, pr.material.metallic_roughness_texture };
        for (int32_t t : tex)
            if (t < -1) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: negative texture index other than -1");
        // a NaN / Inf coordinate would reach the builder's bin index (float -> int of a NaN is undefined) and its comparators
        for (int k = 0; k < 16; ++k)
            if (!std::isfinite(pr.transform[k])) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: primitive " + std::to_string(p) + " has a non-finite transform");
        total_triangles += pr.index_count / 3;
    }
    for (uint32_t v = 0; v < vertex_count; ++v)
        if (!std::isfinite(vertices[v].pos[0]) || !std::isfinite(vertices[v].pos[1]) || !std::isfinite(vertices[v].pos[2]))
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: vertex " + std::to_string(v) + " has a non-finite position");
    if (total_triangles >= (1ull << 29))            // a leaf link packs (first triangle << 2 | count - 1) into 31 bits
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: 2^29 triangles or more");
    if (!ctx->host_only) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
        free_scene(ctx);
    }

    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        *dst = nullptr;
        if (bytes == 0) return hipSuccess;
        hipError_t e = hipMalloc(dst, bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    double upload_ms = 0.0;
    if (!ctx->host_only) {                 // the scene arrays first: the device builder reads them where the walkers will
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<float> nm(size_t(primitive_count) * 9);
        for (uint32_t p = 0; p < primitive_count; ++p) normal_matrix3(primitives[p].transform, &nm[size_t(p) * 9]);
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_vertices), vertices, sizeof(vhr_vertex) * vertex_count));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_indices), indices, sizeof(uint32_t) * index_count));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_primitives), primitives, sizeof(vhr_primitive) * primitive_count));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_normal_matrices), nm.data(), sizeof(float) * nm.size()));
        upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    HostBvh bvh;
    bool device_built = false;
    ctx->bvh_builder_used = 0;
    // ---- "bvh_builder" 1: the tree built on the device (csrc/kernels_bvh.hip), where the reference builds its BLAS / TLAS ----
    if (ctx->bvh_builder >= 1 && !ctx->host_only && total_triangles >= 2ull * uint64_t(ctx->bvh_leaf_tris)) {
        std::vector<uint32_t> prefix(primitive_count);
        uint32_t acc = 0;
        for (uint32_t p = 0; p < primitive_count; ++p) { prefix[p] = acc; acc += primitives[p].index_count / 3; }
        // (a primitive without triangles shares its prefix with the next one: the search below picks the LAST primitive whose prefix is
        // <= t, which is the one that owns triangle t, because an empty primitive's successor starts at the same value)
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = device_build_bvh(ctx, prefix, uint32_t(total_triangles), ctx->bvh_leaf_tris, ctx->bvh_presplit, ctx->bvh_frame_mode);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == VHR_OK) {
            device_built = true;
            ctx->bvh_builder_used = 1;
            ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            // its self-checks ran on the device (k0_check_forms_kernel); "bvh_host_checks" 1 fetches the tree and repeats them with the host's
            // code (tests hold the two equal).  The fingerprint is computed when somebody asks for it (vhr_get_bvh_fingerprint).
            ctx->bvh_fingerprint_valid = false;
            if (ctx->bvh_host_checks) {
                const int frc = fetch_device_tree(ctx, bvh);
                if (frc != VHR_OK) return frc;
            }
        } else {
            hipFree(ctx->d_nodes); hipFree(ctx->d_nodes16); hipFree(ctx->d_nodes_ch); hipFree(ctx->d_nodes48); hipFree(ctx->d_tris);
            ctx->d_nodes = nullptr; ctx->d_nodes16 = nullptr; ctx->d_nodes_ch = nullptr; ctx->d_nodes48 = nullptr; ctx->d_tris = nullptr;
            ctx->node_count = ctx->tri_count = 0;
            if (rc != VHR_ERROR_OUT_OF_SLOTS) return rc;          // (out of slots: deeper than the walkers' stacks, or a one-leaf scene -> the host builder)
        }
    }
    const auto t_build0 = std::chrono::steady_clock::now();
    if (!device_built) {
        build_bvh(vertices, indices, primitives, primitive_count, bvh, ctx->bvh_leaf_tris, ctx->bvh_build_threads, ctx->bvh_presplit, ctx->bvh_frame_mode);          // UpdateBLAS + UpdateTLAS
        ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
        ctx->bvh_presplit_level = bvh.presplit_level;
        for (int i = 0; i < 9; ++i) ctx->bvh_frame[i] = bvh.frame[i];
        ctx->bvh_frame_on = bvh.frame_on;
    }
    if (uint64_t(device_built ? ctx->node_count : bvh.nodes48.size()) * sizeof(BvhNode48) >= (1ull << 31))     // an inner link of the 48-byte nodes is a non-negative 32-bit byte offset
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdateGeometry: more than 44 million BVH nodes");
    const auto t_build1 = std::chrono::steady_clock::now();       // K0 proper ends here; the self-checks below are timed apart
    if (!device_built || ctx->bvh_host_checks) {
        check_node_forms(bvh, ctx->bvh_form_checks, ctx->bvh_build_threads);
        ctx->nodes16_valid = bvh.nodes16_valid && ctx->bvh_form_checks[3] == 0;      // else the walkers stay on the 48-byte nodes
        ctx->bvh_fingerprint = bvh_fingerprint(bvh);
        ctx->bvh_tree_fingerprint = bvh_tree_fingerprint(bvh);
        ctx->bvh_fingerprint_valid = true;
    }
    ctx->bvh_check_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build1).count();
    if (ctx->host_only) {
        ctx->node_count = uint32_t(bvh.nodes.size());
        ctx->tri_count = uint32_t(bvh.tris.size());
        ctx->bvh_depth = bvh.max_depth;
        ctx->geometry_upload_ms = 0.0;
        return VHR_OK;
    }
    if (!device_built) {
        const auto t_upload0 = std::chrono::steady_clock::now();
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_nodes), bvh.nodes.data(), sizeof(BvhNode) * bvh.nodes.size()));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_nodes16), bvh.nodes16.data(), sizeof(BvhNode16) * bvh.nodes16.size()));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_nodes_ch), bvh.nodes_ch.data(), sizeof(BvhNodeCH) * bvh.nodes_ch.size()));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_nodes48), bvh.nodes48.data(), sizeof(BvhNode48) * bvh.nodes48.size()));
        HIP_TRY(ctx, upload(reinterpret_cast<void **>(&ctx->d_tris), bvh.tris.data(), sizeof(BvhTri) * bvh.tris.size()));
        for (int a = 0; a < 3; ++a) ctx->bvh_centre[a] = bvh.centre[a];
        ctx->node_count = uint32_t(bvh.nodes.size());
        ctx->tri_count = uint32_t(bvh.tris.size());
        ctx->bvh_depth = bvh.max_depth;
        upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_upload0).count();
    }
    ctx->vertex_count = vertex_count;
    ctx->index_count = index_count;
    ctx->primitive_count = primitive_count;
    ctx->geometry_upload_ms = upload_ms;
    return VHR_OK;
}

int vhr_get_bvh_builder(vhr_context *ctx, int32_t *used) {
    if (!ctx || !used) return VHR_ERROR_INVALID_ARGUMENT;
    *used = ctx->bvh_builder_used;
    return VHR_OK;
}

int vhr_get_bvh_frame(vhr_context *ctx, float out[9]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    for (int i = 0; i < 9; ++i) out[i] = ctx->bvh_frame[i];
    return VHR_OK;
}

int vhr_get_bvh_presplit_level(vhr_context *ctx, int32_t *level) {
    if (!ctx || !level) return VHR_ERROR_INVALID_ARGUMENT;
    *level = ctx->bvh_presplit_level;
    return VHR_OK;
}

int vhr_get_build_times(vhr_context *ctx, double out[2]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    out[0] = ctx->bvh_build_ms;
    out[1] = ctx->geometry_upload_ms;
    return VHR_OK;
}

int32_t vhr_upload_texture_from_data(vhr_context *ctx, uint32_t width, uint32_t height, const uint8_t *data, int32_t format,
                                     const vhr_sampler_info *sampler_info) {
    // (-1 is the reference's "table exhausted" sentinel, resource_manager.cpp:847-848, and nothing else)
    if (!ctx || !data || !width || !height) return ctx ? ctx->fail(VHR_ERROR_UNSUPPORTED, "UploadTextureFromData: invalid arguments") : VHR_ERROR_UNSUPPORTED;
    if (format != VHR_FORMAT_R8G8B8A8_SRGB && format != VHR_FORMAT_R8G8B8A8_UNORM)
        return ctx->fail(VHR_ERROR_UNSUPPORTED, "UploadTextureFromData: format must be R8G8B8A8_SRGB or R8G8B8A8_UNORM");
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    if (ctx->textures.size() >= vhr_context::kMaxGlobalResources) { ctx->error = "texture table exhausted"; return -1; }   // resource_manager.cpp:847-848
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    vhr_context::Texture t{};
    const size_t bytes = size_t(width) * height * 4;
    HIP_TRY(ctx, hipMalloc(&t.texels, bytes));
    HIP_TRY(ctx, hipMemcpy(t.texels, data, bytes, hipMemcpyHostToDevice));
    t.w = width; t.h = height; t.format = format;
    // default sampler: LINEAR / REPEAT (resource_manager.cpp:58-69)
    t.sampler = sampler_info ? *sampler_info : vhr_sampler_info{ 1, 1, 0, 0 };
    ctx->textures.push_back(t);
    // rebuild the device-side descriptor table (set 0, binding 4: textures[])
    std::vector<DeviceTexture> table(ctx->textures.size());
    for (size_t i = 0; i < table.size(); ++i) {
        const auto &s = ctx->textures[i];
        table[i] = DeviceTexture{ static_cast<const uint8_t *>(s.texels), s.w, s.h, s.format, s.sampler.mag_filter, s.sampler.address_mode_u, s.sampler.address_mode_v, 0 };
    }
    { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    hipFree(ctx->d_textures);
    ctx->d_textures = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_textures), sizeof(DeviceTexture) * table.size()));
    HIP_TRY(ctx, hipMemcpy(ctx->d_textures, table.data(), sizeof(DeviceTexture) * table.size(), hipMemcpyHostToDevice));
    return int32_t(ctx->textures.size() - 1);
}

int32_t vhr_upload_new_storage_image(vhr_context *ctx, uint32_t width, uint32_t height, int32_t format) {
    // -1 is the reference's "pool exhausted" sentinel (resource_manager.cpp:876-877) and nothing else: every failure is < -1
    if (!ctx) return VHR_ERROR_UNSUPPORTED;
    const uint32_t bpp = format_stride(format);
    if (!bpp || !width || !height) return ctx->fail(VHR_ERROR_UNSUPPORTED, "UploadNewStorageImage: unsupported format or empty extent");
    if (!ctx->host_only) HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < vhr_context::kMaxGlobalResources; ++i) {         // first free slot, resource_manager.cpp:866-878
        Image &im = ctx->storage_images[i];
        if (im.used) continue;
        im = Image{};
        im.width = width; im.height = height; im.format = format; im.bpp = bpp;
        if (ctx->host_only) { im.used = true; return int32_t(i); }
        // svgf.comp reads neighbours of, and rewrites, the moments history in one dispatch (svgf.comp:72,140-144): an RG16F image
        // keeps a second buffer so that the dispatch reads a snapshot
        const bool twin = format == VHR_FORMAT_R16G16_SFLOAT;
        hipError_t e = hipMalloc(&im.owned, im.bytes());
        if (e == hipSuccess) e = hipMemsetAsync(im.owned, 0, im.bytes(), ctx->stream);
        if (e == hipSuccess && twin) e = hipMalloc(&im.alt, im.bytes());
        if (e == hipSuccess && twin) e = hipMemsetAsync(im.alt, 0, im.bytes(), ctx->stream);
        if (e != hipSuccess) {                        // nothing half-made stays behind
            hipFree(im.owned);
            hipFree(im.alt);
            im = Image{};
            return ctx->fail(VHR_ERROR_DEVICE, std::string("UploadNewStorageImage: ") + hipGetErrorString(e));
        }
        im.ptr = im.owned;
        im.used = true;
        return int32_t(i);
    }
    ctx->error = "storage image pool exhausted";
    return -1;
}

int vhr_destroy_storage_image(vhr_context *ctx, int32_t id) {
    if (!ctx || id < 0 || uint32_t(id) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[id].used)
        return ctx ? ctx->fail(VHR_ERROR_NOT_FOUND, "DestroyStorageImage: no such image") : VHR_ERROR_INVALID_ARGUMENT;   // assert at resource_manager.cpp:266
    if (ctx->host_only) { ctx->storage_images[id] = Image{}; return VHR_OK; }
    { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    Image &im = ctx->storage_images[id];
    // ptr/alt may have been flipped: free both distinct allocations
    void *a = im.ptr, *b = im.alt;
    hipFree(a);
    if (b && b != a) hipFree(b);
    im = Image{};
    return VHR_OK;
}

int vhr_update_per_frame_ubo(vhr_context *ctx, uint32_t resource_idx, const vhr_per_frame_data *pfd) {
    if (!ctx || !pfd || resource_idx >= 3) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "UpdatePerFrameUBO: invalid arguments") : VHR_ERROR_INVALID_ARGUMENT;
    std::memcpy(&ctx->per_frame[resource_idx], pfd, sizeof *pfd);            // resource_manager.cpp:362-364
    return VHR_OK;
}

int vhr_get_last_per_frame_ubo(vhr_context *ctx, vhr_per_frame_data *out) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    std::memcpy(out, &ctx->per_frame[ctx->last_resource_idx], sizeof *out);
    return VHR_OK;
}

int vhr_set_trace_params(vhr_context *ctx, const vhr_trace_params *p) {
    if (!ctx || !p) return VHR_ERROR_INVALID_ARGUMENT;
    if (p->ao_spp > 64) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "trace params: ao_spp > 64");
    ctx->trace_params = *p;
    return VHR_OK;
}

int vhr_set_tile(vhr_context *ctx, uint32_t col_begin, uint32_t col_end, uint32_t row_begin, uint32_t row_end, uint32_t overlap,
                 uint32_t halo_rows, uint32_t halo_cols) {
    if (!ctx || row_begin > row_end || row_end > ctx->height || col_begin > col_end || col_end > ctx->width || halo_rows < overlap || halo_cols < overlap)
        return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "set_tile: need begin <= end <= extent on both axes and halos >= overlap") : VHR_ERROR_INVALID_ARGUMENT;
    ctx->row_begin = row_begin;
    ctx->row_end = row_end;
    ctx->col_begin = col_begin;
    ctx->col_end = col_end;
    ctx->overlap = overlap;
    ctx->halo = halo_rows;
    ctx->halo_cols = halo_cols;
    return VHR_OK;
}

int vhr_set_strip(vhr_context *ctx, uint32_t row_begin, uint32_t row_end, uint32_t overlap, uint32_t halo) {
    if (!ctx || row_begin > row_end || row_end > ctx->height || halo < overlap)
        return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "set_strip: need row_begin <= row_end <= height and halo >= overlap") : VHR_ERROR_INVALID_ARGUMENT;
    return vhr_set_tile(ctx, 0, ctx->width, row_begin, row_end, overlap, halo, std::max(halo, ctx->width));      // every column: a strip
}

int vhr_set_option(vhr_context *ctx, const char *key, int32_t value) {
    if (!ctx || !key) return VHR_ERROR_INVALID_ARGUMENT;
    if (!std::strcmp(key, "bvh_leaf_triangles")) {          // applies to the next vhr_update_geometry
        if (value < 1 || value > kMaxLeafTris) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_leaf_triangles must be 1..4");
        ctx->bvh_leaf_tris = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_host_checks")) {              // applies to the next vhr_update_geometry
        if (value < 0 || value > 1) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_host_checks must be 0 or 1");
        ctx->bvh_host_checks = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_device_max_depth")) {         // applies to the next vhr_update_geometry
        if (value < 1 || value > kMaxBvhDepth) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_device_max_depth must be 1..40");
        ctx->bvh_device_max_depth = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_builder")) {                  // applies to the next vhr_update_geometry
        if (value < 0 || value > 1) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_builder must be 0 (binned SAH on the host) or 1 (binned SAH on the device)");
        ctx->bvh_builder = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_frame")) {                    // applies to the next vhr_update_geometry
        if (value < 0 || value > 1) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_frame must be 0 (boxes along the world axes) or 1 (along the frame that minimises the triangles' summed box area)");
        ctx->bvh_frame_mode = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_presplit")) {                 // applies to the next vhr_update_geometry
        if (value < 0 || value > 400) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_presplit must be 0 (off) or the budget of extra triangle references in percent (1..400)");
        ctx->bvh_presplit = value;
        return VHR_OK;
    }
    if (!std::strcmp(key, "bvh_build_threads")) {            // applies to the next vhr_update_geometry; the tree does not depend on it
        if (value < 0 || value > 64) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "bvh_build_threads must be 0..64");
        ctx->bvh_build_threads = value;
        return VHR_OK;
    }
    for (int i = 0; i < vhr::kOptCount; ++i) {
        const vhr::OptionInfo &o = vhr::kOptionInfo[i];
        if (std::strcmp(key, o.name)) continue;
        if (value < o.lo || value > o.hi)
            return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, std::string(o.name) + " must be " + std::to_string(o.lo) + ".." + std::to_string(o.hi));
        ctx->options[i] = value;
        return VHR_OK;
    }
    return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("unknown option '") + key + "'");
}

int32_t vhr_option_count(void) { return vhr::kOptCount; }

int vhr_option_info(int32_t index, const char **name, int32_t *default_value, int32_t *min_value, int32_t *max_value) {
    if (index < 0 || index >= vhr::kOptCount) return VHR_ERROR_INVALID_ARGUMENT;
    const vhr::OptionInfo &o = vhr::kOptionInfo[index];
    if (name) *name = o.name;
    if (default_value) *default_value = o.def;
    if (min_value) *min_value = o.lo;
    if (max_value) *max_value = o.hi;
    return VHR_OK;
}

int vhr_get_option(vhr_context *ctx, const char *key, int32_t *value) {
    if (!ctx || !key || !value) return VHR_ERROR_INVALID_ARGUMENT;
    for (int i = 0; i < vhr::kOptCount; ++i)
        if (!std::strcmp(key, vhr::kOptionInfo[i].name)) { *value = ctx->options[i]; return VHR_OK; }
    return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("unknown option '") + key + "'");
}

int vhr_set_ray_statistics(vhr_context *ctx, int32_t enable) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    ctx->ray_stats_enabled = enable != 0;
    return VHR_OK;
}

int vhr_get_ray_statistics(vhr_context *ctx, uint64_t out[4]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    const vhr_trace_params &tp = ctx->trace_params;
    const uint64_t covered = ctx->h_ray_stats.covered_pixels;
    if (ctx->raytraced_pixels) {          // raytraced render path: one primary ray per pixel + one shadow ray per primary hit
        out[0] = out[1] = ctx->raytraced_pixels + covered;
        out[2] = covered;
        out[3] = ctx->h_ray_stats.stack_overflows;
        return VHR_OK;
    }
    const uint64_t second = ctx->h_ray_stats.second_bounce_rays;       // two-bounce extension: one more ray per first-bounce hit
    out[0] = covered * (uint64_t(tp.shadow_enable ? 1 : 0) + tp.ao_spp + (tp.reflections ? 1 : 0)) + second;
    out[1] = covered * (uint64_t(tp.shadow_enable ? 4 : 0) + tp.ao_spp + (tp.reflections ? 1 : 0)) + second;   // raygen.rgen:38-40 duplicates
    out[2] = covered;
    out[3] = ctx->h_ray_stats.stack_overflows;
    return VHR_OK;
}

int vhr_get_traversal_statistics(vhr_context *ctx, uint64_t out[4]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    out[0] = ctx->h_ray_stats.node_visits; out[1] = ctx->h_ray_stats.leaf_visits;
    out[2] = ctx->h_ray_stats.triangle_tests; out[3] = ctx->h_ray_stats.wave_iterations;
    return VHR_OK;
}

int vhr_get_reflection_statistics(vhr_context *ctx, uint64_t out[10]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    const RayStats &r = ctx->h_refl_stats;
    out[0] = r.unique_rays; out[1] = r.second_bounce_rays; out[2] = r.node_visits; out[3] = r.leaf_visits; out[4] = r.triangle_tests;
    out[5] = r.wave_iterations; out[6] = r.refills; out[7] = r.waves; out[8] = r.cycles_total; out[9] = r.cycles_nodes;
    return VHR_OK;
}

int vhr_get_binary64_statistics(vhr_context *ctx, uint64_t out[4]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    out[0] = ctx->h_ray_stats.pending_rays; out[1] = ctx->h_refl_stats.pending_rays; out[2] = 0; out[3] = 0;
    return VHR_OK;
}

int vhr_get_traversal_cycles(vhr_context *ctx, uint64_t out[8]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    const RayStats &r = ctx->h_ray_stats;
    out[0] = r.cycles_total; out[1] = r.cycles_setup; out[2] = r.cycles_refill; out[3] = r.cycles_nodes;
    out[4] = r.cycles_leaves; out[5] = r.refills; out[6] = r.waves; out[7] = r.drain_iterations;
    return VHR_OK;
}

int vhr_get_drain_statistics(vhr_context *ctx, uint64_t out[4]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    const RayStats &r = ctx->h_ray_stats;
    out[0] = r.cut_entries; out[1] = r.drain_le4; out[2] = r.drain_le8; out[3] = r.drain_le16;
    return VHR_OK;
}

int vhr_calibration_stream_read(vhr_context *ctx, int32_t storage_image, uint32_t bytes_per_lane) {
    if (!ctx || ctx->host_only || storage_image < 0 || uint32_t(storage_image) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[storage_image].used)
        return ctx ? ctx->fail(VHR_ERROR_NOT_FOUND, "calibration: no such storage image") : VHR_ERROR_INVALID_ARGUMENT;
    return launch_calibration_read(ctx, ctx->storage_images[storage_image], bytes_per_lane, ctx->d_tile_counter);
}

int vhr_debug_ray_triangle(vhr_context *ctx, const float *pairs, uint32_t count, uint32_t *hit, float *tuv) {
    if (!ctx || (count && (!pairs || !hit || !tuv))) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_debug_ray_triangle: null argument") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_DEVICE, "vhr_debug_ray_triangle: host-only context");
    return launch_ray_triangle_pairs(ctx, pairs, count, hit, tuv);
}

int vhr_get_bvh_statistics(vhr_context *ctx, uint64_t out[5]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    out[0] = ctx->node_count; out[1] = ctx->tri_count; out[2] = ctx->bvh_depth;
    out[3] = uint64_t(ctx->node_count) * sizeof(BvhNode); out[4] = uint64_t(ctx->tri_count) * sizeof(BvhTri);
    return VHR_OK;
}

const char *vhr_source_fingerprint(void) { return VHR_SOURCE_FINGERPRINT; }

/* Diagnostics: the waves' lifetimes (s_memtime ticks) the last ray-tracing launch left for "raygen_cost_order" (index = tile pair * waves + wave). */
int vhr_debug_wave_lifetimes(vhr_context *ctx, uint32_t *out, uint32_t capacity, uint32_t *count) {
    if (!ctx || !out || !count) return VHR_ERROR_INVALID_ARGUMENT;
    const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_;
    const vhr_context::CostOrder &co = ctx->cost_order_raygen;
    // (the words the launch wrote: its blocks x the waves per block it ran with -- "raygen_waves_per_block" 3 runs as 2 -- and never more than the buffer holds)
    const uint32_t n = std::min<uint32_t>(capacity, co.cost_blocks[co.slot] ? std::min(co.cost_waves[co.slot], co.capacity) : 0u);
    *count = n;
    if (n && hipMemcpy(out, co.cost[co.slot], size_t(n) * 4, hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "wave lifetimes: copy failed");
    return VHR_OK;
}

// What the last frame's rays cost, where: the wave lifetimes the queue kernels leave for "raygen_cost_order" -- the any-hit launch's and the mirror ray's --
// summed into a map of 8 x 8-pixel cells (a tile of fewer rows is charged to the cell of its first row).  The input of vhr_tile_plan_make_weighted.
int vhr_get_tile_cost_map(vhr_context *ctx, uint32_t *out, uint32_t cols, uint32_t rows) {
    if (!ctx || !out || cols == 0 || rows == 0) return VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    if (uint64_t(cols) * 8 < ctx->width || uint64_t(rows) * 8 < ctx->height) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "tile cost map: the map does not cover the image");
    const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_;
    std::fill(out, out + size_t(cols) * rows, 0u);
    std::vector<uint32_t> host;
    int found = 0;
    for (const vhr_context::CostOrder *co : { &ctx->cost_order_raygen, &ctx->cost_order_reflection }) {
        const uint32_t slot = co->slot;
        const vhr_context::CostOrder::Shape &sh = co->shape[slot];
        const uint32_t n = co->cost_blocks[slot] ? std::min(co->cost_waves[slot], co->capacity) : 0u;
        if (!n || !sh.tiles_x) continue;
        host.resize(n);
        if (hipMemcpy(host.data(), co->cost[slot], size_t(n) * 4, hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "tile cost map: copy failed");
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t block = i / sh.wv, wave = i % sh.wv;
            const uint32_t by = block / sh.blocks_x, bx = block % sh.blocks_x;
            const uint32_t x = sh.col_begin + (bx * sh.wv + wave) * sh.tile_w, y = sh.row_begin + by * sh.tile_h;
            if (x >= ctx->width || y >= ctx->height) continue;
            uint32_t &cell = out[size_t(y / 8) * cols + x / 8];
            cell = uint32_t(std::min<uint64_t>(0xffffffffull, uint64_t(cell) + host[i]));
        }
        ++found;
    }
    if (!found) return ctx->fail(VHR_ERROR_NOT_FOUND, "tile cost map: no queue-kernel launch has left its wave lifetimes (\"raygen_cost_order\" 0, or a launch below 2 048 workgroups with the option at 1)");
    return VHR_OK;
}

static int fingerprints_of_device_tree(vhr_context *ctx) {      // a device-built tree: fetched when somebody asks, hashed like the host's
    if (ctx->bvh_fingerprint_valid || ctx->host_only || !ctx->d_nodes || !ctx->node_count) return VHR_OK;
    HostBvh bvh;
    const int frc = fetch_device_tree(ctx, bvh);
    if (frc != VHR_OK) return frc;
    ctx->bvh_fingerprint = bvh_fingerprint(bvh);
    ctx->bvh_tree_fingerprint = bvh_tree_fingerprint(bvh);
    ctx->bvh_fingerprint_valid = true;
    return VHR_OK;
}

int vhr_get_bvh_fingerprint(vhr_context *ctx, uint64_t *out) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    const int rc = fingerprints_of_device_tree(ctx);
    if (rc != VHR_OK) return rc;
    *out = ctx->bvh_fingerprint;
    return VHR_OK;
}

int vhr_get_bvh_tree_fingerprint(vhr_context *ctx, uint64_t *out) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    const int rc = fingerprints_of_device_tree(ctx);
    if (rc != VHR_OK) return rc;
    *out = ctx->bvh_tree_fingerprint;
    return VHR_OK;
}


int vhr_get_bvh_form_checks(vhr_context *ctx, uint64_t out[4]) {
    if (!ctx || !out) return VHR_ERROR_INVALID_ARGUMENT;
    for (int i = 0; i < 4; ++i) out[i] = ctx->bvh_form_checks[i];
    return VHR_OK;
}

// VulkanContext::Resize (vulkan_context.cpp:118-120 -> InitSwapchain) as far as it concerns this library: a new display extent.  What depends on
// the extent goes -- the graph with its transient images and pass registry (RenderPath::Build destroys it first anyway, render_path.cpp:14-20), every image
// of the storage pool (the path's SVGF history: the reference's RegisterPath allocates its five again on Build) -- and what does not stays:
// geometry, acceleration structure, textures, options, trace parameters, timers.  The screen tile is the whole image again.
int vhr_resize(vhr_context *ctx, uint32_t width, uint32_t height) {
    if (!ctx) return VHR_ERROR_INVALID_ARGUMENT;
    if (width == 0 || height == 0) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_resize: zero extent");
    if (ctx->recording || ctx->cur_pass) return ctx->fail(VHR_ERROR_GRAPH, "vhr_resize: called from inside a pass");
    if (!ctx->host_only) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }
        { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    }
    const int grc = vhr_graph_destroy_resources(ctx);
    if (grc != VHR_OK) return grc;
    for (Image &im : ctx->storage_images) {
        if (!im.used) continue;
        if (!ctx->host_only) {
            void *a = im.ptr, *b = im.alt;
            hipFree(a);
            if (b && b != a) hipFree(b);
        }
        im = Image{};
    }
    ctx->width = width;
    ctx->height = height;
    ctx->row_begin = 0; ctx->row_end = height;
    ctx->col_begin = 0; ctx->col_end = width;
    ctx->overlap = 0; ctx->halo = 0; ctx->halo_cols = 0;
    ctx->last_resource_idx = 0;
    ctx->error.clear();
    return VHR_OK;
}

int vhr_get_display_size(vhr_context *ctx, uint32_t *width, uint32_t *height) {
    if (!ctx || !width || !height) return VHR_ERROR_INVALID_ARGUMENT;
    *width = ctx->width;
    *height = ctx->height;
    return VHR_OK;
}

static int image_info(vhr_context *ctx, const Image &im, vhr_image_info *out) {
    out->device_ptr = im.ptr; out->width = im.width; out->height = im.height; out->format = im.format; out->bytes_per_pixel = im.bpp;
    (void)ctx;
    return VHR_OK;
}

int vhr_get_transient_image(vhr_context *ctx, const char *name, vhr_image_info *out) {
    if (!ctx || !name || !out) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->images.find(name);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + name + "'");
    // (the caller is about to use the pointer on the context's stream: the mirror ray's pending launch writes this image)
    if (!ctx->host_only && it->second.ptr == ctx->refl_writes) { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) return jrc; }
    return image_info(ctx, it->second, out);
}

int vhr_get_storage_image(vhr_context *ctx, int32_t id, vhr_image_info *out) {
    if (!ctx || !out || id < 0 || uint32_t(id) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[id].used)
        return ctx ? ctx->fail(VHR_ERROR_NOT_FOUND, "no such storage image") : VHR_ERROR_INVALID_ARGUMENT;
    if (!ctx->host_only) { const int jrc = ctx->join_side(); if (jrc != VHR_OK) return jrc; }      // the caller is about to use the pointer on the context's stream
    return image_info(ctx, ctx->storage_images[id], out);
}

static int copy_image(vhr_context *ctx, const Image &im, void *host, uint64_t bytes, bool to_device, bool storage = false) {
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    if (!host || bytes != im.bytes()) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "image copy: byte count does not match the image (" + std::to_string(im.bytes()) + ")");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->recorded.empty()) { const int rc = vhr::flush_recorded(ctx); if (rc != VHR_OK) return rc; }    // called from inside a compute pass
    if (ctx->deferred_raygen) { const int drc = vhr::flush_deferred_raygen(ctx, nullptr); if (drc != VHR_OK) return drc; }
    if (to_device || im.ptr == ctx->refl_writes) { const int jr2 = ctx->join_refl(); if (jr2 != VHR_OK) return jr2; }      // (an upload may rewrite what the mirror ray reads)
    if (storage || to_device) { const int jrc = ctx->join_side(); if (jrc != VHR_OK) return jrc; }      // (a side-stream dispatch reads storage images and the pass's published copies)
    if (to_device) HIP_TRY(ctx, hipMemcpyAsync(im.ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    else HIP_TRY(ctx, hipMemcpyAsync(host, im.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    { const int src_ = ctx->sync_streams(); if (src_ != VHR_OK) return src_; }
    return VHR_OK;
}

int vhr_upload_transient_image(vhr_context *ctx, const char *name, const void *host_data, uint64_t bytes) {
    if (!ctx || !name) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->images.find(name);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + name + "'");
    return copy_image(ctx, it->second, const_cast<void *>(host_data), bytes, true);
}
int vhr_download_transient_image(vhr_context *ctx, const char *name, void *host_data, uint64_t bytes) {
    if (!ctx || !name) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = ctx->images.find(name);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + name + "'");
    return copy_image(ctx, it->second, host_data, bytes, false);
}
int vhr_upload_storage_image(vhr_context *ctx, int32_t id, const void *host_data, uint64_t bytes) {
    if (!ctx || id < 0 || uint32_t(id) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[id].used)
        return ctx ? ctx->fail(VHR_ERROR_NOT_FOUND, "no such storage image") : VHR_ERROR_INVALID_ARGUMENT;
    return copy_image(ctx, ctx->storage_images[id], const_cast<void *>(host_data), bytes, true, true);
}
int vhr_download_storage_image(vhr_context *ctx, int32_t id, void *host_data, uint64_t bytes) {
    if (!ctx || id < 0 || uint32_t(id) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[id].used)
        return ctx ? ctx->fail(VHR_ERROR_NOT_FOUND, "no such storage image") : VHR_ERROR_INVALID_ARGUMENT;
    return copy_image(ctx, ctx->storage_images[id], host_data, bytes, false, true);
}

int vhr_standin_gbuffer(vhr_context *ctx, uint32_t resource_idx, const char *normals_image, const char *motion_image, const char *depth_image) {
    return vhr_standin_gbuffer_with_albedo(ctx, resource_idx, nullptr, normals_image, motion_image, depth_image);
}

int vhr_standin_composition(vhr_context *ctx, uint32_t resource_idx, const vhr_composition_desc *d) {
    if (!ctx || !d || resource_idx >= 3 || !d->albedo_image || !d->normals_image || !d->motion_image || !d->depth_image || !d->shadow_ao_image)
        return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: missing image name") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    auto find = [&](const char *n) -> Image * { auto it = ctx->images.find(n); return it == ctx->images.end() ? nullptr : &it->second; };
    Image *al = find(d->albedo_image), *no = find(d->normals_image), *mo = find(d->motion_image), *de = find(d->depth_image), *sa = find(d->shadow_ao_image);
    Image *re = d->reflections_image ? find(d->reflections_image) : nullptr;
    Image *ss = d->ssao_image ? find(d->ssao_image) : nullptr;
    Image *sm = d->shadow_map_image ? find(d->shadow_map_image) : nullptr;
    if (!al || !no || !mo || !de || !sa || (d->reflections_image && !re) || (d->ssao_image && !ss) || (d->shadow_map_image && !sm))
        return ctx->fail(VHR_ERROR_NOT_FOUND, "composition: unknown transient image");
    if ((d->reflection_mode == 0 || d->reflection_mode == 1) && !re) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: reflection_mode 0 / 1 needs a reflections image");
    if (d->output_storage_image < 0 || uint32_t(d->output_storage_image) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[d->output_storage_image].used)
        return ctx->fail(VHR_ERROR_NOT_FOUND, "composition: output storage image is not allocated");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_composition(ctx, ctx->per_frame[resource_idx], *d, *al, *no, *mo, *de, *sa, re, ss, sm, ctx->storage_images[d->output_storage_image]);
}

int vhr_standin_shadow_map(vhr_context *ctx, uint32_t resource_idx, const char *shadow_map_image) {
    if (!ctx || !shadow_map_image || resource_idx >= 3) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_shadow_map: bad argument") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    auto it = ctx->images.find(shadow_map_image);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, "standin_shadow_map: unknown transient image");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_standin_shadow_map(ctx, ctx->per_frame[resource_idx], it->second);
}

int vhr_standin_raytraced_composition(vhr_context *ctx, const char *raytraced_output_image, int32_t output_storage_image) {
    if (!ctx || !raytraced_output_image) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "raytraced composition: missing image name") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    auto it = ctx->images.find(raytraced_output_image);
    if (it == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, "raytraced composition: unknown transient image");
    if (it->second.format != VHR_FORMAT_B8G8R8A8_UNORM) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "raytraced composition: RaytracedOutput must be B8G8R8A8_UNORM (raytraced_render_path.cpp:15)");
    if (output_storage_image < 0 || uint32_t(output_storage_image) >= vhr_context::kMaxGlobalResources || !ctx->storage_images[output_storage_image].used)
        return ctx->fail(VHR_ERROR_NOT_FOUND, "raytraced composition: output storage image is not allocated");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_raytraced_composition(ctx, it->second, ctx->storage_images[output_storage_image]);
}

int vhr_standin_gbuffer_with_albedo(vhr_context *ctx, uint32_t resource_idx, const char *albedo_image, const char *normals_image,
                                    const char *motion_image, const char *depth_image) {
    if (!ctx || !normals_image || !motion_image || !depth_image || resource_idx >= 3) return VHR_ERROR_INVALID_ARGUMENT;
    auto n = ctx->images.find(normals_image), m = ctx->images.find(motion_image), d = ctx->images.find(depth_image);
    if (n == ctx->images.end() || m == ctx->images.end() || d == ctx->images.end())
        return ctx->fail(VHR_ERROR_NOT_FOUND, "standin_gbuffer: unknown transient image");
    if (n->second.format != VHR_FORMAT_R16G16B16A16_SFLOAT || m->second.format != VHR_FORMAT_R16G16B16A16_SFLOAT || d->second.format != VHR_FORMAT_D32_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_gbuffer: formats must be RGBA16F, RGBA16F, D32F (hybrid_render_path.cpp:16-19)");
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    Image *albedo = nullptr;
    if (albedo_image) {
        auto al = ctx->images.find(albedo_image);
        if (al == ctx->images.end()) return ctx->fail(VHR_ERROR_NOT_FOUND, "standin_gbuffer: unknown albedo image");
        albedo = &al->second;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_standin_gbuffer(ctx, ctx->per_frame[resource_idx], n->second, m->second, d->second, albedo);
}

}  // extern "C"
