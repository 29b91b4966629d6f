// C1 / C2 of SURVEY.md section 2 inside the library: the neighbour halo exchanges and the strip gather of the row-strip
// decomposition (section 8e) as RCCL point-to-point calls -- ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd, one
// process per GPU -- so that a C++ integrator has the multi-GPU path without Python (vulkanhybridrenderer_amd/tiling.py issues the
// same exchanges through torch.distributed for bench.py).  The reference has no counterpart (single GPU, one queue,
// renderer.cpp:135); what is followed is the schedule of its SVGF pass (hybrid_render_path.cpp:288-329), from which the overlap
// and halo sizes derive.
//
// The row arithmetic lives HERE once (vhr_strip_plan_make / _exchanges / vhr_atrous_output_extent) and tiling.py is checked
// against it (tests/test_comm_plan.py), so the two hosts cannot diverge.
//
// RCCL is loaded on first use (dlopen): a single-GPU user of libvhr_amd.so has no dependency on it, and a process that already
// holds an RCCL (torch) shares that copy.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>

#include <rccl/rccl.h>

#include "vhr_internal.hpp"

using namespace vhr;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    // A copy the process already holds first (the loader knows libraries by soname, librccl.so.1: a PyTorch process has its
    // own build loaded), else the ROCm installation's -- with local scope, so that this library's choice never rebinds
    // anybody else's ncclXxx references.
    for (int flags : { RTLD_NOW | RTLD_NOLOAD, RTLD_NOW | RTLD_LOCAL }) {
        for (const char *name : { "librccl.so.1", "librccl.so" }) {
            r.handle = dlopen(name, flags);
            if (r.handle) break;
        }
        if (r.handle) break;
    }
    if (!r.handle) { r.error = std::string("RCCL not found: ") + dlerror(); return r; }
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}

}  // namespace

struct vhr_comm {
    vhr_context *ctx = nullptr;
    vhr_strip_plan plan = {};
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;          // the exchanges' own stream: they run beside the next frame's ray tracing
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false;
    std::string error;
    int fail(int code, const std::string &msg) { error = msg; if (ctx) ctx->error = msg; return code; }
};

#define NCCL_TRY(c, expr)                                                                                  \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return (c)->fail(VHR_ERROR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)
#define HIPC_TRY(c, expr)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return (c)->fail(VHR_ERROR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

// ---- the planner (tiling.atrous_overlap / atrous_output_extent / strip_bounds / make_plan / StripPlan.exchanges) ----
uint32_t vhr_atrous_overlap(uint32_t atrous_steps) {
    // the published image is iteration n-2's output (hybrid_render_path.cpp:322-325); iteration i reads +-2*2^i rows
    // (svgf_atrous_filter.comp:72-75): sum_{i=0}^{n-2} 2*2^i = 2*(2^(n-1) - 1)
    return atrous_steps < 2 ? 0u : 2u * ((1u << (atrous_steps - 1)) - 1u);
}

uint32_t vhr_atrous_output_extent(uint32_t overlap, uint32_t step) {
    const uint32_t used = 4u * step - 2u;          // rows of validity the iterations up to this one have consumed
    return overlap > used ? overlap - used : 0u;
}

int vhr_strip_plan_make(uint32_t height, uint32_t world, uint32_t rank, uint32_t max_motion_rows, uint32_t atrous_steps, vhr_strip_plan *out) {
    if (!out || world == 0 || rank >= world || height == 0) return VHR_ERROR_INVALID_ARGUMENT;
    auto bounds = [&](uint32_t r, uint32_t &a, uint32_t &b) { a = uint32_t(uint64_t(r) * height / world); b = uint32_t(uint64_t(r + 1) * height / world); };
    vhr_strip_plan p = {};
    p.rank = rank; p.world = world; p.height = height;
    bounds(rank, p.row_begin, p.row_end);
    if (world > 1) {
        p.overlap = vhr_atrous_overlap(atrous_steps);
        p.halo = p.overlap + max_motion_rows + 2u;       // svgf.comp reads the reprojected position +-1 (svgf.comp:52-60,81-84)
        uint32_t smallest = height;
        for (uint32_t r = 0; r < world; ++r) { uint32_t a, b; bounds(r, a, b); smallest = std::min(smallest, b - a); }
        if (p.halo > smallest) return VHR_ERROR_OUT_OF_SLOTS;      // strips thinner than the history halo: use fewer GPUs
    }
    *out = p;
    return VHR_OK;
}

int vhr_strip_plan_exchanges(const vhr_strip_plan *p, uint32_t n_rows, vhr_row_exchange out[2]) {
    if (!p || !out) return VHR_ERROR_INVALID_ARGUMENT;
    int n = 0;
    if (p->world <= 1 || n_rows == 0) return 0;
    if (p->rank > 0)
        out[n++] = vhr_row_exchange{ int32_t(p->rank) - 1, p->row_begin, std::min(p->row_end, p->row_begin + n_rows),
                                     p->row_begin > n_rows ? p->row_begin - n_rows : 0u, p->row_begin };
    if (p->rank + 1 < p->world)
        out[n++] = vhr_row_exchange{ int32_t(p->rank) + 1, std::max(p->row_begin, p->row_end > n_rows ? p->row_end - n_rows : 0u), p->row_end,
                                     p->row_end, std::min(p->height, p->row_end + n_rows) };
    return n;
}

// ---- the communicator ----
int vhr_comm_get_unique_id(uint8_t out[VHR_COMM_UNIQUE_ID_BYTES]) {
    static_assert(VHR_COMM_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!out) return VHR_ERROR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    if (!r.error.empty()) return VHR_ERROR_NO_DEVICE;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return VHR_ERROR_DEVICE;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return VHR_OK;
}

int vhr_comm_create(vhr_context *ctx, const vhr_strip_plan *plan, const uint8_t unique_id[VHR_COMM_UNIQUE_ID_BYTES], vhr_comm **out) {
    if (!ctx || !plan || !unique_id || !out || plan->rank >= plan->world) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: invalid arguments") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    if (plan->height != ctx->height) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: the plan's height is not the context's");
    Rccl &r = rccl();
    if (!r.error.empty()) return ctx->fail(VHR_ERROR_NO_DEVICE, r.error);
    vhr_comm *c = new vhr_comm;
    c->ctx = ctx;
    c->plan = *plan;
    auto bail = [&](int code, const std::string &msg) { ctx->error = msg; vhr_comm_destroy(c); return code; };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(VHR_ERROR_DEVICE, "vhr_comm_create: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t rc = r.CommInitRank(&c->nccl, int(plan->world), id, int(plan->rank));
    if (rc != ncclSuccess) { c->nccl = nullptr; return bail(VHR_ERROR_DEVICE, std::string("ncclCommInitRank: ") + r.GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess)
        return bail(VHR_ERROR_DEVICE, "vhr_comm_create: stream / event creation failed");
    // the context computes its strip from now on: owned rows + overlap recomputed, blits extended by the halo
    const int src = vhr_set_strip(ctx, plan->row_begin, plan->row_end, plan->overlap, plan->halo);
    if (src != VHR_OK) { const std::string msg = ctx->error; return bail(src, msg); }
    *out = c;
    return VHR_OK;
}

void vhr_comm_destroy(vhr_comm *c) {
    if (!c) return;
    if (c->ctx && !c->ctx->host_only) hipSetDevice(c->ctx->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->nccl) rccl().CommDestroy(c->nccl);
    if (c->ready) hipEventDestroy(c->ready);
    if (c->done) hipEventDestroy(c->done);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *vhr_comm_last_error(const vhr_comm *c) { return c ? c->error.c_str() : "null communicator"; }

// One grouped batch: for each neighbour, `n_rows` rows of every image, in place (I send rows I own next to the shared boundary and
// receive the rows the peer owns next to it).  Enqueued on `stream`.
static int enqueue_row_exchange(vhr_comm *c, Image *const *images, int n_images, uint32_t n_rows, hipStream_t stream) {
    vhr_row_exchange ex[2];
    const int n = vhr_strip_plan_exchanges(&c->plan, n_rows, ex);
    if (n <= 0) return VHR_OK;
    Rccl &r = rccl();
    NCCL_TRY(c, r.GroupStart());
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n_images; ++i) {
            const Image &im = *images[i];
            const size_t row = size_t(im.width) * im.bpp;
            char *base = static_cast<char *>(im.ptr);
            NCCL_TRY(c, r.Send(base + row * ex[k].send_begin, row * (ex[k].send_end - ex[k].send_begin), ncclUint8, ex[k].peer, c->nccl, stream));
            NCCL_TRY(c, r.Recv(base + row * ex[k].recv_begin, row * (ex[k].recv_end - ex[k].recv_begin), ncclUint8, ex[k].peer, c->nccl, stream));
        }
    NCCL_TRY(c, r.GroupEnd());
    return VHR_OK;
}

// Exchange #1 (only with "trace_overlap" off): the overlap rows of the raw shadow / AO image from the neighbours, in the
// context's stream order -- svgf.comp, enqueued next, reads them.  Call it from the Raytrace Pass's epilogue.
int vhr_comm_exchange_raytraced(vhr_comm *c, const char *raytraced_image) {
    if (!c || !raytraced_image) return VHR_ERROR_INVALID_ARGUMENT;
    auto it = c->ctx->images.find(raytraced_image);
    if (it == c->ctx->images.end()) return c->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + raytraced_image + "'");
    Image *im = &it->second;
    return enqueue_row_exchange(c, &im, 1, c->plan.overlap, c->ctx->stream);
}

// After the SVGF pass (its epilogue): exchange #2 -- `halo` rows of the temporal history and of the moments history just written,
// for the NEXT frame's svgf.comp -- and, if `denoised_image` is given, the gather (C2) of every rank's owned rows of it into
// `gathered_frame` on `root` (a device buffer of the whole image there, ignored elsewhere).  Both run on the communicator's own
// stream behind what the context has enqueued so far, i.e. beside the next frame's ray tracing; nothing waits for them here.
int vhr_comm_start_frame_exchanges(vhr_comm *c, int32_t history_storage_image, int32_t moments_storage_image, const char *denoised_image,
                                   int32_t root, void *gathered_frame) {
    if (!c) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = c->ctx;
    auto storage = [&](int32_t id) -> Image * {
        return (id >= 0 && uint32_t(id) < vhr_context::kMaxGlobalResources && ctx->storage_images[id].used) ? &ctx->storage_images[id] : nullptr;
    };
    Image *imgs[2] = { storage(history_storage_image), storage(moments_storage_image) };
    if (!imgs[0] || !imgs[1]) return c->fail(VHR_ERROR_NOT_FOUND, "vhr_comm_start_frame_exchanges: no such storage image");
    if (c->pending) return c->fail(VHR_ERROR_GRAPH, "vhr_comm_start_frame_exchanges: the previous frame's exchanges were not finished");
    HIPC_TRY(c, hipSetDevice(ctx->device));
    HIPC_TRY(c, hipEventRecord(c->ready, ctx->stream));
    HIPC_TRY(c, hipStreamWaitEvent(c->stream, c->ready, 0));
    if (denoised_image && c->plan.world > 1) {
        auto it = ctx->images.find(denoised_image);
        if (it == ctx->images.end()) return c->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + denoised_image + "'");
        if (root < 0 || uint32_t(root) >= c->plan.world) return c->fail(VHR_ERROR_INVALID_ARGUMENT, "gather root out of range");
        const Image &im = it->second;
        const size_t row = size_t(im.width) * im.bpp;
        Rccl &r = rccl();
        if (int32_t(c->plan.rank) == root) {
            if (!gathered_frame) return c->fail(VHR_ERROR_INVALID_ARGUMENT, "the gather's root needs a destination buffer");
            char *full = static_cast<char *>(gathered_frame);
            NCCL_TRY(c, r.GroupStart());
            for (uint32_t peer = 0; peer < c->plan.world; ++peer) {
                if (int32_t(peer) == root) continue;
                const uint32_t a = uint32_t(uint64_t(peer) * c->plan.height / c->plan.world), b = uint32_t(uint64_t(peer + 1) * c->plan.height / c->plan.world);
                NCCL_TRY(c, r.Recv(full + row * a, row * (b - a), ncclUint8, int(peer), c->nccl, c->stream));
            }
            NCCL_TRY(c, r.GroupEnd());
            HIPC_TRY(c, hipMemcpyAsync(full + row * c->plan.row_begin, static_cast<const char *>(im.ptr) + row * c->plan.row_begin,
                                       row * (c->plan.row_end - c->plan.row_begin), hipMemcpyDeviceToDevice, c->stream));
        } else {
            NCCL_TRY(c, r.GroupStart());
            NCCL_TRY(c, r.Send(static_cast<const char *>(im.ptr) + row * c->plan.row_begin, row * (c->plan.row_end - c->plan.row_begin), ncclUint8, root, c->nccl, c->stream));
            NCCL_TRY(c, r.GroupEnd());
        }
    }
    const int rc = enqueue_row_exchange(c, imgs, 2, c->plan.halo, c->stream);
    if (rc != VHR_OK) return rc;
    HIPC_TRY(c, hipEventRecord(c->done, c->stream));
    c->pending = true;
    return VHR_OK;
}

// Before the next frame's svgf.comp (the Raytrace Pass's epilogue): the context's stream waits for the exchanges started after the
// previous frame's SVGF pass.  No host synchronisation.
int vhr_comm_finish_frame_exchanges(vhr_comm *c) {
    if (!c) return VHR_ERROR_INVALID_ARGUMENT;
    if (!c->pending) return VHR_OK;
    HIPC_TRY(c, hipSetDevice(c->ctx->device));
    HIPC_TRY(c, hipStreamWaitEvent(c->ctx->stream, c->done, 0));
    c->pending = false;
    return VHR_OK;
}

}  // extern "C"
