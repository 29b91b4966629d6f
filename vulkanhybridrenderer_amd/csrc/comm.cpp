// C1 / C2 of SURVEY.md section 2 inside the library: the halo exchanges and the gather of the screen-space decomposition
// (section 8e) as RCCL point-to-point calls -- ncclSend / ncclRecv inside ONE ncclGroupStart / ncclGroupEnd per frame, one process
// per GPU -- so that a C++ integrator has the multi-GPU path without Python (vulkanhybridrenderer_amd/tiling.py issues the same
// exchanges through torch.distributed).  The reference has no counterpart (single GPU, one queue, renderer.cpp:135); what is
// followed is the schedule of its SVGF pass (hybrid_render_path.cpp:288-329), from which the overlap and halo sizes derive.
//
// The decomposition is a grid of grid_rows x grid_cols screen tiles (BASELINE.json: "the framebuffer shards by screen tile"); row
// strips are the grid_cols == 1 case and keep their own entry points.  The rectangle arithmetic lives HERE once
// (vhr_tile_plan_make / _exchanges, vhr_strip_plan_*, vhr_atrous_output_extent) and tiling.py is checked against it
// (tests/test_comm_plan.py), so the two hosts cannot diverge.
//
// RCCL is loaded on first use (dlopen): a single-GPU user of libvhr_amd.so has no dependency on it -- neither at run time nor at
// build time (the few types used are declared below, rccl.h is not included) -- and a process that already holds an RCCL (torch)
// shares that copy.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <vector>

#include "vhr_internal.hpp"

using namespace vhr;

namespace {

// What this file needs of <rccl/rccl.h> (NCCL 2.x ABI, unchanged since 2.0): opaque communicator, 128-byte unique id, result and
// data-type enumerators.
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
typedef int ncclDataType_t;
constexpr ncclDataType_t ncclUint8 = 1;
static_assert(VHR_COMM_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

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
    std::string error, path;
};

std::string &forced_library() { static std::string path; return path; }
std::atomic<bool> g_rccl_resolved{ false };

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy the process already holds first (the loader knows libraries by soname, librccl.so.1: a PyTorch process has its
        // own build loaded), else the ROCm installation's -- with local scope, so that this library's choice never rebinds
        // anybody else's ncclXxx references.
        // vhr_comm_use_library(path), called before this: that library and no other (a site's own RCCL build; tests/rccl_shim's stand-in, which lets the
        // exchanges of N ranks run on ONE GPU, where RCCL itself refuses two ranks per device).  A path that does not load is an error, not a fall-through.
        // (Rounds 4-5 read the path from an environment variable: code loading steered from outside the process's own calls -- gone.)
        if (!forced_library().empty()) {
            r.handle = dlopen(forced_library().c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) { const char *e = dlerror(); r.error = std::string("vhr_comm_use_library: ") + (e ? e : forced_library().c_str()); return; }
        }
        for (int flags : { RTLD_NOW | RTLD_NOLOAD, RTLD_NOW | RTLD_LOCAL }) {
            if (r.handle) break;
            for (const char *name : { "librccl.so.1", "librccl.so" }) {
                r.handle = dlopen(name, flags);
                if (r.handle) break;
            }
            if (r.handle) break;
        }
        if (!r.handle) { const char *e = dlerror(); r.error = std::string("RCCL not found: ") + (e ? e : "librccl.so"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        // which file the entry points came from (what vhr_comm_library reports)
        Dl_info info;
        if (r.GetUniqueId && dladdr(reinterpret_cast<void *>(r.GetUniqueId), &info) && info.dli_fname) r.path = info.dli_fname;
    });
    g_rccl_resolved = true;
    return r;
}

// owned rectangle of tile (tr, tc) of a grid
void tile_bounds(uint32_t width, uint32_t height, uint32_t grid_rows, uint32_t grid_cols, uint32_t tr, uint32_t tc, vhr_rect &out) {
    out.x0 = uint32_t(uint64_t(tc) * width / grid_cols);
    out.x1 = uint32_t(uint64_t(tc + 1) * width / grid_cols);
    out.y0 = uint32_t(uint64_t(tr) * height / grid_rows);
    out.y1 = uint32_t(uint64_t(tr + 1) * height / grid_rows);
}
// ... of a plan: by its cut lines
void plan_tile(const vhr_tile_plan &p, uint32_t tr, uint32_t tc, vhr_rect &out) {
    out.x0 = p.col_cut[tc]; out.x1 = p.col_cut[tc + 1];
    out.y0 = p.row_cut[tc][tr]; out.y1 = p.row_cut[tc][tr + 1];
}
// n + 1 cut lines of [0, extent) at equal cost: `marginal` holds the cost of every cell-pixel-wide slab.  Cut j = the first cell boundary at which the running
// sum reaches j / n of the total, then moved so that every tile is at least min_px wide (the halo) -- left to right, then right to left.  Returns false
// if n tiles of min_px do not fit.
bool balanced_cuts(const std::vector<uint64_t> &marginal, uint32_t cell, uint32_t extent, uint32_t n, uint32_t min_px, uint32_t *cuts) {
    bool weighted = false;
    const uint32_t cells = uint32_t(marginal.size());
    uint64_t total = 0;
    for (uint64_t v : marginal) total += v;
    cuts[0] = 0; cuts[n] = extent;
    if (total == 0 || cells == 0) { for (uint32_t j = 1; j < n; ++j) cuts[j] = uint32_t(uint64_t(j) * extent / n); }
    else {
        weighted = true;
        uint64_t run = 0;
        uint32_t b = 0;
        for (uint32_t j = 1; j < n; ++j) {
            while (b < cells && run * n < total * j) run += marginal[b++];
            cuts[j] = uint32_t(std::min<uint64_t>(extent, uint64_t(b) * cell));
        }
    }
    if (n > 1 && uint64_t(min_px) * n > extent) return false;
    if (weighted) {                                    // (equal-pixel cuts are never moved: a tile thinner than its halo is the caller's to hear about)
        for (uint32_t j = 1; j < n; ++j) cuts[j] = std::max(cuts[j], cuts[j - 1] + std::max(1u, min_px));
        for (uint32_t j = n - 1; j >= 1; --j) cuts[j] = std::min(cuts[j], cuts[j + 1] - std::max(1u, min_px));
    }
    for (uint32_t j = 1; j <= n; ++j) if (cuts[j] <= cuts[j - 1] || cuts[j] - cuts[j - 1] < min_px) return false;
    return true;
}
// what vhr_comm_create checks of a plan it is handed: the cut lines are a partition of the image, the rectangle is the rank's cell, no tile is thinner than a halo
bool plan_consistent(const vhr_tile_plan &p) {
    if (p.world == 0 || p.rank >= p.world || p.grid_rows == 0 || p.grid_cols == 0 || uint64_t(p.grid_rows) * p.grid_cols != p.world) return false;
    if (p.grid_rows > VHR_TILE_MAX_GRID || p.grid_cols > VHR_TILE_MAX_GRID) return false;
    if (p.col_cut[0] != 0 || p.col_cut[p.grid_cols] != p.width) return false;
    for (uint32_t c = 0; c < p.grid_cols; ++c) {
        if (p.col_cut[c + 1] <= p.col_cut[c] || (p.grid_cols > 1 && p.col_cut[c + 1] - p.col_cut[c] < p.halo_cols)) return false;
        if (p.row_cut[c][0] != 0 || p.row_cut[c][p.grid_rows] != p.height) return false;
        for (uint32_t r = 0; r < p.grid_rows; ++r) if (p.row_cut[c][r + 1] <= p.row_cut[c][r] || (p.grid_rows > 1 && p.row_cut[c][r + 1] - p.row_cut[c][r] < p.halo_rows)) return false;
    }
    vhr_rect own;
    plan_tile(p, p.rank / p.grid_cols, p.rank % p.grid_cols, own);
    if (own.x0 != p.col_begin || own.x1 != p.col_end || own.y0 != p.row_begin || own.y1 != p.row_end) return false;
    if (p.world > 1 && (p.halo_rows < p.overlap || p.halo_cols < p.overlap)) return false;
    return true;
}
vhr_rect grown(const vhr_rect &r, uint32_t dx, uint32_t dy, uint32_t width, uint32_t height) {
    return vhr_rect{ r.x0 > dx ? r.x0 - dx : 0u, uint32_t(std::min<uint64_t>(width, uint64_t(r.x1) + dx)), r.y0 > dy ? r.y0 - dy : 0u,
                     uint32_t(std::min<uint64_t>(height, uint64_t(r.y1) + dy)) };
}
bool intersect(const vhr_rect &a, const vhr_rect &b, vhr_rect &out) {
    out = vhr_rect{ std::max(a.x0, b.x0), std::min(a.x1, b.x1), std::max(a.y0, b.y0), std::min(a.y1, b.y1) };
    return out.x0 < out.x1 && out.y0 < out.y1;
}

}  // namespace

struct vhr_comm {
    vhr_context *ctx = nullptr;
    vhr_tile_plan plan = {};
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;          // the exchanges' own stream: they run beside the next frame's ray tracing
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false;                  // exchanges are (or may be) in flight on `stream`: finish has to wait for `done`
    bool broken = false;                   // an enqueue failed half way: the peers may be out of step, nothing more is started
    // staging for rectangles that are not whole rows (RCCL moves contiguous bytes): one buffer per direction, grown on demand
    char *send_stage = nullptr, *recv_stage = nullptr;
    size_t send_capacity = 0, recv_capacity = 0;
    // the rectangle the context had before create applied the plan's (col_begin, col_end, row_begin, row_end, overlap, halo rows, halo cols)
    bool tile_applied = false;
    uint32_t saved_tile[7] = { 0, 0, 0, 0, 0, 0, 0 };
    std::string error;
    int fail(int code, const std::string &msg) { error = msg; if (ctx) ctx->error = msg; return code; }
};

#define HIPC_TRY(c, expr)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return (c)->fail(VHR_ERROR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

// ---- the planner (tiling.atrous_overlap / atrous_output_extent / tile_bounds / make_tile_plan / TilePlan.exchanges) ----
uint32_t vhr_atrous_overlap(uint32_t atrous_steps) {
    // the published image is iteration n-2's output (hybrid_render_path.cpp:322-325); iteration i reads +-2*2^i rows
    // (svgf_atrous_filter.comp:72-75): sum_{i=0}^{n-2} 2*2^i = 2*(2^(n-1) - 1)
    return atrous_steps < 2 ? 0u : 2u * ((1u << (atrous_steps - 1)) - 1u);
}

uint32_t vhr_atrous_output_extent(uint32_t overlap, uint32_t step) {
    const uint32_t used = 4u * step - 2u;          // rows of validity the iterations up to this one have consumed
    return overlap > used ? overlap - used : 0u;
}

// The grid for `world` tiles that recomputes the least: (rows, cols) with rows * cols == world minimising the pixels the SVGF
// kernels compute on the busiest rank (its rectangle grown by E on every cut side, clipped to the image).  1080p, E = 30:
// 8 -> 2 x 4, 540 x 570 = +19 % over the 480 x 540 owned, where 8 row strips compute 1920 x 195 = +44 %.
int vhr_tile_grid_choose(uint32_t width, uint32_t height, uint32_t world, uint32_t overlap, uint32_t *grid_rows, uint32_t *grid_cols) {
    if (!grid_rows || !grid_cols || world == 0 || width == 0 || height == 0) return VHR_ERROR_INVALID_ARGUMENT;
    uint64_t best = ~0ull;
    uint32_t best_skew = ~0u;
    for (uint32_t r = 1; r <= world; ++r) {
        if (world % r) continue;
        const uint32_t c = world / r;
        if (r > height || c > width) continue;
        uint64_t worst = 0;                                  // the slowest rank: its rectangle grown by the overlap, clipped to the image
        for (uint32_t tr = 0; tr < r; ++tr)
            for (uint32_t tc = 0; tc < c; ++tc) {
                vhr_rect t;
                tile_bounds(width, height, r, c, tr, tc, t);
                const vhr_rect g = grown(t, c > 1 ? overlap : 0u, r > 1 ? overlap : 0u, width, height);
                worst = std::max(worst, uint64_t(g.x1 - g.x0) * (g.y1 - g.y0));
            }
        const uint32_t skew = r > c ? r - c : c - r;         // ties: the squarer grid (shorter seams)
        if (worst < best || (worst == best && skew < best_skew)) { best = worst; best_skew = skew; *grid_rows = r; *grid_cols = c; }
    }
    return best == ~0ull ? VHR_ERROR_INVALID_ARGUMENT : VHR_OK;
}

int vhr_tile_plan_make_weighted(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t grid_rows, uint32_t grid_cols, uint32_t max_motion_rows,
                                uint32_t max_motion_cols, uint32_t atrous_steps, const uint32_t *cost, uint32_t cost_cols, uint32_t cost_rows, uint32_t cell,
                                vhr_tile_plan *out) {
    if (!out || world == 0 || rank >= world || height == 0 || width == 0) return VHR_ERROR_INVALID_ARGUMENT;
    const uint32_t overlap = world > 1 ? vhr_atrous_overlap(atrous_steps) : 0u;
    if (grid_rows == 0 || grid_cols == 0) {
        const int rc = vhr_tile_grid_choose(width, height, world, overlap, &grid_rows, &grid_cols);
        if (rc != VHR_OK) return rc;
    }
    if (uint64_t(grid_rows) * grid_cols != world || grid_rows > height || grid_cols > width || grid_rows > VHR_TILE_MAX_GRID || grid_cols > VHR_TILE_MAX_GRID)
        return VHR_ERROR_INVALID_ARGUMENT;
    if (cost && (cell == 0 || uint64_t(cost_cols) * cell < width || uint64_t(cost_rows) * cell < height)) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_tile_plan p = {};
    p.rank = rank; p.world = world; p.width = width; p.height = height;
    p.grid_rows = grid_rows; p.grid_cols = grid_cols;
    if (world > 1) {
        p.overlap = overlap;
        // svgf.comp reads the reprojected position +-1 (svgf.comp:52-60,81-84); an axis that is not cut needs no halo
        p.halo_rows = grid_rows > 1 ? overlap + max_motion_rows + 2u : overlap;
        p.halo_cols = grid_cols > 1 ? overlap + max_motion_cols + 2u : overlap;
    }
    // the cut lines: at equal cost where a map is given, else (and on an all-zero map) at equal pixels; a halo must come from the adjacent tile alone.
    // Columns of tiles first (the column sums of the map), then every column of tiles cuts its own rows (the row sums of the map inside its columns).
    std::vector<uint64_t> mcol;
    const uint32_t ccols = cost ? (width + cell - 1) / cell : 0u, crows = cost ? (height + cell - 1) / cell : 0u;
    if (cost) {
        mcol.assign(ccols, 0);
        for (uint32_t cy = 0; cy < crows; ++cy)
            for (uint32_t cx = 0; cx < ccols; ++cx) mcol[cx] += cost[size_t(cy) * cost_cols + cx];
    }
    if (!balanced_cuts(mcol, cost ? cell : 1u, width, grid_cols, grid_cols > 1 ? p.halo_cols : 0u, p.col_cut)) return VHR_ERROR_OUT_OF_SLOTS;
    for (uint32_t c = 0; c < grid_cols; ++c) {
        std::vector<uint64_t> mrow;
        if (cost) {
            mrow.assign(crows, 0);
            // (a cell belongs to the column of tiles its first pixel column lies in)
            for (uint32_t cx = 0; cx < ccols; ++cx) {
                const uint32_t x = cx * cell;
                if (x < p.col_cut[c] || x >= p.col_cut[c + 1]) continue;
                for (uint32_t cy = 0; cy < crows; ++cy) mrow[cy] += cost[size_t(cy) * cost_cols + cx];
            }
        }
        if (!balanced_cuts(mrow, cost ? cell : 1u, height, grid_rows, grid_rows > 1 ? p.halo_rows : 0u, p.row_cut[c])) return VHR_ERROR_OUT_OF_SLOTS;
    }
    vhr_rect own;
    plan_tile(p, rank / grid_cols, rank % grid_cols, own);
    p.col_begin = own.x0; p.col_end = own.x1; p.row_begin = own.y0; p.row_end = own.y1;
    *out = p;
    return VHR_OK;
}

int vhr_tile_plan_make(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t grid_rows, uint32_t grid_cols, uint32_t max_motion_rows,
                       uint32_t max_motion_cols, uint32_t atrous_steps, vhr_tile_plan *out) {
    return vhr_tile_plan_make_weighted(width, height, world, rank, grid_rows, grid_cols, max_motion_rows, max_motion_cols, atrous_steps, nullptr, 0, 0, 0, out);
}

// For a margin of (halo_rows, halo_cols) pixels: what this rank receives from each peer = the peer's owned pixels inside my
// grown rectangle, and what it sends = my owned pixels inside the peer's grown rectangle (symmetric by construction; up to 8
// peers).  Returns the number of exchanges written (<= capacity), or a negative error.
int vhr_tile_plan_exchanges(const vhr_tile_plan *p, uint32_t halo_rows, uint32_t halo_cols, vhr_rect_exchange *out, uint32_t capacity) {
    if (!p || (!out && capacity)) return VHR_ERROR_INVALID_ARGUMENT;
    if (p->world <= 1 || (halo_rows == 0 && halo_cols == 0)) return 0;
    const vhr_rect mine{ p->col_begin, p->col_end, p->row_begin, p->row_end };
    const vhr_rect my_need = grown(mine, p->grid_cols > 1 ? halo_cols : 0u, p->grid_rows > 1 ? halo_rows : 0u, p->width, p->height);
    uint32_t n = 0;
    for (uint32_t peer = 0; peer < p->world; ++peer) {
        if (peer == p->rank) continue;
        vhr_rect theirs;
        plan_tile(*p, peer / p->grid_cols, peer % p->grid_cols, theirs);
        const vhr_rect their_need = grown(theirs, p->grid_cols > 1 ? halo_cols : 0u, p->grid_rows > 1 ? halo_rows : 0u, p->width, p->height);
        vhr_rect_exchange e{};
        e.peer = int32_t(peer);
        const bool r = intersect(my_need, theirs, e.recv), s = intersect(their_need, mine, e.send);
        if (!r && !s) continue;
        if (!r) e.recv = vhr_rect{ 0, 0, 0, 0 };
        if (!s) e.send = vhr_rect{ 0, 0, 0, 0 };
        if (n >= capacity) return VHR_ERROR_OUT_OF_SLOTS;
        out[n++] = e;
    }
    return int(n);
}

// A re-plan between two frames (tiling.replan_transfers): the cross-frame state follows its pixels, each from the rank that owned it under the old plan.
int vhr_tile_plan_replan(const vhr_tile_plan *old_plan, const vhr_tile_plan *new_plan, vhr_rect_exchange *out, uint32_t capacity) {
    if (!old_plan || !new_plan || (!out && capacity)) return VHR_ERROR_INVALID_ARGUMENT;
    const vhr_tile_plan &o = *old_plan, &p = *new_plan;
    if (o.world != p.world || o.rank != p.rank || o.width != p.width || o.height != p.height) return VHR_ERROR_INVALID_ARGUMENT;
    if (!plan_consistent(o) || !plan_consistent(p)) return VHR_ERROR_INVALID_ARGUMENT;
    const uint32_t dx = p.grid_cols > 1 ? p.halo_cols : 0u, dy = p.grid_rows > 1 ? p.halo_rows : 0u;
    vhr_rect mine_new, mine_old;
    plan_tile(p, p.rank / p.grid_cols, p.rank % p.grid_cols, mine_new);
    plan_tile(o, o.rank / o.grid_cols, o.rank % o.grid_cols, mine_old);
    const vhr_rect my_need = grown(mine_new, dx, dy, p.width, p.height);
    uint32_t n = 0;
    for (uint32_t peer = 0; peer < p.world; ++peer) {
        if (peer == p.rank) continue;
        vhr_rect theirs_old, theirs_new;
        plan_tile(o, peer / o.grid_cols, peer % o.grid_cols, theirs_old);
        plan_tile(p, peer / p.grid_cols, peer % p.grid_cols, theirs_new);
        const vhr_rect their_need = grown(theirs_new, dx, dy, p.width, p.height);
        vhr_rect_exchange e{};
        e.peer = int32_t(peer);
        const bool r = intersect(my_need, theirs_old, e.recv), s = intersect(their_need, mine_old, e.send);
        if (!r && !s) continue;
        if (!r) e.recv = vhr_rect{ 0, 0, 0, 0 };
        if (!s) e.send = vhr_rect{ 0, 0, 0, 0 };
        if (n >= capacity) return VHR_ERROR_OUT_OF_SLOTS;
        out[n++] = e;
    }
    return int(n);
}

// ---- row strips: the one-column grid, in its own vocabulary ----
int vhr_strip_plan_make(uint32_t height, uint32_t world, uint32_t rank, uint32_t max_motion_rows, uint32_t atrous_steps, vhr_strip_plan *out) {
    if (!out || world == 0 || rank >= world || height == 0) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_tile_plan t;
    const int rc = vhr_tile_plan_make(1u << 20, height, world, rank, world, 1, max_motion_rows, 0, atrous_steps, &t);      // (the width plays no part)
    if (rc != VHR_OK) return rc;
    vhr_strip_plan p = {};
    p.rank = rank; p.world = world; p.height = height;
    p.row_begin = t.row_begin; p.row_end = t.row_end;
    p.overlap = t.overlap;
    p.halo = world > 1 ? t.halo_rows : 0u;
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
int vhr_comm_use_library(const char *path) {
    if (g_rccl_resolved) return VHR_ERROR_GRAPH;                     // too late: the entry points are bound
    forced_library() = path ? path : "";
    return VHR_OK;
}

const char *vhr_comm_library(void) {
    Rccl &r = rccl();
    return r.error.empty() ? r.path.c_str() : r.error.c_str();
}

int vhr_comm_get_unique_id(uint8_t out[VHR_COMM_UNIQUE_ID_BYTES]) {
    if (!out) return VHR_ERROR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    if (!r.error.empty()) return VHR_ERROR_NO_DEVICE;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return VHR_ERROR_DEVICE;
    std::memcpy(out, id.internal, sizeof(id.internal));
    return VHR_OK;
}

int vhr_comm_create_tiled(vhr_context *ctx, const vhr_tile_plan *plan, const uint8_t unique_id[VHR_COMM_UNIQUE_ID_BYTES], vhr_comm **out) {
    if (!ctx || !plan || !unique_id || !out || plan->rank >= plan->world)
        return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: invalid arguments") : VHR_ERROR_INVALID_ARGUMENT;
    if (ctx->host_only) return ctx->fail(VHR_ERROR_NO_DEVICE, "host-only context: no device work");
    if (plan->height != ctx->height || plan->width != ctx->width) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: the plan's extent is not the context's");
    // A plan the planner would not have made -- cut lines that are no partition, a rectangle off the grid, a halo that reaches past the adjacent tile -- makes
    // the two ends of an exchange disagree on its size, which hangs RCCL.  (Equal-cost plans have no closed form to recompute: what is checked is
    // the plan's consistency with itself; that every rank holds the SAME cut lines is the caller's contract, as it is for the cost map they come from.)
    if (!plan_consistent(*plan)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: the plan is not one vhr_tile_plan_make / vhr_strip_plan_make returns");
    Rccl &r = rccl();
    if (!r.error.empty()) return ctx->fail(VHR_ERROR_NO_DEVICE, r.error);
    vhr_comm *c = new vhr_comm;
    c->ctx = ctx;
    c->plan = *plan;
    auto bail = [&](int code, const std::string &msg) { vhr_comm_destroy(c); ctx->error = msg; return code; };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(VHR_ERROR_DEVICE, "vhr_comm_create: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, sizeof(id.internal));
    const ncclResult_t rc = r.CommInitRank(&c->nccl, int(plan->world), id, int(plan->rank));
    if (rc != ncclSuccess) { c->nccl = nullptr; return bail(VHR_ERROR_DEVICE, std::string("ncclCommInitRank: ") + r.GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess)
        return bail(VHR_ERROR_DEVICE, "vhr_comm_create: stream / event creation failed");
    // the context computes its tile from now on: owned rectangle + overlap recomputed, blits extended by the halos
    const uint32_t saved[7] = { ctx->col_begin, ctx->col_end, ctx->row_begin, ctx->row_end, ctx->overlap, ctx->halo, ctx->halo_cols };
    const int src = vhr_set_tile(ctx, plan->col_begin, plan->col_end, plan->row_begin, plan->row_end, plan->overlap, plan->halo_rows, plan->halo_cols);
    if (src != VHR_OK) { const std::string msg = ctx->error; return bail(src, msg); }
    for (int i = 0; i < 7; ++i) c->saved_tile[i] = saved[i];
    c->tile_applied = true;
    *out = c;
    return VHR_OK;
}

int vhr_comm_create(vhr_context *ctx, const vhr_strip_plan *plan, const uint8_t unique_id[VHR_COMM_UNIQUE_ID_BYTES], vhr_comm **out) {
    if (!ctx || !plan) return ctx ? ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: invalid arguments") : VHR_ERROR_INVALID_ARGUMENT;
    vhr_tile_plan t = {};
    t.rank = plan->rank; t.world = plan->world; t.width = ctx->width; t.height = plan->height;
    t.grid_rows = plan->world; t.grid_cols = 1;
    t.col_begin = 0; t.col_end = ctx->width; t.row_begin = plan->row_begin; t.row_end = plan->row_end;
    t.overlap = plan->overlap; t.halo_rows = plan->world > 1 ? plan->halo : 0u; t.halo_cols = plan->overlap;
    if (plan->world > VHR_TILE_MAX_GRID) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "vhr_comm_create: more strips than VHR_TILE_MAX_GRID");
    t.col_cut[0] = 0; t.col_cut[1] = ctx->width;
    for (uint32_t r = 0; r <= plan->world; ++r) t.row_cut[0][r] = uint32_t(uint64_t(r) * plan->height / plan->world);
    return vhr_comm_create_tiled(ctx, &t, unique_id, out);
}

void vhr_comm_destroy(vhr_comm *c) {
    if (!c) return;
    if (c->ctx && !c->ctx->host_only) hipSetDevice(c->ctx->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->nccl) rccl().CommDestroy(c->nccl);
    if (c->ready) hipEventDestroy(c->ready);
    if (c->done) hipEventDestroy(c->done);
    if (c->stream) hipStreamDestroy(c->stream);
    hipFree(c->send_stage);
    hipFree(c->recv_stage);
    // the context gets the rectangle back it had before vhr_comm_create[_tiled] applied the plan's (a create that failed before that
    // point leaves whatever the caller had set -- a strip for the torch.distributed route, say -- alone)
    if (c->ctx && c->tile_applied)
        vhr_set_tile(c->ctx, c->saved_tile[0], c->saved_tile[1], c->saved_tile[2], c->saved_tile[3], c->saved_tile[4], c->saved_tile[5], c->saved_tile[6]);
    delete c;
}

const char *vhr_comm_last_error(const vhr_comm *c) { return c ? c->error.c_str() : "null communicator"; }

}  // extern "C"

namespace {

// One rectangle of one image on the wire.  Whole rows are contiguous in the image and go as they are; a column range is packed
// into / unpacked from a staging buffer with a strided device copy on the same stream (RCCL moves contiguous bytes).
struct Piece {
    Image *image;
    vhr_rect rect;
    int peer;
    bool send;
    size_t stage_offset;       // into the send / receive staging buffer; SIZE_MAX: in place
};

size_t rect_bytes(const Image &im, const vhr_rect &r) { return size_t(r.x1 - r.x0) * (r.y1 - r.y0) * im.bpp; }
bool whole_rows(const Image &im, const vhr_rect &r) { return r.x0 == 0 && r.x1 == im.width; }
char *rect_ptr(const Image &im, const vhr_rect &r) { return static_cast<char *>(im.ptr) + (size_t(r.y0) * im.width + r.x0) * im.bpp; }

int grow_stage(vhr_comm *c, char *&buf, size_t &capacity, size_t need) {
    if (need <= capacity) return VHR_OK;
    // (the old buffer may still be read by exchanges in flight: they are behind `done`, which the caller has waited for)
    if (buf) HIPC_TRY(c, hipFree(buf));
    buf = nullptr; capacity = 0;
    HIPC_TRY(c, hipMalloc(reinterpret_cast<void **>(&buf), need));
    capacity = need;
    return VHR_OK;
}

// Everything of one frame in ONE group: packs, sends and receives, unpacks.  The first failure inside the group is remembered, the
// group is closed all the same (a return between ncclGroupStart and ncclGroupEnd would leave every later RCCL call of this thread,
// ncclCommDestroy included, queued into it -- ADVICE r2), and the communicator refuses further work: its peers may be out of step.
int run_pieces_unguarded(vhr_comm *c, std::vector<Piece> &pieces, hipStream_t stream);
// Any failure in here -- a staging buffer that did not grow, a pack or unpack copy that was refused, RCCL itself -- leaves this rank out
// of step with its peers (they have posted, or will post, the matching sends and receives): the communicator is marked unusable on EVERY
// error path, and the job has to be torn down (the peers' pending receives do not complete: vhr_comm.h says so).
int run_pieces(vhr_comm *c, std::vector<Piece> &pieces, hipStream_t stream) {
    const int rc = run_pieces_unguarded(c, pieces, stream);
    if (rc != VHR_OK) c->broken = true;
    return rc;
}
int run_pieces_unguarded(vhr_comm *c, std::vector<Piece> &pieces, hipStream_t stream) {
    if (pieces.empty()) return VHR_OK;
    Rccl &r = rccl();
    size_t send_need = 0, recv_need = 0;
    for (Piece &p : pieces) {
        if (whole_rows(*p.image, p.rect)) { p.stage_offset = SIZE_MAX; continue; }
        size_t &need = p.send ? send_need : recv_need;
        p.stage_offset = need;
        need += (rect_bytes(*p.image, p.rect) + 255) & ~size_t(255);
    }
    int rc = grow_stage(c, c->send_stage, c->send_capacity, send_need);
    if (rc != VHR_OK) return rc;
    rc = grow_stage(c, c->recv_stage, c->recv_capacity, recv_need);
    if (rc != VHR_OK) return rc;
    auto copy2d = [&](char *dst, size_t dpitch, const char *src, size_t spitch, const Image &im, const vhr_rect &q) {
        return hipMemcpy2DAsync(dst, dpitch, src, spitch, size_t(q.x1 - q.x0) * im.bpp, q.y1 - q.y0, hipMemcpyDeviceToDevice, stream);
    };
    for (const Piece &p : pieces)                                   // pack what leaves
        if (p.send && p.stage_offset != SIZE_MAX) {
            const size_t line = size_t(p.rect.x1 - p.rect.x0) * p.image->bpp;
            HIPC_TRY(c, copy2d(c->send_stage + p.stage_offset, line, rect_ptr(*p.image, p.rect), size_t(p.image->width) * p.image->bpp, *p.image, p.rect));
        }
    std::string first_error;
    ncclResult_t g = r.GroupStart();
    if (g != ncclSuccess) return c->fail(VHR_ERROR_DEVICE, std::string("ncclGroupStart: ") + r.GetErrorString(g));
    for (const Piece &p : pieces) {
        char *buf = p.stage_offset == SIZE_MAX ? rect_ptr(*p.image, p.rect) : (p.send ? c->send_stage : c->recv_stage) + p.stage_offset;
        const size_t bytes = rect_bytes(*p.image, p.rect);
        const ncclResult_t e = p.send ? r.Send(buf, bytes, ncclUint8, p.peer, c->nccl, stream) : r.Recv(buf, bytes, ncclUint8, p.peer, c->nccl, stream);
        if (e != ncclSuccess && first_error.empty()) first_error = std::string(p.send ? "ncclSend: " : "ncclRecv: ") + r.GetErrorString(e);
        if (e != ncclSuccess) break;
    }
    g = r.GroupEnd();                                               // always
    if (g != ncclSuccess && first_error.empty()) first_error = std::string("ncclGroupEnd: ") + r.GetErrorString(g);
    if (!first_error.empty()) { c->broken = true; return c->fail(VHR_ERROR_DEVICE, first_error); }
    for (const Piece &p : pieces)                                   // unpack what arrived
        if (!p.send && p.stage_offset != SIZE_MAX) {
            const size_t line = size_t(p.rect.x1 - p.rect.x0) * p.image->bpp;
            HIPC_TRY(c, copy2d(rect_ptr(*p.image, p.rect), size_t(p.image->width) * p.image->bpp, c->recv_stage + p.stage_offset, line, *p.image, p.rect));
        }
    return VHR_OK;
}

// the halo exchange of `images` for a margin of (halo_rows, halo_cols), appended to `pieces`
int add_halo_pieces(vhr_comm *c, Image *const *images, int n_images, uint32_t halo_rows, uint32_t halo_cols, std::vector<Piece> &pieces) {
    vhr_rect_exchange ex[64];
    const int n = vhr_tile_plan_exchanges(&c->plan, halo_rows, halo_cols, ex, 64);
    if (n < 0) return c->fail(n, "vhr_comm: more than 64 peers in a halo exchange");
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n_images; ++i) {
            if (ex[k].send.x1 > ex[k].send.x0) pieces.push_back(Piece{ images[i], ex[k].send, ex[k].peer, true, 0 });
            if (ex[k].recv.x1 > ex[k].recv.x0) pieces.push_back(Piece{ images[i], ex[k].recv, ex[k].peer, false, 0 });
        }
    return VHR_OK;
}

}  // namespace

extern "C" {

// Exchange #1 (only with "trace_overlap" off): the overlap margin of the raw shadow / AO image from the neighbours, in the
// context's stream order -- svgf.comp, enqueued next, reads it.  Call it from the Raytrace Pass's epilogue.
int vhr_comm_exchange_raytraced(vhr_comm *c, const char *raytraced_image) {
    if (!c || !raytraced_image) return VHR_ERROR_INVALID_ARGUMENT;
    if (c->broken) return c->fail(VHR_ERROR_GRAPH, "vhr_comm: an earlier exchange failed half way; the communicator is unusable");
    auto it = c->ctx->images.find(raytraced_image);
    if (it == c->ctx->images.end()) return c->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + raytraced_image + "'");
    Image *im = &it->second;
    std::vector<Piece> pieces;
    const int rc = add_halo_pieces(c, &im, 1, c->plan.overlap, c->plan.overlap, pieces);
    if (rc != VHR_OK) return rc;
    if (c->pending) {          // the staging buffers are shared with the frame exchanges on the other stream: they have to be through
        HIPC_TRY(c, hipStreamWaitEvent(c->ctx->stream, c->done, 0));
        c->pending = false;
    }
    return run_pieces(c, pieces, c->ctx->stream);
}

// After the SVGF pass (its epilogue): exchange #2 -- the halo of the temporal history and of the moments history just written, for
// the NEXT frame's svgf.comp -- and, if `denoised_image` is given, the gather (C2) of every rank's owned rectangle of it into
// `gathered_frame` on `root` (a device buffer of the whole image there, ignored elsewhere).  One grouped batch on the communicator's
// own stream behind what the context has enqueued so far, i.e. beside the next frame's ray tracing; nothing waits for it here.
int vhr_comm_start_frame_exchanges(vhr_comm *c, int32_t history_storage_image, int32_t moments_storage_image, const char *denoised_image,
                                   int32_t root, void *gathered_frame) {
    if (!c) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = c->ctx;
    if (c->broken) return c->fail(VHR_ERROR_GRAPH, "vhr_comm: an earlier exchange failed half way; the communicator is unusable");
    auto storage = [&](int32_t id) -> Image * {
        return (id >= 0 && uint32_t(id) < vhr_context::kMaxGlobalResources && ctx->storage_images[id].used) ? &ctx->storage_images[id] : nullptr;
    };
    Image *imgs[2] = { storage(history_storage_image), storage(moments_storage_image) };
    if (!imgs[0] || !imgs[1]) return c->fail(VHR_ERROR_NOT_FOUND, "vhr_comm_start_frame_exchanges: no such storage image");
    if (c->pending) return c->fail(VHR_ERROR_GRAPH, "vhr_comm_start_frame_exchanges: the previous frame's exchanges were not finished");
    // everything that can be refused is refused before anything is enqueued
    Image gathered{};                         // the root's full-frame destination, addressed like an image
    std::vector<Piece> pieces;
    Image *den = nullptr;
    if (denoised_image && c->plan.world > 1) {
        auto it = ctx->images.find(denoised_image);
        if (it == ctx->images.end()) return c->fail(VHR_ERROR_NOT_FOUND, std::string("no transient image named '") + denoised_image + "'");
        if (root < 0 || uint32_t(root) >= c->plan.world) return c->fail(VHR_ERROR_INVALID_ARGUMENT, "gather root out of range");
        den = &it->second;
        const vhr_rect mine{ c->plan.col_begin, c->plan.col_end, c->plan.row_begin, c->plan.row_end };
        if (int32_t(c->plan.rank) == root) {
            if (!gathered_frame) return c->fail(VHR_ERROR_INVALID_ARGUMENT, "the gather's root needs a destination buffer");
            gathered = *den;
            gathered.ptr = gathered_frame;
            for (uint32_t peer = 0; peer < c->plan.world; ++peer) {
                if (int32_t(peer) == root) continue;
                vhr_rect theirs;
                plan_tile(c->plan, peer / c->plan.grid_cols, peer % c->plan.grid_cols, theirs);
                pieces.push_back(Piece{ &gathered, theirs, int(peer), false, 0 });
            }
        } else {
            pieces.push_back(Piece{ den, mine, root, true, 0 });
        }
    }
    const int arc = add_halo_pieces(c, imgs, 2, c->plan.halo_rows, c->plan.halo_cols, pieces);
    if (arc != VHR_OK) return arc;
    HIPC_TRY(c, hipSetDevice(ctx->device));
    HIPC_TRY(c, hipEventRecord(c->ready, ctx->stream));
    HIPC_TRY(c, hipStreamWaitEvent(c->stream, c->ready, 0));
    // From here on work may be in flight on the communicator's stream whatever happens next: `pending` is set first and `done` is
    // recorded on every path, so that vhr_comm_finish_frame_exchanges always has something true to wait for.
    c->pending = true;
    int rc = run_pieces(c, pieces, c->stream);
    if (rc == VHR_OK && den && int32_t(c->plan.rank) == root) {       // the root's own rectangle: a local copy, same stream
        const vhr_rect mine{ c->plan.col_begin, c->plan.col_end, c->plan.row_begin, c->plan.row_end };
        const size_t pitch = size_t(den->width) * den->bpp;
        if (hipMemcpy2DAsync(rect_ptr(gathered, mine), pitch, rect_ptr(*den, mine), pitch, size_t(mine.x1 - mine.x0) * den->bpp, mine.y1 - mine.y0,
                             hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            rc = c->fail(VHR_ERROR_DEVICE, "vhr_comm_start_frame_exchanges: the root's local copy failed");
    }
    if (hipEventRecord(c->done, c->stream) != hipSuccess && rc == VHR_OK) rc = c->fail(VHR_ERROR_DEVICE, "hipEventRecord(done) failed");
    return rc;
}

// A re-plan between two frames: the communicator takes `new_plan` (the same rank, world and image; every rank calls this between the same two frames with
// plans cut from the same map), the path's cross-frame state -- temporal history, moments history, previous normals -- travels to the new rectangles in one grouped
// batch on the context's stream (vhr_tile_plan_replan: each pixel from the rank that owned it), and the context computes the new rectangle from the next frame on.
// The previous frame's exchanges are finished first.  Everything that can be refused is refused before anything is enqueued.
int vhr_comm_replan(vhr_comm *c, const vhr_tile_plan *new_plan, int32_t history_storage_image, int32_t moments_storage_image, int32_t prev_normals_storage_image) {
    if (!c || !new_plan) return VHR_ERROR_INVALID_ARGUMENT;
    vhr_context *ctx = c->ctx;
    if (c->broken) return c->fail(VHR_ERROR_GRAPH, "vhr_comm: an earlier exchange failed half way; the communicator is unusable");
    auto storage = [&](int32_t id) -> Image * {
        return (id >= 0 && uint32_t(id) < vhr_context::kMaxGlobalResources && ctx->storage_images[id].used) ? &ctx->storage_images[id] : nullptr;
    };
    Image *imgs[3] = { storage(history_storage_image), storage(moments_storage_image), storage(prev_normals_storage_image) };
    if (!imgs[0] || !imgs[1] || !imgs[2]) return c->fail(VHR_ERROR_NOT_FOUND, "vhr_comm_replan: no such storage image");
    vhr_rect_exchange ex[VHR_TILE_MAX_GRID * VHR_TILE_MAX_GRID];
    const int n = vhr_tile_plan_replan(&c->plan, new_plan, ex, VHR_TILE_MAX_GRID * VHR_TILE_MAX_GRID);
    if (n < 0) return c->fail(n, "vhr_comm_replan: the new plan is not a plan of this rank, world and image");
    std::vector<Piece> pieces;
    for (int k = 0; k < n; ++k)
        for (Image *im : imgs) {
            if (ex[k].send.x1 > ex[k].send.x0) pieces.push_back(Piece{ im, ex[k].send, ex[k].peer, true, 0 });
            if (ex[k].recv.x1 > ex[k].recv.x0) pieces.push_back(Piece{ im, ex[k].recv, ex[k].peer, false, 0 });
        }
    HIPC_TRY(c, hipSetDevice(ctx->device));
    const int frc = vhr_comm_finish_frame_exchanges(c);                 // (the staging buffers are shared with the frame exchanges)
    if (frc != VHR_OK) return frc;
    const int rc = run_pieces(c, pieces, ctx->stream);
    if (rc != VHR_OK) return rc;
    const int src = vhr_set_tile(ctx, new_plan->col_begin, new_plan->col_end, new_plan->row_begin, new_plan->row_end, new_plan->overlap, new_plan->halo_rows, new_plan->halo_cols);
    if (src != VHR_OK) return c->fail(src, ctx->error);
    c->plan = *new_plan;
    return VHR_OK;
}

// Before the next frame's svgf.comp (the Raytrace Pass's epilogue): the context's stream waits for the exchanges started after the
// previous frame's SVGF pass -- also after a start that failed half way (what it did enqueue is drained).  No host synchronisation.
int vhr_comm_finish_frame_exchanges(vhr_comm *c) {
    if (!c) return VHR_ERROR_INVALID_ARGUMENT;
    if (!c->pending) return VHR_OK;
    HIPC_TRY(c, hipSetDevice(c->ctx->device));
    HIPC_TRY(c, hipStreamWaitEvent(c->ctx->stream, c->done, 0));
    c->pending = false;
    return VHR_OK;
}

}  // extern "C"
