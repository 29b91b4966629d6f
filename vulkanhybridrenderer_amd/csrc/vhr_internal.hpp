// Internal declarations shared by the host side (context / graph / BVH build) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "vhr_amd.h"

namespace vhr {

// ---------------------------------------------------------------------------------------------
// device-visible scene layout
// ---------------------------------------------------------------------------------------------
// BVH2 node = the two child boxes + two child links, 64 B = 4 x dwordx4 loads.  Each box is stored as
// (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z) so that the two slab planes of an axis sit in one aligned register pair:
// the slab test is then three v_pk_fma_f32 per box.
//   q0 = (b0.lox, b0.hix, b0.loy, b0.hiy)  q1 = (b0.loz, b0.hiz, b1.lox, b1.hix)
//   q2 = (b1.loy, b1.hiy, b1.loz, b1.hiz)  q3 = (child0, child1, 0, 0) as int bits
// child >= 0: inner node index.  child < 0: leaf, v = ~child, first = v >> 2, count = (v & 3) + 1.
// An absent child (single-leaf scene) has an inverted box (lo = +inf, hi = -inf) and is never entered.
struct BvhNode {
    float box0[6];
    float box1[6];
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode");

// The same node in 32 bytes = TWO 16-byte loads per visit (r3c): both boxes as centre and half extent in IEEE HALVES, centres
// relative to the scene centre (the walker shifts the ray origin once per ray).  The slab distances come straight out of
// v_fma_mix_f32, which widens a half operand inside the instruction: no unpacking, 18 plain FMAs for the 9 packed ones of the fp32
// form (the same lane operations).  c is the half nearest to the fp32 box centre, h the smallest normal half for which
// centre + c +- h contains the padded (lo, hi) box in exact arithmetic, plus a few fp32 ulp for the walker's own rounding; boxes only
// cull, so results are unchanged.  No half here is subnormal (c flushes to 0, h starts at 2^-14).
//   q0 = (c.x, c.y, c.z, h.x) words, child 0 in the low half and child 1 in the high half of each; q1 = (h.y, h.z, child0, child1)
// Inner links are BYTE offsets (index * 32).  An absent child has h = -1; a scene whose extent overflows the half range has no
// such form (HostBvh::nodes16_valid = false: the walkers stay on the 48-byte nodes).
struct BvhNode16 {
    uint16_t c[6];                   // (c0.x, c1.x, c0.y, c1.y, c0.z, c1.z)
    uint16_t h[6];                   // (h0.x, h1.x, h0.y, h1.y, h0.z, h1.z)
    int32_t child0, child1;
};
static_assert(sizeof(BvhNode16) == 32, "BvhNode16");

// The node the queue kernels walk (r2): the same two boxes as CENTRE and HALF EXTENT, 64 B.  With c and h the slab distances of an
// axis are near = c*inv + (-o*inv) - h*|inv|, far = ... + h*|inv|: one packed FMA for the centre term of BOTH boxes, then one
// packed FMA per box whose two halves are (near, far) directly (neg_lo on the half extent) -- the per-axis min / max pair of the
// (lo, hi) form disappears: 9 + 8 instead of 6 + 12 + 8 vector instructions per node.  c +- h contains the padded (lo, hi) box
// (h is widened until it does in exact arithmetic, plus 4 ulp of the coordinates' magnitude), and boxes only cull.
//   q0 = (c0.x, c1.x, c0.y, c1.y)  q1 = (c0.z, c1.z, h0.x, h0.y)  q2 = (h0.z, h1.x, h1.y, h1.z)  q3 = (child0, child1, 0, 0)
// An absent child has h = -1 (near > far on every axis).
struct BvhNodeCH {
    float cx[2], cy[2], cz[2];
    float h0[3], h1[3];
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNodeCH) == 64, "BvhNodeCH");

// BvhNodeCH in 48 bytes = THREE 16-byte loads per visit instead of four (the any-hit queue kernel is bound by the number of vector
// memory instructions its waves issue: ~20 cycles of the CU's address unit each, whatever their width -- profiles/r2_pmc_memory.txt).
// Centres stay fp32; the six half extents are stored as the UPPER 16 BITS of their fp32 pattern, rounded up, two per word:
//   hp[0] = (h0.x, h0.y)  hp[1] = (h0.z, h1.x)  hp[2] = (h1.y, h1.z)     (first in bits 31..16, second in bits 15..0)
// A kernel reads the first of a pair by using the word as the fp32 it is (the second's bits only enlarge the mantissa: a still
// larger half extent) and the second with one shift.  Larger half extents only cull less; results are unchanged.
//   q0 = (c0.x, c1.x, c0.y, c1.y)  q1 = (c0.z, c1.z, hp[0], hp[1])  q2 = (hp[2], child0, child1, 0)
struct BvhNode48 {
    float cx[2], cy[2], cz[2];
    uint32_t hp[3];
    int32_t child0, child1;
    int32_t pad;
};
static_assert(sizeof(BvhNode48) == 48, "BvhNode48");

// Leaf triangle, 48 B = 3 x dwordx4: Moeller-Trumbore operands precomputed in world space
// (resource_manager.cpp:608-617 bakes the primitive transform into the BLAS geometry).
struct BvhTri {
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t prim;      // gl_GeometryIndexEXT
    uint32_t tri;       // gl_PrimitiveID
    uint32_t flat;      // primitive-major flat triangle index (closest-hit tie break)
};
static_assert(sizeof(BvhTri) == 48, "BvhTri");

constexpr int kMaxLeafTris = 4;      // encoding limit of a leaf link
constexpr int kDefaultLeafTris = 2;  // r4 (scratch/ab_leaf.py, any-hit launch with 1 / 2 / 3 / 4): sponza_proc 359 / 322 / 325 / 341 us, bistro_proc 467 / 420 / 429 / 448;
                                     // the mirror-ray launch is indifferent between 2 and 3 (244 / 242, 429 / 433); 5.5 % more nodes than with 3
// Depth bound of any tree the walkers are given == the capacity of their traversal stacks (sponza_proc 22, bistro_proc 28; a deeper
// tree from the device builder sends the build to the host builder, whose forced median splits keep any scene below the bound).
constexpr int kMaxBvhDepth = 40;
constexpr int kTraceStack = 40;      // per-pixel kernels: the whole stack in LDS
constexpr int kSpillStack = 64;      // queue kernels: the part of the stack beyond their LDS levels lives in scratch (a power of two: masked indices)

struct DeviceTexture {
    const uint8_t *texels;   // RGBA8
    uint32_t width, height;
    int32_t format;          // VHR_FORMAT_R8G8B8A8_{SRGB,UNORM}
    int32_t mag_filter, address_u, address_v;
    int32_t pad;
};

struct DeviceScene {
    const BvhNode *nodes;
    const BvhNode16 *nodes16;
    const BvhNodeCH *nodes_ch;       // centre / half-extent form of `nodes` (same indices, same links)
    const BvhNode48 *nodes48;        // the same in 48 bytes (same indices, same links)
    float centre[3];                 // origin of the half-precision boxes
    float pad0;
    const BvhTri *tris;
    const vhr_vertex *vertices;
    const uint32_t *indices;
    const vhr_primitive *primitives;
    const float *normal_matrices;    // 9 floats per primitive, column-major inverseTranspose(mat3(transform))
    const DeviceTexture *textures;
    uint32_t node_count, tri_count, primitive_count, texture_count;
    // "bvh_frame": the boxes of all node forms live in this frame (row i = axis i in world coordinates); a walker rotates its ray once for the
    // slab tests -- box_ray() -- and intersects triangles in world space as ever.  frame_on == 0: the world axes, nothing to rotate.
    float frame[9];
    uint32_t frame_on;
};

// ---------------------------------------------------------------------------------------------
// host-side objects
// ---------------------------------------------------------------------------------------------
struct Image {
    void *ptr = nullptr;          // active device pointer (owned or external)
    void *owned = nullptr;        // context-owned allocation (may differ from ptr when bound externally)
    void *alt = nullptr;          // second buffer for images a kernel reads and rewrites in one dispatch
    // frames in flight ("frames_in_flight" n > 1): a graph-owned transient image has one instance per frame slot (slot 0 is
    // `owned`), and an external binding belongs to the slot it was made in; select_slot() points ptr / owned at the slot's
    void *slot_owned[3] = { nullptr, nullptr, nullptr };
    void *slot_external[3] = { nullptr, nullptr, nullptr };
    void select_slot(uint32_t s) { if (slot_owned[s]) owned = slot_owned[s]; ptr = slot_external[s] ? slot_external[s] : owned; }
    uint32_t width = 0, height = 0;
    int32_t format = 0;
    uint32_t bpp = 0;
    bool used = false;
    size_t bytes() const { return size_t(width) * height * bpp; }
};

struct HostBvh {
    std::vector<BvhNode> nodes;
    std::vector<BvhNode16> nodes16;
    std::vector<BvhNodeCH> nodes_ch;
    std::vector<BvhNode48> nodes48;
    float centre[3] = { 0, 0, 0 };
    bool nodes16_valid = false;        // false: some box does not fit the half range around `centre` (the walkers then stay on nodes48)
    float frame[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };     // "bvh_frame": the frame the boxes are in (identity: the world axes)
    bool frame_on = false;
    std::vector<BvhTri> tris;          // one per REFERENCE ("bvh_presplit": a fat triangle is in several leaves)
    uint32_t max_depth = 0;
    int presplit_level = -1;           // the grid level the references were split on, -1 = none were
};

// builds the BVH2 (csrc/bvh_build.cpp)
void check_node_forms(const HostBvh &bvh, uint64_t out[4], int threads = 0);
bool nodes16_in_range(const HostBvh &bvh);      // no inf / NaN / subnormal half in the 32-byte form (a device-built tree: the host decides)
uint64_t bvh_fingerprint(const HostBvh &bvh);
uint64_t bvh_tree_fingerprint(const HostBvh &bvh);      // of the tree, not of its arrays: equal for the host's and the device's build of a scene
void build_bvh(const vhr_vertex *vertices, const uint32_t *indices, const vhr_primitive *primitives,
               uint32_t primitive_count, HostBvh &out, int leaf_tris = kMaxLeafTris, int threads = 0, int presplit_percent = 0, int frame_mode = 0);
// the device-side builder (csrc/kernels_bvh.hip, option "bvh_builder" 1): from ctx->d_vertices / d_indices / d_primitives into the
// context's node and triangle arrays; VHR_ERROR_OUT_OF_SLOTS = fall back to the host builder (tree deeper than the walkers' stacks)
int device_build_bvh(vhr_context *ctx, const std::vector<uint32_t> &tri_prefix, uint32_t total_tris, int leaf_tris, int presplit_percent, int frame_mode);

enum class PassKind { Graphics, Raytracing, Compute };

// Pass time stamps written by the kernels themselves (r3c).  vkCmdWriteTimestamp pairs (render_graph.cpp:167-182) used to ride on the
// dispatch packets as HIP events; a dispatch that carries a completion signal costs ~5.5 us which the next kernel waits for -- 16.5 us
// of a 0.5 ms frame for the two passes' four stamps.  Instead every kernel of the library takes a trailing `Stamps` argument (appended
// by vhr::launch): the first thread of the grid stores the device's wall clock (s_memrealtime, 100 MHz) to `begin` if this is the
// first kernel of its pass, and to `prev_end` if the kernel before it on the stream was the last one of a pass -- on an in-order
// stream a kernel starts when its predecessor has drained, so that instant IS the end of the previous pass (plus the launch gap,
// 1-2 us).  A pass with nothing behind it in the frame gets its end from a one-thread kernel at the end of vhr_graph_execute.
// Used with one frame in flight; with more, passes of one frame sit on two streams and the event pairs stay.
struct Stamps { unsigned long long *begin, *prev_end; };
struct PassStampPair { unsigned long long begin, end; };
constexpr int kMaxStampedPasses = 64;
#if defined(__HIPCC__)
__device__ __forceinline__ void vhr_stamp(const Stamps &st) {
    if ((st.begin || st.prev_end) && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y | threadIdx.z) == 0u) {
        const unsigned long long t = wall_clock64();
        if (st.begin) *st.begin = t;
        if (st.prev_end) *st.prev_end = t;
    }
}
#endif

struct PassDescription {
    std::string name;
    PassKind kind;
    std::vector<vhr_transient_resource> dependencies, outputs;
    std::vector<std::string> resource_names;       // storage for the name strings above
    // raytracing
    std::string pipeline_name, raygen;
    std::vector<std::string> miss, closest_hit, any_hit;
    // compute
    std::vector<std::string> kernels;
    uint32_t push_constant_size = 0;
    vhr_external_pass_callback external_cb = nullptr;
    vhr_raytracing_pass_callback rt_cb = nullptr;
    vhr_compute_pass_callback compute_cb = nullptr;
    void *user = nullptr;
    vhr_external_pass_callback epilogue_cb = nullptr;
    void *epilogue_user = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timed = false;
    bool begin_stamped = false, end_on_last_dispatch = false;     // per Execute: see vhr_context::dispatch_events
    bool stamped_in_kernel = false;  // this Execute's stamps are Stamps stores (vhr_context::take_stamps), not events
    int stamp_index = -1;            // slot of the pass in vhr_context::d_stamps (assigned at Build)
    double ema_ms = 0.0, last_ms = 0.0;
};

// Kernel arguments of the SVGF kernels (csrc/kernels_svgf.hip) and the command record of a compute pass.
struct TemporalArgs {
    const uint2 *normals, *motion, *prev_normals, *history;   // RGBA16F
    const uint32_t *raytraced, *moments_in;                   // RG16F
    uint2 *integrated_out;                                     // RGBA16F
    uint32_t *moments_out;                                     // RG16F
    uint32_t width, height;       // image extent
    uint32_t limit_x, limit_y;    // pixels covered by the dispatch (groups * 8, clamped; limit_x also clamped to the tile's last column)
    uint32_t row_begin, row_end;  // rows this context computes
    uint32_t col_begin;           // first column this context computes (screen tiles, vhr_set_tile; 0 for row strips)
    float display_w, display_h;   // pfd.display_size
};
struct AtrousArgs {
    const uint2 *normals, *in;
    uint2 *out;
    uint2 *out2;                  // second destination of the same texels (a blit fused into the launch), or nullptr
    uint2 *normals_out;           // copy of the normals / ids of the pixels computed (a blit of the launch's `normals` input fused into it), or nullptr
    uint32_t width, height, limit_x, limit_y, row_begin, row_end;
    uint32_t col_begin;           // first column computed (screen tiles); limit_x is the end of the column range
    int32_t step;
    float display_w, display_h;
};
// A compute pass RECORDS its dispatches and blits (the reference records a command buffer, compute_execution_context.cpp)
// and issues them when its callback returns.  That lets a same-extent blit whose source is the output of a recorded a-trous
// dispatch -- and whose destination nothing in between touches -- become a second store of that launch instead of a copy
// kernel of its own: two of the three blits of hybrid_render_path.cpp:310-325 (one launch, its gap and 8 B/px of reads each).
struct SvgfCmd {
    enum Kind { Temporal, Atrous, Copy } kind;
    TemporalArgs t;
    AtrousArgs a;
    const char *copy_src; char *copy_dst; size_t copy_bytes;       // Copy: row range applied (whole rows: one contiguous range)
    size_t copy_pitch, copy_row_bytes; uint32_t copy_rows;         // Copy of a column range (screen tiles): copy_rows pieces of copy_row_bytes, copy_pitch apart; copy_rows == 0: contiguous
    const void *src_base; void *dst_base;                          // Copy: the images' base pointers (hazard checks)
};

struct RayStats {
    unsigned long long unique_rays, covered_pixels, stack_overflows, node_visits, leaf_visits, triangle_tests, wave_iterations, second_bounce_rays;
    unsigned long long cycles_total, cycles_setup, cycles_refill, cycles_nodes, cycles_leaves, refills, waves, drain_iterations;   // per-wave s_memtime sums
    unsigned long long cut_entries;
    unsigned long long drain_le4, drain_le8, drain_le16;     // drain trips made with at most 4 / 8 / 16 rays of the wave still in flight
    unsigned long long pending_rays, pending_retraces;       // decision (vi): pixels a queue kernel computed again by the per-pixel code (binary64 inline); unused
};

// Options (vhr_set_option): ONE table -- name, default, smallest and largest value -- that vhr_set_option, vhr_get_option, vhr_option_info
// (the Python binding, the neutrality test) and the context's defaults all read.  Every setting computes identical results (the a-trous
// and mirror-ray forms within the float tolerance of tests/); the schedule options change what is launched, never what is published.
#define VHR_OPTION_TABLE(X)                                                                                                             \
    /* which form of a shader runs: 0 = the literal per-pixel form (the in-tree cross-check), 1 = the default */                        \
    X(kOptRaygenVariant, "raygen_variant", 1, 0, 1)             /* raygen.rgen's shadow + AO rays: raygen_kernel / raygen_queue_kernel */ \
    X(kOptReflectionVariant, "reflection_variant", 1, 0, 1)     /* the mirror ray: reflection_kernel / reflection_queue_kernel */       \
    X(kOptRaytracedVariant, "raytraced_variant", 1, 0, 1)       /* the raytraced render path: raytraced_kernel / raytraced_queue_kernel */ \
    X(kOptAtrousVariant, "atrous_variant", 1, 0, 1)             /* svgf_atrous_filter.comp: svgf_atrous_kernel / svgf_atrous_tile_kernel */ \
    /* the queue kernels */                                                                                                             \
    X(kOptRefillThreshold, "refill_threshold", 16, 1, 64)       /* idle lanes that trigger a refill from the tile's ray queue */        \
    X(kOptLdsStackLevels, "lds_stack_levels", 8, 1, 32)         /* traversal stack entries kept in LDS (deeper ones live in scratch) */ \
    X(kOptReflectionLdsStackLevels, "reflection_lds_stack_levels", 10, 1, 32) /* the same for the mirror ray's walk (r5: 8 -> 10 = bistro_proc's launch -4 %, sponza_proc's equal; the any-hit launch loses 2-3 % at 10) */ \
    X(kOptWavesPerBlock, "raygen_waves_per_block", 2, 1, 4)     /* tiles (= waves) per workgroup of raygen_queue_kernel: 1, 2 or 4 */    \
    X(kOptCompactNodes, "compact_nodes", 1, 0, 1)               /* the 32-byte half-precision nodes where the tree has them */          \
    X(kOptEarlyExit, "raygen_early_exit", 6, 0, 15)             /* sixteenths of the walkers that entered below which the node loop is left */ \
    X(kOptReflectionEarlyExit, "reflection_early_exit", 8, 0, 15) /* the same for the mirror ray's walk (r4: 6 -> 8 = bistro_proc's launch -4 %, sponza_proc's equal) */ \
    X(kOptRaygenTileRows, "raygen_tile_rows", 0, 0, 8)          /* rows of a wave's tile; 0 = auto (6 for launches that fill < 70 % of the wave slots) */ \
    X(kOptRaygenCostOrder, "raygen_cost_order", 1, 0, 2)        /* start the longest-lived tiles first: 1 = launches of >= 2 048 workgroups, 2 = any */ \
    X(kOptRaygenSteal, "raygen_steal", 8, 0, 63)                /* queue dry and >= n lanes idle: idle lanes take pending subtrees off busy lanes' stacks; 0 = never */ \
    /* the a-trous kernel */                                                                                                            \
    X(kOptAtrousSmallTiles, "atrous_small_tiles", -1, -1, 1)    /* 4-row tiles: -1 auto (below 32 8-row tiles per CU), 0 never, 1 always */ \
    /* the frame's schedule */                                                                                                          \
    X(kOptFuseBlits, "fuse_blits", 1, 0, 1)                     /* a pass's blits as stores of its a-trous launches */                   \
    X(kOptSvgfElideUnread, "svgf_elide_unread", 0, 0, 1)        /* do not launch an a-trous dispatch nothing reads (the reference's fifth) */ \
    X(kOptSvgfAsyncUnread, "svgf_async_unread", 1, 0, 2)        /* ... or issue it on the side stream: 1 = where it pays, 2 = always */  \
    X(kOptReflectionAsync, "reflection_async", 1, 0, 2)         /* the mirror ray's launch on a stream of its own, beside the SVGF pass */  \
    X(kOptFuseTemporal, "fuse_temporal", 0, 0, 1)               /* svgf.comp in the ray-tracing kernel's tile epilogues */              \
    X(kOptFramesInFlight, "frames_in_flight", 1, 1, 3)          /* read by vhr_graph_build */                                            \
    /* screen tiles / row strips (one process per GPU) */                                                                               \
    X(kOptTraceOverlap, "trace_overlap", 0, 0, 1)               /* trace the overlap margin locally instead of exchanging raw visibility */ \
    X(kOptShrinkOverlap, "strip_shrink_overlap", 0, 0, 1)       /* later a-trous iterations compute only the margin still needed */     \
    /* instrumentation */                                                                                                               \
    X(kOptPassTimestamps, "pass_timestamps", 1, 0, 3)           /* 0 off, 1 in-kernel stamps, 2 + a stamp in front of external passes, 3 event pairs */ \
    X(kOptKernelTimingStride, "kernel_timing_stride", 1, 1, 1000000)   /* every n-th launch of a timed kind carries an event pair */

enum Option {
#define X(e, n, d, lo, hi) e,
    VHR_OPTION_TABLE(X)
#undef X
    kOptCount
};
struct OptionInfo { const char *name; int def, lo, hi; };
inline constexpr OptionInfo kOptionInfo[kOptCount] = {
#define X(e, n, d, lo, hi) { n, d, lo, hi },
    VHR_OPTION_TABLE(X)
#undef X
};

// optional per-kernel timing with HIP events on the context stream (vhr_set_kernel_timing)
enum KernelKind { kKernelRaygen = 0, kKernelTemporal = 1, kKernelAtrous = 2, kKernelCopy = 3, kKernelReflection = 4, kKernelSsao = 5, kKernelSsaoBlur = 6, kKernelSsr = 7, kKernelAtrousAsync = 8, kKernelKinds = 9 };
struct KernelTimer {
    std::vector<hipEvent_t> events;     // begin/end pairs
    size_t used = 0;                    // events recorded since the last drain
    uint64_t seen = 0;                  // launches of the kind while its timing was on (kernel_timing_stride samples them)
    double total_ms = 0.0;
    uint64_t launches = 0;
};

}  // namespace vhr

struct vhr_raytracing_execution_context {
    vhr_context *ctx;
    vhr::PassDescription *pass;
    uint32_t resource_idx;
};
struct vhr_compute_execution_context {
    vhr_context *ctx;
    vhr::PassDescription *pass;
    uint32_t resource_idx;
};

struct vhr_context {
    int device = 0;
    bool host_only = false;          // VHR_CREATE_HOST_ONLY: graph bookkeeping without a device
    uint32_t width = 0, height = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string error;

    // ResourceManager state
    static constexpr uint32_t kMaxGlobalResources = 2048;   // resource_manager.h:13
    std::vector<vhr::Image> storage_images;                 // pool of kMaxGlobalResources
    struct Texture { void *texels; uint32_t w, h; int32_t format; vhr_sampler_info sampler; };
    std::vector<Texture> textures;
    vhr::DeviceTexture *d_textures = nullptr;
    bool textures_dirty = false;
    vhr_per_frame_data per_frame[3] = {};
    uint32_t last_resource_idx = 0;    // the slot the last vhr_graph_execute ran with (vhr_get_last_per_frame_ubo)
    vhr_trace_params trace_params = {};

    // scene
    vhr_vertex *d_vertices = nullptr;
    uint32_t *d_indices = nullptr;
    vhr_primitive *d_primitives = nullptr;
    float *d_normal_matrices = nullptr;
    vhr::BvhNode *d_nodes = nullptr;
    vhr::BvhNode16 *d_nodes16 = nullptr;
    vhr::BvhNodeCH *d_nodes_ch = nullptr;
    vhr::BvhNode48 *d_nodes48 = nullptr;
    uint64_t bvh_fingerprint = 0;                   // bvh_fingerprint() of the last build (vhr_get_bvh_fingerprint)
    uint64_t bvh_tree_fingerprint = 0;              // bvh_tree_fingerprint() of the last build (vhr_get_bvh_tree_fingerprint)
    bool bvh_fingerprint_valid = false;             // a device-built tree is hashed when somebody asks (it would have to be fetched first)
    int bvh_host_checks = 0;                        // "bvh_host_checks" 1: a device-built tree is fetched and the host's self-checks repeated on it
    uint64_t bvh_form_checks[4] = { 0, 0, 0, 0 };   // check_node_forms of the last build (vhr_get_bvh_form_checks)
    float bvh_centre[3] = { 0, 0, 0 };
    bool nodes16_valid = false;      // the 32-byte half-precision nodes exist for this tree (extent within the half range) and passed the containment check
    vhr::BvhTri *d_tris = nullptr;
    uint32_t vertex_count = 0, index_count = 0, primitive_count = 0, node_count = 0, tri_count = 0, bvh_depth = 0;
    int bvh_leaf_tris = vhr::kDefaultLeafTris;   // "bvh_leaf_triangles": leaf size of this context's next build
    int bvh_build_threads = 0;                   // "bvh_build_threads": host threads of the next build (0 = up to 16 of the machine's)
    int bvh_frame_mode = 1;                      // "bvh_frame": 1 (default) = the boxes in the frame that minimises the triangles' summed box area (csrc/bvh_frame.hpp; the world axes if none gains 5 %), 0 = world axes
    float bvh_frame[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };      // the current tree's frame
    bool bvh_frame_on = false;
    int bvh_presplit = 0;                        // "bvh_presplit": budget of extra triangle references in percent (csrc/presplit.hpp), 0 = off
    int bvh_presplit_level = -1;                 // the grid level the current tree's references were split on (-1: none)
    int bvh_builder = 1;                         // "bvh_builder": 1 = binned SAH on the device (csrc/kernels_bvh.hip, default), 0 = on the host (csrc/bvh_build.cpp)
    int bvh_builder_used = 0;                    // which one made the current tree (the device builder falls back for trees too deep / too small)
    int bvh_device_max_depth = vhr::kMaxBvhDepth; // "bvh_device_max_depth": the device builder hands a deeper tree to the host builder (kMaxBvhDepth = the walkers' stacks; tests lower it)
    double bvh_build_ms = 0.0, geometry_upload_ms = 0.0;      // K0: host build / device upload of the last vhr_update_geometry
    double bvh_check_ms = 0.0;                                 // the self-checks of the node forms + the fingerprint (not part of K0)

    // RenderGraph state
    std::map<std::string, vhr::PassDescription> pass_descriptions;
    std::vector<std::string> registration_order;
    std::vector<std::string> execution_order;
    std::unordered_map<std::string, vhr::Image> images;
    std::unordered_map<std::string, std::string> compute_kernel_owner;    // shader -> pass (global key)
    bool built = false;

    // strips / screen tiles (vhr_set_strip, vhr_set_tile): the owned rectangle, the margin the SVGF kernels recompute around it
    // (`overlap`, both axes) and the margin the blits copy (`halo` rows, `halo_cols` columns: what next frame's svgf.comp may read)
    uint32_t row_begin = 0, row_end = 0, overlap = 0, halo = 0;
    uint32_t col_begin = 0, col_end = 0, halo_cols = 0;

    // Frames in flight (vulkan_common.h:9 MAX_FRAMES_IN_FLIGHT, renderer.cpp:103-146: the reference's CPU runs up to three
    // frames ahead and its queue overlaps whatever the barriers allow).  With "frames_in_flight" n > 1 (read at vhr_graph_build)
    // the passes up to and including the last ray-tracing pass of the execution order -- the FRONT of the frame: G-buffer,
    // shadow map, Raytrace Pass -- are issued on `front_stream`, the rest (SVGF, composition, ...) on `stream`; every transient
    // image exists once per frame slot (resource_idx mod n).  Dependencies, derived from that split: the back of frame f waits
    // for its front (front_done[slot]); the front of frame f + n, which rewrites slot f mod n, waits for the back of frame f
    // (back_done[slot]).  Persistent storage images (the SVGF history) are touched by back passes only, which stay in order.
    int frames_in_flight = 1;
    hipStream_t front_stream = nullptr;
    hipEvent_t front_done[3] = { nullptr, nullptr, nullptr }, back_done[3] = { nullptr, nullptr, nullptr };
    bool back_pending[3] = { false, false, false };
    size_t front_passes = 0;           // passes [0, front_passes) of execution_order run on front_stream
    uint32_t cur_slot = 0;
    int sync_streams();                // waits for every stream of the context
    // "svgf_async_unread": a compute pass's a-trous dispatch whose output nothing reads (the reference's fifth iteration) is issued on
    // `side_stream` after the pass's other commands, beside whatever the caller's stream does next (the next frame's ray tracing);
    // side_done is recorded behind it and join_side() makes the caller's stream wait for it before anything of the library touches
    // the images it reads or writes again.
    hipStream_t side_stream = nullptr;
    hipEvent_t side_ready = nullptr, side_done = nullptr;
    // (events that only order the library's own streams on one device: no timing, and no system-scope release -- nothing on the host reads what they guard)
#ifndef VHR_JOIN_EVENT_FLAGS
#define VHR_JOIN_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
    bool side_pending = false;
    const void *side_reads[2] = { nullptr, nullptr }, *side_writes = nullptr;      // the images the pending dispatch reads / writes (hazard checks)
    bool async_atrous = false;         // the a-trous launch being issued is the side stream's (kernel kind kKernelAtrousAsync)
    int join_side();                   // the current stream waits for the side stream's pending dispatch (no-op without one)
    // "reflection_async": the mirror ray's launch (not denoised: nothing of the SVGF pass reads it) runs on `refl_stream` beside the SVGF pass;
    // the caller's stream waits for it (join_refl) before the frame's next external pass, at the end of vhr_graph_execute, and wherever the
    // library waits for or hands out the context's images.
    hipStream_t refl_stream = nullptr;
    hipEvent_t refl_ready = nullptr, refl_done = nullptr;
    bool refl_pending = false;
    const void *refl_writes = nullptr; // the image the pending launch writes (the one whose readers must wait)
    const void *refl_reads[2] = { nullptr, nullptr };      // ... and the G-buffer images it reads (whose writers must wait)
    int join_refl();

    // statistics
    bool ray_stats_enabled = false;
    vhr::RayStats *d_ray_stats = nullptr;
    vhr::RayStats h_ray_stats = {};
    vhr::RayStats h_refl_stats = {};      // the mirror-ray launch's counters (d_ray_stats[1])
    uint64_t raytraced_pixels = 0;      // != 0: the last TraceRays was the raytraced render path's (primary rays launched)

    int options[vhr::kOptCount];       // vhr_set_option; defaults from vhr::kOptionInfo (the constructor)
    vhr_context() { for (int i = 0; i < vhr::kOptCount; ++i) options[i] = vhr::kOptionInfo[i].def; }
    int cu_count = 256;
    uint32_t *d_tile_counter = nullptr;
    // "raygen_cost_order" (csrc/kernels_trace.hip): ray-tracing launch f leaves its waves' lifetimes in cost[f & 1], and its FIRST block, before it
    // turns to its own tile, sorts the blocks of launch f - 1 by the lifetimes in cost[(f - 1) & 1] into order[(f + 1) & 1] -- the order launch
    // f + 1 starts its blocks in.  Everything happens inside the launches the frame has anyway: no kernel, stream or event of its own.  One set for
    // the shadow / AO queue kernel, one for the mirror-ray queue kernel, one for the raytraced path's (launches of different shapes).
    struct CostOrder {
        uint32_t *cost[2] = { nullptr, nullptr }, *order[2] = { nullptr, nullptr };
        uint32_t capacity = 0;                     // waves each of the four buffers holds
        uint32_t slot = 0;                         // the slot the last launch wrote its lifetimes to
        hipStream_t stream = nullptr;              // ... and the stream it was issued on (an order only connects launches of one stream)
        uint32_t cost_blocks[2] = { 0, 0 }, cost_key[2] = { 0, 0 };        // the launch shape cost[slot] was written by (0 blocks = nothing)
        uint32_t cost_waves[2] = { 0, 0 };                                 // ... and the words it wrote there (blocks x the waves per block the launch ran with)
        uint32_t order_blocks[2] = { 0, 0 }, order_key[2] = { 0, 0 };      // the launch shape order[slot] is an order of
        // where on the screen the waves of cost[slot] worked (vhr_get_tile_cost_map): wave i = tile (i % tiles_x, i / tiles_x) of tile_w x tile_h pixels from
        // (col_begin, row_begin) -- for the any-hit kernel with its workgroups of wv tiles side by side, i = (by * blocks_x + bx) * wv + wave <-> tile x = bx * wv + wave
        struct Shape { uint32_t tiles_x = 0, blocks_x = 0, wv = 0, tile_w = 8, tile_h = 8, col_begin = 0, row_begin = 0; } shape[2];
    };
    CostOrder cost_order_raygen, cost_order_reflection, cost_order_raytraced;
    // SSAOPushConstants as last pushed by any dispatch of this context: ssao.comp reads its radius although the reference never
    // pushes it to that pipeline (hybrid_render_path.cpp:151-167; the blur pass gets the constants instead, :182-197)
    float ssao_radius = 0.75f;
    uint32_t kernel_timing_mask = 0;   // bit per KernelKind
    vhr::KernelTimer kernel_timers[vhr::kKernelKinds];
    // Timing rides on the dispatch packets: vhr::launch() attaches (start, stop) events to a kernel launch through
    // hipExtLaunchKernelGGL, i.e. the dispatch's own begin / end timestamps.  A hipEventRecord instead puts a barrier packet and
    // a ~4 us bubble on the stream (measured: 33 us per frame for the eight per-pass records alone).
    bool recording = false;            // inside a compute pass callback: SVGF commands are recorded, issued at its end
    std::vector<vhr::SvgfCmd> recorded;
    int timing_kind = -1;              // kernel kind whose launches are being issued (between time_begin and time_end)
    vhr::PassDescription *cur_pass = nullptr;      // pass whose callback is running (vkCmdWriteTimestamp equivalents)
    void time_begin(int kind) { timing_kind = kind; }
    void time_end(int) { timing_kind = -1; }
    void dispatch_events(hipEvent_t &start, hipEvent_t &stop);
    // in-kernel pass time stamps (see vhr::Stamps)
    vhr::PassStampPair *d_stamps = nullptr;          // kMaxStampedPasses pairs in device memory
    unsigned long long *pending_end = nullptr;       // where the next kernel on the stream stores the end of the pass that has just finished
    bool no_stamps = false;                          // the launch being issued is not on the context's stream (side stream): no stamps
    double wall_clock_khz = 100000.0;
    bool in_kernel_stamps() const { return frames_in_flight == 1 && d_stamps != nullptr && options[vhr::kOptPassTimestamps] != 3; }     // 3: event pairs (A-B)
    vhr::Stamps take_stamps();
    // "fuse_temporal": a TraceRays launch held back until the next pass shows its first command (csrc/kernels_trace.hip: launch_raygen,
    // flush_deferred_raygen).  `may_defer_raygen` is set by vhr_graph_execute around a ray-tracing pass that has no epilogue hooked to it.
    bool deferred_raygen = false, may_defer_raygen = false;
    std::vector<unsigned char> deferred_raygen_blob;
    vhr::PassDescription *deferred_pass = nullptr;

    vhr::DeviceScene device_scene() const;
    int fail(int code, const std::string &msg) { error = msg; return code; }
};

namespace vhr {

int flush_deferred_raygen(vhr_context *ctx, const TemporalArgs *fuse);      // csrc/kernels_trace.hip; no-op without a held-back launch
bool deferred_raygen_matches(const vhr_context *ctx, const TemporalArgs &t);

// Every kernel launch of the library goes through here (see vhr_context::dispatch_events).
template <typename K, typename... Args>
inline void launch(vhr_context *ctx, K kernel, dim3 grid, dim3 block, size_t lds, Args... args) {
    if (ctx->deferred_raygen) flush_deferred_raygen(ctx, nullptr);      // a held-back TraceRays goes first (no-op while IT is being issued)
    hipEvent_t start = nullptr, stop = nullptr;
    ctx->dispatch_events(start, stop);
    const Stamps st = ctx->take_stamps();             // every kernel's last parameter
    if (start || stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, ctx->stream, start, stop, 0, args..., st);
    else hipLaunchKernelGGL(kernel, grid, block, lds, ctx->stream, args..., st);
}

uint32_t format_stride(int32_t format);   // VkUtils::FormatStride (vulkan_utils.h:128-148)

// kernel launchers (csrc/kernels_trace.hip, csrc/kernels_svgf.hip).  All enqueue on ctx->stream.
struct ImageView { void *ptr; uint32_t width, height; };
int launch_raygen(vhr_context *ctx, const vhr_per_frame_data &pfd, uint32_t width, uint32_t height,
                  const Image &normals, const Image &depth, Image &shadow_ao, Image *reflections);
int launch_raytraced(vhr_context *ctx, const vhr_per_frame_data &pfd, uint32_t width, uint32_t height, Image &out, bool alpha_test);
int launch_raytraced_composition(vhr_context *ctx, const Image &in, Image &out);
int launch_standin_gbuffer(vhr_context *ctx, const vhr_per_frame_data &pfd, Image &normals, Image &motion, Image &depth, Image *albedo);
int launch_composition(vhr_context *ctx, const vhr_per_frame_data &pfd, const vhr_composition_desc &d, const Image &albedo, const Image &normals,
                       const Image &motion, const Image &depth, const Image &shadow_ao, const Image *reflections, const Image *ssao, const Image *shadow_map, Image &out);
int launch_standin_shadow_map(vhr_context *ctx, const vhr_per_frame_data &pfd, Image &shadow_map);
int launch_svgf_temporal(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &motion,
                         const Image &raytraced, const Image &prev_normals, const Image &history,
                         Image &moments, Image &integrated_out, uint32_t x_groups, uint32_t y_groups);
int launch_svgf_atrous(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &in,
                       Image &out, int32_t step, uint32_t x_groups, uint32_t y_groups);
int copy_image_rows(vhr_context *ctx, const Image &src, Image &dst);
// csrc/kernels_screen.hip: ssao.comp, ssao_blur.comp, ssr.comp
int launch_ssao(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &depth, Image &out, float radius,
                uint32_t x_groups, uint32_t y_groups);
int launch_ssao_blur(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &in, Image &out, uint32_t x_groups, uint32_t y_groups);
int launch_ssr(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &albedo, const Image &normals, const Image &motion,
               const Image &depth, Image &out, const vhr_ssr_push_constants &pc, uint32_t x_groups, uint32_t y_groups);
void launch_stamp(vhr_context *ctx);            // a one-thread kernel that takes the pending pass-end stamp (csrc/kernels_svgf.hip)
int flush_recorded(vhr_context *ctx);          // issue the commands a compute pass recorded (no-op when there are none)
int launch_calibration_read(vhr_context *ctx, const Image &img, uint32_t bytes_per_lane, uint32_t *sink);
int launch_ray_triangle_pairs(vhr_context *ctx, const float *pairs, uint32_t n, uint32_t *hit, float *tuv);

}  // namespace vhr
