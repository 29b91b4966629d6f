// K0 on the device (option "bvh_builder" 1): the acceleration structure built where the reference builds it -- on the GPU
// (ResourceManager::UpdateBLAS / UpdateTLAS, /root/reference/src/rendering_backend/resource_manager.cpp:593-801 record
// vkCmdBuildAccelerationStructuresKHR; the BVH itself is the driver's).  Same semantics as csrc/bvh_build.cpp: one geometry per
// Primitive with its transform baked in (:608-617), all opaque, two-sided, one identity instance => a world-space triangle soup.
//
// Triangles in flat (primitive-major) order -> world-space Moeller-Trumbore records with the host builder's arithmetic (this file is
// compiled without FMA contraction, so the records are bit-identical to the host's) -> 30-bit Morton codes of the box centres ->
// radix sort (rocPRIM through hipCUB) -> agglomerative clustering along that order (PLOC: every round each cluster merges with the
// neighbour that makes the smallest box, if the choice is mutual) -> subtrees of <= leaf_tris triangles collapsed into leaves, the
// triangles in depth-first order -> the 64-byte (lo, hi) nodes -> the derived forms (centre / half extent, 48-byte, half precision)
// with the host's formulas.  Root = node 0, parents before children.
//
// (A first version split the Morton order top-down (Karras 2012).  On sponza_proc its tree cost 39 node visits and 12 triangle tests
// per ray against the host SAH tree's 8.3 and 0.85 -- the walls' big triangles sat deep inside subtrees of small ones -- and
// bistro_proc's came out deeper than the walkers' stacks.  Clustering bottom-up by box surface keeps big triangles near the top.)
// The tree differs from the host's; any-hit results do not depend on the tree and closest hits commit by (t, flat index), so images
// are the same bit for bit with either builder (tests/test_gpu_fuzz.py).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vhr_internal.hpp"

namespace vhr {
namespace {

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t ordered(float f) {          // monotone float -> uint32 (for atomicMin / atomicMax)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unordered(uint32_t u) {
    const uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// ---- 1. triangles: world-space records (bvh_build.cpp:359-381) + boxes + the bounds of the box centres ----
__global__ __launch_bounds__(256) void k0_triangles_kernel(const vhr_vertex *__restrict__ vertices, const uint32_t *__restrict__ indices,
                                                           const vhr_primitive *__restrict__ primitives, const uint32_t *__restrict__ tri_prefix,
                                                           uint32_t primitive_count, uint32_t n, BvhTri *__restrict__ tris, Box6 *__restrict__ boxes,
                                                           uint32_t *__restrict__ centre_bounds) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    float c[3] = { 0.0f, 0.0f, 0.0f };
    if (t < n) {
        uint32_t lo = 0, hi = primitive_count;                   // the primitive whose triangle range holds t: last p with prefix[p] <= t
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (tri_prefix[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t p = lo, local = t - tri_prefix[p];
        const vhr_primitive &pr = primitives[p];
        const float *m = pr.transform;
        float w[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float *v = vertices[pr.vertex_offset + indices[pr.index_offset + 3u * local + uint32_t(k)]].pos;
            // transform * vec4(pos, 1), columns accumulated left to right, no contraction
            w[k][0] = ((m[0] * v[0] + m[4] * v[1]) + m[8] * v[2]) + m[12];
            w[k][1] = ((m[1] * v[0] + m[5] * v[1]) + m[9] * v[2]) + m[13];
            w[k][2] = ((m[2] * v[0] + m[6] * v[1]) + m[10] * v[2]) + m[14];
        }
        BvhTri tri;
        Box6 b;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            tri.v0[a] = w[0][a];
            tri.e1[a] = w[1][a] - w[0][a];
            tri.e2[a] = w[2][a] - w[0][a];
            const float p1 = tri.v0[a] + tri.e1[a], p2 = tri.v0[a] + tri.e2[a];      // the box of what the walkers intersect (bvh_build.cpp:402)
            b.lo[a] = fminf(fminf(tri.v0[a], p1), p2);
            b.hi[a] = fmaxf(fmaxf(tri.v0[a], p1), p2);
            c[a] = 0.5f * (b.lo[a] + b.hi[a]);
        }
        tri.prim = p;
        tri.tri = local;
        tri.flat = t;
        tris[t] = tri;
        boxes[t] = b;
    }
    // bounds of the centres: wave reduction, then one atomic pair per axis per wave
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = t < n ? c[a] : 3.0e38f, mx = t < n ? c[a] : -3.0e38f;
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&centre_bounds[a], ordered(mn)); atomicMax(&centre_bounds[3 + a], ordered(mx)); }
    }
}

// ---- 2. Morton codes of the box centres: 10 bits per axis ----
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void k0_morton_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ centre_bounds, uint32_t n,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = unordered(centre_bounds[a]), hi = unordered(centre_bounds[3 + a]);
        const float c = 0.5f * (boxes[t].lo[a] + boxes[t].hi[a]);
        const float ext = hi - lo;
        const float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
        const uint32_t q = uint32_t(fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f));
        code |= spread10(q) << (2 - a);
    }
    keys[t] = code;
    vals[t] = t;
}

// ---- 3. clustering: PLOC (parallel locally-ordered clustering, Meister & Bittner 2018).  The clusters start as the single triangles
// in Morton order; every round each cluster looks kSearch neighbours up and down that order for the one whose union with it has the
// smallest surface, mutual choices merge into a new node, the survivors are compacted (order kept), until one cluster is left.  A big
// triangle is nobody's cheapest partner, so it stays single until the clusters around it have grown to its size: it ends up high in
// the tree instead of bloating the boxes of a deep subtree -- the failure of a plain Morton-split tree on architectural scenes.
// Node ids: leaves 0 .. n-1 (position in the sorted order), inner nodes n + creation rank (deterministic: ranks come from a scan).
constexpr int kSearch = 16;

__device__ __forceinline__ float union_half_area(const Box6 &a, const Box6 &b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]),
                dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(256) void k0_init_clusters_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ sorted_vals, uint32_t n, Box6 *__restrict__ cbox,
                                                               uint32_t *__restrict__ cid, Box6 *__restrict__ node_box, uint32_t *__restrict__ node_size) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const Box6 b = boxes[sorted_vals[k]];
    cbox[k] = b;
    cid[k] = k;
    node_box[k] = b;
    node_size[k] = 1u;
}
__global__ __launch_bounds__(256) void k0_nearest_kernel(const Box6 *__restrict__ cbox, uint32_t count, uint32_t *__restrict__ nearest, uint32_t search) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const Box6 mine = cbox[i];
    const uint32_t j0 = i > search ? i - search : 0u, j1 = min(count - 1u, i + search);
    float best = 3.0e38f;
    uint32_t bestj = i;
    for (uint32_t j = j0; j <= j1; ++j) {
        if (j == i) continue;
        const float a = union_half_area(mine, cbox[j]);
        if (a < best) { best = a; bestj = j; }              // (ascending j: ties keep the lower index, on both sides of a pair)
    }
    nearest[i] = bestj;
}
// flags: keep[i] = the cluster survives the round (it does not merge, or it is the lower index of a merging pair); merge[i] = it is that lower index
__global__ __launch_bounds__(256) void k0_mark_kernel(const uint32_t *__restrict__ nearest, uint32_t count, uint32_t *__restrict__ keep, uint32_t *__restrict__ merge) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint32_t j = nearest[i];
    const bool mutual = j != i && nearest[j] == i;
    keep[i] = (!mutual || i < j) ? 1u : 0u;
    merge[i] = (mutual && i < j) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k0_merge_kernel(const Box6 *__restrict__ cbox, const uint32_t *__restrict__ cid, const uint32_t *__restrict__ nearest,
                                                       const uint32_t *__restrict__ keep, const uint32_t *__restrict__ keep_pos, const uint32_t *__restrict__ merge,
                                                       const uint32_t *__restrict__ merge_pos, uint32_t count, uint32_t next_node, Box6 *__restrict__ cbox_out,
                                                       uint32_t *__restrict__ cid_out, int2 *__restrict__ node_children, uint32_t *__restrict__ node_parent,
                                                       Box6 *__restrict__ node_box, uint32_t *__restrict__ node_size) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count || !keep[i]) return;
    Box6 b = cbox[i];
    uint32_t id = cid[i];
    if (merge[i]) {
        const uint32_t j = nearest[i], node = next_node + merge_pos[i];
        const Box6 o = cbox[j];
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], o.lo[a]); b.hi[a] = fmaxf(b.hi[a], o.hi[a]); }
        node_children[node] = int2{ int(id), int(cid[j]) };
        node_parent[id] = node;
        node_parent[cid[j]] = node;
        node_box[node] = b;
        node_size[node] = node_size[id] + node_size[cid[j]];
        id = node;
    }
    cbox_out[keep_pos[i]] = b;
    cid_out[keep_pos[i]] = id;
}

// ---- 4. the finished tree -> the walkers' layout.  A subtree of at most `leaf_tris` triangles becomes a leaf (its triangles are
// consecutive in the depth-first order of the tree, which is the order `tris` gets); the inner nodes that remain are numbered in
// reverse creation order -- the root, made last, is node 0, and every parent precedes its children. ----
// each triangle climbs to the root: its depth-first position = the sizes of the left siblings passed on the way; the depth of its leaf
__global__ __launch_bounds__(256) void k0_positions_kernel(const int2 *__restrict__ node_children, const uint32_t *__restrict__ node_parent,
                                                           const uint32_t *__restrict__ node_size, uint32_t n, uint32_t root, uint32_t leaf_tris,
                                                           uint32_t *__restrict__ position, uint32_t *__restrict__ max_depth) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    uint32_t pos = 0, depth = 0, node = k;
    while (node != root) {
        const uint32_t parent = node_parent[node];
        const int2 ch = node_children[parent];
        if (uint32_t(ch.y) == node) pos += node_size[uint32_t(ch.x)];
        if (node_size[parent] > leaf_tris) ++depth;          // an inner node that stays one
        node = parent;
    }
    position[k] = pos;
    atomicMax(max_depth, depth);
}
__global__ __launch_bounds__(256) void k0_place_triangles_kernel(const BvhTri *__restrict__ tris_flat, const uint32_t *__restrict__ sorted_vals,
                                                                 const uint32_t *__restrict__ position, uint32_t n, BvhTri *__restrict__ tris_out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < n) tris_out[position[k]] = tris_flat[sorted_vals[k]];
}
// first triangle (depth-first position) of every node: a leaf's own position; an inner node's = its left child's, resolved by
// descending (<= leaf_tris - 1 steps matter only for the collapsed ones, but any node may ask)
__device__ __forceinline__ uint32_t first_triangle(const int2 *node_children, const uint32_t *position, uint32_t n, uint32_t node) {
    while (node >= n) node = uint32_t(node_children[node].x);
    return position[node];
}
__device__ __forceinline__ void set_child(BvhNode &node, int which, const Box6 &b, int32_t link) {
    float *dst = which == 0 ? node.box0 : node.box1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pad = 1e-3f + 1e-5f * fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a]));       // bvh_build.cpp:292-299
        dst[2 * a] = b.lo[a] - pad;
        dst[2 * a + 1] = b.hi[a] + pad;
    }
    (which == 0 ? node.child0 : node.child1) = link;
}
__global__ __launch_bounds__(256) void k0_emit_kernel(const int2 *__restrict__ node_children, const Box6 *__restrict__ node_box, const uint32_t *__restrict__ node_size,
                                                      const uint32_t *__restrict__ kept_rank, const uint32_t *__restrict__ position, uint32_t n, uint32_t total_nodes,
                                                      uint32_t kept, uint32_t leaf_tris, BvhNode *__restrict__ nodes) {
    const uint32_t node = n + blockIdx.x * 256u + threadIdx.x;
    if (node >= total_nodes || node_size[node] <= leaf_tris) return;
    const int2 ch = node_children[node];
    auto link_of = [&](uint32_t c) -> int32_t {
        const uint32_t size = node_size[c];
        if (size > leaf_tris) return int32_t(kept - 1u - kept_rank[c - n]);                 // reverse creation order
        return ~int32_t((first_triangle(node_children, position, n, c) << 2) | (size - 1u));
    };
    BvhNode out{};
    set_child(out, 0, node_box[uint32_t(ch.x)], link_of(uint32_t(ch.x)));
    set_child(out, 1, node_box[uint32_t(ch.y)], link_of(uint32_t(ch.y)));
    nodes[kept - 1u - kept_rank[node - n]] = out;
}
__global__ __launch_bounds__(256) void k0_kept_flags_kernel(const uint32_t *__restrict__ node_size, uint32_t n, uint32_t total_nodes, uint32_t leaf_tris, uint32_t *__restrict__ flags) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (n + k < total_nodes) flags[k] = node_size[n + k] > leaf_tris ? 1u : 0u;
}

// ---- 6. the derived node forms, with the host's formulas (bvh_build.cpp finalize_ch / finalize16) ----
__device__ __forceinline__ uint32_t upper16(float h) {
    uint32_t bits = __float_as_uint(h);
    if (h > 0.0f && (bits & 0xffffu)) bits += 0x10000u;
    return bits >> 16;
}
__global__ __launch_bounds__(256) void k0_forms_kernel(const BvhNode *__restrict__ nodes, uint32_t count, float cx, float cy, float cz, BvhNodeCH *__restrict__ nodes_ch,
                                                       BvhNode48 *__restrict__ nodes48, BvhNode16 *__restrict__ nodes16) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= count) return;
    const BvhNode nd = nodes[k];
    const float centre[3] = { cx, cy, cz };
    BvhNodeCH c{};
    BvhNode16 h16{};
    const float inf = __builtin_inff();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float *box = which == 0 ? nd.box0 : nd.box1;
        float *hdst = which == 0 ? c.h0 : c.h1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = box[2 * a], hi = box[2 * a + 1];
            float cc = 0.0f, hh = -1.0f;
            if (lo <= hi) {
                cc = 0.5f * lo + 0.5f * hi;
                hh = fmaxf(hi - cc, cc - lo);
                hh += (fabsf(cc) + hh) * 2.4e-7f;
                while (double(cc) - double(hh) > double(lo) || double(cc) + double(hh) < double(hi)) hh = nextafterf(hh, inf);
            }
            (a == 0 ? c.cx : a == 1 ? c.cy : c.cz)[which] = cc;
            hdst[a] = hh;
            // the 32-byte form (BvhNode16): centre relative to the scene centre and half extent as halves, no subnormals; centre + c +- h
            // contains [lo, hi] in exact arithmetic with a few fp32 ulp to spare (the host's half_centre_extent; an overflow leaves inf,
            // which the host sees in the downloaded nodes and then keeps the walkers on the 48-byte form)
            uint16_t cb = 0, hb = 0xbc00;
            if (lo <= hi) {
                const double mid = 0.5 * double(lo) + 0.5 * double(hi) - double(centre[a]);
                cb = __half_as_ushort(__float2half_rn(float(mid)));
                if (((cb >> 10) & 31) == 0) cb = 0;
                const double cv = double(centre[a]) + double(__half2float(__ushort_as_half(cb)));
                double need = fmax(double(hi) - cv, cv - double(lo));
                need += (fabs(double(centre[a])) + fabs(cv - double(centre[a])) + need) * 4.8e-7 + 1e-30;
                float nf = float(need);
                if (double(nf) < need) nf = nextafterf(nf, inf);
                hb = __half_as_ushort(__float2half_ru(nf));
                if (((hb >> 10) & 31) == 0) hb = 0x0400;
            }
            h16.c[2 * a + which] = cb;
            h16.h[2 * a + which] = hb;
        }
    }
    c.child0 = nd.child0; c.child1 = nd.child1;
    h16.child0 = nd.child0 >= 0 ? nd.child0 * int32_t(sizeof(BvhNode16)) : nd.child0;
    h16.child1 = nd.child1 >= 0 ? nd.child1 * int32_t(sizeof(BvhNode16)) : nd.child1;
    nodes_ch[k] = c;
    nodes16[k] = h16;
    BvhNode48 n48{};
    n48.cx[0] = c.cx[0]; n48.cx[1] = c.cx[1]; n48.cy[0] = c.cy[0]; n48.cy[1] = c.cy[1]; n48.cz[0] = c.cz[0]; n48.cz[1] = c.cz[1];
    n48.hp[0] = (upper16(c.h0[0]) << 16) | upper16(c.h0[1]);
    n48.hp[1] = (upper16(c.h0[2]) << 16) | upper16(c.h1[0]);
    n48.hp[2] = (upper16(c.h1[1]) << 16) | upper16(c.h1[2]);
    n48.child0 = c.child0 >= 0 ? c.child0 * int32_t(sizeof(BvhNode48)) : c.child0;
    n48.child1 = c.child1 >= 0 ? c.child1 * int32_t(sizeof(BvhNode48)) : c.child1;
    nodes48[k] = n48;
}

// bounds of all child boxes (the scene centre of the half-precision form): one reduction over the nodes
__global__ __launch_bounds__(256) void k0_node_bounds_kernel(const BvhNode *__restrict__ nodes, uint32_t count, uint32_t *__restrict__ bounds) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (k < count) {
        const BvhNode nd = nodes[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (nd.box0[2 * a] <= nd.box0[2 * a + 1]) { mn[a] = fminf(mn[a], nd.box0[2 * a]); mx[a] = fmaxf(mx[a], nd.box0[2 * a + 1]); }
            if (nd.box1[2 * a] <= nd.box1[2 * a + 1]) { mn[a] = fminf(mn[a], nd.box1[2 * a]); mx[a] = fmaxf(mx[a], nd.box1[2 * a + 1]); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off)); }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&bounds[a], ordered(mn[a])); atomicMax(&bounds[3 + a], ordered(mx[a])); }
    }
}

struct Scratch {            // device allocations of one build, freed together
    std::vector<void *> ptrs;
    template <typename T>
    hipError_t alloc(T **p, size_t count) {
        *p = nullptr;
        if (!count) return hipSuccess;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    ~Scratch() { for (void *p : ptrs) hipFree(p); }
};

}  // namespace

#define K0_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, std::string("device K0: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

// Builds the tree from the scene arrays already on the device (ctx->d_vertices / d_indices / d_primitives) into ctx->d_nodes,
// d_nodes_ch, d_nodes48, d_nodes16, d_tris.  Returns VHR_OK, or VHR_ERROR_OUT_OF_SLOTS when the tree is deeper than the walkers'
// stacks (kMaxBvhDepth) -- the caller then falls back to the host builder.  `tri_prefix`: first flat triangle of every primitive.
int device_build_bvh(vhr_context *ctx, const std::vector<uint32_t> &tri_prefix, uint32_t total_tris, int leaf_tris_in) {
    const uint32_t n = total_tris, leaf_tris = uint32_t(std::max(1, std::min(kMaxLeafTris, leaf_tris_in)));
    if (n <= leaf_tris) return VHR_ERROR_OUT_OF_SLOTS;            // (a scene that fits one leaf: the host builder's special case)
    hipStream_t s = ctx->stream;
    Scratch tmp;
    const uint32_t total_cap = 2u * n;                            // node ids: n leaves + at most n - 1 inner nodes
    uint32_t *d_prefix, *d_bounds, *d_keys, *d_vals, *d_keys2, *d_vals2, *d_cid[2], *d_nearest, *d_keep, *d_keep_pos, *d_merge, *d_merge_pos, *d_parent, *d_size,
             *d_position, *d_counts;
    BvhTri *d_tris_flat;
    Box6 *d_boxes, *d_cbox[2], *d_node_box;
    int2 *d_children;
    K0_TRY(tmp.alloc(&d_prefix, tri_prefix.size()));
    K0_TRY(tmp.alloc(&d_bounds, 12));
    K0_TRY(tmp.alloc(&d_counts, 4));
    K0_TRY(tmp.alloc(&d_keys, n)); K0_TRY(tmp.alloc(&d_vals, n)); K0_TRY(tmp.alloc(&d_keys2, n)); K0_TRY(tmp.alloc(&d_vals2, n));
    K0_TRY(tmp.alloc(&d_tris_flat, n)); K0_TRY(tmp.alloc(&d_boxes, n));
    for (int k = 0; k < 2; ++k) { K0_TRY(tmp.alloc(&d_cbox[k], n)); K0_TRY(tmp.alloc(&d_cid[k], n)); }
    K0_TRY(tmp.alloc(&d_nearest, n)); K0_TRY(tmp.alloc(&d_keep, n)); K0_TRY(tmp.alloc(&d_keep_pos, n)); K0_TRY(tmp.alloc(&d_merge, n)); K0_TRY(tmp.alloc(&d_merge_pos, n));
    K0_TRY(tmp.alloc(&d_parent, total_cap)); K0_TRY(tmp.alloc(&d_size, total_cap)); K0_TRY(tmp.alloc(&d_node_box, total_cap)); K0_TRY(tmp.alloc(&d_children, total_cap));
    K0_TRY(tmp.alloc(&d_position, n));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_tris), sizeof(BvhTri) * n));

    K0_TRY(hipMemcpyAsync(d_prefix, tri_prefix.data(), tri_prefix.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    const uint32_t init_bounds[12] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    K0_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));
    K0_TRY(hipMemsetAsync(d_counts, 0, 4 * sizeof(uint32_t), s));
    const dim3 block(256);
    auto grid = [](uint32_t count) { return dim3((count + 255u) / 256u); };
    hipLaunchKernelGGL(k0_triangles_kernel, grid(n), block, 0, s, ctx->d_vertices, ctx->d_indices, ctx->d_primitives, d_prefix, uint32_t(tri_prefix.size()), n,
                       d_tris_flat, d_boxes, d_bounds);
    hipLaunchKernelGGL(k0_morton_kernel, grid(n), block, 0, s, d_boxes, d_bounds, n, d_keys, d_vals);
    size_t sort_bytes = 0, scan_bytes = 0;
    K0_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys, d_keys2, d_vals, d_vals2, int(n), 0, 30, s));
    K0_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_keep, d_keep_pos, int(n), s));
    char *d_work;
    K0_TRY(tmp.alloc(&d_work, std::max(sort_bytes, scan_bytes)));
    K0_TRY(hipcub::DeviceRadixSort::SortPairs(d_work, sort_bytes, d_keys, d_keys2, d_vals, d_vals2, int(n), 0, 30, s));
    hipLaunchKernelGGL(k0_init_clusters_kernel, grid(n), block, 0, s, d_boxes, d_vals2, n, d_cbox[0], d_cid[0], d_node_box, d_size);
    // ---- the clustering rounds (the cluster count comes back to the host once per round: two words) ----
    uint32_t count = n, next_node = n;
    int cur = 0;
    const char *search_env = std::getenv("VHR_K0_SEARCH");          // (experiments: the search radius along the Morton order)
    const uint32_t search = search_env ? uint32_t(std::max(1, std::atoi(search_env))) : uint32_t(kSearch);
    for (int round = 0; count > 1 && round < 4096; ++round) {
        hipLaunchKernelGGL(k0_nearest_kernel, grid(count), block, 0, s, d_cbox[cur], count, d_nearest, search);
        hipLaunchKernelGGL(k0_mark_kernel, grid(count), block, 0, s, d_nearest, count, d_keep, d_merge);
        K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_work, scan_bytes, d_keep, d_keep_pos, int(count), s));
        K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_work, scan_bytes, d_merge, d_merge_pos, int(count), s));
        hipLaunchKernelGGL(k0_merge_kernel, grid(count), block, 0, s, d_cbox[cur], d_cid[cur], d_nearest, d_keep, d_keep_pos, d_merge, d_merge_pos, count, next_node,
                           d_cbox[cur ^ 1], d_cid[cur ^ 1], d_children, d_parent, d_node_box, d_size);
        uint32_t last[2][2];            // (position, flag) of the last cluster, for keep and merge: total = position + flag
        K0_TRY(hipMemcpyAsync(&last[0][0], d_keep_pos + count - 1, 4, hipMemcpyDeviceToHost, s));
        K0_TRY(hipMemcpyAsync(&last[0][1], d_keep + count - 1, 4, hipMemcpyDeviceToHost, s));
        K0_TRY(hipMemcpyAsync(&last[1][0], d_merge_pos + count - 1, 4, hipMemcpyDeviceToHost, s));
        K0_TRY(hipMemcpyAsync(&last[1][1], d_merge + count - 1, 4, hipMemcpyDeviceToHost, s));
        K0_TRY(hipStreamSynchronize(s));
        const uint32_t kept = last[0][0] + last[0][1], merged = last[1][0] + last[1][1];
        if (merged == 0 || kept + merged != count) return ctx->fail(VHR_ERROR_DEVICE, "device K0: a clustering round made no progress");
        next_node += merged;
        count = kept;
        cur ^= 1;
    }
    if (count != 1 || next_node != 2u * n - 1u) return ctx->fail(VHR_ERROR_DEVICE, "device K0: the clustering did not end in one tree");
    const uint32_t total_nodes = next_node, root = next_node - 1u;
    // ---- layout ----
    hipLaunchKernelGGL(k0_positions_kernel, grid(n), block, 0, s, d_children, d_parent, d_size, n, root, leaf_tris, d_position, d_counts);
    hipLaunchKernelGGL(k0_place_triangles_kernel, grid(n), block, 0, s, d_tris_flat, d_vals2, d_position, n, ctx->d_tris);
    const uint32_t n_inner_all = total_nodes - n;
    uint32_t *d_kept = d_keep, *d_kept_rank = d_keep_pos;        // (the rounds are over: their flag / scan arrays, n entries >= n - 1)
    hipLaunchKernelGGL(k0_kept_flags_kernel, grid(n_inner_all), block, 0, s, d_size, n, total_nodes, leaf_tris, d_kept);
    K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_work, scan_bytes, d_kept, d_kept_rank, int(n_inner_all), s));
    uint32_t last_rank = 0, last_flag = 0, h_depth = 0;
    K0_TRY(hipMemcpyAsync(&last_rank, d_kept_rank + n_inner_all - 1, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipMemcpyAsync(&last_flag, d_kept + n_inner_all - 1, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipMemcpyAsync(&h_depth, d_counts, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    const uint32_t n_inner = last_rank + last_flag;
    if (n_inner == 0 || !last_flag) return VHR_ERROR_OUT_OF_SLOTS;           // (the whole scene collapsed into one leaf)
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes), sizeof(BvhNode) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes_ch), sizeof(BvhNodeCH) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes48), sizeof(BvhNode48) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes16), sizeof(BvhNode16) * n_inner));
    hipLaunchKernelGGL(k0_emit_kernel, grid(n_inner_all), block, 0, s, d_children, d_node_box, d_size, d_kept_rank, d_position, n, total_nodes, n_inner, leaf_tris, ctx->d_nodes);
    hipLaunchKernelGGL(k0_node_bounds_kernel, grid(n_inner), block, 0, s, ctx->d_nodes, n_inner, d_bounds + 6);
    uint32_t h_bounds[12];
    K0_TRY(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    K0_TRY(hipGetLastError());
    for (int a = 0; a < 3; ++a) {
        const float lo = unordered(h_bounds[6 + a]), hi = unordered(h_bounds[9 + a]);
        ctx->bvh_centre[a] = lo <= hi ? 0.5f * (lo + hi) : 0.0f;
    }
    hipLaunchKernelGGL(k0_forms_kernel, grid(n_inner), block, 0, s, ctx->d_nodes, n_inner, ctx->bvh_centre[0], ctx->bvh_centre[1], ctx->bvh_centre[2], ctx->d_nodes_ch,
                       ctx->d_nodes48, ctx->d_nodes16);
    K0_TRY(hipStreamSynchronize(s));
    K0_TRY(hipGetLastError());
    ctx->node_count = n_inner;
    ctx->tri_count = n;
    ctx->bvh_depth = h_depth;                  // inner nodes on the longest root-to-leaf path
    if (std::getenv("VHR_K0_TRACE")) std::fprintf(stderr, "K0 device: %u triangles, %u inner nodes, depth %u\n", n, n_inner, h_depth);
    if (h_depth > uint32_t(kMaxBvhDepth)) return VHR_ERROR_OUT_OF_SLOTS;
    return VHR_OK;
}

}  // namespace vhr
