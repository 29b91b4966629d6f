// K0 on the device (option "bvh_builder" 1, the default): the acceleration structure built where the reference builds it -- on the
// GPU (ResourceManager::UpdateBLAS / UpdateTLAS, /root/reference/src/rendering_backend/resource_manager.cpp:593-801 record
// vkCmdBuildAccelerationStructuresKHR with PREFER_FAST_TRACE; the BVH itself is the driver's).  Same semantics as csrc/bvh_build.cpp:
// one geometry per Primitive with its transform baked in (:608-617), all opaque, two-sided, one identity instance => a world-space
// triangle soup.
//
// The algorithm is the host builder's -- binned SAH, 16 bins per axis over the bounds of the triangles' box centres, cost =
// area_left * n_left + area_right * n_right, split until one triangle is left (the layout stage then turns every subtree of
// <= leaf_tris triangles into a leaf, which is where the host stops splitting) -- run level by level on the GPU:
//   * triangles in flat (primitive-major) order -> world-space Moeller-Trumbore records with the host builder's arithmetic (this file
//     is compiled without FMA contraction, so the records are bit-identical to the host's) + boxes + the bounds of the box centres;
//   * nodes of more than kSmallNode triangles: one pass per level over all triangle positions.  Bin (every active node a workgroup's
//     256 positions touch gets bins in LDS, flushed with one look-first atomic per touched word), one thread per node sweeps its bins
//     and chooses the plane, a device-wide scan of the "goes left" flags gives every triangle its new position, a scatter moves it.
//     A node's triangles stay a contiguous range of `order`, so the finished `order` is the tree's depth-first order;
//   * subtrees of <= kSmallNode triangles: one wave each, its triangles in LDS, an explicit stack, bins in LDS, the sweep spread over
//     the lanes;
//   * layout: subtrees of <= leaf_tris triangles collapse into leaves, the inner nodes that remain are numbered breadth first like the
//     host's (root = node 0, parents before children) -> the 64-byte (lo, hi) nodes -> the derived forms (centre / half extent,
//     48-byte, half precision) with the host's formulas.
// Node ids while building: leaf k = the triangle at position k of the final order, inner nodes n + creation rank (a small subtree
// reserves the ids of all its inner nodes when it is made: one returning atomic per subtree, not per node -- 1.4 M of those on one
// counter cost bistro_proc's build 42 ms).
//
// On sponza_proc and bistro_proc the tree has the host tree's node count, depth and visits per ray; the triangle order inside the
// leaves and the leaves' order in memory differ (depth-first here, by treelets on the host), which costs the any-hit launch <= 1 %.
// Any-hit results do not depend on the tree and closest hits commit by (t, flat index), so images are the same bit for bit with either
// builder (tests/test_gpu_fuzz.py).  r4, MI355X: 7 ms for sponza_proc's 258 k triangles, 22 ms for bistro_proc's 2.9 M (host: 50 /
// 610 ms on the box's cores); a bottom-up clustering builder (PLOC) that stood here until r4 took 7 / 20 ms and its trees cost the
// Raytrace Pass +10 / +22 % (profiles/r4_k0_device.txt).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "vhr_internal.hpp"
#include "presplit.hpp"
#include "bvh_frame.hpp"

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
// min / max into a word many waves aim at: look first (most values no longer move it), then the atomic
__device__ __forceinline__ void atomic_min_checked(uint32_t *p, uint32_t v) {
    if (v < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(p, v);
}
__device__ __forceinline__ void atomic_max_checked(uint32_t *p, uint32_t v) {
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}
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
        if ((threadIdx.x & 63u) == 0u) { atomic_min_checked(&centre_bounds[a], ordered(mn)); atomic_max_checked(&centre_bounds[3 + a], ordered(mx)); }
    }
}

// ---- 2. the finished tree -> the walkers' layout.  A subtree of at most `leaf_tris` triangles becomes a leaf (its triangles are
// consecutive in the depth-first order of the tree, which is the order `tris` gets); the inner nodes that remain are numbered
// breadth first (k0_bfs_keys_kernel below). ----
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
                                                      uint32_t leaf_tris, BvhNode *__restrict__ nodes) {
    const uint32_t node = n + blockIdx.x * 256u + threadIdx.x;
    if (node >= total_nodes || node_size[node] <= leaf_tris) return;
    const int2 ch = node_children[node];
    auto link_of = [&](uint32_t c) -> int32_t {
        const uint32_t size = node_size[c];
        if (size > leaf_tris) return int32_t(kept_rank[c - n]);
        return ~int32_t((first_triangle(node_children, position, n, c) << 2) | (size - 1u));
    };
    BvhNode out{};
    set_child(out, 0, node_box[uint32_t(ch.x)], link_of(uint32_t(ch.x)));
    set_child(out, 1, node_box[uint32_t(ch.y)], link_of(uint32_t(ch.y)));
    nodes[kept_rank[node - n]] = out;
}
__global__ __launch_bounds__(256) void k0_kept_flags_kernel(const uint32_t *__restrict__ node_size, uint32_t n, uint32_t total_nodes, uint32_t leaf_tris, uint32_t *__restrict__ flags) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (n + k < total_nodes) flags[k] = node_size[n + k] > leaf_tris ? 1u : 0u;
}

// ---- 3. the derived node forms, with the host's formulas (bvh_build.cpp finalize_ch / finalize16) ----
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

// The host's self-checks of the derived node forms (bvh_build.cpp check_node_forms + nodes16_in_range), on the device: every form must CONTAIN
// the (lo, hi) boxes in exact arithmetic (doubles hold every value involved exactly).  out[0] boxes checked, out[1] centre / half-extent boxes
// that do not contain theirs, out[2] 48-byte boxes that do not contain the centre / half-extent box (or links that differ), out[3] 32-byte
// boxes that do not contain theirs (or links that differ), out[4] halves of the 32-byte form outside the range its walker reads (inf / NaN /
// subnormal): the caller keeps out[3] only if out[4] == 0, like the host, which checks the 32-byte form only where it is in use.
__device__ __forceinline__ double half_value_d(uint32_t h) {
    const int e = int((h >> 10) & 31u), m = int(h & 1023u);
    const double v = e == 0 ? ldexp(double(m), -24) : (e == 31 ? (m ? __builtin_nan("") : __builtin_inf()) : ldexp(double(1024 + m), e - 25));
    return (h & 0x8000u) ? -v : v;
}
__global__ __launch_bounds__(256) void k0_check_forms_kernel(const BvhNode *__restrict__ nodes, const BvhNodeCH *__restrict__ nodes_ch, const BvhNode48 *__restrict__ nodes48,
                                                             const BvhNode16 *__restrict__ nodes16, uint32_t count, float cx, float cy, float cz,
                                                             unsigned long long *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    uint32_t bad[5] = { 0u, 0u, 0u, 0u, 0u };
    if (k < count) {
        const BvhNode nd = nodes[k];
        const BvhNodeCH ch = nodes_ch[k];
        const BvhNode48 n48 = nodes48[k];
        const BvhNode16 n16 = nodes16[k];
        const double centre[3] = { double(cx), double(cy), double(cz) };
        auto upper = [](uint32_t w16) { return double(__uint_as_float(w16 << 16)); };
        const double h48[6] = { upper(n48.hp[0] >> 16), upper(n48.hp[0] & 0xffffu), upper(n48.hp[1] >> 16), upper(n48.hp[1] & 0xffffu), upper(n48.hp[2] >> 16), upper(n48.hp[2] & 0xffffu) };
        for (int i = 0; i < 6; ++i) {
            const uint32_t ec = (n16.c[i] >> 10) & 31u, eh = (n16.h[i] >> 10) & 31u;
            if (ec == 31u || (ec == 0u && n16.c[i] != 0) || eh == 31u || eh == 0u) ++bad[4];
        }
        for (int which = 0; which < 2; ++which) {
            const float *box = which == 0 ? nd.box0 : nd.box1;
            const float *hh = which == 0 ? ch.h0 : ch.h1;
            ++bad[0];
            for (int a = 0; a < 3; ++a) {
                const double lo = box[2 * a], hi = box[2 * a + 1];
                const double c = (a == 0 ? ch.cx : a == 1 ? ch.cy : ch.cz)[which], h = hh[a];
                const double c48 = (a == 0 ? n48.cx : a == 1 ? n48.cy : n48.cz)[which], hw = h48[3 * which + a];
                const uint32_t cb = n16.c[2 * a + which], hb = n16.h[2 * a + which];
                const double c16 = centre[a] + half_value_d(cb), h16 = half_value_d(hb);
                const bool normal16 = (cb == 0u || (((cb >> 10) & 31u) != 0u && ((cb >> 10) & 31u) != 31u)) && ((hb >> 10) & 31u) != 0u && ((hb >> 10) & 31u) != 31u;
                if (!(lo <= hi)) {                           // an absent child: never entered in any form
                    if (!(h < 0.0)) ++bad[1];
                    if (!(hw < 0.0)) ++bad[2];
                    if (!(h16 < 0.0)) ++bad[3];
                    continue;
                }
                if (c - h > lo || c + h < hi) ++bad[1];
                if (c48 != c || hw < h) ++bad[2];
                if (!normal16 || !(c16 - h16 <= lo) || !(c16 + h16 >= hi)) ++bad[3];
            }
        }
        auto as48 = [](int32_t link) { return link >= 0 ? link * int32_t(sizeof(BvhNode48)) : link; };
        if (n48.child0 != as48(nd.child0) || n48.child1 != as48(nd.child1) || ch.child0 != nd.child0 || ch.child1 != nd.child1) ++bad[2];
        auto as16 = [](int32_t link) { return link >= 0 ? link * int32_t(sizeof(BvhNode16)) : link; };
        if (n16.child0 != as16(nd.child0) || n16.child1 != as16(nd.child1)) ++bad[3];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        uint32_t v = bad[i];
        for (int off = 32; off > 0; off >>= 1) v += uint32_t(__shfl_xor(int(v), off));
        if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&out[i], (unsigned long long)v);
    }
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
        if ((threadIdx.x & 63u) == 0u) { atomic_min_checked(&bounds[a], ordered(mn[a])); atomic_max_checked(&bounds[3 + a], ordered(mx[a])); }
    }
}


// ---- 4. the tree: binned SAH, top-down ----
constexpr uint32_t kSmallNode = 64;
constexpr uint32_t kNoNode = 0xffffffffu;
constexpr int kBinWords = 13;                 // per bin: count, triangle-box lo / hi (6), box-centre lo / hi (6) -- floats as ordered uints

struct SahNode {                              // an active node of the current level
    uint32_t id, first, count, depth;
    float clo[3], chi[3];                     // bounds of its triangles' box centres
};
struct SahSplit {                             // what the sweep decided for it
    int32_t axis;                             // -1: no plane separates anything -> the range is cut in half as it stands
    int32_t bin;
    uint32_t n_left;
    uint32_t next_index[2];                   // the children's index in the NEXT level's active list, kNoNode if not there (small or single)
    uint32_t by_position;                     // axis < 0: children boxes come from the scatter pass
    uint32_t child_id[2];
};
struct SahCounters { uint32_t inner, next_active, small_roots, error; };

__device__ __forceinline__ int sah_bin(float c, float lo, float ext) {                 // bvh_build.cpp: int((c - lo) * (16 / ext)) clamped to 0..15
    const float scale = 16.0f / ext;
    return min(15, max(0, int((c - lo) * scale)));
}
__device__ __forceinline__ float half_area6(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void bins_accumulate(uint32_t *bins, int axis, int b, const Box6 &box, const float c[3]) {
    uint32_t *w = bins + (axis * 16 + b) * kBinWords;
    atomicAdd(&w[0], 1u);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        atomicMin(&w[1 + a], ordered(box.lo[a])); atomicMax(&w[4 + a], ordered(box.hi[a]));
        atomicMin(&w[7 + a], ordered(c[a])); atomicMax(&w[10 + a], ordered(c[a]));
    }
}
__global__ __launch_bounds__(256) void k0_sah_clear_bins_kernel(uint32_t *__restrict__ bins, uint32_t nodes) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= nodes * 48u * kBinWords) return;
    const uint32_t word = k % kBinWords;
    bins[k] = word == 0 ? 0u : ((word >= 1 && word <= 3) || (word >= 7 && word <= 9) ? 0xffffffffu : 0u);
}
// A workgroup's 256 positions lie in at most kBinSegments active nodes (each longer than kSmallNode, contiguous): every one of them gets
// its own bins in LDS, which then go to the node's bins in memory with one look-first atomic per touched word.
constexpr uint32_t kBinSegments = 256u / kSmallNode + 2u;
__global__ __launch_bounds__(256) void k0_sah_bin_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ order, const uint32_t *__restrict__ pos_node,
                                                         const SahNode *__restrict__ active, uint32_t n, uint32_t *__restrict__ bins) {
    __shared__ uint32_t s_bins[kBinSegments][48 * kBinWords];
    __shared__ uint32_t s_wave_starts[4], s_seg_node[kBinSegments];
    const uint32_t t = threadIdx.x, i = blockIdx.x * 256u + t, wave = t >> 6, lane = t & 63u;
    const uint32_t a = i < n ? pos_node[i] : kNoNode;
    const bool starts = a != kNoNode && (t == 0u || pos_node[i - 1u] != a);          // first position of an active node within this workgroup
    const unsigned long long start_mask = __ballot(starts);
    if (lane == 0u) s_wave_starts[wave] = uint32_t(__popcll(start_mask));
    __syncthreads();
    uint32_t seg = uint32_t(__popcll(start_mask & ((2ull << lane) - 1ull)));         // inclusive count of starts up to this position ...
    uint32_t n_segs = 0;
    for (uint32_t w = 0; w < 4u; ++w) { if (w < wave) seg += s_wave_starts[w]; n_segs += s_wave_starts[w]; }
    seg -= 1u;                                                                       // ... = the index of the segment it lies in
    if (n_segs == 0u) return;
    const uint32_t lds_segs = min(n_segs, kBinSegments);
    for (uint32_t k = t; k < lds_segs * 48u * kBinWords; k += 256u) {
        const uint32_t word = (k % (48u * kBinWords)) % kBinWords;
        (&s_bins[0][0])[k] = (word >= 1 && word <= 3) || (word >= 7 && word <= 9) ? 0xffffffffu : 0u;
    }
    if (starts && seg < kBinSegments) s_seg_node[seg] = a;
    __syncthreads();
    if (a != kNoNode) {
        const SahNode nd = active[a];
        const Box6 box = boxes[order[i]];
        float c[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) c[ax] = 0.5f * (box.lo[ax] + box.hi[ax]);
        uint32_t *dst = seg < kBinSegments ? s_bins[seg] : bins + size_t(a) * 48u * kBinWords;      // (more segments than the bound above: cannot happen, stays correct)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = nd.chi[ax] - nd.clo[ax];
            if (!(ext > 0.0f)) continue;
            bins_accumulate(dst, ax, sah_bin(c[ax], nd.clo[ax], ext), box, c);
        }
    }
    __syncthreads();
    for (uint32_t k = t; k < lds_segs * 48u * kBinWords; k += 256u) {
        const uint32_t sg = k / (48u * kBinWords), w = k % (48u * kBinWords), word = w % kBinWords, v = s_bins[sg][w];
        uint32_t *dst = bins + size_t(s_seg_node[sg]) * 48u * kBinWords + w;
        if (word == 0) { if (v) atomicAdd(dst, v); }
        else if ((word >= 1 && word <= 3) || (word >= 7 && word <= 9)) { if (v != 0xffffffffu) atomic_min_checked(dst, v); }
        else if (v != 0u) atomic_max_checked(dst, v);
    }
}
// the sweep over one node's bins (bvh_build.cpp:136-171): best (axis, bin) by strict <, axes and bins in ascending order
struct SahChoice { int axis, bin; uint32_t n_left; float lo[2][3], hi[2][3], clo[2][3], chi[2][3]; };
__device__ __forceinline__ SahChoice sah_sweep(const uint32_t *bins, const float *node_clo, const float *node_chi) {
    SahChoice best;
    best.axis = -1; best.bin = -1; best.n_left = 0;
    float best_cost = __builtin_inff();
    for (int axis = 0; axis < 3; ++axis) {
        if (!(node_chi[axis] - node_clo[axis] > 0.0f)) continue;
        const uint32_t *ab = bins + axis * 16 * kBinWords;
        float right_area[16];
        uint32_t right_count[16];
        float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        uint32_t c = 0;
        for (int b = 15; b > 0; --b) {
            const uint32_t *w = ab + b * kBinWords;
            if (w[0]) { for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], unordered(w[1 + a])); hi[a] = fmaxf(hi[a], unordered(w[4 + a])); } }
            c += w[0];
            right_area[b] = c ? half_area6(lo, hi) : 0.0f;
            right_count[b] = c;
        }
        for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
        c = 0;
        for (int b = 0; b < 15; ++b) {
            const uint32_t *w = ab + b * kBinWords;
            if (w[0]) { for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], unordered(w[1 + a])); hi[a] = fmaxf(hi[a], unordered(w[4 + a])); } }
            c += w[0];
            if (c == 0 || right_count[b + 1] == 0) continue;
            const float cost = half_area6(lo, hi) * float(c) + right_area[b + 1] * float(right_count[b + 1]);
            if (cost < best_cost) { best_cost = cost; best.axis = axis; best.bin = b; best.n_left = c; }
        }
    }
    if (best.axis >= 0) {                      // the children's boxes and centre bounds: unions of the bins on either side of the plane
        const uint32_t *ab = bins + best.axis * 16 * kBinWords;
        for (int side = 0; side < 2; ++side) {
            for (int a = 0; a < 3; ++a) { best.lo[side][a] = best.clo[side][a] = 3.0e38f; best.hi[side][a] = best.chi[side][a] = -3.0e38f; }
            for (int b = side ? best.bin + 1 : 0; b <= (side ? 15 : best.bin); ++b) {
                const uint32_t *w = ab + b * kBinWords;
                if (!w[0]) continue;
                for (int a = 0; a < 3; ++a) {
                    best.lo[side][a] = fminf(best.lo[side][a], unordered(w[1 + a])); best.hi[side][a] = fmaxf(best.hi[side][a], unordered(w[4 + a]));
                    best.clo[side][a] = fminf(best.clo[side][a], unordered(w[7 + a])); best.chi[side][a] = fmaxf(best.chi[side][a], unordered(w[10 + a]));
                }
            }
        }
    }
    return best;
}
struct SmallRoot { uint32_t id, first, count, depth; };
__global__ __launch_bounds__(256) void k0_sah_split_kernel(const SahNode *__restrict__ active, uint32_t n_active, const uint32_t *__restrict__ bins, uint32_t n,
                                                           SahSplit *__restrict__ splits, SahNode *__restrict__ next_active, SmallRoot *__restrict__ small_roots,
                                                           SahCounters *__restrict__ counters, int2 *__restrict__ node_children, uint32_t *__restrict__ node_parent,
                                                           uint32_t *__restrict__ node_size, Box6 *__restrict__ node_box, uint2 *__restrict__ node_place, uint32_t max_depth) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n_active) return;
    const SahNode nd = active[a];
    const SahChoice ch = sah_sweep(bins + size_t(a) * 48u * kBinWords, nd.clo, nd.chi);
    SahSplit sp;
    sp.axis = ch.axis; sp.bin = ch.bin;
    sp.by_position = ch.axis < 0 ? 1u : 0u;
    sp.n_left = ch.axis < 0 ? nd.count / 2u : ch.n_left;
    if (nd.depth + 1u >= max_depth) atomicOr(&counters->error, 1u);          // deeper than the walkers' stacks allow: the host builder's forced median takes over
    const uint32_t counts[2] = { sp.n_left, nd.count - sp.n_left }, firsts[2] = { nd.first, nd.first + sp.n_left };
    for (int side = 0; side < 2; ++side) {
        const uint32_t cnt = counts[side];
        uint32_t cid;
        if (cnt == 1u) cid = firsts[side];                                    // a leaf's id = its triangle's final position
        else cid = n + atomicAdd(&counters->inner, cnt > kSmallNode ? 1u : cnt - 1u);      // (a small subtree's inner nodes: its own id and the cnt - 2 after it)
        sp.child_id[side] = cid;
        node_parent[cid] = nd.id;
        node_size[cid] = cnt;
        node_place[cid] = uint2{ nd.depth + 1u, firsts[side] };
        Box6 cb;                                                              // (by_position: ordered-uint identities, grown by the scatter pass's atomics)
        for (int k = 0; k < 3; ++k) { cb.lo[k] = ch.axis < 0 ? __uint_as_float(0xffffffffu) : ch.lo[side][k]; cb.hi[k] = ch.axis < 0 ? __uint_as_float(0u) : ch.hi[side][k]; }
        {
            uint32_t *w = reinterpret_cast<uint32_t *>(&node_box[cid]);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&cb);
            for (int k = 0; k < 6; ++k) w[k] = src[k];
        }
        sp.next_index[side] = kNoNode;
        if (cnt > kSmallNode) {
            const uint32_t slot = atomicAdd(&counters->next_active, 1u);
            SahNode nx;
            nx.id = cid; nx.first = firsts[side]; nx.count = cnt; nx.depth = nd.depth + 1u;
            for (int k = 0; k < 3; ++k) { nx.clo[k] = ch.axis < 0 ? nd.clo[k] : ch.clo[side][k]; nx.chi[k] = ch.axis < 0 ? nd.chi[k] : ch.chi[side][k]; }
            next_active[slot] = nx;
            sp.next_index[side] = slot;
        } else if (cnt > 1u) {
            const uint32_t slot = atomicAdd(&counters->small_roots, 1u);
            small_roots[slot] = SmallRoot{ cid, firsts[side], cnt, nd.depth + 1u };
        }
    }
    node_children[nd.id] = int2{ int(sp.child_id[0]), int(sp.child_id[1]) };
    splits[a] = sp;
}
// "goes left" per position of an active node (0 elsewhere): the input of the device-wide scan
__global__ __launch_bounds__(256) void k0_sah_flags_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ order, const uint32_t *__restrict__ pos_node,
                                                           const SahNode *__restrict__ active, const SahSplit *__restrict__ splits, uint32_t n, uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = pos_node[i];
    uint32_t f = 0;
    if (a != kNoNode) {
        const SahNode nd = active[a];
        const SahSplit sp = splits[a];
        if (sp.axis < 0) f = (i - nd.first) < sp.n_left ? 1u : 0u;
        else {
            const Box6 box = boxes[order[i]];
            const float c = 0.5f * (box.lo[sp.axis] + box.hi[sp.axis]);
            f = sah_bin(c, nd.clo[sp.axis], nd.chi[sp.axis] - nd.clo[sp.axis]) <= sp.bin ? 1u : 0u;
        }
    }
    flags[i] = f;
}
__global__ __launch_bounds__(256) void k0_sah_scatter_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ order, const uint32_t *__restrict__ pos_node,
                                                             const SahNode *__restrict__ active, const SahSplit *__restrict__ splits, const uint32_t *__restrict__ flags,
                                                             const uint32_t *__restrict__ scan, uint32_t n, uint32_t *__restrict__ order_out,
                                                             uint32_t *__restrict__ pos_node_out, Box6 *__restrict__ node_box, SahNode *__restrict__ next_active) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = pos_node[i], t = order[i];
    if (a == kNoNode) { order_out[i] = t; pos_node_out[i] = kNoNode; return; }
    const SahNode nd = active[a];
    const SahSplit sp = splits[a];
    const uint32_t lefts_before = scan[i] - scan[nd.first], side = flags[i] ? 0u : 1u;
    const uint32_t dst = side == 0u ? nd.first + lefts_before : nd.first + sp.n_left + ((i - nd.first) - lefts_before);
    order_out[dst] = t;
    pos_node_out[dst] = sp.next_index[side];
    if (sp.by_position) {                                                     // (rare: no plane separated anything) the children's boxes, by atomics
        const Box6 box = boxes[t];
        uint32_t *w = reinterpret_cast<uint32_t *>(&node_box[sp.child_id[side]]);
        for (int k = 0; k < 3; ++k) { atomicMin(&w[k], ordered(box.lo[k])); atomicMax(&w[3 + k], ordered(box.hi[k])); }
    }
}
// by_position children hold their boxes as ordered uints until here
__global__ __launch_bounds__(256) void k0_sah_fix_boxes_kernel(const SahSplit *__restrict__ splits, uint32_t n_active, Box6 *__restrict__ node_box) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n_active || !splits[a].by_position) return;
    for (int side = 0; side < 2; ++side) {
        uint32_t *w = reinterpret_cast<uint32_t *>(&node_box[splits[a].child_id[side]]);
        for (int k = 0; k < 6; ++k) { const float f = unordered(w[k]); w[k] = __float_as_uint(f); }
    }
}

// ---- subtrees of <= kSmallNode triangles: one wave each ----
// One wave per subtree of <= kSmallNode triangles, depth first over an explicit stack in LDS.  Bins here hold (count, box): the bounds of
// the centres are wave reductions at every node.  The sweep is spread over the lanes -- lane 16 * axis + b prices the plane after bin b
// -- and the lowest lane among the cheapest wins, which is the host's "strict <, axes and bins ascending".  Nodes of <= leaf_tris
// triangles end up inside one leaf of the final tree, so their splits are never looked at: they are cut in halves by position.
constexpr uint32_t kSmallBinWords = 7;
__global__ __launch_bounds__(256) void k0_sah_small_kernel(const Box6 *__restrict__ boxes, uint32_t *__restrict__ order, const SmallRoot *__restrict__ roots, uint32_t n_roots,
                                                           uint32_t n, SahCounters *__restrict__ counters, int2 *__restrict__ node_children, uint32_t *__restrict__ node_parent,
                                                           uint32_t *__restrict__ node_size, Box6 *__restrict__ node_box, uint2 *__restrict__ node_place, uint32_t max_depth,
                                                           uint32_t leaf_tris) {
    constexpr int WAVES = 4;
    __shared__ float s_box[WAVES][6][kSmallNode], s_cent[WAVES][3][kSmallNode];
    __shared__ uint32_t s_tri[WAVES][2][kSmallNode];                  // the subtree's triangles in their current order (double buffered)
    __shared__ uint32_t s_bins[WAVES][48 * kSmallBinWords];
    __shared__ uint32_t s_stack[WAVES][kSmallNode + 1][4];            // (id, first (within the subtree), count, depth); the last row passes the stack pointer
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * WAVES + wave;
    if (r >= n_roots) return;
    const SmallRoot root = roots[r];
    uint32_t cur = 0;
    const uint32_t my_tri = lane < root.count ? order[root.first + lane] : 0u;
    if (lane < root.count) {
        const Box6 b = boxes[my_tri];
        s_tri[wave][0][lane] = lane;                                  // indices into s_box / s_cent (the loaded copy stays where it is)
        for (int a = 0; a < 3; ++a) { s_box[wave][a][lane] = b.lo[a]; s_box[wave][3 + a][lane] = b.hi[a]; s_cent[wave][a][lane] = 0.5f * (b.lo[a] + b.hi[a]); }
    }
    uint32_t sp = 1, next_id = root.id + 1u;                          // ids root.id .. root.id + root.count - 2 are this subtree's (reserved by the split that made it)
    if (lane == 0) { s_stack[wave][0][0] = root.id; s_stack[wave][0][1] = 0u; s_stack[wave][0][2] = root.count; s_stack[wave][0][3] = root.depth; }
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    wave_sync();
    while (sp > 0) {
        --sp;
        const uint32_t id = s_stack[wave][sp][0], first = s_stack[wave][sp][1], count = s_stack[wave][sp][2], depth = s_stack[wave][sp][3];
        if (count <= leaf_tris) {
            // the inside of a future leaf: halves by position, all of it by lane 0, no boxes (nothing reads them)
            static_assert(kMaxLeafTris <= 4, "two levels of halves cover a future leaf");
            if (lane == 0) {
                const uint32_t cc[2] = { count / 2u, count - count / 2u }, cf[2] = { first, first + count / 2u };
                uint32_t taken = 0, cid[2];
                const uint32_t base = next_id;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    cid[side] = cc[side] == 1u ? root.first + cf[side] : base + taken++;
                    node_parent[cid[side]] = id;
                    node_size[cid[side]] = cc[side];
                    if (cc[side] == 2u) {                                                // (1 or 2: count <= 4)
                        const uint32_t l0 = root.first + cf[side];
                        node_parent[l0] = cid[side]; node_parent[l0 + 1u] = cid[side];
                        node_size[l0] = 1u; node_size[l0 + 1u] = 1u;
                        node_children[cid[side]] = int2{ int(l0), int(l0 + 1u) };
                    }
                }
                node_children[id] = int2{ int(cid[0]), int(cid[1]) };
            }
            next_id += count - 2u;
            continue;
        }
        const bool in = lane >= first && lane < first + count;
        const uint32_t me = in ? s_tri[wave][cur][lane] : 0u;
        // centre bounds of the node (wave reductions over its lanes)
        float clo[3], chi[3], c[3] = { 0.0f, 0.0f, 0.0f };
        for (int a = 0; a < 3; ++a) {
            if (in) c[a] = s_cent[wave][a][me];
            float mn = in ? c[a] : 3.0e38f, mx = in ? c[a] : -3.0e38f;
            for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
            clo[a] = mn; chi[a] = mx;
        }
        for (uint32_t k = lane; k < 48u * kSmallBinWords; k += 64u) {
            const uint32_t word = k % kSmallBinWords;
            s_bins[wave][k] = word >= 1 && word <= 3 ? 0xffffffffu : 0u;
        }
        wave_sync();
        int my_bin[3] = { 0, 0, 0 };
        if (in) {
            for (int a = 0; a < 3; ++a) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                my_bin[a] = sah_bin(c[a], clo[a], ext);
                uint32_t *w = s_bins[wave] + (a * 16 + my_bin[a]) * kSmallBinWords;
                atomicAdd(&w[0], 1u);
                for (int k = 0; k < 3; ++k) { atomicMin(&w[1 + k], ordered(s_box[wave][k][me])); atomicMax(&w[4 + k], ordered(s_box[wave][3 + k][me])); }
            }
        }
        wave_sync();
        // lane 16 * axis + b: the plane after bin b of that axis
        float cost = __builtin_inff();
        float llo[3], lhi[3], rlo[3], rhi[3];
        uint32_t cnt_left = 0;
        {
            const int axis = int(lane >> 4), b = int(lane & 15u);
            const uint32_t *ab = s_bins[wave] + (axis < 3 ? axis : 0) * 16 * kSmallBinWords;
            uint32_t cnt_right = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { llo[k] = rlo[k] = 3.0e38f; lhi[k] = rhi[k] = -3.0e38f; }
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) {
                const uint32_t *w = ab + bb * kSmallBinWords;
                const uint32_t c0 = w[0];
                const bool to_left = bb <= b && c0 != 0u, to_right = bb > b && c0 != 0u;
                cnt_left += to_left ? c0 : 0u;
                cnt_right += to_right ? c0 : 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float lo = unordered(w[1 + k]), hi = unordered(w[4 + k]);
                    llo[k] = to_left ? fminf(llo[k], lo) : llo[k]; lhi[k] = to_left ? fmaxf(lhi[k], hi) : lhi[k];
                    rlo[k] = to_right ? fminf(rlo[k], lo) : rlo[k]; rhi[k] = to_right ? fmaxf(rhi[k], hi) : rhi[k];
                }
            }
            const bool axis_ok = axis < 3 && (axis == 0 ? chi[0] - clo[0] : axis == 1 ? chi[1] - clo[1] : chi[2] - clo[2]) > 0.0f;
            if (axis_ok && b < 15 && cnt_left && cnt_right) cost = half_area6(llo, lhi) * float(cnt_left) + half_area6(rlo, rhi) * float(cnt_right);
        }
        float best = cost;
        uint32_t best_lane = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off);
            const uint32_t ol = uint32_t(__shfl_xor(int(best_lane), off));
            if (oc < best || (oc == best && ol < best_lane)) { best = oc; best_lane = ol; }
        }
        const bool planar = best < __builtin_inff();                             // a plane separates something
        const int axis = planar ? int(best_lane >> 4) : -1, bin = int(best_lane & 15u);
        const uint32_t n_left = planar ? uint32_t(__shfl(int(cnt_left), int(best_lane))) : count / 2u;
        const bool left = in && (planar ? (axis == 0 ? my_bin[0] : axis == 1 ? my_bin[1] : my_bin[2]) <= bin : (lane - first) < n_left);
        const unsigned long long in_mask = __ballot(in), left_mask = __ballot(left);
        const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
        const uint32_t lefts_before = uint32_t(__popcll(left_mask & below)), ins_before = uint32_t(__popcll(in_mask & below));
        if (in) {
            const uint32_t dst = left ? first + lefts_before : first + n_left + (ins_before - lefts_before);
            s_tri[wave][cur ^ 1][dst] = me;
        } else if (lane < root.count) {
            s_tri[wave][cur ^ 1][lane] = s_tri[wave][cur][lane];
        }
        cur ^= 1;
        // the children's boxes: the winning lane's two unions (or, when no plane separated anything, reductions over the two halves)
        float blo[2][3], bhi[2][3];
        for (int side = 0; side < 2; ++side)
            for (int a = 0; a < 3; ++a) {
                if (planar) { blo[side][a] = __shfl(side ? rlo[a] : llo[a], int(best_lane)); bhi[side][a] = __shfl(side ? rhi[a] : lhi[a], int(best_lane)); continue; }
                const bool mine = in && (left ? side == 0 : side == 1);
                float mn = mine ? s_box[wave][a][me] : 3.0e38f, mx = mine ? s_box[wave][3 + a][me] : -3.0e38f;
                for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
                blo[side][a] = mn; bhi[side][a] = mx;
            }
        if (lane == 0) {
            if (depth + 1u >= max_depth) atomicOr(&counters->error, 1u);
            const uint32_t counts[2] = { n_left, count - n_left }, firsts[2] = { first, first + n_left };
            uint32_t cid[2];
            for (int side = 0; side < 2; ++side) {
                cid[side] = counts[side] == 1u ? root.first + firsts[side] : next_id + uint32_t(side && counts[0] > 1u);
                node_parent[cid[side]] = id;
                node_size[cid[side]] = counts[side];
                node_place[cid[side]] = uint2{ depth + 1u, root.first + firsts[side] };
                Box6 cb;
                for (int a = 0; a < 3; ++a) { cb.lo[a] = blo[side][a]; cb.hi[a] = bhi[side][a]; }
                node_box[cid[side]] = cb;
            }
            node_children[id] = int2{ int(cid[0]), int(cid[1]) };
            uint32_t top = sp;
            for (int side = 1; side >= 0; --side)                            // (left child on top: depth first, left to right)
                if (counts[side] > 1u) { s_stack[wave][top][0] = cid[side]; s_stack[wave][top][1] = firsts[side]; s_stack[wave][top][2] = counts[side]; s_stack[wave][top][3] = depth + 1u; ++top; }
            s_stack[wave][kSmallNode][0] = top;                              // the new stack pointer, for the other lanes
        }
        next_id += uint32_t(n_left > 1u) + uint32_t(count - n_left > 1u);
        wave_sync();
        sp = s_stack[wave][kSmallNode][0];
    }
    // the subtree's final order back into `order`: position p holds the triangle that lane s_tri[p] loaded
    const uint32_t src = lane < root.count ? s_tri[wave][cur][lane] : 0u;
    const uint32_t tri = uint32_t(__shfl(int(my_tri), int(src)));
    if (lane < root.count) order[root.first + lane] = tri;
}
// every triangle is a leaf node of the intermediate tree: leaf k = the triangle at position k of the final order
__global__ __launch_bounds__(256) void k0_sah_leaves_kernel(const Box6 *__restrict__ boxes, const uint32_t *__restrict__ order, uint32_t n, Box6 *__restrict__ node_box,
                                                            uint32_t *__restrict__ node_size) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    node_box[k] = boxes[order[k]];
    node_size[k] = 1u;
}
// breadth-first numbering of the kept inner nodes, the host's (bvh_build.cpp "breadth-first numbering"): by depth, then left to right
__global__ __launch_bounds__(256) void k0_bfs_keys_kernel(const uint32_t *__restrict__ node_size, const uint2 *__restrict__ node_place, uint32_t n, uint32_t total_nodes,
                                                              uint32_t leaf_tris, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (n + k >= total_nodes) return;
    const uint2 place = node_place[n + k];
    keys[k] = node_size[n + k] > leaf_tris ? (static_cast<unsigned long long>(place.x) << 32) | place.y : ~0ull;
    vals[k] = k;
}
__global__ __launch_bounds__(256) void k0_bfs_rank_kernel(const uint32_t *__restrict__ sorted_vals, uint32_t kept, uint32_t *__restrict__ rank) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < kept) rank[sorted_vals[k]] = k;
}
// ---- "bvh_frame": the search's pass over the triangles (bvh_frame.hpp), and the boxes in the frame it found ----
// sums[k] += sum over the triangles of cost_term(candidate k): integers, so the order of the atomics does not matter
__global__ __launch_bounds__(256) void k0_frame_cost_kernel(const BvhTri *__restrict__ tris, uint32_t n, uint32_t stride, bvh_frame::Candidates c, unsigned long long *__restrict__ sums) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;            // sample t = triangle t * stride
    BvhTri tri{};
    if (t < n) tri = tris[size_t(t) * stride];
    for (int k = 0; k < c.n; ++k) {
        unsigned long long e = t < n ? bvh_frame::cost_term(c.r[k], tri) : 0ull;
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if ((threadIdx.x & 63u) == 0u && e) atomicAdd(&sums[k], e);
    }
}
struct FrameMatrix { float r[9]; };
__global__ __launch_bounds__(256) void k0_frame_boxes_kernel(const BvhTri *__restrict__ tris, uint32_t n, FrameMatrix frame, Box6 *__restrict__ boxes,
                                                             uint32_t *__restrict__ centre_bounds) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    float c[3] = { 0.0f, 0.0f, 0.0f };
    if (t < n) {
        Box6 b;
        bvh_frame::box_in_frame(frame.r, tris[t], b.lo, b.hi);
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (b.lo[a] + b.hi[a]);
        boxes[t] = b;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = t < n ? c[a] : 3.0e38f, mx = t < n ? c[a] : -3.0e38f;
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if ((threadIdx.x & 63u) == 0u) { atomic_min_checked(&centre_bounds[a], ordered(mn)); atomic_max_checked(&centre_bounds[3 + a], ordered(mx)); }
    }
}

// ---- "bvh_presplit": the references of very fat triangles, one per grid cell they pass through (presplit.hpp, shared with the host builder) ----
struct PresplitGrids { presplit::Grid g[presplit::kLevels]; };

// per level: the estimated number of references, summed (64 bit) over the triangles
__global__ __launch_bounds__(256) void k0_presplit_estimate_kernel(const BvhTri *__restrict__ tris, const Box6 *__restrict__ boxes, uint32_t n, PresplitGrids grids,
                                                                   unsigned long long *__restrict__ estimates) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    BvhTri tri{};
    Box6 b{};
    if (t < n) { tri = tris[t]; b = boxes[t]; }
    for (int k = 0; k < presplit::kLevels; ++k) {
        unsigned long long e = t < n ? presplit::estimate(grids.g[k], tri, b.lo, b.hi) : 0ull;
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if ((threadIdx.x & 63u) == 0u && e) atomicAdd(&estimates[k], e);
    }
}
// the exact number of references of every triangle on one grid
__global__ __launch_bounds__(256) void k0_presplit_count_kernel(const BvhTri *__restrict__ tris, const Box6 *__restrict__ boxes, uint32_t n, presplit::Grid grid,
                                                                uint32_t *__restrict__ refs, unsigned long long *__restrict__ total) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    unsigned long long count = 0ull;
    if (t < n) {
        const BvhTri tri = tris[t];
        const Box6 b = boxes[t];
        count = refs[t] = presplit::references(grid, tri, b.lo, b.hi, [](const float *, const float *) {});
    }
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    if ((threadIdx.x & 63u) == 0u && count) atomicAdd(total, count);
}
// ... and the references themselves, triangle t's from start[t] on; the bounds of their box centres like k0_triangles_kernel's
__global__ __launch_bounds__(256) void k0_presplit_emit_kernel(const BvhTri *__restrict__ tris, const Box6 *__restrict__ boxes, uint32_t n, presplit::Grid grid,
                                                               const uint32_t *__restrict__ start, BvhTri *__restrict__ out_tris, Box6 *__restrict__ out_boxes,
                                                               uint32_t *__restrict__ centre_bounds) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (t < n) {
        const BvhTri tri = tris[t];
        const Box6 b = boxes[t];
        uint32_t at = start[t];
        presplit::references(grid, tri, b.lo, b.hi, [&](const float *lo, const float *hi) {
            Box6 r;
            for (int a = 0; a < 3; ++a) {
                r.lo[a] = lo[a]; r.hi[a] = hi[a];
                const float c = 0.5f * (lo[a] + hi[a]);
                mn[a] = fminf(mn[a], c); mx[a] = fmaxf(mx[a], c);
            }
            out_tris[at] = tri;
            out_boxes[at] = r;
            ++at;
        });
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = mn[a], hi = mx[a];
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
        if ((threadIdx.x & 63u) == 0u && lo <= hi) { atomic_min_checked(&centre_bounds[a], ordered(lo)); atomic_max_checked(&centre_bounds[3 + a], ordered(hi)); }
    }
}

__global__ __launch_bounds__(256) void k0_iota_kernel(uint32_t *__restrict__ order, uint32_t *__restrict__ pos_node, uint32_t n) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < n) { order[k] = k; pos_node[k] = 0u; }
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
int device_build_bvh(vhr_context *ctx, const std::vector<uint32_t> &tri_prefix, uint32_t total_tris, int leaf_tris_in, int presplit_percent, int frame_mode) {
    uint32_t n = total_tris;
    const uint32_t leaf_tris = uint32_t(std::max(1, std::min(kMaxLeafTris, leaf_tris_in)));
    if (n <= leaf_tris || n < 2u) return VHR_ERROR_OUT_OF_SLOTS;
    hipStream_t s = ctx->stream;
    Scratch tmp;
    const bool trace = std::getenv("VHR_K0_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "K0 device %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const dim3 block(256);
    auto grid = [](uint32_t count) { return dim3((count + 255u) / 256u); };
    // ---- the triangles (world space, their boxes, the bounds of the box centres), then -- "bvh_presplit" -- their references ----
    uint32_t *d_prefix, *d_bounds, *d_counts;
    BvhTri *d_tris_flat;
    Box6 *d_boxes;
    K0_TRY(tmp.alloc(&d_prefix, tri_prefix.size()));
    K0_TRY(tmp.alloc(&d_bounds, 12));
    K0_TRY(tmp.alloc(&d_counts, 4));
    K0_TRY(tmp.alloc(&d_tris_flat, n)); K0_TRY(tmp.alloc(&d_boxes, n));
    K0_TRY(hipMemcpyAsync(d_prefix, tri_prefix.data(), tri_prefix.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    const uint32_t init_bounds[12] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    K0_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));
    K0_TRY(hipMemsetAsync(d_counts, 0, 4 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k0_triangles_kernel, grid(n), block, 0, s, ctx->d_vertices, ctx->d_indices, ctx->d_primitives, d_prefix, uint32_t(tri_prefix.size()), n,
                       d_tris_flat, d_boxes, d_bounds);
    uint32_t h_bounds[12];
    K0_TRY(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    // ---- "bvh_frame" 1: the frame search (the host builder's, with a kernel as its pass over the triangles), then the boxes in that frame ----
    ctx->bvh_frame_on = false;
    { const float identity[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; std::memcpy(ctx->bvh_frame, identity, sizeof(identity)); }
    if (frame_mode == 1) {
        unsigned long long *d_sums;
        K0_TRY(tmp.alloc(&d_sums, size_t(bvh_frame::kMaxCandidates)));
        hipError_t pass_error = hipSuccess;
        float frame[9];
        const uint32_t stride = bvh_frame::sample_stride(n), samples = (n + stride - 1u) / stride;
        const bool found = bvh_frame::choose([&](const bvh_frame::Candidates &c, uint64_t *sums) {
            static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "64-bit sums");
            hipError_t e = hipMemsetAsync(d_sums, 0, sizeof(unsigned long long) * bvh_frame::kMaxCandidates, s);
            if (e == hipSuccess) { hipLaunchKernelGGL(k0_frame_cost_kernel, grid(samples), block, 0, s, d_tris_flat, samples, stride, c, d_sums); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipMemcpyAsync(sums, d_sums, sizeof(uint64_t) * size_t(c.n), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { pass_error = e; for (int k = 0; k < c.n; ++k) sums[k] = ~0ull; }
        }, frame);
        K0_TRY(pass_error);
        if (found) {
            FrameMatrix fm;
            std::memcpy(fm.r, frame, sizeof(frame));
            K0_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k0_frame_boxes_kernel, grid(n), block, 0, s, d_tris_flat, n, fm, d_boxes, d_bounds);
            K0_TRY(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, s));
            K0_TRY(hipStreamSynchronize(s));
            K0_TRY(hipGetLastError());
            std::memcpy(ctx->bvh_frame, frame, sizeof(frame));
            ctx->bvh_frame_on = true;
        }
        lap("frame search");
    }
    ctx->bvh_presplit_level = -1;
    if (presplit_percent > 0 && !ctx->bvh_frame_on) {           // (a rotated frame takes the place of splitting: the option is not combined with it)
        float clo[3], chi[3];
        for (int a = 0; a < 3; ++a) { clo[a] = unordered(h_bounds[a]); chi[a] = unordered(h_bounds[3 + a]); }
        PresplitGrids grids;
        for (int k = 0; k < presplit::kLevels; ++k) grids.g[k] = presplit::make_grid(clo, chi, k);
        unsigned long long *d_estimates;
        uint32_t *d_refs, *d_start;
        K0_TRY(tmp.alloc(&d_estimates, size_t(presplit::kLevels)));
        K0_TRY(tmp.alloc(&d_refs, n)); K0_TRY(tmp.alloc(&d_start, n));
        K0_TRY(hipMemsetAsync(d_estimates, 0, sizeof(unsigned long long) * presplit::kLevels, s));
        hipLaunchKernelGGL(k0_presplit_estimate_kernel, grid(n), block, 0, s, d_tris_flat, d_boxes, n, grids, d_estimates);
        uint64_t estimates[presplit::kLevels];
        static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "64-bit sums");
        K0_TRY(hipMemcpyAsync(estimates, d_estimates, sizeof(estimates), hipMemcpyDeviceToHost, s));
        K0_TRY(hipStreamSynchronize(s));
        int level = presplit::choose_level(estimates, n, uint32_t(presplit_percent), clo, chi);
        const uint64_t hard_limit = uint64_t(n) + 2ull * uint64_t(n) * uint64_t(presplit_percent) / 100ull;
        size_t ps_bytes = 0;
        K0_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, ps_bytes, d_refs, d_start, int(n), s));
        char *d_ps_work;
        K0_TRY(tmp.alloc(&d_ps_work, ps_bytes));
        uint64_t total = n;
        unsigned long long *d_total;
        K0_TRY(tmp.alloc(&d_total, size_t(1)));
        for (; level >= 0; --level) {                 // the exact count; an estimate that was too low by more than 2x goes one level up
            // (the total in 64 bits beside the counts: every count is below 2^30, but a 32-bit sum that wrapped could look small)
            K0_TRY(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k0_presplit_count_kernel, grid(n), block, 0, s, d_tris_flat, d_boxes, n, grids.g[level], d_refs, d_total);
            unsigned long long h_total = 0;
            K0_TRY(hipMemcpyAsync(&h_total, d_total, sizeof(h_total), hipMemcpyDeviceToHost, s));
            K0_TRY(hipStreamSynchronize(s));
            total = h_total;
            if (total <= hard_limit && total < (1ull << 31)) break;
        }
        if (level >= 0 && total > n) {
            K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_ps_work, ps_bytes, d_refs, d_start, int(n), s));
            BvhTri *d_tris_refs;
            Box6 *d_boxes_refs;
            K0_TRY(tmp.alloc(&d_tris_refs, size_t(total))); K0_TRY(tmp.alloc(&d_boxes_refs, size_t(total)));
            K0_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k0_presplit_emit_kernel, grid(n), block, 0, s, d_tris_flat, d_boxes, n, grids.g[level], d_start, d_tris_refs, d_boxes_refs, d_bounds);
            K0_TRY(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, s));
            K0_TRY(hipStreamSynchronize(s));
            K0_TRY(hipGetLastError());
            if (trace) std::fprintf(stderr, "K0 device presplit: level %d, %u triangles -> %llu references\n", level, n, (unsigned long long)total);
            d_tris_flat = d_tris_refs;
            d_boxes = d_boxes_refs;
            n = uint32_t(total);
            ctx->bvh_presplit_level = level;
        }
        lap("presplit");
    }
    const uint32_t total_cap = 2u * n;
    const uint32_t max_active = n / kSmallNode + 2u, max_small = n / 2u + 2u;
    uint32_t *d_order[2], *d_pos_node[2], *d_flags, *d_scan, *d_bins, *d_parent, *d_size, *d_position, *d_kept, *d_kept_rank;
    Box6 *d_node_box;
    int2 *d_children;
    SahNode *d_active[2];
    SahSplit *d_splits;
    SmallRoot *d_small;
    SahCounters *d_counters;
    K0_TRY(tmp.alloc(&d_counters, 1));
    for (int k = 0; k < 2; ++k) { K0_TRY(tmp.alloc(&d_order[k], n)); K0_TRY(tmp.alloc(&d_pos_node[k], n)); K0_TRY(tmp.alloc(&d_active[k], max_active)); }
    K0_TRY(tmp.alloc(&d_flags, n)); K0_TRY(tmp.alloc(&d_scan, n));
    K0_TRY(tmp.alloc(&d_bins, size_t(max_active) * 48u * kBinWords));
    K0_TRY(tmp.alloc(&d_splits, max_active)); K0_TRY(tmp.alloc(&d_small, max_small));
    K0_TRY(tmp.alloc(&d_parent, total_cap)); K0_TRY(tmp.alloc(&d_size, total_cap)); K0_TRY(tmp.alloc(&d_node_box, total_cap)); K0_TRY(tmp.alloc(&d_children, total_cap));
    K0_TRY(tmp.alloc(&d_position, n)); K0_TRY(tmp.alloc(&d_kept, n)); K0_TRY(tmp.alloc(&d_kept_rank, n));
    uint2 *d_place;
    unsigned long long *d_keys[2];
    uint32_t *d_vals[2];
    K0_TRY(tmp.alloc(&d_place, total_cap));
    for (int k = 0; k < 2; ++k) { K0_TRY(tmp.alloc(&d_keys[k], n)); K0_TRY(tmp.alloc(&d_vals[k], n)); }
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_tris), sizeof(BvhTri) * n));
    K0_TRY(hipMemsetAsync(d_place + n, 0, sizeof(uint2), s));                 // the root: depth 0, first 0
    hipLaunchKernelGGL(k0_iota_kernel, grid(n), block, 0, s, d_order[0], d_pos_node[0], n);
    K0_TRY(hipStreamSynchronize(s));
    lap("allocations + triangles");
    size_t scan_bytes = 0;
    K0_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_scan, int(n), s));
    char *d_work;
    K0_TRY(tmp.alloc(&d_work, scan_bytes));
    // the root: inner node n (creation rank 0), every triangle, the bounds of the box centres from the triangles pass
    SahNode root{};
    root.id = n; root.first = 0; root.count = n; root.depth = 0;
    for (int a = 0; a < 3; ++a) { root.clo[a] = unordered(h_bounds[a]); root.chi[a] = unordered(h_bounds[3 + a]); }
    SahCounters h_counters{ 1u, 0u, 0u, 0u };
    uint32_t n_active = 0, n_small = 0;
    if (n > kSmallNode) {
        K0_TRY(hipMemcpyAsync(d_active[0], &root, sizeof(root), hipMemcpyHostToDevice, s));
        n_active = 1;
    } else {
        const SmallRoot r{ n, 0u, n, 0u };
        K0_TRY(hipMemcpyAsync(d_small, &r, sizeof(r), hipMemcpyHostToDevice, s));
        h_counters.small_roots = n_small = 1;
        h_counters.inner = n - 1u;
        hipLaunchKernelGGL(k0_iota_kernel, grid(n), block, 0, s, d_order[0], d_pos_node[0], n);
    }
    K0_TRY(hipMemcpyAsync(d_counters, &h_counters, sizeof(h_counters), hipMemcpyHostToDevice, s));
    const uint32_t depth_guard = 2u * uint32_t(kMaxBvhDepth);       // (the layout stage measures the real depth; this only stops a runaway)
    int cur = 0;
    for (int level = 0; n_active > 0 && level < 256; ++level) {
        hipLaunchKernelGGL(k0_sah_clear_bins_kernel, grid(n_active * 48u * kBinWords), block, 0, s, d_bins, n_active);
        hipLaunchKernelGGL(k0_sah_bin_kernel, grid(n), block, 0, s, d_boxes, d_order[cur], d_pos_node[cur], d_active[cur], n, d_bins);
        K0_TRY(hipMemsetAsync(&d_counters->next_active, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k0_sah_split_kernel, grid(n_active), block, 0, s, d_active[cur], n_active, d_bins, n, d_splits, d_active[cur ^ 1], d_small, d_counters, d_children,
                           d_parent, d_size, d_node_box, d_place, depth_guard);
        hipLaunchKernelGGL(k0_sah_flags_kernel, grid(n), block, 0, s, d_boxes, d_order[cur], d_pos_node[cur], d_active[cur], d_splits, n, d_flags);
        K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_work, scan_bytes, d_flags, d_scan, int(n), s));
        hipLaunchKernelGGL(k0_sah_scatter_kernel, grid(n), block, 0, s, d_boxes, d_order[cur], d_pos_node[cur], d_active[cur], d_splits, d_flags, d_scan, n, d_order[cur ^ 1],
                           d_pos_node[cur ^ 1], d_node_box, d_active[cur ^ 1]);
        hipLaunchKernelGGL(k0_sah_fix_boxes_kernel, grid(n_active), block, 0, s, d_splits, n_active, d_node_box);
        K0_TRY(hipMemcpyAsync(&h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, s));
        K0_TRY(hipStreamSynchronize(s));
        if (h_counters.error) return VHR_ERROR_OUT_OF_SLOTS;
        if (h_counters.next_active > max_active || h_counters.small_roots > max_small) return ctx->fail(VHR_ERROR_DEVICE, "device K0: a node list overflowed");
        if (trace) std::fprintf(stderr, "K0 device level %d: %u nodes -> %u, %u small roots\n", level, n_active, h_counters.next_active, h_counters.small_roots);
        n_active = h_counters.next_active;
        n_small = h_counters.small_roots;
        cur ^= 1;
    }
    if (n_active) return VHR_ERROR_OUT_OF_SLOTS;
    lap("levels");
    if (n_small) hipLaunchKernelGGL(k0_sah_small_kernel, dim3((n_small + 3u) / 4u), block, 0, s, d_boxes, d_order[cur], d_small, n_small, n, d_counters, d_children, d_parent,
                                    d_size, d_node_box, d_place, depth_guard, leaf_tris);
    hipLaunchKernelGGL(k0_sah_leaves_kernel, grid(n), block, 0, s, d_boxes, d_order[cur], n, d_node_box, d_size);
    K0_TRY(hipMemcpyAsync(&h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    K0_TRY(hipGetLastError());
    if (h_counters.error) return VHR_ERROR_OUT_OF_SLOTS;
    lap("small subtrees");
    if (h_counters.inner != n - 1u) return ctx->fail(VHR_ERROR_DEVICE, "device K0: the splits did not end in one binary tree");
    const uint32_t total_nodes = 2u * n - 1u, root_id = n;
    uint32_t h_size_root = n;
    K0_TRY(hipMemcpyAsync(d_size + root_id, &h_size_root, 4, hipMemcpyHostToDevice, s));
    // ---- layout: the clustering builder's stage, inner nodes numbered in creation order (parents first) ----
    hipLaunchKernelGGL(k0_positions_kernel, grid(n), block, 0, s, d_children, d_parent, d_size, n, root_id, leaf_tris, d_position, d_counts);
    hipLaunchKernelGGL(k0_place_triangles_kernel, grid(n), block, 0, s, d_tris_flat, d_order[cur], d_position, n, ctx->d_tris);
    const uint32_t n_inner_all = total_nodes - n;
    hipLaunchKernelGGL(k0_kept_flags_kernel, grid(n_inner_all), block, 0, s, d_size, n, total_nodes, leaf_tris, d_kept);
    K0_TRY(hipcub::DeviceScan::ExclusiveSum(d_work, scan_bytes, d_kept, d_kept_rank, int(n_inner_all), s));
    uint32_t last_rank = 0, last_flag = 0, h_depth = 0;
    K0_TRY(hipMemcpyAsync(&last_rank, d_kept_rank + n_inner_all - 1, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipMemcpyAsync(&last_flag, d_kept + n_inner_all - 1, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipMemcpyAsync(&h_depth, d_counts, 4, hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    const uint32_t n_inner = last_rank + last_flag;
    if (n_inner == 0) return VHR_ERROR_OUT_OF_SLOTS;
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes), sizeof(BvhNode) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes_ch), sizeof(BvhNodeCH) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes48), sizeof(BvhNode48) * n_inner));
    K0_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_nodes16), sizeof(BvhNode16) * n_inner));
    {
        size_t sort_bytes = 0;
        K0_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys[0], d_keys[1], d_vals[0], d_vals[1], int(n_inner_all), 0, 40, s));
        char *d_sort_work;
        K0_TRY(tmp.alloc(&d_sort_work, sort_bytes));
        hipLaunchKernelGGL(k0_bfs_keys_kernel, grid(n_inner_all), block, 0, s, d_size, d_place, n, total_nodes, leaf_tris, d_keys[0], d_vals[0]);
        K0_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_work, sort_bytes, d_keys[0], d_keys[1], d_vals[0], d_vals[1], int(n_inner_all), 0, 40, s));
        hipLaunchKernelGGL(k0_bfs_rank_kernel, grid(n_inner), block, 0, s, d_vals[1], n_inner, d_kept_rank);
    }
    hipLaunchKernelGGL(k0_emit_kernel, grid(n_inner_all), block, 0, s, d_children, d_node_box, d_size, d_kept_rank, d_position, n, total_nodes, leaf_tris, ctx->d_nodes);
    hipLaunchKernelGGL(k0_node_bounds_kernel, grid(n_inner), block, 0, s, ctx->d_nodes, n_inner, d_bounds + 6);
    K0_TRY(hipMemcpyAsync(h_bounds, d_bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, s));
    K0_TRY(hipStreamSynchronize(s));
    K0_TRY(hipGetLastError());
    for (int a = 0; a < 3; ++a) {
        const float lo = unordered(h_bounds[6 + a]), hi = unordered(h_bounds[9 + a]);
        ctx->bvh_centre[a] = lo <= hi ? 0.5f * (lo + hi) : 0.0f;
    }
    hipLaunchKernelGGL(k0_forms_kernel, grid(n_inner), block, 0, s, ctx->d_nodes, n_inner, ctx->bvh_centre[0], ctx->bvh_centre[1], ctx->bvh_centre[2], ctx->d_nodes_ch,
                       ctx->d_nodes48, ctx->d_nodes16);
    {   // the self-checks of the node forms, where the nodes are (the host builder's tree is checked on the host)
        unsigned long long *d_checks;
        K0_TRY(tmp.alloc(&d_checks, 5));
        K0_TRY(hipMemsetAsync(d_checks, 0, 5 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k0_check_forms_kernel, grid(n_inner), block, 0, s, ctx->d_nodes, ctx->d_nodes_ch, ctx->d_nodes48, ctx->d_nodes16, n_inner,
                           ctx->bvh_centre[0], ctx->bvh_centre[1], ctx->bvh_centre[2], d_checks);
        unsigned long long h_checks[5];
        K0_TRY(hipMemcpyAsync(h_checks, d_checks, sizeof(h_checks), hipMemcpyDeviceToHost, s));
        K0_TRY(hipStreamSynchronize(s));
        const bool in_range = h_checks[4] == 0ull && size_t(n_inner) * sizeof(BvhNode16) < (size_t(1) << 31);
        for (int i = 0; i < 3; ++i) ctx->bvh_form_checks[i] = h_checks[i];
        ctx->bvh_form_checks[3] = in_range ? h_checks[3] : 0ull;
        ctx->nodes16_valid = in_range && h_checks[3] == 0ull;      // else the walkers stay on the 48-byte nodes
    }
    K0_TRY(hipStreamSynchronize(s));
    K0_TRY(hipGetLastError());
    lap("layout + node forms + self-checks");
    ctx->node_count = n_inner;
    ctx->tri_count = n;
    ctx->bvh_depth = h_depth;
    if (std::getenv("VHR_K0_TRACE")) std::fprintf(stderr, "K0 device: %u triangles, %u inner nodes, depth %u\n", n, n_inner, h_depth);
    if (h_depth > uint32_t(std::min(kMaxBvhDepth, ctx->bvh_device_max_depth))) return VHR_ERROR_OUT_OF_SLOTS;
    return VHR_OK;
}

}  // namespace vhr
