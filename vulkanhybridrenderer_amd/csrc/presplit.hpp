// K0, before the tree: the references of very fat triangles split on a uniform grid (option "bvh_presplit", default off).
//
// The reference hands its triangles to the driver's BLAS builder (resource_manager.cpp:593-689, PREFER_FAST_TRACE), and what a
// driver does with a triangle whose box is much larger than the triangle is its own business; a binned-SAH tree over one box per
// triangle pays for such a triangle on every ray that crosses its box.  The remedy is the one of the split-BVH literature, in its
// build-independent form: the triangle is entered into the build several times, each REFERENCE with the box of one grid cell the
// triangle passes through (cut to the triangle's own box).  The tree builders, the leaf records and the walkers do not change: a
// reference is a BvhTri like any other (same prim / tri / flat, so a ray that meets two references of one triangle computes the same
// hit twice and the closest-hit tie break still orders by flat index).
//
// What it buys and what it costs is in profiles/r5_sponza_hard.txt: on the stand-in built to need it (sponza_hard turned off the world
// axes: two-triangle walls 40 m long) the any-hit launch gains 18 % for 6 % more references, and the mirror ray's closest-hit launch
// LOSES 18 % -- an unsplit wall is a leaf next to the root, which a closest-hit ray tests first and which then bounds everything
// behind it.  Splitting the merely thin triangles (column slivers) as well loses on both.  Hence: only triangles that waste a
// noticeable share of the whole scene, and off by default.
//
// One implementation for the host builder (bvh_build.cpp) and the device builder (kernels_bvh.hip): both must produce the same
// references in the same order (the two builders make the same tree, tests/test_gpu_fuzz.py), so everything here is +, -, *, /,
// comparisons and integer work, compiled without contraction on both sides.
//
//   S                      = half area of the box of the triangle-box centres ("the scene"), E its longest edge
//   fatness of a triangle  = half area of its box - (|c.x| + |c.y| + |c.z|), c = e1 x e2: the half area of the flattest box a
//                            triangle with these three projections can have (equal for any triangle in an axis plane)
//   a triangle is split iff fatness > kFatShare * S and its box spans more than one cell of the grid
//   grid                   = cells of edge h = E / 2^level from the low corner of the scene box
//   its references         = the cells its box spans that the triangle touches (separating-axis test, conservative), found by
//                            halving the cell range along its longest axis, lower half first
//   level                  = the finest one with h * h >= kMinShare * S whose estimated references fit the budget (the option's value,
//                            percent of the triangle count); the exact count is checked afterwards and the level lowered if it does not
#pragma once

#include <cstdint>

#include "vhr_internal.hpp"

namespace vhr {
namespace presplit {

constexpr int kLevels = 13;                 // cell edge E / 2^0 .. E / 2^12
constexpr int kMaxSpan = 4096;              // cells per axis a split triangle's box may span (a range fits 16 bits; the stack below 3 * 12 + 4)
constexpr int kStack = 40;
constexpr float kMinShare = 1.0e-4f;        // cell faces no smaller than this share of the scene box's half area
constexpr float kFatShare = 1.0e-2f;        // a triangle is split when its box wastes more than this share of it

struct Grid {
    float origin[3];
    float h, inv_h;
    float fat;          // a triangle is split iff its fatness exceeds this
};

#define VHR_PS __host__ __device__ inline

VHR_PS float absf(float x) { return x < 0.0f ? -x : x; }
VHR_PS float minf(float a, float b) { return a < b ? a : b; }
VHR_PS float maxf(float a, float b) { return a > b ? a : b; }

VHR_PS float scene_half_area(const float centre_lo[3], const float centre_hi[3]) {
    const float dx = centre_hi[0] - centre_lo[0], dy = centre_hi[1] - centre_lo[1], dz = centre_hi[2] - centre_lo[2];
    return dx * dy + dy * dz + dz * dx;
}

VHR_PS Grid make_grid(const float centre_lo[3], const float centre_hi[3], int level) {
    Grid g;
    float e = centre_hi[0] - centre_lo[0];
    e = maxf(e, centre_hi[1] - centre_lo[1]);
    e = maxf(e, centre_hi[2] - centre_lo[2]);
    for (int a = 0; a < 3; ++a) g.origin[a] = centre_lo[a];
    const float cells = float(1u << level);
    g.h = e / cells;
    g.inv_h = e > 0.0f ? cells / e : 0.0f;
    g.fat = kFatShare * scene_half_area(centre_lo, centre_hi);
    return g;
}

VHR_PS float fatness(const BvhTri &t, const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float cx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
    const float cy = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
    const float cz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    return (dx * dy + dy * dz + dz * dx) - (absf(cx) + absf(cy) + absf(cz));
}

VHR_PS int floor_int(float x) {             // x is finite and far inside the int range (spans are checked before anything is indexed)
    int i = int(x);
    if (float(i) > x) --i;
    return i;
}
VHR_PS int cell_of(const Grid &g, float x, int axis) {
    const float r = (x - g.origin[axis]) * g.inv_h;
    return r > 1.0e9f ? 1000000000 : r < -1.0e9f ? -1000000000 : floor_int(r);
}

// about how many cells a flat triangle touches: its three projections over the cell face, plus the cells along its outline
VHR_PS float pieces(const Grid &g, const BvhTri &t, const float lo[3], const float hi[3]) {
    const float cx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
    const float cy = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
    const float cz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    const float faces = (absf(cx) + absf(cy) + absf(cz)) * 0.5f * g.inv_h * g.inv_h;
    const float outline = ((hi[0] - lo[0]) + (hi[1] - lo[1]) + (hi[2] - lo[2])) * g.inv_h;
    return 1.0f + faces + outline;
}

// the first cell and the spans of a triangle's box; false when the triangle is not to be split on this grid
VHR_PS bool wants_split(const Grid &g, const BvhTri &t, const float lo[3], const float hi[3], int first[3], int span[3]) {
    if (!(g.inv_h > 0.0f) || !(fatness(t, lo, hi) > g.fat)) return false;
    bool many = false;
    for (int a = 0; a < 3; ++a) {
        first[a] = cell_of(g, lo[a], a);
        const int last = cell_of(g, hi[a], a);
        if (last - first[a] >= kMaxSpan || last < first[a]) return false;
        span[a] = last - first[a] + 1;
        many = many || span[a] > 1;
    }
    return many;
}

// what the level selection adds up
VHR_PS uint32_t estimate(const Grid &g, const BvhTri &t, const float lo[3], const float hi[3]) {
    int first[3], span[3];
    if (!wants_split(g, t, lo, hi, first, span)) return 1u;
    const float e = pieces(g, t, lo, hi);
    return e < 1.0e9f ? uint32_t(e) : 1000000000u;
}

// triangle (three corners) against the box centre +- half: the separating-axis test over the box's three axes, the triangle's plane and
// the nine edge x axis directions.  `half` is taken a little larger than the cell, so a rounding error can only add a reference.
VHR_PS bool touches(const float p0[3], const float p1[3], const float p2[3], const float centre[3], const float half[3]) {
    float v0[3], v1[3], v2[3];
    for (int a = 0; a < 3; ++a) { v0[a] = p0[a] - centre[a]; v1[a] = p1[a] - centre[a]; v2[a] = p2[a] - centre[a]; }
    for (int a = 0; a < 3; ++a) {
        const float mn = minf(minf(v0[a], v1[a]), v2[a]), mx = maxf(maxf(v0[a], v1[a]), v2[a]);
        if (mn > half[a] || mx < -half[a]) return false;
    }
    const float f[3][3] = { { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, { v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2] }, { v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2] } };
    for (int e = 0; e < 3; ++e) {
        for (int a = 0; a < 3; ++a) {
            // axis = unit(a) x f[e]: components in the other two coordinates b, c
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            const float ab = -f[e][c], ac = f[e][b];
            const float q0 = ab * v0[b] + ac * v0[c], q1 = ab * v1[b] + ac * v1[c], q2 = ab * v2[b] + ac * v2[c];
            const float r = half[b] * absf(ab) + half[c] * absf(ac);
            const float slack = 1.0e-6f * (absf(q0) + absf(q1) + absf(q2)) + 1.0e-30f;
            if (minf(minf(q0, q1), q2) > r + slack || maxf(maxf(q0, q1), q2) < -(r + slack)) return false;
        }
    }
    const float nx = f[0][1] * f[1][2] - f[0][2] * f[1][1], ny = f[0][2] * f[1][0] - f[0][0] * f[1][2], nz = f[0][0] * f[1][1] - f[0][1] * f[1][0];
    const float d = nx * v0[0] + ny * v0[1] + nz * v0[2];
    const float r = half[0] * absf(nx) + half[1] * absf(ny) + half[2] * absf(nz);
    const float slack = 1.0e-5f * (absf(nx * v0[0]) + absf(ny * v0[1]) + absf(nz * v0[2])) + 1.0e-30f;
    return !(absf(d) > r + slack);
}

// The references of one triangle: emit(lo, hi) once per reference, in the order both builders store them.  Returns their number
// (1 and the triangle's own box when it is not split).
template <typename Emit>
VHR_PS uint32_t references(const Grid &g, const BvhTri &t, const float lo[3], const float hi[3], Emit &&emit) {
    int first[3], span[3];
    if (!wants_split(g, t, lo, hi, first, span)) { emit(lo, hi); return 1u; }
    float p0[3], p1[3], p2[3];
    for (int a = 0; a < 3; ++a) { p0[a] = t.v0[a]; p1[a] = t.v0[a] + t.e1[a]; p2[a] = t.v0[a] + t.e2[a]; }
    struct Range { uint16_t a[3], b[3]; };                 // cells a .. b per axis, relative to `first`
    Range stack[kStack];
    int top = 0;
    stack[top++] = Range{ { 0, 0, 0 }, { uint16_t(span[0] - 1), uint16_t(span[1] - 1), uint16_t(span[2] - 1) } };
    uint32_t count = 0;
    const float grow = 0.5f * g.h * 1.0e-3f;
    while (top > 0) {
        const Range r = stack[--top];
        float blo[3], bhi[3], centre[3], half[3];
        for (int a = 0; a < 3; ++a) {
            blo[a] = g.origin[a] + float(first[a] + int(r.a[a])) * g.h;
            bhi[a] = g.origin[a] + float(first[a] + int(r.b[a]) + 1) * g.h;
            centre[a] = 0.5f * (blo[a] + bhi[a]);
            half[a] = 0.5f * (bhi[a] - blo[a]) + grow + 1.0e-6f * absf(centre[a]);
        }
        if (!touches(p0, p1, p2, centre, half)) continue;
        int axis = 0, cells = int(r.b[0]) - int(r.a[0]);
        for (int a = 1; a < 3; ++a)
            if (int(r.b[a]) - int(r.a[a]) > cells) { axis = a; cells = int(r.b[a]) - int(r.a[a]); }
        if (cells == 0) {                                   // one cell: a reference, the cell cut to the triangle's box
            float rlo[3], rhi[3];
            for (int a = 0; a < 3; ++a) {
                rlo[a] = r.a[a] == 0 ? lo[a] : maxf(lo[a], blo[a]);
                rhi[a] = int(r.b[a]) == span[a] - 1 ? hi[a] : minf(hi[a], bhi[a]);
                if (rlo[a] > rhi[a]) { const float m = rlo[a]; rlo[a] = rhi[a]; rhi[a] = m; }      // (a cell edge and a box face within rounding of each other)
            }
            emit(rlo, rhi);
            ++count;
            continue;
        }
        const uint16_t mid = uint16_t((int(r.a[axis]) + int(r.b[axis])) / 2);
        Range lower = r, upper = r;
        lower.b[axis] = mid;
        upper.a[axis] = uint16_t(mid + 1);
        if (top + 2 > kStack) { emit(lo, hi); return count + 1u; }      // cannot happen (depth <= 3 * 12): the whole triangle rather than a hole
        stack[top++] = upper;
        stack[top++] = lower;
    }
    if (count == 0) { emit(lo, hi); return 1u; }                        // every cell refused (a degenerate triangle on a cell face): itself
    return count;
}

#undef VHR_PS

// Level selection, host side of both builders: `estimates[k]` = sum over the triangles of estimate(grid(k)).  The finest level whose
// cells are large enough and whose estimate stays within n + budget; -1 = no splitting.
inline int choose_level(const uint64_t estimates[kLevels], uint32_t n, uint32_t budget_percent, const float centre_lo[3], const float centre_hi[3]) {
    const uint64_t limit = uint64_t(n) + uint64_t(n) * budget_percent / 100u;
    const float floor_h2 = kMinShare * scene_half_area(centre_lo, centre_hi);
    int best = -1;
    for (int k = 0; k < kLevels; ++k) {
        const Grid g = make_grid(centre_lo, centre_hi, k);
        if (!(g.inv_h > 0.0f) || g.h * g.h < floor_h2) break;
        if (estimates[k] > uint64_t(n) && estimates[k] <= limit) best = k;
    }
    return best;
}

}  // namespace presplit
}  // namespace vhr
