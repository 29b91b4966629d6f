// K0: acceleration-structure build.  Replaces ResourceManager::UpdateBLAS / UpdateTLAS
// (/root/reference/src/rendering_backend/resource_manager.cpp:593-801): one geometry per Primitive with
// its transform baked in (:608-617), all opaque (:633), triangle count index_count / 3 (:637), indices
// offset by index_offset and vertices by vertex_offset (:638-639), one identity instance with face
// culling disabled (:704-718)  ==>  a world-space, two-sided triangle soup.
//
// Output: a binned-SAH BVH2 laid out for the CDNA4 traversal kernel -- 64-byte nodes holding BOTH child
// boxes (one node fetch = 4 dwordx4 loads decides both children), emitted in breadth-first order so the
// first K nodes are the top of the tree (the part the traversal kernel stages in LDS), leaves of <= 4
// triangles stored contiguously as 48-byte Moeller-Trumbore records.  Depth is bounded by kMaxBvhDepth,
// which is also the capacity of the traversal stack.
//
// Built with -ffp-contract=off: the world-space vertex transform and the edge subtraction below are part
// of the bit-exact visibility contract ("exact arithmetic").
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <atomic>
#include <limits>
#include <queue>
#include <thread>
#include <mutex>
#include <system_error>

#include "vhr_internal.hpp"
#include "presplit.hpp"
#include "bvh_frame.hpp"

namespace vhr {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;
    uint32_t first = 0, count = 0;     // leaf range in `order`
    uint32_t depth = 0;
};

// [0, n) cut into one contiguous range per thread (the per-node / per-triangle loops around the tree build: independent items)
template <typename F>
void parallel_for(size_t n, unsigned threads, F &&fn) {
    if (threads <= 1 || n < 16384) { fn(size_t(0), n); return; }
    std::vector<std::thread> pool;
    const size_t chunk = (n + threads - 1) / threads;
    for (unsigned t = 1; t < threads; ++t) {
        const size_t b = std::min(n, size_t(t) * chunk), e = std::min(n, b + chunk);
        if (b >= e) continue;
        // a thread that cannot be created (std::system_error) must not escape through the C boundary: its range runs here instead
        try { pool.emplace_back([&fn, b, e]() { fn(b, e); }); } catch (const std::system_error &) { fn(b, e); }
    }
    fn(size_t(0), std::min(n, chunk));
    for (auto &th : pool) th.join();
}
unsigned host_threads(int threads) { return threads > 0 ? unsigned(std::min(threads, 64)) : std::max(1u, std::min(16u, std::thread::hardware_concurrency())); }

struct Builder {
    std::vector<BvhTri> tris;          // flat order
    std::vector<Box> tri_box;
    std::vector<float> centroid;       // 3 per tri
    std::vector<uint32_t> order;
    std::vector<TmpNode> nodes;
    uint32_t max_depth = 0;
    int leaf_tris = kMaxLeafTris;      // per build (vhr_set_option "bvh_leaf_triangles" of the context that builds)
    unsigned top_threads = 1;          // > 1 while the one thread that builds the top of the tree may spread a big node's passes over triangles

    uint32_t levels_needed(uint32_t count) const {
        uint32_t leaves = (count + leaf_tris - 1) / leaf_tris;
        uint32_t l = 0;
        while ((1u << l) < leaves) ++l;
        return l;
    }

    // Subtrees of at most `defer_below` triangles are not built here but recorded in `tasks` (their link is the code
    // kDeferred - task index): the top of the tree is built by one thread, the subtrees below it by a pool, each into a node
    // vector of its own, and spliced in afterwards (build_parallel).  `order` is partitioned in place on disjoint ranges, so the
    // workers share it without locks; the result is the tree the one-thread build makes (same splits: nothing depends on node ids).
    struct Task { uint32_t first, count, depth; };
    static constexpr int32_t kDeferred = -1000000000;

    int32_t build(std::vector<TmpNode> &nodes, uint32_t &max_depth, uint32_t first, uint32_t count, uint32_t depth, uint32_t defer_below = 0,
                  std::vector<Task> *tasks = nullptr) {
        if (tasks && count <= defer_below && count > uint32_t(leaf_tris)) {
            tasks->push_back(Task{ first, count, depth });
            return kDeferred - int32_t(tasks->size() - 1);
        }
        int32_t id = int32_t(nodes.size());
        nodes.emplace_back();
        TmpNode n;
        n.box.reset();
        Box cb;
        cb.reset();
        const bool wide = top_threads > 1 && count >= 131072u;      // min / max and counts: the same values in any order
        if (wide) {
            std::mutex m;
            parallel_for(count, top_threads, [&](size_t i0, size_t i1) {
                Box pb, pc;
                pb.reset(); pc.reset();
                for (size_t i = first + i0; i < first + i1; ++i) { pb.grow(tri_box[order[i]]); pc.grow(&centroid[size_t(order[i]) * 3]); }
                std::lock_guard<std::mutex> g(m);
                n.box.grow(pb); cb.grow(pc);
            });
        } else {
            for (uint32_t i = first; i < first + count; ++i) {
                n.box.grow(tri_box[order[i]]);
                cb.grow(&centroid[size_t(order[i]) * 3]);
            }
        }
        n.first = first;
        n.count = count;
        n.depth = depth;
        max_depth = std::max(max_depth, depth);
        if (count <= uint32_t(leaf_tris)) {
            nodes[id] = n;
            return id;
        }
        uint32_t mid = 0;
        bool force_median = depth + levels_needed(count) + 1 >= uint32_t(kMaxBvhDepth);
        if (!force_median) {
            constexpr int kBins = 16;
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.0f)) continue;
                Box bin_box[kBins];
                uint32_t bin_count[kBins] = {};
                for (auto &b : bin_box) b.reset();
                float scale = float(kBins) / ext;
                auto bin_range = [&](size_t i0, size_t i1, Box *bb, uint32_t *bc) {
                    for (size_t i = first + i0; i < first + i1; ++i) {
                        uint32_t t = order[i];
                        int b = std::min(kBins - 1, std::max(0, int((centroid[size_t(t) * 3 + axis] - cb.lo[axis]) * scale)));
                        bb[b].grow(tri_box[t]);
                        ++bc[b];
                    }
                };
                if (wide) {
                    std::mutex m;
                    parallel_for(count, top_threads, [&](size_t i0, size_t i1) {
                        Box bb[kBins];
                        uint32_t bc[kBins] = {};
                        for (auto &b : bb) b.reset();
                        bin_range(i0, i1, bb, bc);
                        std::lock_guard<std::mutex> g(m);
                        for (int b = 0; b < kBins; ++b) { bin_box[b].grow(bb[b]); bin_count[b] += bc[b]; }
                    });
                } else {
                    bin_range(0, count, bin_box, bin_count);
                }
                float right_area[kBins];
                uint32_t right_count[kBins];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    c += bin_count[b];
                    right_area[b] = acc.half_area();
                    right_count[b] = c;
                }
                acc.reset();
                c = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bin_box[b]);
                    c += bin_count[b];
                    if (c == 0 || right_count[b + 1] == 0) continue;
                    float cost = acc.half_area() * float(c) + right_area[b + 1] * float(right_count[b + 1]);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                float ext = cb.hi[best_axis] - cb.lo[best_axis];
                float scale = 16.0f / ext;
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                    int b = std::min(15, std::max(0, int((centroid[size_t(t) * 3 + best_axis] - cb.lo[best_axis]) * scale)));
                    return b <= best_bin;
                });
                mid = uint32_t(it - order.begin());
            }
        }
        if (mid <= first || mid >= first + count) {      // forced or degenerate: split by count along the widest axis
            int axis = 0;
            float ext = cb.hi[0] - cb.lo[0];
            if (cb.hi[1] - cb.lo[1] > ext) { axis = 1; ext = cb.hi[1] - cb.lo[1]; }
            if (cb.hi[2] - cb.lo[2] > ext) { axis = 2; }
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) {
                                 float ca = centroid[size_t(a) * 3 + axis], cb2 = centroid[size_t(b) * 3 + axis];
                                 return ca < cb2 || (!(cb2 < ca) && a < b);      // a strict weak order whatever the values (inputs are validated finite)
                             });
        }
        n.count = 0;
        nodes[id] = n;
        int32_t l = build(nodes, max_depth, first, mid - first, depth + 1, defer_below, tasks);
        int32_t r = build(nodes, max_depth, mid, first + count - mid, depth + 1, defer_below, tasks);
        nodes[id].left = l;
        nodes[id].right = r;
        return id;
    }

    // splice: a subtree's nodes keep their relative links, its root (its first node) replaces the deferred code in dst[0 .. top)
    static void splice(std::vector<TmpNode> &dst, std::vector<std::vector<TmpNode>> &subs) {
        const size_t top = dst.size();
        std::vector<int32_t> root_of(subs.size());
        size_t total = top;
        for (const auto &v : subs) total += v.size();
        dst.reserve(total);
        for (size_t k = 0; k < subs.size(); ++k) {
            const int32_t offset = int32_t(dst.size());
            root_of[k] = offset;
            for (TmpNode t : subs[k]) {
                if (t.left >= 0) { t.left += offset; t.right += offset; }
                dst.push_back(t);
            }
            std::vector<TmpNode>().swap(subs[k]);
        }
        for (size_t i = 0; i < top; ++i) {
            TmpNode &t = dst[i];
            if (t.left <= kDeferred) t.left = root_of[size_t(kDeferred - t.left)];
            if (t.right <= kDeferred) t.right = root_of[size_t(kDeferred - t.right)];
        }
    }
    template <typename F>
    static void run_pool(unsigned threads, size_t n, F &&fn) {
        std::atomic<size_t> next{ 0 };
        auto worker = [&]() { for (size_t k = next.fetch_add(1); k < n; k = next.fetch_add(1)) fn(k); };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads && t < n; ++t) {
            try { pool.emplace_back(worker); } catch (const std::system_error &) { break; }      // fewer workers: the queue is drained all the same
        }
        worker();
        for (auto &t : pool) t.join();
    }

    // Three phases.  A: one thread splits the top of the tree down to regions of <= n / 8 triangles (the passes over a big node's
    // triangles spread over the threads).  B: the regions' own tops, one region per thread, down to subtrees of a few thousand
    // triangles.  C: those subtrees, from one queue.  Every piece is built into a node vector of its own and spliced in afterwards;
    // `order` is partitioned in place on disjoint ranges, so nothing is shared, and the tree is the one thread's tree.
    void build_parallel(uint32_t n, int threads) {
        const unsigned hw = host_threads(threads);
        if (hw == 1 || n < 32768u) {                     // small scenes: one thread
            build(nodes, max_depth, 0, n, 0);
            return;
        }
        const uint32_t region = std::max<uint32_t>(16384u, n / 8u), small = std::max<uint32_t>(4096u, n / (8u * hw));
        std::vector<Task> regions;
        top_threads = hw;
        build(nodes, max_depth, 0, n, 0, region, &regions);
        top_threads = 1;
        std::vector<std::vector<TmpNode>> mid(regions.size());
        std::vector<std::vector<Task>> tasks(regions.size());
        std::vector<uint32_t> mid_depth(regions.size(), 0);
        run_pool(hw, regions.size(), [&](size_t k) {
            const Task &r = regions[k];
            build(mid[k], mid_depth[k], r.first, r.count, r.depth, small, r.count > small ? &tasks[k] : nullptr);
        });
        struct Ref { uint32_t region, task; };
        std::vector<Ref> flat;
        for (size_t k = 0; k < regions.size(); ++k)
            for (size_t j = 0; j < tasks[k].size(); ++j) flat.push_back(Ref{ uint32_t(k), uint32_t(j) });
        std::vector<std::vector<std::vector<TmpNode>>> sub(regions.size());
        for (size_t k = 0; k < regions.size(); ++k) sub[k].resize(tasks[k].size());
        std::vector<uint32_t> sub_depth(flat.size(), 0);
        run_pool(hw, flat.size(), [&](size_t f) {
            const Task &t = tasks[flat[f].region][flat[f].task];
            auto &v = sub[flat[f].region][flat[f].task];
            v.reserve(size_t(t.count));
            build(v, sub_depth[f], t.first, t.count, t.depth);
        });
        for (size_t k = 0; k < regions.size(); ++k) splice(mid[k], sub[k]);
        splice(nodes, mid);
        for (uint32_t d : mid_depth) max_depth = std::max(max_depth, d);
        for (uint32_t d : sub_depth) max_depth = std::max(max_depth, d);
    }
};

inline void padded(const Box &b, float lo[3], float hi[3]) {
    // Conservative padding: box culling must never change which triangles are accepted.
    for (int a = 0; a < 3; ++a) {
        float pad = 1e-3f + 1e-5f * std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
        lo[a] = b.lo[a] - pad;
        hi[a] = b.hi[a] + pad;
    }
}

// value of a half bit pattern (exact)
inline double half_value(uint16_t h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const double v = e == 0 ? std::ldexp(double(m), -24) : (e == 31 ? (m ? std::nan("") : HUGE_VAL) : std::ldexp(double(1024 + m), e - 25));
    return (h & 0x8000) ? -v : v;
}

uint16_t half_directed(float x, bool down);

// One axis of one child box of the 32-byte node form (BvhNode16): centre (relative to `origin`) and half extent as halves such that
// origin + c +- h contains [lo, hi] in exact arithmetic with a few fp32 ulp to spare (the walker rounds o - origin and the FMAs).
// Neither half is subnormal.  An absent child (lo > hi) gets c = 0, h = -1.  false: the half range does not reach (the form is unusable).
inline bool half_centre_extent(float lo, float hi, float origin, uint16_t &c16, uint16_t &h16) {
    if (!(lo <= hi)) { c16 = 0; h16 = 0xbc00; return true; }
    const double mid = 0.5 * double(lo) + 0.5 * double(hi) - double(origin);
    // nearest half of the centre: directed conversion both ways, the closer one (ties: either is fine)
    const uint16_t dn = half_directed(float(mid), true), up = half_directed(float(mid), false);
    c16 = std::fabs(half_value(dn) - mid) <= std::fabs(half_value(up) - mid) ? dn : up;
    if (((c16 >> 10) & 31) == 0) c16 = 0;                                 // subnormal centre: 0 (h below absorbs the difference)
    if (((c16 >> 10) & 31) == 31) return false;
    const double c = double(origin) + half_value(c16);
    double need = std::max(double(hi) - c, c - double(lo));
    need += (std::fabs(double(origin)) + std::fabs(half_value(c16)) + need) * 4.8e-7 + 1e-30;
    float nf = float(need);
    if (double(nf) < need) nf = std::nextafter(nf, std::numeric_limits<float>::infinity());
    h16 = half_directed(nf, false);
    if (((h16 >> 10) & 31) == 0) h16 = 0x0400;                            // smallest normal half
    if (((h16 >> 10) & 31) == 31) return false;
    return true;
}

// float -> half bits, rounded toward -inf (down = true) or +inf; |x| beyond the half range saturates outward to +-inf
inline uint16_t half_directed(float x, bool down) {
    if (std::isnan(x)) return 0x7e00;
    if (std::isinf(x)) return x < 0 ? 0xfc00 : 0x7c00;
    // nearest-even conversion first, then step one ulp outward if it landed on the wrong side
    uint32_t u;
    std::memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const float ax = std::fabs(x);
    uint16_t h;
    if (ax >= 65520.0f) h = uint16_t(sign | 0x7c00u);
    else if (ax < 5.9604645e-8f * 0.5f) h = uint16_t(sign);
    else {
        int e;
        const float m = std::frexp(ax, &e);                 // ax = m * 2^e, m in [0.5, 1)
        int he = e + 14;                                     // half exponent field for normals
        if (he <= 0) {                                       // subnormal half: units of 2^-24
            const uint32_t q = uint32_t(std::nearbyint(std::ldexp(ax, 24)));
            h = uint16_t(sign | q);
        } else {
            uint32_t q = uint32_t(std::nearbyint(std::ldexp(m, 11)));   // 11-bit significand incl. implicit bit
            if (q == 2048u) { q = 1024u; ++he; }
            h = (he >= 31) ? uint16_t(sign | 0x7c00u) : uint16_t(sign | (uint32_t(he) << 10) | (q & 0x3ffu));
        }
    }
    auto value = [](uint16_t hb) -> float {
        const uint32_t s2 = hb & 0x8000u, ex = (hb >> 10) & 0x1fu, ma = hb & 0x3ffu;
        float v;
        if (ex == 0) v = std::ldexp(float(ma), -24);
        else if (ex == 31) v = ma ? NAN : INFINITY;
        else v = std::ldexp(float(ma | 0x400u), int(ex) - 25);
        return s2 ? -v : v;
    };
    auto step = [](uint16_t hb, bool up) -> uint16_t {     // next representable half toward +inf (up) or -inf
        if ((hb & 0x7fffu) == 0) return up ? uint16_t(0x0001) : uint16_t(0x8001);
        const bool neg = hb & 0x8000u;
        return (neg == up) ? uint16_t(hb - 1) : uint16_t(hb + 1);
    };
    const float v = value(h);
    if (down && v > x) h = step(h, false);
    if (!down && v < x) h = step(h, true);
    return h;
}

inline int32_t leaf_link(uint32_t first, uint32_t count) { return ~int32_t((first << 2) | (count - 1)); }


}  // namespace

void build_bvh(const vhr_vertex *vertices, const uint32_t *indices, const vhr_primitive *primitives,
               uint32_t primitive_count, HostBvh &out, int leaf_tris, int threads, int presplit_percent, int frame_mode) {
    const bool k0trace = std::getenv("VHR_K0_TRACE") != nullptr; auto k0t = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (k0trace) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "K0 %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - k0t).count()); k0t = t; } };
    Builder b;
    b.leaf_tris = std::max(1, std::min(kMaxLeafTris, leaf_tris));
    size_t total = 0;
    for (uint32_t p = 0; p < primitive_count; ++p) total += primitives[p].index_count / 3;
    b.tris.reserve(total);
    for (uint32_t p = 0; p < primitive_count; ++p) {
        const vhr_primitive &pr = primitives[p];
        const float *m = pr.transform;
        for (uint32_t t = 0; t < pr.index_count / 3; ++t) {
            float w[3][3];
            for (int c = 0; c < 3; ++c) {
                const float *v = vertices[pr.vertex_offset + indices[pr.index_offset + 3 * t + c]].pos;
                // transform * vec4(pos, 1), columns accumulated left to right, no contraction
                w[c][0] = ((m[0] * v[0] + m[4] * v[1]) + m[8] * v[2]) + m[12];
                w[c][1] = ((m[1] * v[0] + m[5] * v[1]) + m[9] * v[2]) + m[13];
                w[c][2] = ((m[2] * v[0] + m[6] * v[1]) + m[10] * v[2]) + m[14];
            }
            BvhTri tri;
            for (int a = 0; a < 3; ++a) {
                tri.v0[a] = w[0][a];
                tri.e1[a] = w[1][a] - w[0][a];
                tri.e2[a] = w[2][a] - w[0][a];
            }
            tri.prim = p;
            tri.tri = t;
            tri.flat = uint32_t(b.tris.size());
            b.tris.push_back(tri);
        }
    }
    uint32_t n = uint32_t(b.tris.size());
    out.nodes.clear();
    out.nodes16.clear();
    out.nodes_ch.clear();
    out.nodes48.clear();
    out.tris.clear();
    out.max_depth = 0;
    if (n == 0) return;

    b.tri_box.resize(n);
    b.centroid.resize(size_t(n) * 3);
    b.order.resize(n);
    const unsigned hw = host_threads(threads);
    // "bvh_frame" 1: the frame the boxes are built in (bvh_frame.hpp; the device builder runs the same search with a kernel as the pass)
    out.frame_on = false;
    { const float identity[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; std::memcpy(out.frame, identity, sizeof(identity)); }
    if (frame_mode == 1 && n >= 2u) {
        float frame[9];
        const bool found = bvh_frame::choose([&](const bvh_frame::Candidates &c, uint64_t *sums) {
            for (int k = 0; k < c.n; ++k) sums[k] = 0;
            std::mutex m;
            const size_t stride = bvh_frame::sample_stride(n), samples = (size_t(n) + stride - 1) / stride;
            parallel_for(samples, hw, [&](size_t i0, size_t i1) {
                uint64_t part[bvh_frame::kMaxCandidates] = {};
                for (size_t i = i0; i < i1; ++i)
                    for (int k = 0; k < c.n; ++k) part[k] += bvh_frame::cost_term(c.r[k], b.tris[i * stride]);
                std::lock_guard<std::mutex> lock(m);
                for (int k = 0; k < c.n; ++k) sums[k] += part[k];
            });
        }, frame);
        if (found) { std::memcpy(out.frame, frame, sizeof(frame)); out.frame_on = true; }
        lap("frame search");
    }
    const bool framed = out.frame_on;
    parallel_for(n, hw, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
        const BvhTri &t = b.tris[i];
        Box bx;
        bx.reset();
        if (framed) {
            bvh_frame::box_in_frame(out.frame, t, bx.lo, bx.hi);
        } else {
            float p1[3], p2[3];
            for (int a = 0; a < 3; ++a) { p1[a] = t.v0[a] + t.e1[a]; p2[a] = t.v0[a] + t.e2[a]; }
            bx.grow(t.v0);
            bx.grow(p1);
            bx.grow(p2);
        }
        b.tri_box[i] = bx;
        for (int a = 0; a < 3; ++a) b.centroid[size_t(i) * 3 + a] = 0.5f * (bx.lo[a] + bx.hi[a]);
        b.order[i] = uint32_t(i);
    }
    });
    lap("triangles + boxes");
    out.presplit_level = -1;
    if (presplit_percent > 0 && n >= 2u && !framed) {           // (a rotated frame takes the place of splitting: the option is not combined with it)
        // fat triangles entered once per grid cell they pass through (presplit.hpp; the device builder's k0_presplit_* kernels do the same)
        float clo[3] = { b.centroid[0], b.centroid[1], b.centroid[2] }, chi[3] = { b.centroid[0], b.centroid[1], b.centroid[2] };
        for (size_t i = 1; i < n; ++i)
            for (int a = 0; a < 3; ++a) { clo[a] = std::min(clo[a], b.centroid[i * 3 + a]); chi[a] = std::max(chi[a], b.centroid[i * 3 + a]); }
        uint64_t estimates[presplit::kLevels] = {};
        for (int k = 0; k < presplit::kLevels; ++k) {
            const presplit::Grid g = presplit::make_grid(clo, chi, k);
            std::mutex m;
            parallel_for(n, hw, [&](size_t i0, size_t i1) {
                uint64_t sum = 0;
                for (size_t i = i0; i < i1; ++i) sum += presplit::estimate(g, b.tris[i], b.tri_box[i].lo, b.tri_box[i].hi);
                std::lock_guard<std::mutex> lock(m);
                estimates[k] += sum;
            });
        }
        int level = presplit::choose_level(estimates, n, uint32_t(presplit_percent), clo, chi);
        const uint64_t hard_limit = uint64_t(n) + 2ull * uint64_t(n) * uint64_t(presplit_percent) / 100ull;
        std::vector<uint32_t> refs(n);
        uint64_t total = n;
        for (; level >= 0; --level) {                 // the exact count; an estimate that was too low by more than 2x goes one level up
            const presplit::Grid g = presplit::make_grid(clo, chi, level);
            parallel_for(n, hw, [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; ++i) refs[i] = presplit::references(g, b.tris[i], b.tri_box[i].lo, b.tri_box[i].hi, [](const float *, const float *) {});
            });
            total = 0;
            for (uint32_t r : refs) total += r;
            if (total <= hard_limit && total < (1ull << 31)) break;
        }
        if (level >= 0 && total > n) {
            const presplit::Grid g = presplit::make_grid(clo, chi, level);
            std::vector<uint64_t> start(size_t(n) + 1, 0);
            for (size_t i = 0; i < n; ++i) start[i + 1] = start[i] + refs[i];
            std::vector<BvhTri> tris2(total);
            std::vector<Box> box2(total);
            parallel_for(n, hw, [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; ++i) {
                    uint64_t at = start[i];
                    presplit::references(g, b.tris[i], b.tri_box[i].lo, b.tri_box[i].hi, [&](const float *lo, const float *hi) {
                        tris2[at] = b.tris[i];
                        for (int a = 0; a < 3; ++a) { box2[at].lo[a] = lo[a]; box2[at].hi[a] = hi[a]; }
                        ++at;
                    });
                }
            });
            b.tris.swap(tris2);
            b.tri_box.swap(box2);
            n = uint32_t(total);
            b.centroid.resize(size_t(n) * 3);
            b.order.resize(n);
            for (size_t i = 0; i < n; ++i) {
                for (int a = 0; a < 3; ++a) b.centroid[i * 3 + a] = 0.5f * (b.tri_box[i].lo[a] + b.tri_box[i].hi[a]);
                b.order[i] = uint32_t(i);
            }
            out.presplit_level = level;
        }
        if (k0trace) std::fprintf(stderr, "K0 presplit: level %d, %zu triangles -> %u references\n", level, refs.size(), n);
        lap("presplit");
    }
    b.nodes.reserve(size_t(n));
    b.build_parallel(n, threads);
    lap("tree");
    out.max_depth = b.max_depth;

    // Triangle order: by TREELETS.  Breadth first over groups of up to four subtrees -- an inner node with its two children, of which the
    // inner one with the largest box is opened (replaced by its two children, in place) until there are four or only leaves are left --
    // and the leaves of one group are neighbours in `tris`: the two to four leaves a ray reaches within two steps of each other share
    // cache lines (against the plain breadth-first order of the binary tree's leaves, r4: equal on sponza_proc, the any-hit launch -3 % on
    // bistro_proc).
    std::vector<uint32_t> leaf_pos(b.nodes.size(), 0u);
    if (b.nodes[0].left >= 0) {
        uint32_t pos = 0;
        std::vector<int32_t> queue{ 0 };
        for (size_t head = 0; head < queue.size(); ++head) {
            const int32_t id = queue[head];
            int32_t child[4] = { b.nodes[id].left, b.nodes[id].right, -1, -1 };
            int nc = 2;
            while (nc < 4) {
                int best = -1;
                float best_area = -1.0f;
                for (int c = 0; c < nc; ++c) {
                    const TmpNode &t = b.nodes[child[c]];
                    if (t.left < 0) continue;
                    const float area = t.box.half_area();
                    if (area > best_area) { best_area = area; best = c; }
                }
                if (best < 0) break;
                const TmpNode &t = b.nodes[child[best]];
                for (int c = nc; c > best + 1; --c) child[c] = child[c - 1];
                child[best] = t.left;
                child[best + 1] = t.right;
                ++nc;
            }
            for (int c = 0; c < nc; ++c) {
                if (b.nodes[child[c]].left >= 0) queue.push_back(child[c]);
                else { leaf_pos[child[c]] = pos; pos += b.nodes[child[c]].count; }
            }
        }
    }
    out.tris.resize(n);
    parallel_for(b.nodes.size(), hw, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const TmpNode &t = b.nodes[i];
            if (t.left >= 0) continue;
            for (uint32_t j = 0; j < t.count; ++j) out.tris[leaf_pos[i] + j] = b.tris[b.order[t.first + j]];
        }
    });
    lap("triangle order");

    const float inf = std::numeric_limits<float>::infinity();
    auto set_child = [&](BvhNode &node, int which, const TmpNode &child, int32_t link) {
        float lo[3], hi[3];
        padded(child.box, lo, hi);
        float *dst = which == 0 ? node.box0 : node.box1;
        for (int a = 0; a < 3; ++a) { dst[2 * a] = lo[a]; dst[2 * a + 1] = hi[a]; }
        (which == 0 ? node.child0 : node.child1) = link;
    };

    // centre / half-extent twin of every node (BvhNodeCH): c +- h must contain [lo, hi] in exact arithmetic
    auto finalize_ch = [&]() {
        out.nodes_ch.resize(out.nodes.size());
        parallel_for(out.nodes.size(), hw, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const BvhNode &nd = out.nodes[k];
            BvhNodeCH c{};
            for (int which = 0; which < 2; ++which) {
                const float *box = which == 0 ? nd.box0 : nd.box1;
                float *hdst = which == 0 ? c.h0 : c.h1;
                for (int a = 0; a < 3; ++a) {
                    const float lo = box[2 * a], hi = box[2 * a + 1];
                    float cc = 0.0f, hh = -1.0f;                   // absent child: never entered
                    if (lo <= hi) {
                        cc = 0.5f * lo + 0.5f * hi;
                        hh = std::max(hi - cc, cc - lo);
                        hh += (std::fabs(cc) + hh) * 2.4e-7f;      // 4 ulp of the magnitudes involved
                        while (double(cc) - double(hh) > double(lo) || double(cc) + double(hh) < double(hi)) hh = std::nextafter(hh, inf);
                    }
                    (a == 0 ? c.cx : a == 1 ? c.cy : c.cz)[which] = cc;
                    hdst[a] = hh;
                }
            }
            c.child0 = nd.child0;
            c.child1 = nd.child1;
            out.nodes_ch[k] = c;
        }
        });
        // the 48-byte form: half extents as the upper half of their fp32 pattern, rounded up (away from zero for the -1 of an
        // absent child, which is exact anyway)
        auto upper16 = [](float h) -> uint32_t {
            uint32_t bits;
            std::memcpy(&bits, &h, 4);
            if (h > 0.0f && (bits & 0xffffu)) bits += 0x10000u;      // next value with 16 zero bits below (an overflow would give +inf: no finite h gets there)
            return bits >> 16;
        };
        out.nodes48.resize(out.nodes_ch.size());
        parallel_for(out.nodes_ch.size(), hw, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const BvhNodeCH &c = out.nodes_ch[k];
            BvhNode48 n{};
            for (int w = 0; w < 2; ++w) { n.cx[w] = c.cx[w]; n.cy[w] = c.cy[w]; n.cz[w] = c.cz[w]; }
            n.hp[0] = (upper16(c.h0[0]) << 16) | upper16(c.h0[1]);
            n.hp[1] = (upper16(c.h0[2]) << 16) | upper16(c.h1[0]);
            n.hp[2] = (upper16(c.h1[1]) << 16) | upper16(c.h1[2]);
            // inner links as BYTE offsets (index * 48): the walkers add them to the base address as they are
            n.child0 = c.child0 >= 0 ? c.child0 * int32_t(sizeof(BvhNode48)) : c.child0;
            n.child1 = c.child1 >= 0 ? c.child1 * int32_t(sizeof(BvhNode48)) : c.child1;
            out.nodes48[k] = n;
        }
        });
    };

    auto finalize16 = [&]() {
        finalize_ch();
        float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
        for (const BvhNode &nd : out.nodes)
            for (int a = 0; a < 3; ++a) {
                if (nd.box0[2 * a] <= nd.box0[2 * a + 1]) { lo[a] = std::min(lo[a], nd.box0[2 * a]); hi[a] = std::max(hi[a], nd.box0[2 * a + 1]); }
                if (nd.box1[2 * a] <= nd.box1[2 * a + 1]) { lo[a] = std::min(lo[a], nd.box1[2 * a]); hi[a] = std::max(hi[a], nd.box1[2 * a + 1]); }
            }
        for (int a = 0; a < 3; ++a) out.centre[a] = (lo[a] <= hi[a]) ? 0.5f * (lo[a] + hi[a]) : 0.0f;
        out.nodes16.resize(out.nodes.size());
        std::atomic<uint32_t> overflow{ 0 };
        parallel_for(out.nodes.size(), hw, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const BvhNode &nd = out.nodes[k];
            BvhNode16 c{};
            for (int which = 0; which < 2; ++which) {
                const float *box = which == 0 ? nd.box0 : nd.box1;
                for (int a = 0; a < 3; ++a) {
                    uint16_t c16, h16;
                    if (!half_centre_extent(box[2 * a], box[2 * a + 1], out.centre[a], c16, h16)) overflow = 1;
                    c.c[2 * a + which] = c16;
                    c.h[2 * a + which] = h16;
                }
            }
            // inner links as BYTE offsets (index * 32), like the 48-byte nodes'
            c.child0 = nd.child0 >= 0 ? nd.child0 * int32_t(sizeof(BvhNode16)) : nd.child0;
            c.child1 = nd.child1 >= 0 ? nd.child1 * int32_t(sizeof(BvhNode16)) : nd.child1;
            out.nodes16[k] = c;
        }
        });
        out.nodes16_valid = overflow == 0 && out.nodes.size() * sizeof(BvhNode16) < (size_t(1) << 31);
    };

    const TmpNode &root = b.nodes[0];
    if (root.left < 0) {                       // whole scene fits one leaf
        BvhNode node{};
        set_child(node, 0, root, leaf_link(leaf_pos[0], root.count));
        for (int a = 0; a < 3; ++a) { node.box1[2 * a] = inf; node.box1[2 * a + 1] = -inf; }
        node.child1 = node.child0;
        out.nodes.push_back(node);
        finalize16();
        return;
    }
    // breadth-first numbering of the inner nodes
    std::vector<int32_t> bfs_index(b.nodes.size(), -1);
    std::vector<int32_t> bfs_order;
    std::queue<int32_t> q;
    q.push(0);
    while (!q.empty()) {
        int32_t id = q.front();
        q.pop();
        bfs_index[id] = int32_t(bfs_order.size());
        bfs_order.push_back(id);
        const TmpNode &t = b.nodes[id];
        if (b.nodes[t.left].left >= 0) q.push(t.left);
        if (b.nodes[t.right].left >= 0) q.push(t.right);
    }
    out.nodes.resize(bfs_order.size());
    parallel_for(bfs_order.size(), hw, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
        const TmpNode &t = b.nodes[bfs_order[k]];
        BvhNode node{};
        const TmpNode &l = b.nodes[t.left], &r = b.nodes[t.right];
        set_child(node, 0, l, l.left >= 0 ? bfs_index[t.left] : leaf_link(leaf_pos[t.left], l.count));
        set_child(node, 1, r, r.left >= 0 ? bfs_index[t.right] : leaf_link(leaf_pos[t.right], r.count));
        out.nodes[k] = node;
    }
    });
    lap("numbering + (lo, hi) nodes");
    finalize16();
    lap("derived node forms");
}

// A 64-bit multiplicative hash over the (lo, hi) nodes and the leaf triangles in their final order, eight bytes at a time: the
// identity of a build (tests: the tree must not depend on the number of build threads)
uint64_t bvh_fingerprint(const HostBvh &bvh) {
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const void *p, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i + 8 <= bytes; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ull; h ^= h >> 29; }
    };
    static_assert(sizeof(BvhNode) % 8 == 0 && sizeof(BvhTri) % 8 == 0, "whole words");
    eat(bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode));
    eat(bvh.tris.data(), bvh.tris.size() * sizeof(BvhTri));
    return h;
}

// A hash of the TREE rather than of its arrays: per inner node the bits of its two child boxes and its children's hashes (left, right),
// per leaf the flat ids of its triangles in ascending order -- whatever the numbering of the nodes, the order of the leaves in memory and
// the order of the triangles inside a leaf.  Two builders that make the same tree agree on it (the host's and the device's: tests).
// Needs parents before children in `nodes` (both builders number breadth first).
uint64_t bvh_tree_fingerprint(const HostBvh &bvh) {
    auto mix = [](uint64_t h, uint64_t w) { h = (h ^ w) * 1099511628211ull; return h ^ (h >> 29); };
    auto leaf_hash = [&](int32_t link) {
        const uint32_t v = ~uint32_t(link), first = v >> 2, count = (v & 3u) + 1u;
        uint32_t ids[4] = { 0, 0, 0, 0 };
        for (uint32_t i = 0; i < count; ++i) ids[i] = bvh.tris[first + i].flat;
        std::sort(ids, ids + count);
        uint64_t h = mix(14695981039346656037ull, count);
        for (uint32_t i = 0; i < count; ++i) h = mix(h, ids[i]);
        return h;
    };
    std::vector<uint64_t> sig(bvh.nodes.size(), 0);
    for (size_t k = bvh.nodes.size(); k-- > 0;) {
        const BvhNode &nd = bvh.nodes[k];
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 6; ++i) { uint32_t b0, b1; std::memcpy(&b0, &nd.box0[i], 4); std::memcpy(&b1, &nd.box1[i], 4); h = mix(h, (uint64_t(b0) << 32) | b1); }
        const bool absent1 = !(nd.box1[0] <= nd.box1[1]);                   // (a one-leaf scene: child 1 is a copy of child 0 behind an empty box)
        h = mix(h, nd.child0 >= 0 ? sig[size_t(nd.child0)] : leaf_hash(nd.child0));
        h = mix(h, absent1 ? 0ull : (nd.child1 >= 0 ? sig[size_t(nd.child1)] : leaf_hash(nd.child1)));
        sig[k] = h;
    }
    return sig.empty() ? 0ull : sig[0];
}

// Every derived node form must CONTAIN the (lo, hi) boxes of `nodes` in exact arithmetic -- that is all the walkers' bit-identity
// rests on (boxes only cull).  out: boxes checked, centre / half-extent boxes that do not contain theirs, 48-byte boxes that do not
// contain the centre / half-extent box, half-precision 32-byte (compact) boxes that do not contain theirs.  (vhr_get_bvh_form_checks)
void check_node_forms(const HostBvh &bvh, uint64_t out[4], int threads) {
    auto upper = [](uint32_t w16) -> double { const uint32_t bits = w16 << 16; float f; std::memcpy(&f, &bits, 4); return double(f); };
    out[0] = out[1] = out[2] = out[3] = 0;
    std::atomic<uint64_t> total[4];
    for (auto &t : total) t = 0;
    parallel_for(bvh.nodes.size(), host_threads(threads), [&](size_t k0, size_t k1) {
    uint64_t out[4] = { 0, 0, 0, 0 };                  // this thread's counts
    for (size_t k = k0; k < k1; ++k) {
        const BvhNode &nd = bvh.nodes[k];
        const BvhNodeCH &ch = bvh.nodes_ch[k];
        const BvhNode48 &n48 = bvh.nodes48[k];
        const BvhNode16 &n16 = bvh.nodes16[k];
        const double h48[6] = { upper(n48.hp[0] >> 16), upper(n48.hp[0] & 0xffffu), upper(n48.hp[1] >> 16), upper(n48.hp[1] & 0xffffu), upper(n48.hp[2] >> 16), upper(n48.hp[2] & 0xffffu) };
        for (int which = 0; which < 2; ++which) {
            const float *box = which == 0 ? nd.box0 : nd.box1;
            const float *hh = which == 0 ? ch.h0 : ch.h1;
            ++out[0];
            for (int a = 0; a < 3; ++a) {
                const double lo = box[2 * a], hi = box[2 * a + 1];
                const double c = (a == 0 ? ch.cx : a == 1 ? ch.cy : ch.cz)[which], h = hh[a];
                const double c48 = (a == 0 ? n48.cx : a == 1 ? n48.cy : n48.cz)[which], hw = h48[3 * which + a];
                // the 32-byte form: centre + c16 +- h16 in exact arithmetic, no subnormal / inf / NaN half (checked only when the form is in use)
                const uint16_t cb = n16.c[2 * a + which], hb = n16.h[2 * a + which];
                const double c16 = double(bvh.centre[a]) + half_value(cb), h16 = half_value(hb);
                const bool normal16 = (cb == 0 || (((cb >> 10) & 31) != 0 && ((cb >> 10) & 31) != 31)) && ((hb >> 10) & 31) != 0 && ((hb >> 10) & 31) != 31;
                if (!(lo <= hi)) {                           // an absent child: never entered in any form
                    if (!(h < 0.0)) ++out[1];
                    if (!(hw < 0.0)) ++out[2];
                    if (bvh.nodes16_valid && !(h16 < 0.0)) ++out[3];
                    continue;
                }
                if (c - h > lo || c + h < hi) ++out[1];
                if (c48 != c || hw < h) ++out[2];
                if (bvh.nodes16_valid && (!normal16 || !(c16 - h16 <= lo) || !(c16 + h16 >= hi))) ++out[3];
            }
        }
        auto as48 = [](int32_t link) { return link >= 0 ? link * int32_t(sizeof(BvhNode48)) : link; };
        if (n48.child0 != as48(nd.child0) || n48.child1 != as48(nd.child1) || ch.child0 != nd.child0 || ch.child1 != nd.child1) ++out[2];
        auto as16 = [](int32_t link) { return link >= 0 ? link * int32_t(sizeof(BvhNode16)) : link; };
        if (bvh.nodes16_valid && (n16.child0 != as16(nd.child0) || n16.child1 != as16(nd.child1))) ++out[3];
    }
    for (int i = 0; i < 4; ++i) total[i] += out[i];
    });
    for (int i = 0; i < 4; ++i) out[i] = total[i];
}

bool nodes16_in_range(const HostBvh &bvh) {
    if (bvh.nodes16.size() != bvh.nodes.size() || bvh.nodes16.size() * sizeof(BvhNode16) >= (size_t(1) << 31)) return false;
    for (const BvhNode16 &n : bvh.nodes16)
        for (int i = 0; i < 6; ++i) {
            const int ec = (n.c[i] >> 10) & 31, eh = (n.h[i] >> 10) & 31;
            if (ec == 31 || (ec == 0 && n.c[i] != 0) || eh == 31 || eh == 0) return false;
        }
    return true;
}

}  // namespace vhr
