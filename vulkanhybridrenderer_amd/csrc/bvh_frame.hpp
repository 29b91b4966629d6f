// K0: the frame the boxes live in (option "bvh_frame", default 1 = found by the builder).
//
// A box hierarchy pays for geometry that is not aligned with its axes: the same scene turned 34 deg off the world axes costs +50 .. +140 % on the
// any-hit launch and doubles the mirror ray's node visits (profiles/r5_sponza_hard.txt), and no splitting removes that.  What does, where the
// scene HAS a dominant orientation (a building, a street), is to give the hierarchy that orientation: the boxes are built around the triangles'
// corners in a rotated frame R, and a walker rotates its ray once (18 FMAs) before the slab tests.  Only the boxes move: the leaf records stay
// the world-space records, the triangle test stays raygen.rgen's arithmetic on the world-space ray, so what a ray hits, at which t and with
// which barycentrics is untouched (boxes only cull; the padding of a box, 1e-3 + 1e-5 |x|, is two orders above the rotation's rounding).
//
// The frame is found, not given: R = the rotation (Euler angles about Y, X, Z, coarse to fine) that minimises the sum over the triangles of the
// half area of the triangle's box in the frame -- the quantity every level of a surface-area tree is made of.  The sum is taken in 64-bit
// integers (each term rounded once), so it does not depend on the order of summation and the host's and the device's builder pick the same
// frame (they make the same tree, tests/test_gpu_fuzz.py).  A frame is only used if it beats the world axes by kMinGain; otherwise R = identity
// and nothing changes, bit for bit.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "vhr_internal.hpp"

namespace vhr {
namespace bvh_frame {

constexpr int kMaxCandidates = 24;          // rotations priced per pass over the triangles
constexpr double kMinGain = 0.05;           // a frame has to lower the summed half area by at least this share to be used
constexpr uint32_t kSampleTarget = 262144;  // the search prices every stride-th triangle, stride = the count over this (a scene's orientation does not hide in a sample)

inline uint32_t sample_stride(uint32_t n) { return n > kSampleTarget ? n / kSampleTarget : 1u; }

struct Candidates {
    float r[kMaxCandidates][9];             // row-major: row i = the frame's axis i in world coordinates
    int n;
};

#define VHR_BF __host__ __device__ inline

// R p, every product and sum rounded on its own, in this order (host and device builder: the same bits)
VHR_BF void rotate(const float *R, const float p[3], float out[3]) {
    out[0] = (R[0] * p[0] + R[1] * p[1]) + R[2] * p[2];
    out[1] = (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2];
    out[2] = (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2];
}

// the box of a triangle's three corners in the frame (the corners the walkers intersect: v0, v0 + e1, v0 + e2)
VHR_BF void box_in_frame(const float *R, const BvhTri &t, float lo[3], float hi[3]) {
    float p[3][3], q[3];
    for (int a = 0; a < 3; ++a) { p[0][a] = t.v0[a]; p[1][a] = t.v0[a] + t.e1[a]; p[2][a] = t.v0[a] + t.e2[a]; }
    for (int k = 0; k < 3; ++k) {
        rotate(R, p[k], q);
        for (int a = 0; a < 3; ++a) {
            if (k == 0 || q[a] < lo[a]) lo[a] = q[a];
            if (k == 0 || q[a] > hi[a]) hi[a] = q[a];
        }
    }
}

// one triangle's term of the cost: the half area of its box in the frame, as an integer (2^-20 m^2 units, capped: the sum of 2^22 terms fits 64 bits)
VHR_BF unsigned long long cost_term(const float *R, const BvhTri &t) {
    float lo[3], hi[3];
    box_in_frame(R, t, lo, hi);
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const double ha = double(dx * dy + dy * dz + dz * dx) * 1048576.0;
    if (!(ha >= 0.0)) return 0ull;
    return ha >= 2199023255552.0 ? 2199023255552ull : (unsigned long long)(ha + 0.5);
}

#undef VHR_BF

// R = Rz(roll) Rx(pitch) Ry(yaw), degrees; the trigonometry in double on the host (both builders get the same nine floats)
inline void from_euler(double yaw, double pitch, double roll, float out[9]) {
    const double d2r = 3.14159265358979323846 / 180.0;
    const double cy = std::cos(yaw * d2r), sy = std::sin(yaw * d2r), cp = std::cos(pitch * d2r), sp = std::sin(pitch * d2r), cr = std::cos(roll * d2r), sr = std::sin(roll * d2r);
    const double Ry[9] = { cy, 0, -sy, 0, 1, 0, sy, 0, cy }, Rx[9] = { 1, 0, 0, 0, cp, sp, 0, -sp, cp }, Rz[9] = { cr, sr, 0, -sr, cr, 0, 0, 0, 1 };
    double t[9], m[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { t[3 * i + j] = 0; for (int k = 0; k < 3; ++k) t[3 * i + j] += Rx[3 * i + k] * Ry[3 * k + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { m[3 * i + j] = 0; for (int k = 0; k < 3; ++k) m[3 * i + j] += Rz[3 * i + k] * t[3 * k + j]; }
    for (int i = 0; i < 9; ++i) out[i] = float(m[i]);
}

inline bool is_identity(const float R[9]) {
    static const float I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    return std::memcmp(R, I, sizeof(I)) == 0;
}

// The search.  `evaluate(candidates, sums)`: sums[k] = sum of cost_term(candidates.r[k], t) over the triangles (a pass of the builder that
// calls: host loop or device kernel).  Coordinate descent over (yaw, pitch, roll): yaw in [0, 90) -- a quarter turn about an axis maps the
// axes onto themselves --, pitch and roll in (-45, 45], steps of 6, 1.5, 0.5, 1/8, 1/32 and 1/128 degrees (a wall 40 m long tilted by 0.06 deg has a box 4 cm thick: the last steps are worth +10 % node
// visits on the mirror ray).  Returns true and the frame if it gains kMinGain.
template <typename Evaluate>
inline bool choose(Evaluate &&evaluate, float out[9]) {
    double best[3] = { 0.0, 0.0, 0.0 };
    Candidates c;
    uint64_t sums[kMaxCandidates];
    from_euler(0, 0, 0, c.r[0]);
    c.n = 1;
    evaluate(c, sums);
    const uint64_t world = sums[0];
    uint64_t best_cost = world;
    if (world == 0) return false;
    const double steps[6] = { 6.0, 1.5, 0.5, 0.125, 0.03125, 0.0078125 };
    for (int level = 0; level < 6; ++level) {
        for (int round = 0; round < (level == 0 ? 2 : 1); ++round) {
            for (int angle = 0; angle < 3; ++angle) {
                // candidates along this angle: the whole range at the coarsest level, +-4 steps around the best afterwards
                double values[kMaxCandidates];
                int n = 0;
                if (level == 0) {
                    const double lo = angle == 0 ? 0.0 : -42.0, hi = angle == 0 ? 84.0 : 42.0;
                    for (double v = lo; v <= hi + 1e-9 && n < kMaxCandidates; v += steps[0]) values[n++] = v;
                    // ... and half a step either side of the world axes: a scene tilted by a few degrees lies BETWEEN the grid's points, where both
                    // neighbours cost about what the world axes cost, and the early exit below would keep the world axes (3 deg about y and x on
                    // sponza_proc: 10.7 % of the summed half area lost)
                    values[n++] = 3.0;
                    values[n++] = angle == 0 ? 87.0 : -3.0;
                } else {
                    for (int k = -4; k <= 4; ++k) if (k) values[n++] = best[angle] + k * steps[level];
                }
                c.n = n;
                for (int k = 0; k < n; ++k) {
                    double e[3] = { best[0], best[1], best[2] };
                    e[angle] = values[k];
                    from_euler(e[0], e[1], e[2], c.r[k]);
                }
                evaluate(c, sums);
                for (int k = 0; k < n; ++k)
                    if (sums[k] < best_cost) { best_cost = sums[k]; best[angle] = values[k]; }
            }
            // a scene along the world axes (or without an orientation) shows it in the first coarse round: nothing within 3 degrees of any
            // rotation comes near the gain a frame must have -- the search ends there (a quarter of its passes)
            if (level == 0 && round == 0 && double(best_cost) > double(world) * (1.0 - 0.25 * kMinGain)) return false;
        }
    }
    if (double(best_cost) > double(world) * (1.0 - kMinGain)) return false;
    from_euler(best[0], best[1], best[2], out);
    return !is_identity(out);
}

}  // namespace bvh_frame
}  // namespace vhr
