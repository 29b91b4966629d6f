/*
 * vhr_exact.h -- TEST INFRASTRUCTURE ONLY (part of the oracle; see vhr_oracle.h for the contract).
 *
 * The arbiter behind decision (vi): the comparisons of Moeller-Trumbore (vhr_oracle.c ray_triangle(); the reference's
 * traceRayEXT, raygen.rgen:39,51,64, leaves the arithmetic to the driver) decided WITHOUT rounding on the same fp32
 * inputs (o, d, v0, e1, e2, tmin, tmax).  "The triangle" is the closed triangle v0, v0 + e1, v0 + e2 with e1, e2 the
 * stored fp32 edges; "hit" is
 *      det != 0,  0 <= u,  u <= 1,  0 <= v,  u + v <= 1,  tmin < t < tmax
 * for the exact rationals det = e1 . (d x e2), u = (o - v0) . (d x e2) / det, v = d . ((o - v0) x e1) / det,
 * t = e2 . ((o - v0) x e1) / det.  No division is performed: every comparison is the sign of a polynomial of degree
 * <= 4 in the inputs.
 *
 * Method: each polynomial is evaluated in binary64 together with the same polynomial over the absolute values (its
 * "permanent" P).  An fp32 operand is exact in binary64; every binary64 operation adds a relative error <= 2^-53, and a
 * tree of n operations is off by at most ((1 + 2^-53)^n - 1) P (Higham, Accuracy and Stability of Numerical Algorithms,
 * section 3.1); the longest tree here has 23 operations, the bound used is 64 * 2^-53 * P.  A sign is CERTAIN when the
 * value's magnitude exceeds that bound, or when P itself is 0 (every term is exactly 0).  Anything else is reported as
 * undecided (-1) and the caller settles it in exact rational arithmetic (tests/exact_rational.py, python Fractions on the
 * same bits) -- binary64 has 29 more bits than the inputs, so this happens on exact ties (a ray through a vertex or an
 * edge of axis-aligned geometry) and practically nowhere else.
 */
#ifndef VHR_EXACT_H
#define VHR_EXACT_H

#include <math.h>

typedef struct {
    int decision;            /* 1 hit, 0 miss, -1 undecided in binary64 (settle with exact rationals) */
    double det, u, v, t;     /* binary64 values of the quantities (u, v, t are quotients: reporting only, not part of the decision) */
} orc_exact_result;

/* sign of val given its permanent: +1 / -1 / 0 certain, 2 undecided */
static inline int exact_sign(double val, double perm) {
    const double bound = perm * (64.0 * 0x1p-53);
    if (val > bound) return 1;
    if (val < -bound) return -1;
    if (perm == 0.0) return 0;
    return 2;
}

static inline orc_exact_result exact_ray_triangle(const float of[3], const float df[3], const float v0f[3], const float e1f[3],
                                                  const float e2f[3], float tminf, float tmaxf) {
    orc_exact_result r = { 0, 0.0, 0.0, 0.0, 0.0 };
    double o[3], d[3], v0[3], e1[3], e2[3], tv[3], atv[3];
    for (int i = 0; i < 3; ++i) {
        o[i] = of[i]; d[i] = df[i]; v0[i] = v0f[i]; e1[i] = e1f[i]; e2[i] = e2f[i];
        tv[i] = o[i] - v0[i];                              /* one rounding when the exponents are > 29 apart */
        atv[i] = fabs(o[i]) + fabs(v0[i]);
    }
    const double tmin = tminf, tmax = tmaxf;
    /* pvec = d x e2, qvec = tvec x e1 and their permanents */
    double pv[3], ppv[3], qv[3], pqv[3];
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        pv[i] = d[j] * e2[k] - d[k] * e2[j];
        ppv[i] = fabs(d[j] * e2[k]) + fabs(d[k] * e2[j]);
        qv[i] = tv[j] * e1[k] - tv[k] * e1[j];
        pqv[i] = atv[j] * fabs(e1[k]) + atv[k] * fabs(e1[j]);
    }
    const double det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2];
    const double pdet = (fabs(e1[0]) * ppv[0] + fabs(e1[1]) * ppv[1]) + fabs(e1[2]) * ppv[2];
    const double un = (tv[0] * pv[0] + tv[1] * pv[1]) + tv[2] * pv[2];
    const double pun = (atv[0] * ppv[0] + atv[1] * ppv[1]) + atv[2] * ppv[2];
    const double vn = (d[0] * qv[0] + d[1] * qv[1]) + d[2] * qv[2];
    const double pvn = (fabs(d[0]) * pqv[0] + fabs(d[1]) * pqv[1]) + fabs(d[2]) * pqv[2];
    const double tn = (e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2];
    const double ptn = (fabs(e2[0]) * pqv[0] + fabs(e2[1]) * pqv[1]) + fabs(e2[2]) * pqv[2];
    r.det = det;
    if (det != 0.0) { r.u = un / det; r.v = vn / det; r.t = tn / det; }

    const int s = exact_sign(det, pdet);
    if (s == 0) return r;                                  /* det == 0 exactly: a miss */
    /* each test as "sign(det) * value >= 0" (or > 0); with sign(det) undecided a test is still certain when the value is certainly 0
     * in a closed comparison -- not worth the case: undecided */
    if (s == 2) { r.decision = -1; return r; }
    const double sg = (double)s;
    int undecided = 0;
#define VHR_EXACT_TEST(value, perm, closed)                                            \
    do {                                                                                \
        const int sv_ = exact_sign(sg * (value), (perm));                               \
        if (sv_ == 2) undecided = 1;                                                    \
        else if (sv_ < 0 || (sv_ == 0 && !(closed))) return r;      /* certainly false: a miss whatever the others say */ \
    } while (0)
    VHR_EXACT_TEST(un, pun, 1);                                                   /* u >= 0 */
    VHR_EXACT_TEST(det - un, pdet + pun, 1);                                      /* u <= 1 */
    VHR_EXACT_TEST(vn, pvn, 1);                                                   /* v >= 0 */
    VHR_EXACT_TEST((det - un) - vn, (pdet + pun) + pvn, 1);                       /* u + v <= 1 */
    VHR_EXACT_TEST(tn - tmin * det, ptn + fabs(tmin) * pdet, 0);                  /* t > tmin */
    VHR_EXACT_TEST(tmax * det - tn, fabs(tmax) * pdet + ptn, 0);                  /* t < tmax */
#undef VHR_EXACT_TEST
    r.decision = undecided ? -1 : 1;
    return r;
}

#endif
