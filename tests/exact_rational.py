"""Exact rational ray / triangle decision (python Fractions on the fp32 bit patterns): the last word behind oracle/vhr_exact.h's
binary64 filter.  TEST INFRASTRUCTURE.  Same definition as the header: closed triangle v0, v0 + e1, v0 + e2; hit iff det != 0,
0 <= u <= 1, 0 <= v, u + v <= 1, tmin < t < tmax for Moeller-Trumbore's quantities evaluated without rounding."""
from fractions import Fraction

import numpy as np


def _fr(x):
    return [Fraction(float(np.float32(v))) for v in np.ravel(x)]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def ray_triangle(o, d, v0, e1, e2, tmin, tmax):
    """-> (hit: bool, (det, u, v, t) as Fractions or None when det == 0)."""
    o, d, v0, e1, e2 = _fr(o), _fr(d), _fr(v0), _fr(e1), _fr(e2)
    tmin, tmax = Fraction(float(np.float32(tmin))), Fraction(float(np.float32(tmax)))
    pvec = _cross(d, e2)
    det = _dot(e1, pvec)
    if det == 0:
        return False, None
    tvec = [o[i] - v0[i] for i in range(3)]
    u = _dot(tvec, pvec) / det
    qvec = _cross(tvec, e1)
    v = _dot(d, qvec) / det
    t = _dot(e2, qvec) / det
    hit = u >= 0 and u <= 1 and v >= 0 and u + v <= 1 and t > tmin and t < tmax
    return bool(hit), (det, u, v, t)
