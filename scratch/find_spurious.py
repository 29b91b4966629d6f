"""The r5 debug ray (raytraced path, rotated sponza_proc(0.3), frame 1, pixel (165, 118)): which triangle does Moeller-Trumbore accept although the point is beside it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import scenes
f = np.float32
sc = scenes.rotated(scenes.sponza_proc(0.3), rot_y=0.6, rot_x=0.25)
o = np.array([float.fromhex("0x1.4216bp+4"), float.fromhex("0x1.42c918p+3"), float.fromhex("-0x1.7fbd2p+2")], f)
d = np.array([float.fromhex("-0x1.bcaf8cp-5"), float.fromhex("0x1.fda0e4p-1"), float.fromhex("-0x1.44ff66p-4")], f)
V, I, P = sc.vertices, sc.indices, sc.primitives
out = []
for p in P:
    m = np.asarray(p["transform"], f).reshape(4, 4).T if np.asarray(p["transform"]).size == 16 else None
    idx = I[p["index_offset"]:p["index_offset"] + p["index_count"]].reshape(-1, 3) + p["vertex_offset"]
    pos = V["pos"][idx].astype(f)                     # (T, 3 corners, 3)
    M = np.asarray(p["transform"], f)                 # column-major 16
    w = np.empty_like(pos)
    for a in range(3):
        w[..., a] = ((M[a] * pos[..., 0] + M[4 + a] * pos[..., 1]) + M[8 + a] * pos[..., 2]) + M[12 + a]
    out.append(w)
W = np.concatenate(out)                               # world-space corners
v0, e1, e2 = W[:, 0], W[:, 1] - W[:, 0], W[:, 2] - W[:, 0]
def cross(a, b): return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2], a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(f)
def dot(a, b): return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]).astype(f)
with np.errstate(all="ignore"):
    pvec = cross(np.broadcast_to(d, e2.shape), e2); det = dot(e1, pvec); inv = (f(1) / det).astype(f)
    tvec = (o - v0).astype(f); uu = (dot(tvec, pvec) * inv).astype(f); qvec = cross(tvec, e1)
    vv = (dot(np.broadcast_to(d, qvec.shape), qvec) * inv).astype(f); tt = (dot(e2, qvec) * inv).astype(f)
    ok = (det != 0) & (uu >= 0) & ~(uu > 1) & (vv >= 0) & ~((uu + vv).astype(f) > 1) & (tt > f(0.1)) & (tt < f(10000))
for k in np.nonzero(ok)[0]:
    p = (o + d * tt[k]).astype(f); q = ((v0[k] + e1[k] * uu[k]).astype(f) + e2[k] * vv[k]).astype(f)
    print("accepted triangle", k, "t", float(tt[k]).hex(), "u", float(uu[k]).hex(), "v", float(vv[k]).hex(), "det", float(det[k]).hex(), "|p - q|", np.abs(p - q).tolist())
    print("  v0", [float(x).hex() for x in v0[k]], "e1", [float(x).hex() for x in e1[k]], "e2", [float(x).hex() for x in e2[k]])
