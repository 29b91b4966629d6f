"""Procedural stand-ins for the scenes BASELINE.json names (Sponza, Amazon Bistro).

Neither asset exists in this image or on the GPU box (the reference git-ignores data/models), so the
benchmark and parity inputs are generated: `sponza_proc` (~258 k triangles in 103 primitives, SURVEY.md
section 8(d) config 2/3) and `bistro_proc` (~2.8 M triangles in 3000 primitives with 64 textures, config
4/5; > 2048 primitives so the fp16 object-id aliasing of gbuf.frag:43 is exercised).  The output is
exactly what SceneLoader hands to ResourceManager::UpdateGeometry
(/root/reference/src/scene/scene_loader.cpp:104-231,331): one flat Vertex array, one flat uint32 index
array and one Primitive per glTF primitive carrying its world transform, material and offsets.

This module is input generation (synthetic data), not part of the hot path.
"""
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .camera import directional_light


@dataclass
class Scene:
    name: str
    vertices: np.ndarray            # abi.vertex_dtype
    indices: np.ndarray             # uint32
    primitives: np.ndarray          # abi.primitive_dtype
    textures: list = field(default_factory=list)   # dicts: rgba8 (H,W,4) uint8, format, mag, min, address_u, address_v
    camera: dict = field(default_factory=dict)
    light: np.ndarray = None

    @property
    def triangle_count(self):
        return int((self.primitives["index_count"] // 3).sum())


# --------------------------------------------------------------------------------------------
# mesh helpers: every helper returns (pos[N,3], normal[N,3], uv[N,2], tri[M,3])
# --------------------------------------------------------------------------------------------
def _grid_indices(nu, nv, wrap_u=False):
    cols = nu if wrap_u else nu + 1
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    i0 = i
    i1 = (i + 1) % cols if wrap_u else i + 1
    a = j * cols + i0
    b = j * cols + i1
    c = (j + 1) * cols + i0
    d = (j + 1) * cols + i1
    tris = np.stack([np.stack([a, b, d], -1), np.stack([a, d, c], -1)], -2)
    return tris.reshape(-1, 3).astype(np.uint32)


def param_surface(fn, nu, nv, wrap_u=False, flip=False):
    """Tessellate fn(u, v) -> (pos, normal) over [0,1]^2 into nu x nv quads."""
    us = np.arange(nu if wrap_u else nu + 1) / nu
    vs = np.arange(nv + 1) / nv
    u, v = np.meshgrid(us, vs, indexing="xy")
    pos, nrm = fn(u.reshape(-1), v.reshape(-1))
    uv = np.stack([u.reshape(-1), v.reshape(-1)], -1)
    tri = _grid_indices(nu, nv, wrap_u)
    if flip:                      # reverse the winding only; normals stay as fn() gave them
        tri = tri[:, ::-1].copy()
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    return pos, nrm, uv, tri


def plane(origin, eu, ev, nu, nv):
    origin, eu, ev = (np.asarray(a, float) for a in (origin, eu, ev))
    n = np.cross(eu, ev)          # the side the normal (and the winding) faces: floors up, walls into the room

    def fn(u, v):
        return origin + u[:, None] * eu + v[:, None] * ev, np.broadcast_to(n, (u.size, 3)).copy()
    return param_surface(fn, nu, nv)


def cylinder(radius, height, sides, segs, caps=True):
    def fn(u, v):
        a = 2 * np.pi * u
        p = np.stack([radius * np.cos(a), height * v, radius * np.sin(a)], -1)
        n = np.stack([np.cos(a), 0 * a, np.sin(a)], -1)
        return p, n
    pos, nrm, uv, tri = param_surface(fn, sides, segs, wrap_u=True, flip=True)
    if caps:
        parts = [(pos, nrm, uv, tri)]
        for y, ny in ((0.0, -1.0), (height, 1.0)):
            a = 2 * np.pi * np.arange(sides) / sides
            ring = np.stack([radius * np.cos(a), np.full(sides, y), radius * np.sin(a)], -1)
            cp = np.concatenate([[[0.0, y, 0.0]], ring])
            cn = np.tile([[0.0, ny, 0.0]], (sides + 1, 1))
            cuv = np.concatenate([[[0.5, 0.5]], 0.5 + 0.5 * np.stack([np.cos(a), np.sin(a)], -1)])
            k = np.arange(sides)
            ct = np.stack([np.zeros(sides, int), 1 + k, 1 + (k + 1) % sides], -1).astype(np.uint32)
            if ny > 0:
                ct = ct[:, ::-1].copy()
            parts.append((cp, cn, cuv, ct))
        return merge(parts)
    return pos, nrm, uv, tri


def half_torus(major, minor, nu, nv):
    """Arch: tube of radius `minor` swept over the upper half circle of radius `major` in the XY plane."""
    def fn(u, v):
        t = np.pi * v
        a = 2 * np.pi * u
        cx, cy = np.cos(t), np.sin(t)
        r = major + minor * np.cos(a)
        p = np.stack([r * cx, r * cy, minor * np.sin(a)], -1)
        n = np.stack([np.cos(a) * cx, np.cos(a) * cy, np.sin(a)], -1)
        return p, n
    return param_surface(fn, nu, nv, wrap_u=True, flip=True)


def sphere(radius, nu, nv):
    def fn(u, v):
        a = 2 * np.pi * u
        t = np.pi * (v * 0.998 + 0.001)
        n = np.stack([np.sin(t) * np.cos(a), np.cos(t), np.sin(t) * np.sin(a)], -1)
        return radius * n, n
    return param_surface(fn, nu, nv, wrap_u=True)


def curtain(width, height, nu, nv, waves, depth):
    def fn(u, v):
        ph = 2 * np.pi * waves * u
        amp = depth * (0.3 + 0.7 * v)
        p = np.stack([width * (u - 0.5), -height * v, amp * np.sin(ph)], -1)
        dz_du = amp * 2 * np.pi * waves * np.cos(ph)
        dz_dv = depth * 0.7 * np.sin(ph)
        tu = np.stack([np.full_like(u, width), 0 * u, dz_du], -1)
        tv = np.stack([0 * u, np.full_like(u, -height), dz_dv], -1)
        return p, np.cross(tv, tu)
    return param_surface(fn, nu, nv, flip=True)


def box(size, n):
    sx, sy, sz = (s * 0.5 for s in size)
    faces = [
        plane([-sx, -sy, sz], [2 * sx, 0, 0], [0, 2 * sy, 0], n, n),
        plane([sx, -sy, -sz], [-2 * sx, 0, 0], [0, 2 * sy, 0], n, n),
        plane([sx, -sy, sz], [0, 0, -2 * sz], [0, 2 * sy, 0], n, n),
        plane([-sx, -sy, -sz], [0, 0, 2 * sz], [0, 2 * sy, 0], n, n),
        plane([-sx, sy, sz], [2 * sx, 0, 0], [0, 0, -2 * sz], n, n),
        plane([-sx, -sy, -sz], [2 * sx, 0, 0], [0, 0, 2 * sz], n, n),
    ]
    fixed = []
    for (p, nrm, uv, tri) in faces:
        c = p.mean(0)
        if np.dot(nrm[0], c) < 0:     # make normals point outward
            nrm = -nrm
            tri = tri[:, ::-1].copy()
        fixed.append((p, nrm, uv, tri))
    return merge(fixed)


def merge(parts):
    pos, nrm, uv, tri = [], [], [], []
    base = 0
    for p, n, t, i in parts:
        pos.append(p)
        nrm.append(n)
        uv.append(t)
        tri.append(i.astype(np.int64) + base)
        base += len(p)
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(uv), np.concatenate(tri).astype(np.uint32)


def trs(translate=(0, 0, 0), rot_y=0.0, rot_x=0.0, scale=(1, 1, 1)):
    t = np.eye(4)
    t[:3, 3] = translate
    cy, sy = np.cos(rot_y), np.sin(rot_y)
    cx, sx = np.cos(rot_x), np.sin(rot_x)
    ry = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1.0]])
    rx = np.array([[1, 0, 0, 0], [0, cx, -sx, 0], [0, sx, cx, 0], [0, 0, 0, 1.0]])
    s = np.diag([scale[0], scale[1], scale[2], 1.0])
    return t @ ry @ rx @ s


class _Builder:
    def __init__(self):
        self.v, self.i, self.p = [], [], []
        self.nv = 0
        self.ni = 0

    def add(self, mesh, transform=None, base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=0.8,
            base_color_texture=-1, metallic_roughness_texture=-1, uv_scale=1.0):
        pos, nrm, uv, tri = mesh
        v = np.zeros(len(pos), abi.vertex_dtype)
        v["pos"] = pos
        v["normal"] = nrm
        v["tangent"] = [1, 0, 0, 1]
        v["uv0"] = uv * uv_scale
        v["uv1"] = uv
        idx = tri.reshape(-1).astype(np.uint32)
        pr = np.zeros((), abi.primitive_dtype)
        pr["transform"] = abi.mat_to_glm(np.eye(4) if transform is None else transform)
        m = pr["material"]
        m["base_color"] = base_color
        m["base_color_texture"] = base_color_texture
        m["metallic_roughness_texture"] = metallic_roughness_texture
        m["normal_map"] = -1
        m["metallic_factor"] = metallic
        m["roughness_factor"] = roughness
        m["alpha_mask"] = 0
        m["alpha_cutoff"] = 0.5
        pr["vertex_offset"] = self.nv
        pr["index_offset"] = self.ni
        pr["index_count"] = idx.size
        self.v.append(v)
        self.i.append(idx)
        self.p.append(pr)
        self.nv += len(v)
        self.ni += idx.size

    def finish(self, name, camera, light, textures=()):
        return Scene(name, np.concatenate(self.v), np.concatenate(self.i), np.stack(self.p), list(textures), camera, light)


def _palette(k):
    """Deterministic per-primitive colour (golden-ratio hue walk)."""
    h = (k * 0.61803398875) % 1.0
    s, v = 0.45, 0.85
    i = int(h * 6)
    f = h * 6 - i
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    r, g, b = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)][i % 6]
    return (r, g, b, 1.0)


# --------------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------------
REFERENCE_LIGHT_DIRECTION = (0.0, -0.97, 0.35)   # the value left in scene_loader.cpp:87


def tiny_scene():
    """~300 triangles: floor, back wall, a box, a sphere and a tilted quad -- small enough for the brute-force oracle."""
    b = _Builder()
    b.add(plane([-4, 0, 4], [8, 0, 0], [0, 0, -8], 4, 4), base_color=(0.7, 0.7, 0.7, 1))
    b.add(plane([-4, 0, -4], [8, 0, 0], [0, 5, 0], 3, 3), base_color=(0.6, 0.7, 0.8, 1))
    b.add(box((1.2, 1.6, 1.0), 2), trs((-1.4, 0.8, -0.5), rot_y=0.5), base_color=(0.8, 0.3, 0.2, 1), metallic=0.2, roughness=0.5)
    b.add(sphere(0.7, 12, 8), trs((1.3, 0.9, 0.2), scale=(1.0, 1.3, 1.0)), base_color=(0.2, 0.6, 0.3, 1), metallic=0.8, roughness=0.3)
    b.add(plane([-0.8, 0, 0.6], [1.6, 0, 0], [0, 0, -1.2], 2, 2), trs((0.2, 2.4, -1.0), rot_x=0.4), base_color=(0.9, 0.8, 0.2, 1))
    camera = dict(position=(0.0, 1.6, 5.5), yaw=0.0, pitch=-0.12, yfov=0.9, znear=0.1, dolly=(0.0, 0.0, -0.05))
    return b.finish("tiny", camera, directional_light(REFERENCE_LIGHT_DIRECTION))


def sponza_proc(detail=1.0):
    """Procedural atrium: floor, walls, gallery slabs, two storeys of 24-gon columns, arches, hanging curtains, props.

    detail=1.0 gives 257 536 triangles in 103 primitives (object ids 0..102)."""
    d = lambda n: max(2, int(round(n * detail)))   # noqa: E731
    b = _Builder()
    L, Wd, Ht = 20.0, 8.0, 14.0
    k = 0
    b.add(plane([-L, 0, Wd], [2 * L, 0, 0], [0, 0, -2 * Wd], d(64), d(32)), base_color=(0.62, 0.58, 0.52, 1), roughness=0.9); k += 1
    b.add(plane([-L, 0, -Wd], [2 * L, 0, 0], [0, Ht, 0], d(64), d(24)), base_color=(0.75, 0.7, 0.62, 1)); k += 1
    b.add(plane([L, 0, Wd], [-2 * L, 0, 0], [0, Ht, 0], d(64), d(24)), base_color=(0.75, 0.7, 0.62, 1)); k += 1
    b.add(plane([-L, 0, Wd], [0, 0, -2 * Wd], [0, Ht, 0], d(24), d(24)), base_color=(0.7, 0.66, 0.6, 1)); k += 1
    b.add(plane([L, 0, -Wd], [0, 0, 2 * Wd], [0, Ht, 0], d(24), d(24)), base_color=(0.7, 0.66, 0.6, 1)); k += 1
    gz = 4.5
    for sgn in (-1.0, 1.0):        # gallery slabs: top and underside
        z0, z1 = (sgn * Wd, sgn * gz)
        b.add(plane([-L, 6.0, max(z0, z1)], [2 * L, 0, 0], [0, 0, -(Wd - gz)], d(64), d(6)), base_color=(0.66, 0.62, 0.55, 1)); k += 1
        b.add(plane([-L, 5.7, min(z0, z1)], [2 * L, 0, 0], [0, 0, (Wd - gz)], d(64), d(6)), base_color=(0.6, 0.56, 0.5, 1)); k += 1
    xs = np.linspace(-17.5, 17.5, 12)
    col_lo = cylinder(0.35, 5.2, d(24), d(40))
    col_hi = cylinder(0.28, 5.0, d(24), d(40))
    for sgn in (-1.0, 1.0):
        for x in xs:
            b.add(col_lo, trs((x, 0.0, sgn * gz)), base_color=_palette(k), roughness=0.7); k += 1
        for x in xs:
            b.add(col_hi, trs((x, 6.0, sgn * gz)), base_color=_palette(k), roughness=0.7); k += 1
    spacing = xs[1] - xs[0]
    arch = half_torus(spacing * 0.5, 0.3, d(32), d(48))
    for sgn in (-1.0, 1.0):
        for a, c in zip(xs[:-1], xs[1:]):
            b.add(arch, trs(((a + c) * 0.5, 5.2, sgn * gz), scale=(1.0, 0.55, 1.0)), base_color=_palette(k), roughness=0.6); k += 1
    cur = curtain(2.6, 4.2, d(48), d(48), 3.0, 0.18)
    for j in range(12):
        x = -16.0 + j * (32.0 / 11.0)
        z = (-1.0 if j % 2 else 1.0) * (1.2 + 0.35 * (j % 3))
        b.add(cur, trs((x, 11.5 - 0.4 * (j % 4), z), rot_y=0.35 * ((j % 5) - 2)), base_color=_palette(k), roughness=0.95); k += 1
    urn = sphere(0.55, d(32), d(32))
    for j in range(12):
        x = -15.0 + j * (30.0 / 11.0)
        z = (1.0 if j % 2 else -1.0) * 2.6
        b.add(urn, trs((x, 0.88, z), scale=(1.0, 1.6, 1.0)), base_color=_palette(k), metallic=0.6, roughness=0.35); k += 1
    camera = dict(position=(-17.0, 2.0, 0.4), yaw=-np.pi / 2, pitch=0.06, yfov=0.9, znear=0.1, aspect=16.0 / 9.0,
                  dolly=(0.05, 0.0, 0.0))
    return b.finish("sponza_proc", camera, directional_light(REFERENCE_LIGHT_DIRECTION))


def sponza_hard(detail=1.0):
    """sponza_proc's atrium with the triangle budget spent the way a modelled building spends it (the uniform tessellation of
    sponza_proc is kind to a BVH): floor, walls and gallery slabs are TWO triangles each (boxes 40 m long), the 48 columns are 24-gons of two
    segments (slivers 9 cm x 2.6 m), and what that frees goes into dense drapery -- six curtains in the nave, six banners hung flush against
    the long walls (inside the walls' huge boxes), two layers of overlapping folds each -- and finer urns.  ~254 k triangles in 103
    primitives, same camera, light and dolly as sponza_proc.  An extra beside the headline, not a BASELINE configuration."""
    d = lambda n: max(2, int(round(n * detail)))   # noqa: E731
    b = _Builder()
    L, Wd, Ht = 20.0, 8.0, 14.0
    k = 0
    b.add(plane([-L, 0, Wd], [2 * L, 0, 0], [0, 0, -2 * Wd], 1, 1), base_color=(0.62, 0.58, 0.52, 1), roughness=0.9); k += 1
    b.add(plane([-L, 0, -Wd], [2 * L, 0, 0], [0, Ht, 0], 1, 1), base_color=(0.75, 0.7, 0.62, 1)); k += 1
    b.add(plane([L, 0, Wd], [-2 * L, 0, 0], [0, Ht, 0], 1, 1), base_color=(0.75, 0.7, 0.62, 1)); k += 1
    b.add(plane([-L, 0, Wd], [0, 0, -2 * Wd], [0, Ht, 0], 1, 1), base_color=(0.7, 0.66, 0.6, 1)); k += 1
    b.add(plane([L, 0, -Wd], [0, 0, 2 * Wd], [0, Ht, 0], 1, 1), base_color=(0.7, 0.66, 0.6, 1)); k += 1
    gz = 4.5
    for sgn in (-1.0, 1.0):
        z0, z1 = (sgn * Wd, sgn * gz)
        b.add(plane([-L, 6.0, max(z0, z1)], [2 * L, 0, 0], [0, 0, -(Wd - gz)], 1, 1), base_color=(0.66, 0.62, 0.55, 1)); k += 1
        b.add(plane([-L, 5.7, min(z0, z1)], [2 * L, 0, 0], [0, 0, (Wd - gz)], 1, 1), base_color=(0.6, 0.56, 0.5, 1)); k += 1
    xs = np.linspace(-17.5, 17.5, 12)
    col_lo = cylinder(0.35, 5.2, 24, 2)
    col_hi = cylinder(0.28, 5.0, 24, 2)
    for sgn in (-1.0, 1.0):
        for x in xs:
            b.add(col_lo, trs((x, 0.0, sgn * gz)), base_color=_palette(k), roughness=0.7); k += 1
        for x in xs:
            b.add(col_hi, trs((x, 6.0, sgn * gz)), base_color=_palette(k), roughness=0.7); k += 1
    spacing = xs[1] - xs[0]
    arch = half_torus(spacing * 0.5, 0.3, d(32), d(48))
    for sgn in (-1.0, 1.0):
        for a, c in zip(xs[:-1], xs[1:]):
            b.add(arch, trs(((a + c) * 0.5, 5.2, sgn * gz), scale=(1.0, 0.55, 1.0)), base_color=_palette(k), roughness=0.6); k += 1
    # drapery: each primitive is two layers of folds 4 cm apart (overlapping boxes all the way down the tree)
    layer_a = curtain(2.6, 4.2, d(51), d(51), 3.0, 0.18)
    layer_b = curtain(2.6, 4.2, d(51), d(51), 3.5, 0.15)
    pb = layer_b[0].copy(); pb[:, 2] += 0.04
    drape = merge([layer_a, (pb, layer_b[1], layer_b[2], layer_b[3])])
    for j in range(6):             # in the nave
        x = -16.0 + j * (32.0 / 5.0)
        z = (-1.0 if j % 2 else 1.0) * (1.2 + 0.35 * (j % 3))
        b.add(drape, trs((x, 11.5 - 0.4 * (j % 4), z), rot_y=0.35 * ((j % 5) - 2)), base_color=_palette(k), roughness=0.95); k += 1
    for j in range(6):             # banners flush against the long walls
        x = -15.0 + j * 6.0
        sgn = -1.0 if j % 2 else 1.0
        b.add(drape, trs((x, 10.5, sgn * (Wd - 0.25)), scale=(1.6, 1.6, 1.0)), base_color=_palette(k), roughness=0.95); k += 1
    urn = sphere(0.55, d(48), d(48))
    for j in range(12):
        x = -15.0 + j * (30.0 / 11.0)
        z = (1.0 if j % 2 else -1.0) * 2.6
        b.add(urn, trs((x, 0.88, z), scale=(1.0, 1.6, 1.0)), base_color=_palette(k), metallic=0.6, roughness=0.35); k += 1
    camera = dict(position=(-17.0, 2.0, 0.4), yaw=-np.pi / 2, pitch=0.06, yfov=0.9, znear=0.1, aspect=16.0 / 9.0,
                  dolly=(0.05, 0.0, 0.0))
    return b.finish("sponza_hard", camera, directional_light(REFERENCE_LIGHT_DIRECTION))


def rotated(scene, rot_y=0.6, rot_x=0.25, name=None):
    """The same scene turned as a whole -- geometry, camera and light -- about y, then x: every image is the unrotated scene's up to rounding, but
    no wall, floor or column is axis-aligned any more, so every large triangle's BOX holds far more than the triangle (what a model that was not
    built along the world axes does to a BVH).  An extra beside the headline (profiles/r5_sponza_hard.txt)."""
    import copy
    R = trs(rot_y=rot_y, rot_x=rot_x)
    out = copy.copy(scene)
    out.name = name or (scene.name + "_rot")
    out.primitives = scene.primitives.copy()
    for i in range(len(out.primitives)):
        m = abi.glm_to_mat(out.primitives["transform"][i]) if hasattr(abi, "glm_to_mat") else np.asarray(out.primitives["transform"][i], np.float64).reshape(4, 4).T
        out.primitives["transform"][i] = abi.mat_to_glm(R @ m)
    out.camera = dict(scene.camera, world=R)
    light = scene.light.copy()
    d = np.asarray(light["direction"][:3], np.float64)
    out.light = directional_light(tuple((R[:3, :3] @ d).tolist()), color=tuple(light["color"][:3].tolist()), intensity=float(light["intensity"][0]))
    return out


def _procedural_texture(k, size=512):
    """Deterministic RGBA8 texture (checker + stripes + hash noise), sRGB base-colour content."""
    y, x = np.mgrid[0:size, 0:size].astype(np.uint32)
    with np.errstate(over="ignore"):
        h = (x * np.uint32(73856093)) ^ (y * np.uint32(19349663)) ^ np.uint32((k * 83492791 + 12345) & 0xffffffff)
    h ^= h >> np.uint32(13)
    with np.errstate(over="ignore"):
        h = h * np.uint32(0x5bd1e995)
    h ^= h >> np.uint32(15)
    noise = (h & np.uint32(63)).astype(np.int32)
    cell = 8 << (k % 4)
    checker = (((x // cell) + (y // cell)) & 1).astype(np.int32)
    stripes = ((x + 2 * y) // (cell // 2 + 1) & 1).astype(np.int32)
    base = np.array(_palette(k)[:3]) * 255.0
    img = np.zeros((size, size, 4), np.uint8)
    for c in range(3):
        ch = base[c] * (0.55 + 0.35 * checker + 0.10 * stripes) + noise - 32
        img[..., c] = np.clip(ch, 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


def bistro_proc(detail=1.0, n_primitives=3000, n_textures=64, texture_size=512):
    """Street canyon: ground, two facades and thousands of ledges, awnings, posts and props.

    detail=1.0 gives ~2.8 M triangles in 3000 primitives; 64 procedural RGBA8 textures (sRGB base colour,
    REPEAT/LINEAR samplers) shared round-robin."""
    d = lambda n: max(2, int(round(n * np.sqrt(detail))))   # noqa: E731
    b = _Builder()
    textures = [dict(rgba8=_procedural_texture(t, texture_size), format=abi.FORMAT_R8G8B8A8_SRGB,
                     mag=abi.FILTER_LINEAR, min=abi.FILTER_LINEAR, address_u=abi.ADDRESS_REPEAT,
                     address_v=abi.ADDRESS_REPEAT) for t in range(n_textures)]
    L, Wd, Ht = 60.0, 7.0, 18.0
    k = 0
    b.add(plane([-L, 0, Wd], [2 * L, 0, 0], [0, 0, -2 * Wd], d(192), d(24)), base_color_texture=k % n_textures, uv_scale=24.0, roughness=0.85); k += 1
    b.add(plane([-L, 0, -Wd], [2 * L, 0, 0], [0, Ht, 0], d(192), d(32)), base_color_texture=k % n_textures, uv_scale=12.0); k += 1
    b.add(plane([L, 0, Wd], [-2 * L, 0, 0], [0, Ht, 0], d(192), d(32)), base_color_texture=k % n_textures, uv_scale=12.0); k += 1
    templates = [
        box((1.6, 0.25, 0.5), d(9)),              # ledge
        box((1.1, 1.7, 0.12), d(9)),              # shutter / window frame
        curtain(2.4, 1.4, d(22), d(22), 2.0, 0.1),  # awning
        cylinder(0.09, 3.2, d(16), d(28)),        # post
        sphere(0.3, d(22), d(22)),                # lamp / planter
        half_torus(0.8, 0.1, d(14), d(34)),       # arch
    ]
    rng = np.uint32(0x9e3779b9)

    def rnd():
        nonlocal rng
        rng ^= np.uint32(rng << np.uint32(13))
        rng ^= np.uint32(rng >> np.uint32(17))
        rng ^= np.uint32(rng << np.uint32(5))
        return float(rng) / 4294967296.0

    with np.errstate(over="ignore"):
        while k < n_primitives:
            t = k % len(templates)
            side = -1.0 if (k // len(templates)) % 2 else 1.0
            x = -L + 2 * L * rnd()
            if t in (0, 1):
                y, z = 1.5 + 15.0 * rnd(), side * (Wd - 0.3)
                tr = trs((x, y, z), rot_y=0.0 if side > 0 else np.pi)
            elif t == 2:
                y, z = 3.0 + 1.0 * rnd(), side * (Wd - 0.9)
                tr = trs((x, y, z), rot_x=side * 1.1)
            elif t == 3:
                y, z = 0.0, side * (Wd - 1.8 - 0.5 * rnd())
                tr = trs((x, y, z))
            elif t == 4:
                y, z = 0.3 + 3.0 * rnd(), side * (Wd - 2.2 * rnd() - 0.5)
                tr = trs((x, y, z), scale=(1.0, 1.0 + rnd(), 1.0))
            else:
                y, z = 2.6 + 10.0 * rnd(), side * (Wd - 0.25)
                tr = trs((x, y, z), rot_y=0.0)
            b.add(templates[t], tr, base_color=_palette(k), base_color_texture=(k % n_textures) if k % 3 else -1,
                  metallic=0.1 + 0.6 * (k % 5 == 0), roughness=0.35 + 0.5 * rnd(), uv_scale=2.0)
            k += 1
    camera = dict(position=(-50.0, 1.7, 0.6), yaw=-np.pi / 2, pitch=0.04, yfov=0.9, znear=0.1, aspect=16.0 / 9.0,
                  dolly=(0.05, 0.0, 0.0))
    return b.finish("bistro_proc", camera, directional_light(REFERENCE_LIGHT_DIRECTION), textures)


def sponza_proc_rot():
    return rotated(sponza_proc())


def sponza_hard_rot():
    return rotated(sponza_hard())


def bistro_proc_rot():
    return rotated(bistro_proc())


def tiny_rot():
    return rotated(tiny_scene())
