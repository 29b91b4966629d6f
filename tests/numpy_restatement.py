"""Independent numpy restatement of the SVGF kernels and the RNG (TEST INFRASTRUCTURE), written from the GLSL
(/root/reference/data/shaders/hybrid_render_path/svgf.comp, svgf_atrous_filter.comp, ../common.glsl) without
looking at oracle/vhr_oracle.c's structure: a second pin for the oracle (SURVEY.md section 8c 'Fixtures policy').
Vectorised over the image, so the operation order per pixel is the shader's but taps are processed image-wide."""
import numpy as np

F = np.float32


def h2f(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


def f2h(x):
    with np.errstate(over="ignore", invalid="ignore"):
        return np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def seed_thread(seed):
    m = 0xffffffff
    seed &= m
    seed = ((seed ^ 61) ^ (seed >> 16)) & m
    seed = (seed * 9) & m
    seed = (seed ^ (seed >> 4)) & m
    seed = (seed * 0x27d4eb2d) & m
    seed = (seed ^ (seed >> 15)) & m
    return seed


def xorshift(state):
    m = 0xffffffff
    state ^= (state << 13) & m
    state ^= state >> 17
    state ^= (state << 5) & m
    return state & m


def random01(state):
    state = xorshift(state)
    bits = np.array([0x3f800000 | (state >> 9)], np.uint32)
    return state, np.float32(bits.view(np.float32)[0] - F(1.0))


def _shift(img, dx, dy, fill=0):
    """out[y, x] = img[y + dy, x + dx] where in range, else `fill`; also returns the in-range mask."""
    H, W = img.shape[:2]
    out = np.full_like(img, fill)
    ys = slice(max(0, -dy), min(H, H - dy))
    xs = slice(max(0, -dx), min(W, W - dx))
    yd = slice(max(0, -dy) + dy, min(H, H - dy) + dy)
    xd = slice(max(0, -dx) + dx, min(W, W - dx) + dx)
    mask = np.zeros((H, W), bool)
    if ys.start < ys.stop and xs.start < xs.stop:
        out[ys, xs] = img[yd, xd]
        mask[ys, xs] = True
    return out, mask


def atrous(normals_bits, in_bits, step):
    n = h2f(normals_bits)
    p = h2f(in_bits)
    H, W = p.shape[:2]
    normal_p, id_p = n[..., :3], n[..., 3].astype(np.int32)
    var = np.zeros((H, W, 2), F)
    gw = [F(1 / 16), F(1 / 8), F(1 / 16), F(1 / 8), F(1 / 4), F(1 / 8), F(1 / 16), F(1 / 8), F(1 / 16)]
    for y in (-1, 0, 1):
        for x in (-1, 0, 1):
            q, m = _shift(p, x, y)
            w = gw[3 * (y + 1) + (x + 1)]
            var = var + np.where(m[..., None], w * q[..., 2:4], F(0)).astype(F)
    k1 = [F(1 / 16), F(1 / 4), F(3 / 8), F(1 / 4), F(1 / 16)]
    sum_w = np.ones((H, W, 2), F)
    acc = p.copy()
    denom = (F(4.0) * np.sqrt(var) + F(1e-6)).astype(F)
    for y in range(-2, 3):
        for x in range(-2, 3):
            if x == 0 and y == 0:
                continue
            q, m = _shift(p, x * step, y * step)
            nq, _ = _shift(n, x * step, y * step)
            kernel = F(k1[y + 2] * k1[x + 2])
            d = ((normal_p[..., 0] * nq[..., 0] + normal_p[..., 1] * nq[..., 1]) + normal_p[..., 2] * nq[..., 2]).astype(F)
            with np.errstate(over="ignore", under="ignore"):
                pw = np.where(d > 0, d, F(0)).astype(F)
                for _ in range(7):
                    pw = (pw * pw).astype(F)
            wid = (id_p == nq[..., 3].astype(np.int32)).astype(F)
            w = ((kernel * pw) * wid).astype(F)
            e = (np.abs(p[..., 0:2] - q[..., 0:2]) / denom).astype(F)
            lw = np.exp(-e).astype(F)
            wxy = (w[..., None] * lw).astype(F)
            wxy = np.where(m[..., None], wxy, F(0)).astype(F)
            sum_w = (sum_w + wxy).astype(F)
            acc[..., 0:2] = (acc[..., 0:2] + wxy * q[..., 0:2]).astype(F)
            acc[..., 2:4] = (acc[..., 2:4] + (wxy * wxy) * q[..., 2:4]).astype(F)
    out = np.empty_like(acc)
    out[..., 0:2] = acc[..., 0:2] / sum_w
    out[..., 2:4] = acc[..., 2:4] / (sum_w * sum_w)
    return f2h(out)


def temporal(W, H, normals_bits, motion_bits, rt_bits, prev_normals_bits, history_bits, moments_bits):
    n = h2f(normals_bits)
    mv = h2f(motion_bits)
    rt = h2f(rt_bits)
    pn = h2f(prev_normals_bits)
    hist = h2f(history_bits)
    mom = h2f(moments_bits)
    ys, xs = np.mgrid[0:H, 0:W]
    cur_n, cur_id = n[..., :3], n[..., 3].astype(np.int32)
    pcx = ((xs.astype(F) - mv[..., 0] * F(W)) + F(0.5)).astype(F)
    pcy = ((ys.astype(F) - mv[..., 1] * F(H)) + F(0.5)).astype(F)
    fx = (pcx - np.floor(pcx)).astype(F)
    fy = (pcy - np.floor(pcy)).astype(F)
    with np.errstate(invalid="ignore"):
        ax = np.where(np.isnan(pcx), 0, np.trunc(pcx)).astype(np.int64)
        ay = np.where(np.isnan(pcy), 0, np.trunc(pcy)).astype(np.int64)
    weights = [(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]

    def gather(sx, sy):
        inb = (sx >= 0) & (sy >= 0) & (sx < W) & (sy < H)
        cx, cy = np.clip(sx, 0, W - 1), np.clip(sy, 0, H - 1)
        p = pn[cy, cx]
        ok = inb & (cur_id == p[..., 3].astype(np.int32))
        d = ((cur_n[..., 0] * p[..., 0] + cur_n[..., 1] * p[..., 1]) + cur_n[..., 2] * p[..., 2]).astype(F)
        ok &= ~(d < F(0.70710678118654752440084))
        return ok, hist[cy, cx], mom[cy, cx]

    z = np.zeros((H, W), F)
    ps, pa, s = z.copy(), z.copy(), z.copy()
    m0, m1, a0, a1 = z.copy(), z.copy(), z.copy(), z.copy()
    for i, (ox, oy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        ok, h, m = gather(ax + ox, ay + oy)
        w = weights[i].astype(F)
        ps = np.where(ok, ps + w * h[..., 0], ps).astype(F)
        pa = np.where(ok, pa + w * h[..., 1], pa).astype(F)
        m0 = np.where(ok, m0 + w * m[..., 0], m0).astype(F)
        m1 = np.where(ok, m1 + w * m[..., 1], m1).astype(F)
        a0 = np.where(ok, a0 + w * F(0), a0).astype(F)
        a1 = np.where(ok, a1 + w * F(1), a1).astype(F)
        s = np.where(ok, s + w, s).astype(F)
    with np.errstate(invalid="ignore"):
        valid = s > F(1e-6)
    retry = ~valid
    for oy in (-1, 0, 1):
        for ox in (-1, 0, 1):
            ok, h, m = gather(ax + ox, ay + oy)
            ok &= retry
            ps = np.where(ok, ps + h[..., 0], ps).astype(F)
            pa = np.where(ok, pa + h[..., 1], pa).astype(F)
            m0 = np.where(ok, m0 + m[..., 0], m0).astype(F)
            m1 = np.where(ok, m1 + m[..., 1], m1).astype(F)
            a0 = np.where(ok, a0 + F(0), a0).astype(F)
            a1 = np.where(ok, a1 + F(1), a1).astype(F)
            s = np.where(ok, s + F(1), s).astype(F)
    with np.errstate(invalid="ignore"):
        valid = np.where(retry, s > F(1e-6), valid)
    cs, ca = rt[..., 0], rt[..., 1]
    sm0, sm1, am0, am1 = cs, (cs * cs).astype(F), ca, (ca * ca).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        mix = lambda a, b, t: (a * (F(1) - F(t)) + b * F(t)).astype(F)   # noqa: E731
        v_sm0, v_sm1 = mix(m0 / s, sm0, 0.2), mix(m1 / s, sm1, 0.2)
        v_am0, v_am1 = mix(a0 / s, am0, 0.2), mix(a1 / s, am1, 0.2)
        v_s, v_a = mix(ps / s, cs, 0.2), mix(pa / s, ca, 0.2)
    sm0 = np.where(valid, v_sm0, sm0)
    sm1 = np.where(valid, v_sm1, sm1)
    am0 = np.where(valid, v_am0, am0)
    am1 = np.where(valid, v_am1, am1)
    out_s = np.where(valid, v_s, cs)
    out_a = np.where(valid, v_a, ca)
    sv = np.maximum(F(0), sm1 - sm0 * sm0).astype(F)
    av = np.maximum(F(0), am1 - am0 * am0).astype(F)
    return f2h(np.stack([out_s, out_a, sv, av], -1)), f2h(np.stack([sm0, sm1], -1))


# ---------------------------------------------------------------------------------------------------------
# K2 in float64, written from the GLSL (raygen.rgen:59-65, reflection_hit.rchit:10-72, common.glsl:116-150) for untextured
# materials: a second derivation of the shading formulas, compared with the oracle at fp16 resolution.
# ---------------------------------------------------------------------------------------------------------
def _mat(pfd, name):
    return np.asarray(pfd[name], np.float64).reshape(4, 4).T          # column-major 16 floats -> math matrix


def world_position(pfd, depth, u, v):                                  # glsl_common.h:118-122
    r = _mat(pfd, "camera_viewproj_inverse") @ np.array([u * 2.0 - 1.0, v * 2.0 - 1.0, depth, 1.0])
    return r[:3] / r[3]


def reflection_ray(pfd, depth, normal, x, y, W, H, bias=0.1):           # raygen.rgen:15-16,26-29,60-63
    P = world_position(pfd, float(depth), (x + 0.5) / W, (y + 0.5) / H)
    N = np.asarray(normal, np.float64)
    cam = _mat(pfd, "camera_view_inverse")[:3, 3]
    I = (P - cam) / np.linalg.norm(P - cam)
    return P + bias * N, I - 2.0 * np.dot(N, I) * N


def interpolate_hit(scene, prim_index, tri, u, v):                      # reflection_hit.rchit:11-24
    pr = scene.primitives[prim_index]
    idx = scene.indices[int(pr["index_offset"]) + 3 * tri:int(pr["index_offset"]) + 3 * tri + 3]
    vs = scene.vertices[int(pr["vertex_offset"]) + idx]
    b = np.array([1.0 - u - v, u, v])
    normal = (np.asarray(vs["normal"], np.float64) * b[:, None]).sum(0)           # object space, unnormalised (:23)
    opos = (np.asarray(vs["pos"], np.float64) * b[:, None]).sum(0)
    M = np.asarray(pr["transform"], np.float64).reshape(4, 4).T
    return pr, normal, (M @ np.append(opos, 1.0))[:3]


def brdf_terms(albedo, metallic, roughness, N, V, L):                   # common.glsl:116-150
    H = (L + V) / np.linalg.norm(L + V)
    f0 = 0.04 * (1.0 - metallic) + albedo * metallic                    # mix(vec3(0.04), albedo, metallic)
    hv = max(np.dot(H, V), 0.0)
    F = f0 + (1.0 - f0) * (1.0 - hv) ** 5
    a2 = roughness * roughness
    nh = max(np.dot(N, H), 0.0)
    f = nh * nh * (a2 - 1.0) + 1.0
    D = a2 / (np.pi * f * f)
    k = (roughness + 1.0) ** 2 * 0.125
    nv, nl = max(np.dot(N, V), 0.0), max(np.dot(N, L), 0.0)
    G = (nv / (nv * (1.0 - k) + k)) * (nl / (nl * (1.0 - k) + k))
    specular = D * G * F / max(4.0 * nv * nl, 1e-6)
    diffuse = (1.0 - F) * (1.0 - metallic) * albedo / np.pi
    return diffuse, specular, nl


def reflection_hit(scene, pfd, prim_index, tri, u, v):                  # reflection_hit.rchit:26-71, untextured materials
    pr, N, position = interpolate_hit(scene, prim_index, tri, u, v)
    m = pr["material"]
    assert int(m["base_color_texture"]) == -1 and int(m["metallic_roughness_texture"]) == -1
    albedo = np.asarray(m["base_color"], np.float64)[:3]
    roughness = min(max(float(m["roughness_factor"]), 0.04), 1.0)
    metallic = min(max(float(m["metallic_factor"]), 0.0), 1.0)
    cam = _mat(pfd, "camera_view_inverse")[:3, 3]
    V = (cam - position) / np.linalg.norm(cam - position)
    light = pfd["directional_light"]
    L = -np.asarray(light["direction"], np.float64)[:3]
    diffuse, specular, nl = brdf_terms(albedo, metallic, roughness, N, V, L)
    return albedo * (0.2 / np.pi) + (diffuse + specular) * nl * np.asarray(light["intensity"], np.float64)[:3] * np.asarray(light["color"], np.float64)[:3]


# ---------------------------------------------------------------------------------------------------------
# Screen-space alternatives in float64, vectorised over the image, written from the GLSL (ssao.comp:14-53,
# ssao_blur.comp:11-26, ssr.comp:16-137, glsl_common.h:111-122, common.glsl:116-150): a second derivation compared with
# the oracle at a tolerance (float64 here, fp32 there; np.sin / np.cos instead of the shared polynomial).
# ---------------------------------------------------------------------------------------------------------
def seed_thread_v(seed):
    seed = np.asarray(seed, np.uint64) & 0xffffffff
    seed = ((seed ^ 61) ^ (seed >> 16)) & 0xffffffff
    seed = (seed * 9) & 0xffffffff
    seed = (seed ^ (seed >> 4)) & 0xffffffff
    seed = (seed * 0x27d4eb2d) & 0xffffffff
    seed = (seed ^ (seed >> 15)) & 0xffffffff
    return seed


def random01_v(state):
    state = state ^ ((state << 13) & 0xffffffff)
    state = state ^ (state >> 17)
    state = state ^ ((state << 5) & 0xffffffff)
    state &= 0xffffffff
    bits = (0x3f800000 | (state >> 9)).astype(np.uint32)
    return state, bits.view(np.float32).astype(np.float64) - 1.0


def sample_linear_repeat(img, u, v):
    """texture() with the default sampler (LINEAR, REPEAT, resource_manager.cpp:58-69) on an (H, W[, C]) float array."""
    H, W = img.shape[:2]
    with np.errstate(invalid="ignore"):
        fx, fy = u * W - 0.5, v * H - 0.5
        x0f, y0f = np.floor(fx), np.floor(fy)
        ax, ay = fx - x0f, fy - y0f
        bad = ~(np.isfinite(x0f) & np.isfinite(y0f))
        x0 = np.mod(np.where(bad, 0, np.clip(x0f, -2**31, 2**31 - 1)).astype(np.int64), W)
        y0 = np.mod(np.where(bad, 0, np.clip(y0f, -2**31, 2**31 - 1)).astype(np.int64), H)
    x1, y1 = (x0 + 1) % W, (y0 + 1) % H
    if img.ndim == 3:
        ax, ay = ax[..., None], ay[..., None]
    return (img[y0, x0] * (1 - ax) + img[y0, x1] * ax) * (1 - ay) + (img[y1, x0] * (1 - ax) + img[y1, x1] * ax) * ay


def _unproject(M, depth, u, v):
    p = np.stack([u * 2.0 - 1.0, v * 2.0 - 1.0, depth, np.ones_like(depth)], -1) @ M.T
    with np.errstate(divide="ignore", invalid="ignore"):
        return p[..., :3] / p[..., 3:4]


def _fmax(a, b):          # IEEE maxNum: a NaN operand loses (oracle decision xii)
    return np.fmax(a, b)


def ssao(pfd, normals_bits, depth, radius=0.75):
    H, W = depth.shape
    depth = depth.astype(np.float64)
    nrm = h2f(normals_bits).astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    inv = np.asarray(pfd["display_size_inverse"], np.float64)
    cu, cv = xs * inv[0], ys * inv[1]
    cur = sample_linear_repeat(depth, cu, cv)
    Pinv = _mat(pfd, "camera_proj_inverse")
    P = _unproject(Pinv, cur, cu, cv)
    N = sample_linear_repeat(nrm, cu, cv)[..., :3] @ _mat(pfd, "camera_view")[:3, :3].T
    with np.errstate(divide="ignore", invalid="ignore"):
        pr = radius / P[..., 2]
    rng = seed_thread_v((ys.astype(np.uint64) * np.uint64(int(pfd["display_size"][1])) + xs.astype(np.uint64)) * np.uint64(int(pfd["frame_index"])))
    total = np.zeros((H, W))
    for _ in range(16):
        rng, r1 = random01_v(rng)
        rng, r2 = random01_v(rng)
        ang, dist = r1 * 2 * np.pi, r2 * pr
        su, sv = cu + np.cos(ang) * dist, cv + np.sin(ang) * dist
        V = _unproject(Pinv, sample_linear_repeat(depth, su, sv), su, sv) - P
        with np.errstate(invalid="ignore", over="ignore"):
            total = total + _fmax((V * N).sum(-1) - 1e-4, 0.0) / ((V * V).sum(-1) + 1e-4)
    ao = _fmax(1.0 - (2.0 / 16.0) * total, 0.0)
    return np.where(cur == 0.0, 0.0, ao)


def ssao_blur_f32(raw_bits, display_w, display_h):
    """ssao_blur.comp in fp32 with the shader's summation order (rows top to bottom, taps left to right): bit-comparable."""
    H, W = raw_bits.shape[:2]
    src = h2f(raw_bits[..., 0])
    acc = np.zeros((H, W), np.float32)
    for dy in range(-6, 7):
        for dx in range(-6, 7):
            tap, mask = _shift(src, dx, dy)
            ys, xs = np.mgrid[0:H, 0:W]
            mask &= ((xs + dx) < display_w) & ((ys + dy) < display_h)
            acc = np.where(mask, (acc + tap).astype(np.float32), acc)
    return (acc / np.float32(13.0 * 13.0)).astype(np.float32)


def ssr(pfd, albedo_bgra8, normals_bits, motion_bits, depth, ray_distance=25.0, step_size=0.1, thickness=0.5, bsearch_steps=10):
    """Returns (found mask, lighting (H, W, 3), final step offset) in float64."""
    H, W = depth.shape
    depth = depth.astype(np.float64)
    nrm = h2f(normals_bits).astype(np.float64)
    mot = h2f(motion_bits).astype(np.float64)
    alb = albedo_bgra8[..., [2, 1, 0]].astype(np.float64) / 255.0
    ys, xs = np.mgrid[0:H, 0:W]
    inv = np.asarray(pfd["display_size_inverse"], np.float64)
    cu, cv = xs * inv[0], ys * inv[1]
    VPinv = _mat(pfd, "camera_viewproj_inverse")
    PV = _mat(pfd, "camera_proj") @ _mat(pfd, "camera_view")
    cam = _mat(pfd, "camera_view_inverse")[:3, 3]
    P = _unproject(VPinv, sample_linear_repeat(depth, cu, cv), cu, cv)
    N = sample_linear_repeat(nrm, cu, cv)[..., :3]
    with np.errstate(invalid="ignore", divide="ignore"):
        I = (P - cam) / np.linalg.norm(P - cam, axis=-1, keepdims=True)
        R = I - 2.0 * (N * I).sum(-1, keepdims=True) * N
        R = R / np.linalg.norm(R, axis=-1, keepdims=True)

    def probe(offset):
        rp = P + R * offset[..., None]
        clip = np.concatenate([rp, np.ones((H, W, 1))], -1) @ PV.T
        with np.errstate(invalid="ignore", divide="ignore"):
            uv = clip[..., :2] / clip[..., 3:4] * 0.5 + 0.5
            sp = _unproject(VPinv, sample_linear_repeat(depth, uv[..., 0], uv[..., 1]), uv[..., 0], uv[..., 1])
            delta = np.linalg.norm(cam - rp, axis=-1) - np.linalg.norm(cam - sp, axis=-1)
        return delta, uv

    found = np.zeros((H, W), bool)
    prev = np.zeros((H, W))
    final = np.zeros((H, W))
    for i in range(int(np.float32(ray_distance) / np.float32(step_size))):
        off = np.full((H, W), float(np.float32(step_size) * np.float32(i)))
        with np.errstate(invalid="ignore"):
            delta, _ = probe(off)
            hit = ~found & (delta > 0.3) & (delta < thickness)
        final = np.where(hit, off, final)
        prev = np.where(~found & ~hit, off, prev)
        found |= hit
    mid = (prev + final) * 0.5
    uv = np.zeros((H, W, 2))
    for _ in range(bsearch_steps):
        with np.errstate(invalid="ignore"):
            delta, uv = probe(mid)
            inside = (delta > 0.3) & (delta < thickness)
        new_mid = np.where(inside, (prev + mid) * 0.5, mid + (mid - prev))
        prev = np.where(inside, prev, mid)
        mid = new_mid
    fu, fv = uv[..., 0], uv[..., 1]
    albedo = sample_linear_repeat(alb, fu, fv)
    with np.errstate(invalid="ignore", divide="ignore"):
        position = _unproject(VPinv, sample_linear_repeat(depth, fu, fv), fu, fv)
        mr = sample_linear_repeat(mot, fu, fv)[..., 2:4]
        V = (cam - position) / np.linalg.norm(cam - position, axis=-1, keepdims=True)
        light = pfd["directional_light"]
        L = -np.asarray(light["direction"], np.float64)[:3]
        Nl = sample_linear_repeat(nrm, fu, fv)[..., :3]
        Hh = (L + V) / np.linalg.norm(L + V, axis=-1, keepdims=True)
        metallic = np.clip(mr[..., 0:1], 0.0, 1.0)
        rough = np.clip(mr[..., 1:2], 0.04, 1.0)
        f0 = 0.04 * (1.0 - metallic) + albedo * metallic
        hv = np.fmax((Hh * V).sum(-1, keepdims=True), 0.0)
        F = f0 + (1.0 - f0) * (1.0 - hv) ** 5
        a2 = rough * rough
        nh = np.fmax((Nl * Hh).sum(-1, keepdims=True), 0.0)
        f = nh * nh * (a2 - 1.0) + 1.0
        D = a2 / (np.pi * f * f)
        k = (rough + 1.0) ** 2 * 0.125
        nv, nl = np.fmax((Nl * V).sum(-1, keepdims=True), 0.0), np.fmax((Nl * L).sum(-1, keepdims=True), 0.0)
        G = (nv / (nv * (1.0 - k) + k)) * (nl / (nl * (1.0 - k) + k))
        spec = D * G * F / np.fmax(4.0 * nv * nl, 1e-6)
        diff = (1.0 - F) * (1.0 - metallic) * albedo / np.pi
        lit = albedo * (0.2 / np.pi) + (diff + spec) * nl * np.asarray(light["intensity"], np.float64)[:3] * np.asarray(light["color"], np.float64)[:3]
    return found, np.where(found[..., None], lit, 0.0), np.where(found, final, -1.0)
