"""glTF scene host against a file written here: the arrays must be the ones
SceneLoader::ParseglTF / ParseNode (scene_loader.cpp:40-332) would hand to the resource manager."""
import base64
import io
import json
import struct

import numpy as np
import pytest

from vulkanhybridrenderer_amd import abi, camera, gltf


def _quat(yaw, pitch):
    """rotation of yawPitchRoll(yaw, pitch, 0) as a glTF quaternion (x, y, z, w)."""
    r = camera.yaw_pitch_roll(yaw, pitch, 0.0)[:3, :3]
    w = np.sqrt(1.0 + r[0, 0] + r[1, 1] + r[2, 2]) / 2
    return [float((r[2, 1] - r[1, 2]) / (4 * w)), float((r[0, 2] - r[2, 0]) / (4 * w)), float((r[1, 0] - r[0, 1]) / (4 * w)), float(w)]


def _png(rgba):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgba, "RGBA").save(b, format="PNG")
    return b.getvalue()


def _write(tmp_path, glb=False):
    quad_pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    quad_nrm = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    quad_tan = np.tile(np.array([[1, 0, 0, 1]], np.float32), (4, 1))
    quad_uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    quad_idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    tri_pos = np.array([[0, 0, 1], [2, 0, 1], [0, 2, 1]], np.float32)
    tri_uv8 = np.array([[0, 0], [255, 0], [0, 255]], np.uint8)              # normalised UNSIGNED_BYTE texcoords, stride 4
    tri_uv8_padded = np.zeros((3, 4), np.uint8); tri_uv8_padded[:, :2] = tri_uv8
    tri_idx = np.array([0, 1, 2], np.uint32)
    chunks, views = [], []

    def view(arr, stride=None):
        raw = arr.tobytes()
        off = sum(len(c) for c in chunks)
        chunks.append(raw + b"\0" * (-len(raw) % 4))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    acc = []

    def accessor(arr, ctype, typ, stride=None, normalized=False, count=None):
        a = {"bufferView": view(arr, stride), "componentType": ctype, "count": count if count is not None else len(arr), "type": typ}
        if normalized:
            a["normalized"] = True
        acc.append(a)
        return len(acc) - 1

    a_qp = accessor(quad_pos, 5126, "VEC3"); a_qn = accessor(quad_nrm, 5126, "VEC3"); a_qt = accessor(quad_tan, 5126, "VEC4")
    a_qu = accessor(quad_uv, 5126, "VEC2"); a_qi = accessor(quad_idx, 5123, "SCALAR")
    a_tp = accessor(tri_pos, 5126, "VEC3"); a_tu = accessor(tri_uv8_padded, 5121, "VEC2", stride=4, normalized=True, count=3)
    a_ti = accessor(tri_idx, 5125, "SCALAR")
    base = np.zeros((4, 4, 4), np.uint8); base[..., 0] = 200; base[..., 3] = 255; base[0, 0] = [10, 20, 30, 255]
    mr = np.full((2, 2, 4), 128, np.uint8)
    nm = np.zeros((2, 2, 4), np.uint8); nm[..., 2] = 255; nm[..., 3] = 255
    (tmp_path / "base.png").write_bytes(_png(base))
    png_nm = _png(nm)
    nm_view = view(np.frombuffer(png_nm, np.uint8))
    doc = {
        "asset": {"version": "2.0"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 5.0}]}},
        "scene": 0, "scenes": [{"nodes": [0, 2, 3]}],
        "nodes": [
            {"name": "parent", "translation": [1, 2, 3], "scale": [2, 2, 2], "children": [1]},
            {"name": "child", "mesh": 0, "translation": [0, 0, -1]},
            {"name": "cam", "camera": 0, "translation": [0, 1.5, 6], "rotation": _quat(0.5, -0.2), "scale": [3, 3, 3]},
            {"name": "sun", "rotation": _quat(0.3, -1.0), "extensions": {"KHR_lights_punctual": {"light": 0}}},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.05, "aspectRatio": 1.5}}],
        "meshes": [{"primitives": [
            {"attributes": {"POSITION": a_qp, "NORMAL": a_qn, "TANGENT": a_qt, "TEXCOORD_0": a_qu}, "indices": a_qi, "material": 0},
            {"attributes": {"POSITION": a_tp, "TEXCOORD_0": a_tu}, "indices": a_ti, "material": 1}]}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1}, "metallicFactor": 0.25,
                                      "roughnessFactor": 0.5, "baseColorFactor": [0.1, 0.2, 0.3, 1.0]}, "normalTexture": {"index": 2}},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0]}, "alphaMode": "MASK", "alphaCutoff": 0.3}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1}, {"source": 2, "sampler": 1}],
        "samplers": [{"magFilter": 0x2600, "minFilter": 0x2702, "wrapS": 0x812F, "wrapT": 0x8370}, {"magFilter": 0x2601, "minFilter": 0x2601}],
        "images": [{"uri": "base.png", "name": "base"}, {"uri": "data:image/png;base64," + base64.b64encode(_png(mr)).decode(), "name": "mr"},
                   {"bufferView": nm_view, "mimeType": "image/png", "name": "nm"}],
        "accessors": acc, "bufferViews": views,
    }
    blob = b"".join(chunks)
    if glb:
        doc["buffers"] = [{"byteLength": len(blob)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        body = struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(blob), b"BIN\0") + blob
        path = tmp_path / "scene.glb"
        path.write_bytes(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)
    else:
        (tmp_path / "scene.bin").write_bytes(blob)
        doc["buffers"] = [{"uri": "scene.bin", "byteLength": len(blob)}]
        path = tmp_path / "scene.gltf"
        path.write_text(json.dumps(doc))
    return str(path), dict(base=base, mr=mr, nm=nm)


@pytest.mark.parametrize("glb", [False, True])
def test_arrays_match_scene_loader_semantics(tmp_path, glb):
    path, imgs = _write(tmp_path, glb)
    s = gltf.load(path)
    assert s.vertices.dtype == abi.vertex_dtype and len(s.vertices) == 7 and len(s.indices) == 9
    p0, p1 = s.primitives
    assert (p0["vertex_offset"], p0["index_offset"], p0["index_count"]) == (0, 0, 6)
    assert (p1["vertex_offset"], p1["index_offset"], p1["index_count"]) == (4, 6, 3)
    assert s.indices.tolist() == [0, 1, 2, 0, 2, 3, 0, 1, 2]                        # primitive-relative (scene_loader.cpp:176-177)
    # world transform of the child: T(1,2,3) * S(2) * T(0,0,-1)
    m = abi.glm_to_mat(p0["transform"])
    assert np.allclose(m, [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 1], [0, 0, 0, 1]])
    assert np.array_equal(s.primitives[1]["transform"], p0["transform"])
    assert np.allclose(s.vertices["normal"][:4], [0, 0, 1]) and np.allclose(s.vertices["normal"][4:], 0)      # absent -> zeros
    assert np.allclose(s.vertices["tangent"][:4], [1, 0, 0, 1])
    assert np.allclose(s.vertices["uv0"][4:], [[0, 0], [1, 0], [0, 1]])              # normalised bytes, strided view
    # materials (:178-211)
    m0, m1 = p0["material"], p1["material"]
    assert (m0["base_color_texture"], m0["metallic_roughness_texture"], m0["normal_map"]) == (0, 1, 2)
    assert np.allclose(m0["base_color"], 1.0)                                        # factor ignored when a texture is present (:191-197)
    assert np.isclose(m0["metallic_factor"], 0.25) and np.isclose(m0["roughness_factor"], 0.5) and m0["alpha_mask"] == 0
    assert (m1["base_color_texture"], m1["metallic_roughness_texture"], m1["normal_map"]) == (-1, -1, -1)
    assert np.allclose(m1["base_color"], [0.9, 0.8, 0.7, 1.0]) and m1["alpha_mask"] == 1 and np.isclose(m1["alpha_cutoff"], 0.3)
    assert np.isclose(m1["metallic_factor"], 1.0) and np.isclose(m1["roughness_factor"], 1.0)
    # textures: format by use, sampler enums (:8-38, :222-259)
    t0, t1, t2 = s.textures
    assert t0["format"] == abi.FORMAT_R8G8B8A8_SRGB and t1["format"] == abi.FORMAT_R8G8B8A8_UNORM and t2["format"] == abi.FORMAT_R8G8B8A8_UNORM
    assert (t0["mag"], t0["min"], t0["address_u"], t0["address_v"]) == (0, 1, 2, 1)
    assert (t1["mag"], t1["min"], t1["address_u"], t1["address_v"]) == (1, 1, 0, 0)
    assert np.array_equal(t0["rgba8"], imgs["base"]) and np.array_equal(t1["rgba8"], imgs["mr"]) and np.array_equal(t2["rgba8"], imgs["nm"])
    # camera: yaw / pitch recovered from the world transform, node scale dropped (:58-69)
    c = s.camera
    assert np.allclose([c["yaw"], c["pitch"], c["roll"]], [0.5, -0.2, 0.0], atol=1e-6)
    assert np.allclose(c["position"], [0, 1.5, 6]) and np.isclose(c["yfov"], 0.8) and np.isclose(c["znear"], 0.05) and np.isclose(c["aspect"], 1.5)
    # light: rot * (0, 0, -1), colour from the file, intensity hard-wired to 30 (:86-97)
    want = camera.yaw_pitch_roll(0.3, -1.0, 0.0)[:3, :3] @ np.array([0, 0, -1.0])
    assert np.allclose(s.light["direction"][:3], want, atol=1e-6) and s.light["direction"][3] == 0
    assert np.allclose(s.light["color"], [1.0, 0.9, 0.8, 1.0]) and np.allclose(s.light["intensity"], 30.0)
    # the frame driver accepts it
    pfds = camera.dolly_frames(s, 96, 64, 2)
    assert pfds[1]["frame_index"] == 1


def test_euler_extraction_round_trip():
    rng = np.random.default_rng(0)
    for _ in range(50):
        y, p, r = rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)
        assert np.allclose(gltf.extract_euler_yxz(camera.yaw_pitch_roll(y, p, r)), [y, p, r], atol=1e-9)


def test_unsupported_inputs_raise(tmp_path):
    path, _ = _write(tmp_path)
    doc = json.loads(open(path).read())
    doc["meshes"][0]["primitives"][0]["mode"] = 1
    bad = tmp_path / "lines.gltf"
    bad.write_text(json.dumps(doc))
    with pytest.raises(gltf.GltfError):
        gltf.load(str(bad))
    doc["meshes"][0]["primitives"][0]["mode"] = 4
    del doc["meshes"][0]["primitives"][0]["indices"]
    bad.write_text(json.dumps(doc))
    with pytest.raises(gltf.GltfError):
        gltf.load(str(bad))


def test_default_light_without_punctual_lights(tmp_path):
    path, _ = _write(tmp_path)
    doc = json.loads(open(path).read())
    doc["nodes"][3] = {"name": "sun"}
    p = tmp_path / "nolight.gltf"
    p.write_text(json.dumps(doc))
    s = gltf.load(str(p))
    assert np.allclose(s.light["direction"], [0, -1, 0.01, 0]) and np.allclose(s.light["color"], [1, 1, 1, 0])    # :324-329
    assert np.allclose(s.light["intensity"], 0)


@pytest.mark.parametrize("which", ["tiny", "bistro_small"])
def test_export_import_round_trip(tmp_path, which):
    """save() then load() returns the arrays bit for bit: what the hot path is fed does not depend on the container."""
    from vulkanhybridrenderer_amd import scenes
    s = scenes.tiny_scene() if which == "tiny" else scenes.bistro_proc(detail=0.002, n_primitives=40, n_textures=3, texture_size=16)
    path = str(tmp_path / "scene.glb")
    gltf.save(s, path)
    t = gltf.load(path)
    assert np.array_equal(t.vertices, s.vertices) and np.array_equal(t.indices, s.indices)
    for a, b in zip(t.primitives, s.primitives):
        b = b.copy()
        if b["material"]["base_color_texture"] >= 0:
            b["material"]["base_color"] = 1.0        # the factor is ignored when a texture is present (scene_loader.cpp:191-197)
        if not b["material"]["alpha_mask"]:
            b["material"]["alpha_cutoff"] = 0.0      # only read for MASK materials (:208-211)
        assert a.tobytes() == b.tobytes()
    assert len(t.textures) == len(s.textures)
    used = sorted({int(x) for p in s.primitives for x in (p["material"]["base_color_texture"], p["material"]["metallic_roughness_texture"],
                                                        p["material"]["normal_map"]) if x >= 0})
    for i in used:
        assert np.array_equal(t.textures[i]["rgba8"], s.textures[i]["rgba8"]) and t.textures[i]["format"] == s.textures[i]["format"]
    assert np.allclose(t.camera["position"], s.camera["position"]) and np.isclose(t.camera["yaw"], s.camera["yaw"], atol=1e-6)
    assert np.isclose(t.camera["pitch"], s.camera["pitch"], atol=1e-6)
    assert np.allclose(t.light["direction"], s.light["direction"], atol=1e-6)


def test_light_projview_is_the_reference_shadow_frustum():
    """scene_loader.cpp:85-94: ortho(-8, 8, -8, 8, 12, 0.1) * lookAt(-direction * 12, origin, +y) with zero-to-one depth:
    the origin sits in the middle of the shadow map at depth 0 (12 m from the light's eye point), a point 0.1 m in front
    of the eye at depth 1, and 8 m sideways is the edge of the map."""
    import numpy as np
    from vulkanhybridrenderer_amd import abi, camera
    d = np.array([0.0, -0.97, 0.35]) / np.linalg.norm([0.0, -0.97, 0.35])
    light = camera.directional_light(d)
    pv = abi.glm_to_mat(light["projview"])

    def ndc(p):
        r = pv @ np.append(np.asarray(p, np.float64), 1.0)
        return r[:3] / r[3]
    assert np.allclose(ndc([0, 0, 0]), [0, 0, 0], atol=1e-6)
    assert np.allclose(ndc(-d * 12 + d * 0.1), [0, 0, 1], atol=1e-6)
    side = np.cross(d, [0.0, 1.0, 0.0])
    side /= np.linalg.norm(side)
    assert np.allclose(np.abs(ndc(side * 8.0)[:2]).max(), 1.0, atol=1e-6)
    assert np.allclose(light["direction"][:3], d, atol=1e-7) and light["intensity"][0] == 30.0
