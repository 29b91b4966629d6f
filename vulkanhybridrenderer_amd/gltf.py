"""Vulkan-free glTF 2.0 scene host: produces the flat Vertex / index / Primitive arrays,
texture list, camera and directional light that SceneLoader::ParseglTF / ParseNode hand to
ResourceManager::UpdateGeometry and UploadTextureFromData (/root/reference/src/scene/scene_loader.cpp:40-332), so that
a real Sponza / Bistro file drops into the hot path the moment the asset is supplied (none ships with the reference:
.gitignore:3).  The reference parses with cgltf and decodes with stb_image; here: json + numpy; PNG through Pillow's inflate,
JPEG through a restatement of stb_image's decoder (stb_jpeg.py).  Pinned against cgltf 1.9, glm 0.9.9.8 and stb_image 2.26
themselves -- the reference's vendored copies, compiled in the build container -- by tests/test_reference_pins.py.

Semantics kept from the reference (file:line = scene_loader.cpp):
  * nodes are visited in ARRAY order, each with its WORLD transform (cgltf_node_transform_world, :58,:75,:108);
  * a camera node sets the reverse-Z infinite projection from yfov / aspectRatio / znear and the transform
    T * yawPitchRoll(extractEulerAngleYXZ(world)) (:43-71) -- the angles are taken from the world matrix as it stands, so
    a scaled or sheared camera node bends them exactly as in the reference (camera_from_world);
  * a KHR_lights_punctual directional light gives direction = normalize(rot * (0, 0, -1)) with rot from glm::decompose
    (Gram-Schmidt on the world matrix' axes, light_direction_from_world), its colour, and the
    hard-wired intensity 30 (2 for "Pica.glb") (:73-99); without one: direction (0, -1, 0.01), colour (1, 1, 1, 0),
    intensity 0 (:324-329);
  * every mesh primitive becomes one Primitive {world transform, material, vertex_offset, index_offset, index_count};
    vertices keep POSITION / NORMAL / TANGENT / TEXCOORD_0 / TEXCOORD_1 (zeros when absent), indices stay
    primitive-relative (:105-212); primitives must be indexed triangle lists (asserts :113, :174);
  * material: base colour texture OR factor (:191-197), metallic-roughness texture, the two factors, normal map,
    alpha MASK flag + cutoff (:199-211); defaults of :178-187;
  * texture formats by use: base colour R8G8B8A8_SRGB, metallic-roughness and normal maps R8G8B8A8_UNORM (:222-259),
    RGBA8 (stbi STBI_rgb_alpha), sampler filter / wrap enums mapped as :8-38.
Differences: texture indices are assigned in first-use order (the reference assigns them inside an OpenMP loop, i.e.
nondeterministically, :261-294, and keys its map by the image-name POINTER, so unnamed images collide); sparse
accessors and non-triangle primitives raise instead of asserting.
"""
import base64
import json
import os
import struct

import numpy as np

from . import abi
from .scenes import Scene

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}
_FILTER = {0x2600: 0, 0x2700: 0, 0x2701: 0, 0x2601: 1, 0x2702: 1, 0x2703: 1}            # :8-22 (VkFilter NEAREST 0 / LINEAR 1)
_WRAP = {0x2901: 0, 0x8370: 1, 0x812F: 2, 0x812D: 3}                                     # :24-38 (VkSamplerAddressMode)


class GltfError(ValueError):
    pass


def _read_container(path):
    """-> (json dict, binary chunk or None)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] == b"glTF":                                          # .glb: 12-byte header, then chunks
        _, _, total = struct.unpack_from("<4sII", raw, 0)
        off, doc, blob = 12, None, None
        while off + 8 <= min(total, len(raw)):
            n, kind = struct.unpack_from("<I4s", raw, off)
            body = raw[off + 8: off + 8 + n]
            if kind == b"JSON":
                doc = json.loads(body.decode("utf-8"))
            elif kind == b"BIN\x00" and blob is None:
                blob = body
            off += 8 + n + (-n % 4)
        if doc is None:
            raise GltfError("glb without a JSON chunk")
        return doc, blob
    return json.loads(raw.decode("utf-8")), None


def _load_buffers(doc, base_dir, blob):
    out = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if blob is None:
                raise GltfError(f"buffer {i} has no uri and the file has no binary chunk")
            out.append(blob)
        elif uri.startswith("data:"):
            out.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(os.path.join(base_dir, uri), "rb") as f:
                out.append(f.read())
    return out


def _accessor(doc, buffers, index):
    """cgltf_accessor_read_float / read_index: -> float32 [count, ncomp] (normalised integers scaled) or raw integers."""
    acc = doc["accessors"][index]
    if "sparse" in acc:
        raise GltfError("sparse accessors are not supported")
    dt = np.dtype(_COMPONENT[acc["componentType"]]).newbyteorder("<")
    ncomp, count = _NCOMP[acc["type"]], acc["count"]
    if "bufferView" not in acc:
        return np.zeros((count, ncomp), dt)
    view = doc["bufferViews"][acc["bufferView"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride") or dt.itemsize * ncomp
    data = buffers[view["buffer"]]
    arr = np.ndarray((count, ncomp), dt, buffer=data, offset=start, strides=(stride, dt.itemsize))
    if acc.get("normalized") and dt.kind in "iu":
        info = np.iinfo(dt)
        # cgltf 1.9's cgltf_component_read_float: value / max as it stands -- the most negative integer maps to just below
        # -1 (-32768 / 32767), unlike the glTF specification's max(c / max, -1); pinned by tests/golden/ref_gltf/nested.gltf
        return arr.astype(np.float32) / np.float32(info.max)
    return np.array(arr)


def _local_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T            # glTF stores column-major
    t = np.array(node.get("translation", [0, 0, 0]), np.float64)
    x, y, z, w = np.array(node.get("rotation", [0, 0, 0, 1]), np.float64)
    s = np.array(node.get("scale", [1, 1, 1]), np.float64)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m


def _world_matrices(doc):
    nodes = doc.get("nodes", [])
    parent = [None] * len(nodes)
    for i, n in enumerate(nodes):
        for c in n.get("children", []):
            parent[c] = i
    world = [None] * len(nodes)

    def solve(i):
        if world[i] is None:
            m = _local_matrix(nodes[i])
            world[i] = m if parent[i] is None else solve(parent[i]) @ m
        return world[i]

    for i in range(len(nodes)):
        solve(i)
    return world


def extract_euler_yxz(m):
    """glm::extractEulerAngleYXZ on a math matrix m[row, col] (glm's M[c][r] = m[r, c])."""
    t1 = np.arctan2(m[0, 2], m[2, 2])
    c2 = np.sqrt(m[1, 0] ** 2 + m[1, 1] ** 2)
    t2 = np.arctan2(-m[1, 2], c2)
    s1, c1 = np.sin(t1), np.cos(t1)
    t3 = np.arctan2(s1 * m[2, 1] - c1 * m[0, 1], c1 * m[0, 0] - s1 * m[2, 0])
    return float(t1), float(t2), float(t3)


def decode_image_bytes(data):
    """stbi_load_from_memory(..., STBI_rgb_alpha) (scene_loader.cpp:277-290): encoded PNG / JPEG bytes -> RGBA8 [h, w, 4].
    PNG goes through Pillow's inflate + unfilter and stb_image's own colour-type expansion rules (16-bit samples keep their
    high byte, grey replicates, missing alpha = 255); JPEG through the restatement of stb_image's decoder in stb_jpeg.py --
    JPEG decoders differ in IDCT, chroma upsampling and colour conversion, and the reference's textures are what stb makes of
    them.  tests/test_reference_pins.py compares both with stb_image 2.26 itself, texel for texel."""
    if data[:2] == b"\xff\xd8":
        from . import stb_jpeg
        return stb_jpeg.decode_rgba(data)
    try:
        from PIL import Image
    except ImportError as e:                                   # no silent stand-in pixels
        raise GltfError("decoding glTF images needs Pillow (PIL)") from e
    import io
    im = Image.open(io.BytesIO(data))
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):             # 16-bit grey: stbi__convert_16_to_8 keeps the high byte
        g = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
        return np.stack([g, g, g, np.full_like(g, 255)], -1)
    return np.array(im.convert("RGBA"), np.uint8)


def _decode_image(doc, buffers, base_dir, image_index):
    img = doc["images"][image_index]
    if "uri" in img:
        uri = img["uri"]
        if uri.startswith("data:"):
            data = base64.b64decode(uri.split(",", 1)[1])
        else:
            with open(os.path.join(base_dir, uri), "rb") as f:
                data = f.read()
    else:
        view = doc["bufferViews"][img["bufferView"]]
        off = view.get("byteOffset", 0)
        data = buffers[view["buffer"]][off: off + view["byteLength"]]
    return decode_image_bytes(bytes(data))


def camera_from_world(m):
    """scene_loader.cpp:58-66 on the camera node's WORLD matrix m[row, col]: the Euler angles are extracted from the matrix
    as it stands (scale and shear included -- glm::extractEulerAngleYXZ does not normalise), the camera transform is
    T(translation) * yawPitchRoll(angles), the view its inverse.  -> (yaw, pitch, roll, transform, view)."""
    from .camera import yaw_pitch_roll
    yaw, pitch, roll = extract_euler_yxz(np.asarray(m, np.float64))
    t = np.eye(4)
    t[:3, 3] = np.asarray(m, np.float64)[:3, 3]
    transform = t @ yaw_pitch_roll(yaw, pitch, roll)
    return yaw, pitch, roll, transform, np.linalg.inv(transform)


def light_direction_from_world(m):
    """scene_loader.cpp:74-86: glm::decompose(world) -> rotation quaternion -> normalize(rot * (0, 0, -1)).  decompose
    orthonormalises the three axes in order (x; y minus its x part; z minus its x and y parts: Gram-Schmidt, i.e. shear
    is removed from the LATER axes) and flips all three if the basis is left-handed; the direction is minus the third."""
    a = np.asarray(m, np.float64)[:3, :3] / np.asarray(m, np.float64)[3, 3]
    x = a[:, 0] / np.linalg.norm(a[:, 0])
    y = a[:, 1] - x * np.dot(x, a[:, 1])
    y /= np.linalg.norm(y)
    z = a[:, 2] - x * np.dot(x, a[:, 2])
    z = z - y * np.dot(y, z)
    z /= np.linalg.norm(z)
    if np.dot(x, np.cross(y, z)) < 0:
        z = -z
    return -z


def load(path, dolly_step=0.05):
    """-> scenes.Scene with the arrays scene_loader.cpp would hand to the resource manager."""
    doc, blob = _read_container(path)
    base_dir = os.path.dirname(os.path.abspath(path))
    buffers = _load_buffers(doc, base_dir, blob)
    name = os.path.basename(path)
    world = _world_matrices(doc)
    materials = doc.get("materials", [])

    # ---- textures in first-use order, format by use (:222-259) ----
    textures, slot_of = [], {}

    def texture_slot(tex_index, fmt):
        if tex_index is None:
            return -1
        if tex_index in slot_of:
            return slot_of[tex_index]
        tex = doc["textures"][tex_index]
        if "source" not in tex:
            raise GltfError(f"texture {tex_index} has no image source")
        rgba = _decode_image(doc, buffers, base_dir, tex["source"])
        smp = doc.get("samplers", [{}])[tex["sampler"]] if "sampler" in tex else {}
        try:
            entry = dict(rgba8=rgba, format=fmt, mag=_FILTER[smp.get("magFilter", 0x2601)], min=_FILTER[smp.get("minFilter", 0x2601)],
                         address_u=_WRAP[smp.get("wrapS", 0x2901)], address_v=_WRAP[smp.get("wrapT", 0x2901)],
                         name=doc["images"][tex["source"]].get("name"))
        except KeyError as e:
            raise GltfError(f"sampler enum {e} is not one the reference maps (scene_loader.cpp:8-38)") from e
        slot_of[tex_index] = len(textures)
        textures.append(entry)
        return slot_of[tex_index]

    for mesh in doc.get("meshes", []):
        for prim in mesh["primitives"]:
            if "material" not in prim:
                continue
            mat = materials[prim["material"]]
            pbr = mat.get("pbrMetallicRoughness", {})
            texture_slot(pbr.get("baseColorTexture", {}).get("index"), abi.FORMAT_R8G8B8A8_SRGB)
            texture_slot(pbr.get("metallicRoughnessTexture", {}).get("index"), abi.FORMAT_R8G8B8A8_UNORM)
            texture_slot(mat.get("normalTexture", {}).get("index"), abi.FORMAT_R8G8B8A8_UNORM)

    vertices, indices, primitives = [], [], []
    camera = None
    light = None
    lights = doc.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    n_vertices = n_indices = 0
    for ni, node in enumerate(doc.get("nodes", [])):
        m = world[ni]
        if "camera" in node:                                                                   # :43-71
            cam = doc["cameras"][node["camera"]]
            if cam.get("type") != "perspective":
                raise GltfError("only perspective cameras are supported (scene_loader.cpp:44)")
            p = cam["perspective"]
            yaw, pitch, roll, transform, _ = camera_from_world(m)
            forward = -(transform[:3, 2])
            camera = dict(position=[float(v) for v in m[:3, 3]], yaw=yaw, pitch=pitch, roll=roll, yfov=float(p["yfov"]),
                          znear=float(p["znear"]), aspect=(float(p["aspectRatio"]) if "aspectRatio" in p else None),
                          dolly=[float(v) * dolly_step for v in forward])
            continue
        ext = node.get("extensions", {}).get("KHR_lights_punctual")
        if ext is not None and lights[ext["light"]].get("type") == "directional":               # :73-99
            lt = lights[ext["light"]]
            d = light_direction_from_world(m)
            from .camera import directional_light
            light = directional_light(d, tuple(lt.get("color", [1.0, 1.0, 1.0])), 2.0 if name == "Pica.glb" else 30.0)
            continue
        if "mesh" not in node:
            continue
        for prim in doc["meshes"][node["mesh"]]["primitives"]:                                  # :110-212
            if prim.get("mode", 4) != 4:
                raise GltfError("only triangle-list primitives are supported (scene_loader.cpp:113)")
            if "indices" not in prim:
                raise GltfError("primitives must be indexed (scene_loader.cpp:174)")
            attrs = prim["attributes"]
            pos = _accessor(doc, buffers, attrs["POSITION"]).astype(np.float32)
            v = np.zeros(len(pos), abi.vertex_dtype)
            v["pos"] = pos
            for key, field, n in (("NORMAL", "normal", 3), ("TANGENT", "tangent", 4), ("TEXCOORD_0", "uv0", 2), ("TEXCOORD_1", "uv1", 2)):
                if key in attrs:
                    v[field] = _accessor(doc, buffers, attrs[key]).astype(np.float32)[:, :n]
            idx = _accessor(doc, buffers, prim["indices"]).reshape(-1).astype(np.uint32)
            if len(idx) % 3:
                raise GltfError("index count is not a multiple of 3")
            p = np.zeros((), abi.primitive_dtype)
            p["transform"] = abi.mat_to_glm(m)
            mtl = p["material"]
            mtl["base_color"] = [1.0, 1.0, 1.0, 1.0]                                            # defaults :178-187
            mtl["base_color_texture"] = mtl["metallic_roughness_texture"] = mtl["normal_map"] = -1
            mtl["metallic_factor"] = mtl["roughness_factor"] = 1.0
            if "material" in prim:
                mat = materials[prim["material"]]
                pbr = mat.get("pbrMetallicRoughness")
                if pbr is None:
                    raise GltfError("only the PBR metallic-roughness model is supported (scene_loader.cpp:199)")
                bt = pbr.get("baseColorTexture", {}).get("index")
                if bt is not None:
                    mtl["base_color_texture"] = texture_slot(bt, abi.FORMAT_R8G8B8A8_SRGB)      # :191-193
                else:
                    mtl["base_color"] = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])        # :195-197
                mtl["metallic_roughness_texture"] = texture_slot(pbr.get("metallicRoughnessTexture", {}).get("index"), abi.FORMAT_R8G8B8A8_UNORM)
                mtl["metallic_factor"] = pbr.get("metallicFactor", 1.0)
                mtl["roughness_factor"] = pbr.get("roughnessFactor", 1.0)
                nt = mat.get("normalTexture", {}).get("index")
                if nt is not None:
                    if "TANGENT" not in attrs:
                        raise GltfError("normal map without vertex tangents (scene_loader.cpp:207)")
                    mtl["normal_map"] = texture_slot(nt, abi.FORMAT_R8G8B8A8_UNORM)
                if mat.get("alphaMode") == "MASK":                                              # :208-211
                    mtl["alpha_mask"] = 1
                    mtl["alpha_cutoff"] = mat.get("alphaCutoff", 0.5)
            p["vertex_offset"], p["index_offset"], p["index_count"] = n_vertices, n_indices, len(idx)
            vertices.append(v)
            indices.append(idx)
            primitives.append(p)
            n_vertices += len(v)
            n_indices += len(idx)

    if not primitives:
        raise GltfError("the file contains no mesh primitives")
    if light is None:                                                                           # :324-329 (intensity stays 0)
        light = np.zeros((), abi.directional_light_dtype)
        light["projview"] = abi.mat_to_glm(np.eye(4))
        light["direction"] = [0.0, -1.0, 0.01, 0.0]
        light["color"] = [1.0, 1.0, 1.0, 0.0]
    if camera is None:                       # the reference keeps its zero-initialised camera; a usable default instead
        lo, hi = _bounds(vertices, primitives)
        centre, size = (lo + hi) / 2, float(np.max(hi - lo))
        camera = dict(position=[float(centre[0]), float(centre[1]), float(hi[2] + size)], yaw=0.0, pitch=0.0, roll=0.0, yfov=0.9,
                      znear=0.1, aspect=None, dolly=[0.0, 0.0, -dolly_step])
    return Scene(name, np.concatenate(vertices), np.concatenate(indices), np.array(primitives, abi.primitive_dtype), textures, camera, light)


def _bounds(vertices, primitives):
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for v, p in zip(vertices, primitives):
        m = abi.glm_to_mat(p["transform"])
        w = v["pos"].astype(np.float64) @ m[:3, :3].T + m[:3, 3]
        lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
    return lo, hi


# ---------------------------------------------------------------------------------------------
# Exporter: any scenes.Scene -> .glb (tooling for tests and for handing the procedural stand-in scenes to other
# viewers; one node + mesh per Primitive, node matrix = the primitive's transform).
# ---------------------------------------------------------------------------------------------
def _quaternion(r):
    """3x3 rotation -> (x, y, z, w)."""
    t = r[0, 0] + r[1, 1] + r[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s, 0.25 * s]
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        s = np.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2
        q = [0.25 * s, (r[0, 1] + r[1, 0]) / s, (r[0, 2] + r[2, 0]) / s, (r[2, 1] - r[1, 2]) / s]
    elif r[1, 1] > r[2, 2]:
        s = np.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2
        q = [(r[0, 1] + r[1, 0]) / s, 0.25 * s, (r[1, 2] + r[2, 1]) / s, (r[0, 2] - r[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2
        q = [(r[0, 2] + r[2, 0]) / s, (r[1, 2] + r[2, 1]) / s, 0.25 * s, (r[1, 0] - r[0, 1]) / s]
    return [float(v) for v in q]


def save(scene, path):
    """Write `scene` as a binary glTF (.glb) that load() maps back to the same arrays."""
    import io
    from PIL import Image
    from .camera import yaw_pitch_roll
    chunks, views, accessors = [], [], []

    def add_view(raw, target=None):
        off = sum(len(c) for c in chunks)
        chunks.append(raw + b"\0" * (-len(raw) % 4))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if target:
            v["target"] = target
        views.append(v)
        return len(views) - 1

    def add_accessor(arr, ctype, typ, target, minmax=False):
        a = {"bufferView": add_view(np.ascontiguousarray(arr).tobytes(), target), "componentType": ctype, "count": len(arr), "type": typ}
        if minmax:
            a["min"] = [float(v) for v in arr.min(0)]
            a["max"] = [float(v) for v in arr.max(0)]
        accessors.append(a)
        return len(accessors) - 1

    inv_filter = {0: 0x2600, 1: 0x2601}
    inv_wrap = {v: k for k, v in _WRAP.items()}
    images, textures, samplers = [], [], []
    for i, t in enumerate(scene.textures):
        b = io.BytesIO()
        Image.fromarray(np.asarray(t["rgba8"], np.uint8), "RGBA").save(b, format="PNG")
        images.append({"bufferView": add_view(b.getvalue()), "mimeType": "image/png", "name": t.get("name") or f"texture{i}"})
        samplers.append({"magFilter": inv_filter[t["mag"]], "minFilter": inv_filter[t["min"]], "wrapS": inv_wrap[t["address_u"]], "wrapT": inv_wrap[t["address_v"]]})
        textures.append({"source": i, "sampler": i})

    nodes, meshes, materials = [], [], []
    for pi, p in enumerate(scene.primitives):
        v = scene.vertices[int(p["vertex_offset"]):]
        idx = scene.indices[int(p["index_offset"]): int(p["index_offset"]) + int(p["index_count"])]
        v = v[: int(idx.max()) + 1] if len(idx) else v[:0]
        # the vertex range of a primitive ends where the next one starts
        nxt = [int(q["vertex_offset"]) for q in scene.primitives if int(q["vertex_offset"]) > int(p["vertex_offset"])]
        if nxt:
            v = scene.vertices[int(p["vertex_offset"]): min(nxt)]
        else:
            v = scene.vertices[int(p["vertex_offset"]):]
        attrs = {"POSITION": add_accessor(v["pos"], 5126, "VEC3", 34962, True), "NORMAL": add_accessor(v["normal"], 5126, "VEC3", 34962),
                 "TANGENT": add_accessor(v["tangent"], 5126, "VEC4", 34962), "TEXCOORD_0": add_accessor(v["uv0"], 5126, "VEC2", 34962),
                 "TEXCOORD_1": add_accessor(v["uv1"], 5126, "VEC2", 34962)}
        m = p["material"]
        pbr = {"metallicFactor": float(m["metallic_factor"]), "roughnessFactor": float(m["roughness_factor"])}
        if m["base_color_texture"] >= 0:
            pbr["baseColorTexture"] = {"index": int(m["base_color_texture"])}
        else:
            pbr["baseColorFactor"] = [float(x) for x in m["base_color"]]
        if m["metallic_roughness_texture"] >= 0:
            pbr["metallicRoughnessTexture"] = {"index": int(m["metallic_roughness_texture"])}
        mat = {"pbrMetallicRoughness": pbr}
        if m["normal_map"] >= 0:
            mat["normalTexture"] = {"index": int(m["normal_map"])}
        if m["alpha_mask"]:
            mat["alphaMode"], mat["alphaCutoff"] = "MASK", float(m["alpha_cutoff"])
        materials.append(mat)
        meshes.append({"primitives": [{"attributes": attrs, "indices": add_accessor(idx.astype(np.uint32), 5125, "SCALAR", 34963), "material": pi}]})
        nodes.append({"mesh": pi, "matrix": [float(x) for x in p["transform"]]})
    cam = scene.camera
    rot = yaw_pitch_roll(cam["yaw"], cam["pitch"], cam.get("roll", 0.0))[:3, :3]
    persp = {"yfov": float(cam["yfov"]), "znear": float(cam["znear"])}
    if cam.get("aspect"):
        persp["aspectRatio"] = float(cam["aspect"])
    nodes.append({"camera": 0, "translation": [float(x) for x in cam["position"]], "rotation": _quaternion(rot)})
    doc = {"asset": {"version": "2.0", "generator": "vulkanhybridrenderer_amd.gltf"}, "scene": 0, "nodes": nodes, "meshes": meshes,
           "materials": materials, "accessors": accessors, "bufferViews": views,
           "cameras": [{"type": "perspective", "perspective": persp}]}
    if scene.light is not None and float(np.abs(scene.light["intensity"]).max()) > 0:
        d = np.asarray(scene.light["direction"][:3], np.float64)
        d /= np.linalg.norm(d)
        # a rotation taking (0, 0, -1) to d
        z = -d
        x = np.cross([0.0, 1.0, 0.0], z)
        x = x / np.linalg.norm(x) if np.linalg.norm(x) > 1e-9 else np.array([1.0, 0.0, 0.0])
        y = np.cross(z, x)
        doc["extensionsUsed"] = ["KHR_lights_punctual"]
        doc["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "directional", "color": [float(c) for c in scene.light["color"][:3]]}]}}
        nodes.append({"rotation": _quaternion(np.stack([x, y, z], 1)), "extensions": {"KHR_lights_punctual": {"light": 0}}})
    doc["scenes"] = [{"nodes": list(range(len(nodes)))}]
    if images:
        doc["images"], doc["textures"], doc["samplers"] = images, textures, samplers
    blob = b"".join(chunks)
    doc["buffers"] = [{"byteLength": len(blob)}]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    body = struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(blob), b"BIN\0") + blob
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)
