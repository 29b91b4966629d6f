"""Generates the REFERENCE-derived fixtures of tests/golden/ (run in the build container from the repo root:
`python tests/golden/make_ref_pins.py`; needs /root/reference, which never travels to the GPU box).

What is pinned -- the only pieces of the reference that compile in this image without stand-ins, each compiled from where it
lies by `make -C oracle ref` into oracle/_ref/ (git-ignored):
  ref_abi_layout.json   sizeof / offsetof of every struct of src/rendering_backend/glsl_common.h:22-105 (with the vendored glm)
  ref_glm_cases.json    dependencies/glm called as scene_loader.cpp:58-66,74-94 and renderer.cpp:191-196 call it
  ref_gltf/*.json       dependencies/cgltf (parse, world transforms, accessor reads) on the glTF files written into ref_gltf/
  ref_stb_decodes.npz   dependencies/stb/stb_image.h decodes of the encoded images in ref_stb/
Fixtures are DATA: the probes' outputs and the inputs that produced them.  The shaders' arithmetic stays unpinned (no GLSL
compiler, no Vulkan driver in this image): DESIGN.md section 4.
"""
import io
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vulkanhybridrenderer_amd import abi, camera      # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def run(probe, *args, stdin=None, binary=False):
    r = subprocess.run([os.path.join(REF, probe), *args], input=stdin, capture_output=True, check=True, text=not binary)
    return r.stdout


def glm_line(op, values):
    return op + " " + " ".join(repr(float(np.float32(v))) for v in values)


def glm_cases():
    """Inputs chosen here, outputs from glm: {op, in, out}."""
    rng = np.random.default_rng(7)
    lines = []
    dirs = [np.array([0.0, -0.97, 0.35]), np.array([0.3, -0.8, 0.1]), np.array([-0.5, -0.5, -0.7]), np.array([0.7, -0.2, 0.68])]
    for d in dirs:
        d = (d / np.linalg.norm(d)).astype(np.float32)
        lines.append(("ortho_lookat", d.tolist()))
    for _ in range(6):                                   # camera nodes: T * R * S world matrices (scene_loader.cpp:58-66)
        y, p, r = rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)
        m = np.eye(4)
        m[:3, 3] = rng.uniform(-10, 10, 3)
        m = m @ camera.yaw_pitch_roll(y, p, r) @ np.diag([*rng.uniform(0.5, 3.0, 3), 1.0])
        lines.append(("camera", abi.mat_to_glm(m).tolist()))
    for _ in range(6):                                   # light nodes: rotation (x uniform scale) (scene_loader.cpp:74-86)
        y, p, r = rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)
        m = camera.yaw_pitch_roll(y, p, r) @ np.diag([*([rng.uniform(0.5, 2.0)] * 3), 1.0])
        m[:3, 3] = rng.uniform(-5, 5, 3)
        lines.append(("lightdir", abi.mat_to_glm(m).tolist()))
    for yfov, aspect, znear in ((0.9, 16 / 9, 0.1), (0.8, 1.5, 0.05), (1.2, 1.0, 0.01)):     # renderer.cpp:194-195
        proj = camera.infinite_reverse_depth_projection(yfov, aspect, znear)
        t = np.eye(4)
        t[:3, 3] = rng.uniform(-10, 10, 3)
        view = np.linalg.inv(t @ camera.yaw_pitch_roll(rng.uniform(-3, 3), rng.uniform(-1.0, 1.0), 0.0))
        lines.append(("inverse", abi.mat_to_glm(proj).tolist()))
        lines.append(("inverse_product", abi.mat_to_glm(proj).tolist() + abi.mat_to_glm(view).tolist()))
    for _ in range(4):
        lines.append(("yaw_pitch_roll", [rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)]))
    out = run("ref_glm_probe", stdin="\n".join(glm_line(op, v) for op, v in lines) + "\n")
    cases = []
    for (op, v), line in zip(lines, out.strip().split("\n")):
        tok = line.split()
        assert tok[0] == op, line
        cases.append({"op": op, "in": [float(np.float32(x)) for x in v], "out": [float(x) for x in tok[1:]]})
    return cases


def write_gltf_files(out_dir):
    """The file of tests/test_gltf.py (both containers) plus a second one with the accessor / transform forms that file
    does not have: a `matrix` node above a rotated TRS child, UNSIGNED_BYTE indices, normalised SHORT normals in a strided
    view, TEXCOORD_1, a light below a rotated parent, a mesh instanced by two nodes."""
    import pathlib
    from tests.test_gltf import _png, _quat, _write
    os.makedirs(out_dir, exist_ok=True)
    p = pathlib.Path(out_dir)
    _write(p, glb=False)
    _write(p, glb=True)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5]], np.float32)
    nrm16 = np.zeros((4, 4), np.int16)                   # VEC3 of normalised shorts, stride 8
    nrm16[:, :3] = [[0, 0, 32767], [0, -32768, 0], [16384, 0, -16384], [-32767, 32767, 1]]
    uv1 = np.array([[0.25, 0.5], [0.75, 0.5], [0.25, 1.0], [2.0, -1.0]], np.float32)
    idx8 = np.array([0, 1, 2, 2, 1, 3], np.uint8)
    chunks, views, acc = [], [], []

    def view(raw, stride=None):
        off = sum(len(c) for c in chunks)
        chunks.append(raw + b"\0" * (-len(raw) % 4))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    def accessor(arr, ctype, typ, count, stride=None, normalized=False, offset=0):
        a = {"bufferView": view(arr.tobytes(), stride), "componentType": ctype, "count": count, "type": typ}
        if normalized:
            a["normalized"] = True
        if offset:
            a["byteOffset"] = offset
        acc.append(a)
        return len(acc) - 1

    a_p = accessor(pos, 5126, "VEC3", 4)
    a_n = accessor(nrm16, 5122, "VEC3", 4, stride=8, normalized=True)
    a_u = accessor(uv1, 5126, "VEC2", 4)
    a_i = accessor(idx8, 5121, "SCALAR", 6)
    shear = np.array([[1, 0.5, 0, 2], [0, 2, 0, -1], [0, 0, 1.5, 3], [0, 0, 0, 1]], np.float64)
    doc = {
        "asset": {"version": "2.0"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1, 0, 0]}, {"type": "directional", "color": [0.5, 0.6, 0.7]}]}},
        "scene": 0, "scenes": [{"nodes": [0, 4]}],
        "nodes": [
            {"name": "root", "matrix": [float(x) for x in shear.T.reshape(16)], "children": [1, 2]},
            {"name": "a", "mesh": 0, "translation": [0.5, 0.25, -2], "rotation": _quat(0.7, 0.3), "scale": [1, 2, 3], "children": [3]},
            {"name": "b", "mesh": 0, "rotation": _quat(-1.1, -0.4)},
            {"name": "sun", "rotation": _quat(2.0, -0.8), "extensions": {"KHR_lights_punctual": {"light": 1}}},
            {"name": "cam", "camera": 0, "translation": [3, 2, 1], "rotation": _quat(-2.5, 0.6)},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 1.1, "znear": 0.2, "zfar": 50.0, "aspectRatio": 2.0}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": a_p, "NORMAL": a_n, "TEXCOORD_1": a_u}, "indices": a_i, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"metallicFactor": 0.0, "roughnessFactor": 0.75}, "alphaMode": "BLEND"}],
        "accessors": acc, "bufferViews": views,
    }
    blob = b"".join(chunks)
    doc["buffers"] = [{"uri": "nested.bin", "byteLength": len(blob)}]
    (p / "nested.bin").write_bytes(blob)
    (p / "nested.gltf").write_text(json.dumps(doc))
    return ["scene.gltf", "scene.glb", "nested.gltf"]


def write_images(out_dir):
    """Small encoded images of the kinds glTF assets carry (PNG colour types, baseline / progressive JPEG at the usual
    chroma subsamplings), encoded once here; the committed files are the inputs of both decoders."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:24, 0:40]
    smooth = np.stack([(x * 6) % 256, (y * 10) % 256, ((x + y) * 4) % 256, 255 - (x * 3) % 200], -1).astype(np.uint8)
    noisy = rng.integers(0, 256, (17, 23, 4), dtype=np.uint8)
    files = {}

    def save(name, img, **kw):
        img.save(os.path.join(out_dir, name), **kw)
        files[name] = True

    save("rgba.png", Image.fromarray(smooth, "RGBA"))
    save("rgba_noise_odd.png", Image.fromarray(noisy, "RGBA"))
    save("rgb.png", Image.fromarray(smooth[..., :3], "RGB"))
    save("gray.png", Image.fromarray(smooth[..., 0], "L"))
    save("gray_alpha.png", Image.fromarray(smooth[..., [0, 3]], "LA"))
    save("palette.png", Image.fromarray(smooth[..., :3], "RGB").quantize(16))
    save("rgb16.png", Image.fromarray((smooth[..., 0].astype(np.uint16) * 257), "I;16"))
    big = np.kron(smooth[..., :3], np.ones((2, 2, 1), np.uint8))
    save("baseline_420.jpg", Image.fromarray(big, "RGB"), quality=85, subsampling=2)
    save("baseline_422.jpg", Image.fromarray(big, "RGB"), quality=85, subsampling=1)
    save("baseline_444.jpg", Image.fromarray(big, "RGB"), quality=92, subsampling=0)
    save("baseline_odd_420.jpg", Image.fromarray(noisy[..., :3], "RGB"), quality=75, subsampling=2)
    save("gray.jpg", Image.fromarray(big[..., 0], "L"), quality=80)
    save("progressive_420.jpg", Image.fromarray(big, "RGB"), quality=85, subsampling=2, progressive=True)
    return sorted(files)


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], check=True)
    with open(os.path.join(HERE, "ref_abi_layout.json"), "w") as f:
        f.write(run("ref_abi_probe"))
    with open(os.path.join(HERE, "ref_glm_cases.json"), "w") as f:
        json.dump(glm_cases(), f, indent=0)
    gdir = os.path.join(HERE, "ref_gltf")
    for name in write_gltf_files(gdir):
        with open(os.path.join(gdir, name + ".cgltf.json"), "w") as f:
            f.write(run("ref_cgltf_probe", os.path.join(gdir, name)))
    sdir = os.path.join(HERE, "ref_stb")
    decodes = {}
    for name in write_images(sdir):
        raw = run("ref_stb_probe", os.path.join(sdir, name), binary=True)
        head, body = raw.split(b"\n", 1)
        w, h = (int(v) for v in head.split())
        decodes[name] = np.frombuffer(body, np.uint8).reshape(h, w, 4)
    np.savez_compressed(os.path.join(HERE, "ref_stb_decodes.npz"), **decodes)
    print("wrote ref_abi_layout.json, ref_glm_cases.json, ref_gltf/, ref_stb/, ref_stb_decodes.npz")


if __name__ == "__main__":
    main()
