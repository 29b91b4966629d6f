"""Reference-derived pins: the product against the OUTPUTS of the only pieces of the reference that compile
in the build image without stand-ins -- its shared CPU/GPU header `src/rendering_backend/glsl_common.h` (with the vendored
glm), and its vendored glm / cgltf / stb_image called the way `scene_loader.cpp` and `renderer.cpp` call them.  The probes
live in oracle/ref_probes/, are built by `make -C oracle ref` into oracle/_ref/ (build container only) and their outputs are
committed under tests/golden/ by tests/golden/make_ref_pins.py.  Where oracle/_ref/ exists the probes are re-run against
the fixtures (drift check).  What stays unpinned: the shaders' arithmetic (no GLSL compiler / Vulkan driver here)."""
import json
import os
import subprocess

import numpy as np
import pytest

from vulkanhybridrenderer_amd import abi, camera, gltf

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
REF = os.path.join(ROOT, "oracle", "_ref")


def _layout():
    with open(os.path.join(GOLD, "ref_abi_layout.json")) as f:
        return json.load(f)


# reference struct -> (numpy mirror, C struct of include/vhr_types.h)
STRUCTS = {
    "Vertex": (abi.vertex_dtype, "vhr_vertex"), "Material": (abi.material_dtype, "vhr_material"),
    "Primitive": (abi.primitive_dtype, "vhr_primitive"), "DirectionalLight": (abi.directional_light_dtype, "vhr_directional_light"),
    "PerFrameData": (abi.per_frame_dtype, "vhr_per_frame_data"), "SVGFPushConstants": (abi.svgf_push_constants_dtype, "vhr_svgf_push_constants"),
    "SSRPushConstants": (None, "vhr_ssr_push_constants"), "SSAOPushConstants": (None, "vhr_ssao_push_constants"),
}


def test_numpy_mirrors_match_the_reference_header_field_by_field():
    """abi.py's dtypes against sizeof / offsetof of glsl_common.h:22-105 as g++ lays it out with the reference's glm."""
    ref = _layout()
    assert ref["mat4_col1_row2_float_index"] == 6                    # glm column-major: what abi.mat_to_glm assumes
    m = np.zeros((4, 4)); m[2, 1] = 7.0                               # math (row 2, col 1)
    assert int(np.flatnonzero(abi.mat_to_glm(m))[0]) == 6
    for name, (dt, _) in STRUCTS.items():
        if dt is None:
            continue
        assert dt.itemsize == ref[name]["size"], name
        assert list(dt.names) == list(ref[name]["fields"]), name      # same fields, same order
        for field, (off, size) in ref[name]["fields"].items():
            assert dt.fields[field][1] == off and dt.fields[field][0].itemsize == size, (name, field)
    # HybridPushConstants / DefaultPushConstants belong to the raster G-buffer / forward passes (out of scope): not mirrored
    assert set(ref) - set(STRUCTS) == {"DefaultPushConstants", "HybridPushConstants", "mat4_col1_row2_float_index"}


def test_c_header_matches_the_reference_header_field_by_field(tmp_path):
    """include/vhr_types.h compiled by gcc: every field of every mirrored struct at the reference's offset and size."""
    ref = _layout()
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "vhr_types.h"', 'int main(void) {']
    for name, (_, cname) in STRUCTS.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({cname}));')
        for field in ref[name]["fields"]:
            lines.append(f'printf("{name}.{field} %zu %zu\\n", offsetof({cname}, {field}), sizeof((({cname} *)0)->{field}));')
    lines += ["return 0; }"]
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_probe"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n"):
        if line:
            k, *v = line.split()
            got[k] = [int(x) for x in v]
    for name in STRUCTS:
        assert got[name] == [ref[name]["size"]], name
        for field, want in ref[name]["fields"].items():
            assert got[f"{name}.{field}"] == want, (name, field)


def test_oracle_header_matches_the_reference_header_field_by_field(tmp_path):
    """oracle/vhr_oracle.h's own copies of the structs (the checker must read the same bytes the product does)."""
    ref = _layout()
    names = {"Vertex": "orc_vertex", "Material": "orc_material", "Primitive": "orc_primitive", "DirectionalLight": "orc_directional_light",
             "PerFrameData": "orc_per_frame_data"}
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "vhr_oracle.h"', 'int main(void) {']
    for name, cname in names.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({cname}));')
        for field in ref[name]["fields"]:
            lines.append(f'printf("{name}.{field} %zu %zu\\n", offsetof({cname}, {field}), sizeof((({cname} *)0)->{field}));')
    lines += ["return 0; }"]
    src = tmp_path / "orc_abi_probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "orc_abi_probe"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n"):
        if line:
            k, *v = line.split()
            got[k] = [int(x) for x in v]
    for name in names:
        assert got[name] == [ref[name]["size"]], name
        for field, want in ref[name]["fields"].items():
            assert got[f"{name}.{field}"] == want, (name, field)


def _glm_cases():
    with open(os.path.join(GOLD, "ref_glm_cases.json")) as f:
        return json.load(f)


def _close32(a, b, scale=1.0):
    """agreement to fp32 rounding of a short chain of operations (glm computes in fp32, the host code in fp64)."""
    return np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=2e-6, atol=4e-6 * scale)


def test_host_matrices_match_glm_as_the_reference_calls_it():
    """camera.py / gltf.py against glm 0.9.9.8 under GLM_FORCE_DEPTH_ZERO_TO_ONE | GLM_FORCE_RADIANS (pch.h:37-38):
    the shadow frustum (scene_loader.cpp:85-94), the camera's Euler round trip (:58-66), the light direction through
    glm::decompose (:74-86), inverse(proj) and inverse(proj * view) (renderer.cpp:194-195), yawPitchRoll."""
    seen = set()
    for c in _glm_cases():
        op, a, want = c["op"], np.array(c["in"], np.float64), np.array(c["out"], np.float64)
        seen.add(op)
        if op == "ortho_lookat":
            assert _close32(abi.mat_to_glm(camera.light_projview(a)), want)
            assert _close32(camera.directional_light(a)["projview"], want)
        elif op == "camera":
            yaw, pitch, roll, transform, view = gltf.camera_from_world(abi.glm_to_mat(a))
            assert _close32([yaw, pitch, roll], want[:3])
            assert _close32(abi.mat_to_glm(transform), want[3:19], 10.0) and _close32(abi.mat_to_glm(view), want[19:35], 10.0)
        elif op == "lightdir":
            assert _close32(gltf.light_direction_from_world(abi.glm_to_mat(a)), want)
        elif op == "inverse":
            assert _close32(abi.mat_to_glm(np.linalg.inv(abi.glm_to_mat(a))), want, 10.0)
        elif op == "inverse_product":
            p, v = abi.glm_to_mat(a[:16]), abi.glm_to_mat(a[16:])
            got = abi.mat_to_glm(np.linalg.inv(p @ v))
            # entries of the inverse reach 1 / znear: compare relative to the matrix' own scale
            assert _close32(got, want, float(np.abs(want).max()))
        elif op == "yaw_pitch_roll":
            assert _close32(abi.mat_to_glm(camera.yaw_pitch_roll(*a)), want)
    assert seen == {"ortho_lookat", "camera", "lightdir", "inverse", "inverse_product", "yaw_pitch_roll"}


def _primitive_rows(scene, k):
    p = scene.primitives[k]
    nxt = [int(q["vertex_offset"]) for q in scene.primitives if int(q["vertex_offset"]) > int(p["vertex_offset"])]
    v = scene.vertices[int(p["vertex_offset"]): (min(nxt) if nxt else len(scene.vertices))]
    idx = scene.indices[int(p["index_offset"]): int(p["index_offset"]) + int(p["index_count"])]
    return p, v, idx


@pytest.mark.parametrize("name", ["scene.gltf", "scene.glb", "nested.gltf"])
def test_gltf_host_matches_cgltf_on_the_same_files(name):
    """gltf.load() against cgltf 1.9's own reading of the committed files, in ParseNode's order (scene_loader.cpp:40-231):
    world matrices, every vertex attribute through cgltf_accessor_read_float (normalised integers, strides), indices
    through cgltf_accessor_read_index, material / sampler fields, camera parameters, light colour."""
    path = os.path.join(GOLD, "ref_gltf", name)
    with open(path + ".cgltf.json") as f:
        ref = json.load(f)
    s = gltf.load(path)
    k = 0
    for node in ref["nodes"]:
        if "camera" in node:                       # ParseNode returns after the camera / light branch (:70, :99)
            cam = node["camera"]
            assert np.isclose(s.camera["yfov"], cam["yfov"]) and np.isclose(s.camera["znear"], cam["znear"]) and np.isclose(s.camera["aspect"], cam["aspect_ratio"])
            assert _close32(s.camera["position"], node["world"][12:15], 10.0)
            continue
        if node.get("light", {}).get("directional"):
            assert _close32(s.light["color"][:3], node["light"]["color"]) and s.light["color"][3] == 1.0
            assert _close32(s.light["direction"][:3], gltf.light_direction_from_world(abi.glm_to_mat(node["world"])))
            continue
        for rp in node.get("primitives", []):
            p, v, idx = _primitive_rows(s, k)
            k += 1
            assert rp["triangles"]
            assert _close32(p["transform"], node["world"], 10.0)
            assert np.array_equal(idx, np.array(rp["indices"], np.uint32))
            assert len(v) == len(rp["pos"])
            for field, key, n in (("pos", "pos", 3), ("normal", "normal", 3), ("tangent", "tangent", 4), ("uv0", "uv0", 2), ("uv1", "uv1", 2)):
                want = np.zeros((len(v), n), np.float32) if rp[key] is None else np.array(rp[key], np.float32)     # Vertex v {} (:152)
                assert np.array_equal(v[field], want), (field, v[field], want)                                        # bit for bit
            m, rm = p["material"], rp["material"]
            assert rm["has_pbr"]
            assert np.isclose(m["metallic_factor"], rm["metallic_factor"]) and np.isclose(m["roughness_factor"], rm["roughness_factor"])
            assert int(m["alpha_mask"]) == int(rm["alpha_mask"])
            assert np.isclose(m["alpha_cutoff"], rm["alpha_cutoff"] if rm["alpha_mask"] else 0.0)                       # :208-211
            if rm["base_color_texture"] is None:
                assert m["base_color_texture"] == -1 and _close32(m["base_color"], rm["base_color_factor"])            # :195-197
            else:
                assert _close32(m["base_color"], [1, 1, 1, 1])                                                          # :179,191-193
            for field, key in (("base_color_texture", "base_color_texture"), ("metallic_roughness_texture", "metallic_roughness_texture"),
                               ("normal_map", "normal_texture")):
                if rm[key] is None:
                    assert m[field] == -1
                else:
                    t = s.textures[int(m[field])]
                    assert t["name"] == rm[key]["image_name"]
                    smp = rm[key]["sampler"]
                    if smp is not None:            # GetVkFilter / GetVkAddressMode (:8-38) on cgltf's raw GL enums
                        assert (t["mag"], t["min"]) == (gltf._FILTER[smp[0] or 0x2601], gltf._FILTER[smp[1] or 0x2601])
                        assert (t["address_u"], t["address_v"]) == (gltf._WRAP[smp[2]], gltf._WRAP[smp[3]])
    assert k == len(s.primitives)
    if ref["directional_lights"] == 0:
        assert np.allclose(s.light["direction"], [0, -1, 0.01, 0])


def _stb():
    return np.load(os.path.join(GOLD, "ref_stb_decodes.npz"))


LOSSLESS = ["rgba.png", "rgba_noise_odd.png", "rgb.png", "gray.png", "gray_alpha.png", "palette.png", "rgb16.png"]
JPEGS = ["baseline_420.jpg", "baseline_422.jpg", "baseline_444.jpg", "baseline_odd_420.jpg", "gray.jpg", "progressive_420.jpg"]


@pytest.mark.parametrize("name", LOSSLESS + JPEGS)
def test_texture_decode_matches_stb_image(name):
    """gltf.decode_image_bytes() against stb_image 2.26's stbi_load(..., STBI_rgb_alpha) (scene_loader.cpp:277-290) on the
    committed files: texel for texel, PNG colour types and JPEG (stb's integer IDCT, its 'fancy' chroma upsampling and its
    fixed-point YCbCr conversion are restated in vulkanhybridrenderer_amd/stb_jpeg.py)."""
    with open(os.path.join(GOLD, "ref_stb", name), "rb") as f:
        got = gltf.decode_image_bytes(f.read())
    want = _stb()[name]
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want), f"{name}: {(got != want).any(-1).sum()} texels differ, max {np.abs(got.astype(int) - want.astype(int)).max()}"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_abi_probe")), reason="oracle/_ref is built in the build container only (needs /root/reference)")
def test_fixtures_are_what_the_reference_gives_today():
    """Drift check where the reference is present: re-run the probes and compare with the committed fixtures."""
    out = subprocess.run([os.path.join(REF, "ref_abi_probe")], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == _layout()
    for name in ("scene.gltf", "scene.glb", "nested.gltf"):
        path = os.path.join(GOLD, "ref_gltf", name)
        out = subprocess.run([os.path.join(REF, "ref_cgltf_probe"), path], capture_output=True, text=True, check=True).stdout
        with open(path + ".cgltf.json") as f:
            assert json.loads(out) == json.load(f), name
    stb = _stb()
    for name in LOSSLESS + JPEGS:
        raw = subprocess.run([os.path.join(REF, "ref_stb_probe"), os.path.join(GOLD, "ref_stb", name)], capture_output=True, check=True).stdout
        head, body = raw.split(b"\n", 1)
        w, h = (int(v) for v in head.split())
        assert np.array_equal(np.frombuffer(body, np.uint8).reshape(h, w, 4), stb[name]), name
    cases = _glm_cases()
    lines = "\n".join(c["op"] + " " + " ".join(repr(float(v)) for v in c["in"]) for c in cases) + "\n"
    out = subprocess.run([os.path.join(REF, "ref_glm_probe")], input=lines, capture_output=True, text=True, check=True).stdout
    for c, line in zip(cases, out.strip().split("\n")):
        assert [float(x) for x in line.split()[1:]] == c["out"], c["op"]
