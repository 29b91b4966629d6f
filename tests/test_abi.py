"""The C ABI library loads without a GPU, exports every symbol include/vhr_amd.h declares, agrees with the
numpy mirrors of the data ABI, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vulkanhybridrenderer_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vhr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vhr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(vhr):
    L = vhr.load()
    declared = _declared_symbols()
    assert len(declared) >= 50
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == declared


def test_struct_sizes_match_reference_layout(vhr):
    out = (C.c_uint32 * 8)()
    assert vhr.load().vhr_abi_struct_sizes(out) == 7
    assert list(out)[:7] == [56, 44, 120, 112, 584, 24, 32]          # glsl_common.h:22-99
    assert [abi.vertex_dtype.itemsize, abi.material_dtype.itemsize, abi.primitive_dtype.itemsize, abi.directional_light_dtype.itemsize,
            abi.per_frame_dtype.itemsize, abi.svgf_push_constants_dtype.itemsize, abi.trace_params_dtype.itemsize] == list(out)[:7]
    off = {n: abi.per_frame_dtype.fields[n][1] for n in abi.per_frame_dtype.names}
    assert (off["camera_viewproj_inverse"], off["camera_view_prev_frame"], off["directional_light"], off["display_size"],
            off["display_size_inverse"], off["frame_index"], off["blue_noise_texture_index"]) == (256, 320, 448, 560, 568, 576, 580)
    poff = {n: abi.primitive_dtype.fields[n][1] for n in abi.primitive_dtype.names}
    assert (poff["material"], poff["vertex_offset"], poff["index_offset"], poff["index_count"]) == (64, 108, 112, 116)


def test_default_trace_params_are_the_shader_constants(vhr):
    tp = np.zeros((), abi.trace_params_dtype)
    vhr.load().vhr_default_trace_params(tp.ctypes.data_as(C.c_void_p))
    assert tp.tobytes() == abi.default_trace_params().tobytes()
    assert (tp["ao_spp"], float(tp["ao_tmax"]), float(tp["tmin"]), float(tp["tmax"])) == (2, 5.0, np.float32(0.01), 10000.0)


def test_no_cpu_fallback(vhr):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.VhrError, match="no HIP device"):
        lib.Context(64, 64)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vulkanhybridrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vhr_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
