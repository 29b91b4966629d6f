"""The C++ integration example (examples/hybrid_frames.cpp): the reference's host-side classes -- DeviceContext /
ResourceManager / RenderGraph / HybridRenderPath on the facade of include/vhr_render_graph.hpp -- driven the way
Renderer::Render drives them.  CPU: it compiles and links against the public headers + libvhr_amd.so only.
GPU: it runs and finds a plausible picture (a cube's shadow on a lit floor)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "hybrid_frames")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "vulkanhybridrenderer_amd", "csrc"), "-s", "examples"], check=True)
    assert os.path.exists(EXE)


def test_example_builds_against_public_headers_only():
    _build()
    src = open(os.path.join(ROOT, "examples", "hybrid_frames.cpp")).read()
    assert "hip/" not in src and "vhr_internal" not in src          # no HIP, no internals: facade + C ABI only


@pytest.mark.gpu
def test_example_runs(tmp_path):
    _build()
    out = tmp_path / "frame.ppm"
    r = subprocess.run([EXE, "8", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
    assert out.stat().st_size > 640 * 360 * 3
    # the run above saved the path's state half way (HybridRenderPath::SaveState), rebuilt the path and restored it (LoadState): its last frame is the
    # uninterrupted run's, bit for bit
    assert "checkpoint after frame 4" in r.stdout
    assert "resized to 640 x 360 after frame 1" in r.stdout            # DeviceContext::Resize + RenderPath::Build (renderer.cpp:113-118) ran first
    straight = subprocess.run([EXE, "8", str(tmp_path / "straight.ppm"), "straight"], capture_output=True, text=True, timeout=300)
    assert straight.returncode == 0 and "checkpoint" not in straight.stdout.replace("denoised checksum", ""), straight.stdout + straight.stderr
    pick = lambda text: [l for l in text.splitlines() if l.startswith("denoised checksum")]
    assert pick(r.stdout) and pick(r.stdout) == pick(straight.stdout), (pick(r.stdout), pick(straight.stdout))
    assert (tmp_path / "straight.ppm").read_bytes() == out.read_bytes()
